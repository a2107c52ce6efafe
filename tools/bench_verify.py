"""bench.py's `verify` leg: after the timed region every rank checks the results of the LAST timed
step — its rows of Y, dX and the all-reduced dW — against the CPU oracle (oracle/tmgcn_ref.c through
oracle/c_ref.py), starting from the REGENERATED seeded inputs, not from anything the exchange
produced.  A sharded step whose exchange, row windows or layouts were wrong cannot pass:

  Y    per local slice k, S sampled rows: their CSR segments (resident adjacency) name the neighbour
       columns; the tube fibres X[:, cols] are regenerated slice by slice from the seeds of WHOEVER
       holds them (all ranks' shards), then the oracle chain ref_mtransform_rows -> ref_spmm ->
       ref_gemm gives Y_ref[k][rows]                                  (ehf:204, 206-207, 222)
  dX   S sampled nodes: for every slice k that reaches this rank's rows of dX through Mᵀ, the
       adjacency slice and the upstream gradient slice are regenerated from their seeds (they
       belong to other ranks), the entries of column n are collected, and
       ref_spmm -> ref_gemm(trans_w) -> ref_mtransform(transpose) gives dX_ref[:, n]
                                                        (autograd of ehf:222, 206-207, 204)
  dW   an fp64 product formed independently on the device from the regenerated inputs with stock
       torch ops (Xt = Σ_j M[k][j]·X[j], AX = Â_k·Xt in row blocks, P += AXᵀ·dY), all-reduced in fp64
       and compared with the fp32 dW the step produced (after ITS all-reduce); plus the adjoint
       identities <Y,dY> = <W,dW> = <X,dX> summed over the ranks.

The oracle is the checker here, never the thing measured.  Everything is plain torch + ctypes and
runs on CPU tensors too (tests/test_bench_verify.py drives it under gloo with the oracle kernels)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _rel(got, ref):
    ref = ref.double()
    scale = max(float(ref.abs().max()), 1e-30)
    return float((got.double().cpu() - ref.cpu()).abs().max()) / scale


def _segments(rowptr, col, val, rows):
    """CSR segments of `rows` (global row indices into rowptr): (sub-rowptr on the host, positions)."""
    lo, hi = rowptr[rows], rowptr[rows + 1]
    cnt = hi - lo
    sub = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=rows.device)
    torch.cumsum(cnt, 0, out=sub[1:])
    total = int(sub[-1])
    seg = torch.repeat_interleave(torch.arange(rows.numel(), device=rows.device), cnt)
    off = torch.arange(total, device=rows.device) - sub[seg] + lo[seg]
    return sub.cpu(), off


def _oracle_rows(lib, cptr, sub, val, gathered, W, trans_w):
    """ref_gemm(ref_spmm(segments, gathered rows), W): `gathered` holds, for every stored non-zero
    of the sampled rows in order, the dense row it multiplies (the sub-problem's column index is the
    position)."""
    nnz, Fk = gathered.shape
    n_rows = sub.numel() - 1
    Nsub = max(nnz, n_rows, 1)
    Xs = torch.zeros(Nsub, Fk, dtype=torch.float32)
    Xs[:nnz] = gathered
    col = torch.arange(nnz, dtype=torch.int32)
    AX = torch.empty(n_rows, Fk, dtype=torch.float32)
    val = val.contiguous()
    lib.ref_spmm(cptr(sub), cptr(col), cptr(val), cptr(Xs), cptr(AX), n_rows, Nsub, Fk)
    Nf = W.shape[0] if trans_w else W.shape[1]
    Y = torch.empty(n_rows, Nf, dtype=torch.float32)
    lib.ref_gemm(cptr(AX), cptr(W), cptr(Y), n_rows, Fk, Nf, int(trans_w), 0, 0)
    return Y


def _spmm64(A, kk, X64, row_block=125_000):
    """Â_kk · X64 in fp64 with stock torch ops (gather + segment sums in row blocks): independent of the
    product's SpMM kernels.  torch.segment_reduce over the CSR's own row lengths — no atomics, so a hub row
    of 10^5 entries costs a loop, not 10^5 colliding atomic adds (index_add_ took 0.5 s per block there)."""
    N, F = A.N, X64.shape[1]
    rp = A.rowptr[kk * N:(kk + 1) * N + 1]
    out = torch.empty(N, F, dtype=torch.float64, device=X64.device)
    for r0 in range(0, N, row_block):
        r1 = min(N, r0 + row_block)
        a, b = int(rp[r0]), int(rp[r1])
        cnt = rp[r0 + 1:r1 + 1] - rp[r0:r1]
        contrib = X64[A.col[a:b].long()] * A.val[a:b].double()[:, None]
        out[r0:r1] = torch.segment_reduce(contrib, "sum", lengths=cnt, axis=0, unsafe=True)
        del contrib
    return out


def verify_layer(*, dist, rank, world, dev, node_sharded_input, A, M64, T, k0, N, W, X, dY, Y, dX, dW,
                 x_slice, dy_slice, a_slice, rows=128, tol=1e-5, seed=4242, fp64_dw=True, y_slices=None):
    """See the module docstring.  A: this rank's resident BatchedCSR (Tl slices); M64: [T,T] fp64
    numpy/torch; X, dY, Y, dX, dW: the step's tensors on this rank (X/dX node-sharded [T, N/G, F] when
    `node_sharded_input`, else slice-sharded [Tl, N, F]; dY/Y slice-sharded).  x_slice(j) -> [N,F] of
    input slice j, dy_slice(k) -> [N,F] of upstream-gradient slice k, a_slice(k) -> one-slice
    BatchedCSR of adjacency slice k — all REGENERATED from seeds, for any rank's data.
    y_slices: check the rows of Y on that many evenly spaced local slices (the first and the last among them) instead of
    on every one — the T = 128 leg of bench.py; dX (which every slice reaches through Mᵀ), dW and the identities always
    cover all of them.
    Returns the `verify` record (errors are MAX over ranks); `ok` is the collective verdict."""
    # all ranks of a node check at the same time on the same host cores: give each its share
    # (libgomp reads OMP_NUM_THREADS when the oracle library is first loaded)
    share = max(1, (os.cpu_count() or 1) // max(1, world))
    os.environ.setdefault("OMP_NUM_THREADS", str(min(share, 32)))
    from oracle import c_ref
    # a stale / missing oracle library is rebuilt by rank 0 alone (c_ref.load holds a file lock around the build);
    # the other ranks wait for it and then only load
    if world > 1:
        if rank == 0:
            c_ref.load()
        dist.barrier()
    lib, cptr = c_ref.load(build_if_stale=(world == 1 or rank == 0)), c_ref.cptr
    t_start = time.perf_counter()
    lap = {}

    def mark(name):
        if dev.type == "cuda":
            torch.cuda.synchronize(dev)
        lap[name] = round(time.perf_counter() - t_start - sum(lap.values()), 1)
    Tl, F = A.T, X.shape[-1]
    F1 = W.shape[1]
    M = torch.as_tensor(M64).double().contiguous()
    Wc = W.detach().float().cpu().contiguous()
    g = torch.Generator(device="cpu").manual_seed(seed + 7919 * rank)
    errs = {}

    # ------------------------------------------------------------------ Y (and the fp64 dW partial)
    S = min(rows, N)
    ids = [torch.randperm(N, generator=g)[:S].sort().values.to(dev) for _ in range(Tl)]
    check = list(range(Tl))
    if y_slices is not None and 1 < y_slices < Tl:
        check = sorted({round(i * (Tl - 1) / (y_slices - 1)) for i in range(y_slices)})
    segs = {}
    for kk in check:
        sub, off = _segments(A.rowptr, A.col, A.val, ids[kk] + kk * N)
        segs[kk] = (sub, A.col[off].long(), A.val[off].cpu())
    local_rows = M[k0:k0 + Tl]                                         # this rank's rows of M
    needed_j = torch.nonzero((local_rows != 0).any(0)).reshape(-1).tolist()
    # fibres of the neighbour columns, only the input slices M actually mixes into this rank's
    # output slices (band-M: Tl + b - 1 of T), in `needed_j` order
    Tn = len(needed_j)
    fib = {kk: torch.zeros(Tn, int(segs[kk][0][-1]), F, dtype=torch.float32) for kk in check}
    want64 = fp64_dw
    if want64 and dev.type == "cuda":
        free_b, _ = torch.cuda.mem_get_info(dev)
        want64 = free_b > Tl * N * F * 8 + (16 << 30)
    if world > 1:
        # a COLLECTIVE decision (as bench.fits() makes its own): the fp64 leg all-reduces P, so either every rank
        # builds it or none does — a rank deciding alone from its own free memory would leave the others waiting in
        # a collective it never enters
        flag = torch.tensor([1 if want64 else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        want64 = bool(int(flag.item()))
    Xt64 = torch.zeros(Tl, N, F, dtype=torch.float64, device=dev) if want64 else None
    for jj, j in enumerate(needed_j):
        Xj = x_slice(j)
        hit = [kk for kk in check if float(local_rows[kk, j]) != 0.0]
        if hit:
            # one gather and one device-to-host copy per input slice (T = 128: 20 output slices share it), then split
            rows_j = Xj[torch.cat([segs[kk][1] for kk in hit])].cpu()
            at = 0
            for kk in hit:
                n = int(segs[kk][1].numel())
                fib[kk][jj] = rows_j[at:at + n]
                at += n
            del rows_j
        if Xt64 is not None:
            Xd = Xj.double()
            for kk in range(Tl):
                m = float(local_rows[kk, j])
                if m != 0.0:
                    Xt64[kk].add_(Xd, alpha=m)
            del Xd
        del Xj
    worst = 0.0
    for kk in check:
        sub, _cols, val = segs[kk]
        nnz = int(sub[-1])
        xt = torch.empty(1, nnz, F, dtype=torch.float32)
        Mrow = torch.zeros(Tn, Tn, dtype=torch.float64)             # row 0 = row k0+kk of M restricted to `needed_j`
        Mrow[0] = local_rows[kk, needed_j]
        lib.ref_mtransform_rows(cptr(Mrow), Tn, 0, 0, 1, cptr(fib[kk]), cptr(xt), nnz * F)
        y_ref = _oracle_rows(lib, cptr, sub, val, xt[0], Wc, False)
        worst = max(worst, _rel(Y[kk][ids[kk]], y_ref))
        fib[kk] = None
    errs["max_rel_err_Y"] = worst
    mark("Y_and_regenerated_P1")

    # ------------------------------------------------------------------ dX
    if node_sharded_input:
        Nl = X.shape[1]
        loc = torch.randperm(Nl, generator=g)[:min(rows, Nl)].sort().values
        nodes = (loc + rank * Nl).to(dev)
        my_j = list(range(T))
    else:
        loc = torch.randperm(N, generator=g)[:S].sort().values
        nodes = loc.to(dev)
        my_j = list(range(k0, k0 + Tl))
    Sx = int(nodes.numel())
    needed_k = torch.nonzero((M[:, my_j] != 0).any(1)).reshape(-1).tolist()
    dXt_ref = torch.zeros(T, Sx, F, dtype=torch.float32)
    for k in needed_k:
        Ak = a_slice(k)
        pos = torch.nonzero(torch.isin(Ak.col, nodes.to(Ak.col.dtype))).reshape(-1)     # entries of the sampled columns
        src = torch.searchsorted(Ak.rowptr, pos, right=True) - 1                          # their rows (in-slice)
        which = torch.searchsorted(nodes, Ak.col[pos].long())                             # index into `nodes`
        order = torch.argsort(which * N + src)                                            # by sampled node, then row
        pos, src, which = pos[order], src[order], which[order]
        sub = torch.zeros(Sx + 1, dtype=torch.int64)
        sub[1:] = torch.cumsum(torch.bincount(which, minlength=Sx), 0).cpu()
        dYk = dy_slice(k)
        dXt_ref[k] = _oracle_rows(lib, cptr, sub, Ak.val[pos].cpu(), dYk[src].cpu(), Wc, True)
        del Ak, dYk
    dX_ref = torch.empty_like(dXt_ref)
    lib.ref_mtransform(cptr(M), T, 1, cptr(dXt_ref), cptr(dX_ref), Sx * F)
    if node_sharded_input:
        errs["max_rel_err_dX"] = _rel(dX[:, loc.to(dev)], dX_ref)
    else:
        errs["max_rel_err_dX"] = _rel(dX[:, nodes], dX_ref[k0:k0 + Tl])
    mark("dX")

    # ------------------------------------------------------------------ dW
    def allsum(t):
        if world > 1:
            dist.all_reduce(t)
        return t

    # adjoint identities <Y,dY> = <W,dW> = <X,dX>, summed over the ranks.  dY is random-sign, so the
    # inner products themselves are small against the norms of their factors (cancellation): the
    # differences are measured against ||W||·||dW|| and ||X||·||dX|| (what an error of relative size
    # eps in dW / dX can move them by), not against the inner product.
    dots = torch.zeros(4, dtype=torch.float64, device=dev)           # <Y,dY>, <X,dX>, ||X||^2, ||dX||^2
    for kk in range(Tl):
        dots[0] += (Y[kk].double() * dY[kk].double()).sum()
    for j in range(X.shape[0]):
        xj, dxj = X[j].detach().double(), dX[j].double()
        dots[1] += (xj * dxj).sum()
        dots[2] += (xj * xj).sum()
        dots[3] += (dxj * dxj).sum()
    dots = allsum(dots)
    ydy, xdx = float(dots[0]), float(dots[1])
    Wd, dWd = W.detach().double(), dW.double()
    wdw = float((Wd * dWd).sum())
    errs["identity_YdY_vs_WdW"] = abs(ydy - wdw) / max(float(Wd.norm() * dWd.norm()), 1e-300)
    errs["identity_YdY_vs_XdX"] = abs(ydy - xdx) / max(float(dots[2].sqrt() * dots[3].sqrt()), 1e-300)
    if Xt64 is not None:
        P = torch.zeros(F, F1, dtype=torch.float64, device=dev)
        for kk in range(Tl):
            AX64 = _spmm64(A, kk, Xt64[kk])
            P += AX64.t() @ dY[kk].double()
            del AX64
        del Xt64
        P = allsum(P)
        errs["max_rel_err_dW"] = _rel(dW, P)
    else:
        errs["max_rel_err_dW"] = None

    mark("dW_fp64_and_identities")

    # ------------------------------------------------------------------ collective verdict
    # the SAME keys in the same order on every rank (a skipped entry travels as -1): the all-reduced vector has one
    # length everywhere whatever each rank measured
    keys = sorted(errs)
    vec = torch.tensor([-1.0 if errs[k] is None else errs[k] for k in keys], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(vec, op=dist.ReduceOp.MAX)
    out = {k: (None if v < 0 else float(v)) for k, v in zip(keys, vec.tolist())}
    ok = all(v <= tol for k, v in out.items() if v is not None)
    out.update({"rows_Y_per_slice": S, "slices_checked": len(check) * world, "slices_local": Tl, "nodes_dX": Sx, "tol": tol, "ok": bool(ok),
                "dW_reference": "fp64 product of the regenerated inputs on the device (torch ops), all-reduced in fp64"
                if out.get("max_rel_err_dW") is not None else "skipped: not enough free memory for the fp64 [T/G,N,F] buffer",
                "reference": "oracle/tmgcn_ref.c (ref_mtransform_rows, ref_spmm, ref_gemm, ref_mtransform) on inputs regenerated "
                             "from their seeds; errors are max|Δ|/max|ref|, MAX over ranks",
                "seconds": round(time.perf_counter() - t_start, 1), "seconds_by_part": lap,
                "slices_regenerated": {"input_X": len(needed_j), "adjacency_and_dY": len(needed_k)}})
    return out

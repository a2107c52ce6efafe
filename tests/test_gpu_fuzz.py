"""GPU: seeded random shapes AND deliberately misaligned operands through every kernel family.

The parametrised tests elsewhere pin chosen shapes; the kernels, however, choose among vector / scalar
and narrow / wide code paths from widths, row counts and pointer alignment at launch time.  Here
every operand is a contiguous view that starts 1..3 floats into its allocation (4-byte aligned only),
shapes are drawn at random, and each result is compared with a torch fp64 restatement of the
reference statement it replaces (ehf:204-207, 222, 228-232) — plus the aligned run, which must agree
with the misaligned one to the same tolerance.  Tolerance: 1e-5 of max|ref|.
"""
import numpy as np
import pytest
import torch

from _util import REL_TOL, max_rel_err
from tmgcn_amd import ops
from tmgcn_amd.csr import BatchedCSR
from tmgcn_amd.losses import weighted_ce

pytestmark = pytest.mark.gpu
DEV = "cuda"


def off(t: torch.Tensor, k: int) -> torch.Tensor:
    """The same values as a contiguous CUDA tensor whose data pointer is k elements past an allocation."""
    flat = torch.empty(t.numel() + k, dtype=t.dtype, device=DEV)
    v = flat[k:].view(t.shape)
    v.copy_(t)
    assert v.is_contiguous() and (k == 0 or v.data_ptr() % 16 != 0)
    return v


def rand_csr(rng, T, N, deg):
    nnz = max(1, int(T * N * deg))
    g = torch.Generator().manual_seed(int(rng.integers(1 << 30)))
    k = torch.randint(0, T, (nnz,), generator=g)
    i = torch.randint(0, N, (nnz,), generator=g)
    j = torch.randint(0, N, (nnz,), generator=g)
    v = torch.randn(nnz, generator=g, dtype=torch.float64)
    return BatchedCSR.from_coo(k, i, j, v, T, N)


def dense_slices(A: BatchedCSR):
    rp, col, val = A.rowptr.cpu(), A.col.cpu().long(), A.val.cpu().double()
    D = torch.zeros(A.T, A.N, A.N, dtype=torch.float64)
    rows = torch.repeat_interleave(torch.arange(A.T * A.N), rp[1:] - rp[:-1])
    D.view(-1, A.N).index_put_((rows, col), val, accumulate=True)
    return D


ACT64 = {None: lambda x: x, "relu": torch.relu, "leaky": lambda x: torch.nn.functional.leaky_relu(x, 0.01),
         "selu": torch.selu}


@pytest.mark.parametrize("seed", range(40))
def test_spmm_gemm_chain_random_shapes_misaligned(seed):
    rng = np.random.default_rng(seed)
    T, N = int(rng.integers(1, 5)), int(rng.integers(3, 400))
    K = int(rng.choice([1, 2, 3, 6, 8, 12, 16, 24, 40, 64, 100, 128, 132]))
    Nf = int(rng.choice([1, 2, 6, 16, 33, 64, 128]))
    act = [None, "relu", "leaky", "selu"][int(rng.integers(4))]
    per_slice = bool(rng.integers(2))
    A = rand_csr(rng, T, N, float(rng.uniform(0.5, 6.0))).to(DEV)
    D = dense_slices(A)
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(T, N, K, generator=g)
    W = torch.randn(*((T, K, Nf) if per_slice else (K, Nf)), generator=g) * 0.3
    ref_ax = torch.matmul(D, X.double())
    ref = ACT64[act](torch.matmul(ref_ax, W.double()))
    for k in (0, int(rng.integers(1, 4))):
        Xd, Wd = off(X, k), off(W, k)
        assert max_rel_err(ops.spmm(A, Xd), ref_ax) <= REL_TOL, ("spmm", T, N, K, k)
        got = ops.spmm_feature_gemm(A, Xd, Wd, act=act)
        assert max_rel_err(got, ref) <= REL_TOL, ("spmm_gemm", T, N, K, Nf, act, per_slice, k)
        got2 = ops.feature_gemm(ops.spmm(A, Xd), Wd, act=act)
        assert max_rel_err(got2, ref) <= REL_TOL, ("gemm", T, N, K, Nf, act, per_slice, k)
        # backward through the registered autograd, upstream gradient misaligned too
        Xg, Wg = Xd.clone().requires_grad_(True), Wd.clone().requires_grad_(True)
        if k:
            Xg, Wg = off(X, k).requires_grad_(True), off(W, k).requires_grad_(True)
        dY = torch.randn(T, N, Nf, generator=g)
        ops.spmm_feature_gemm(A, Xg, Wg, act=act).backward(off(dY, k))
        X64, W64 = X.double().requires_grad_(True), W.double().requires_grad_(True)
        ACT64[act](torch.matmul(torch.matmul(D, X64), W64)).backward(dY.double())
        assert max_rel_err(Xg.grad, X64.grad) <= REL_TOL, ("dX", T, N, K, Nf, act, per_slice, k)
        assert max_rel_err(Wg.grad, W64.grad) <= REL_TOL, ("dW", T, N, K, Nf, act, per_slice, k)
        if K % 4 == 0 and 16 <= K <= 128:   # bf16-stored weight through the bf16-operand kernel
            Wh = Wd.to(torch.bfloat16)
            refh = ACT64[act](torch.matmul(ref_ax, Wh.double().cpu()))
            assert max_rel_err(ops.feature_gemm(off(ref_ax.float(), k), off(Wh, k), act=act), refh) <= REL_TOL


@pytest.mark.parametrize("seed", range(30))
def test_gemm_backward_random_shapes_misaligned(seed):
    rng = np.random.default_rng(100 + seed)
    T, N = int(rng.integers(1, 4)), int(rng.integers(5, 3000))
    K = int(rng.choice([2, 4, 6, 8, 12, 16, 64, 100, 128]))
    Nf = int(rng.choice([2, 4, 6, 8, 16, 36, 128]))
    per_slice = bool(rng.integers(2))
    g = torch.Generator().manual_seed(seed)
    A = torch.randn(T, N, K, generator=g)
    dY = torch.randn(T, N, Nf, generator=g)
    W = torch.randn(*((T, K, Nf) if per_slice else (K, Nf)), generator=g)
    ref_dw = torch.einsum("tnk,tnf->tkf" if per_slice else "tnk,tnf->kf", A.double(), dY.double())
    ref_da = torch.matmul(dY.double(), W.double().transpose(-1, -2))
    for k in (0, int(rng.integers(1, 4))):
        Ad, dYd, Wd = off(A, k), off(dY, k), off(W, k)
        assert max_rel_err(ops.kernels.gemm_dw(Ad, dYd, per_slice), ref_dw) <= REL_TOL, ("dW", T, N, K, Nf, per_slice, k)
        assert max_rel_err(ops.kernels.gemm(dYd, Wd, trans_w=True), ref_da) <= REL_TOL, ("dA", T, N, K, Nf, per_slice, k)


@pytest.mark.parametrize("seed", range(30))
def test_mtransform_random_shapes_misaligned(seed):
    rng = np.random.default_rng(200 + seed)
    T, N, F = int(rng.integers(1, 140)), int(rng.integers(1, 60)), int(rng.choice([1, 2, 3, 4, 6, 8, 16]))
    b = int(rng.integers(1, T + 1))
    dense = bool(rng.integers(2))
    M = torch.tril(torch.randn(T, T, generator=torch.Generator().manual_seed(seed), dtype=torch.float64))
    if not dense:
        M = M - torch.tril(M, -b)       # lower band of width b
    op = ops.MOperator(M, DEV)
    X = torch.randn(T, N, F, generator=torch.Generator().manual_seed(seed + 1))
    ref = torch.einsum("kj,jnf->knf", M, X.double())
    ref_t = torch.einsum("jk,jnf->knf", M, X.double())
    for k in (0, int(rng.integers(1, 4))):
        Xd = off(X, k)
        assert max_rel_err(ops.kernels.mtransform(op, Xd), ref) <= REL_TOL, ("M", T, N, F, b, dense, k)
        assert max_rel_err(ops.kernels.mtransform(op, Xd, transpose=True), ref_t) <= REL_TOL, ("Mt", T, N, F, b, dense, k)


@pytest.mark.parametrize("seed", range(30))
def test_mtransform_column_windows_random_shapes_misaligned(seed):
    """tmgcn_mtransform_ld_f32 under random shapes: a random split of the columns into windows, a random
    row window of a band or dense M, grouped row storage on either side, operands that start 1..3
    floats into their allocation — every window written must carry the bits of the one-shot product
    (same kernel, same arithmetic per element), forward and adjoint; columns outside stay untouched."""
    rng = np.random.default_rng(900 + seed)
    G = int(rng.choice([1, 2, 4]))
    Tl = int(rng.integers(1, 9))
    T = G * Tl
    N, F = int(rng.integers(2, 50)), int(rng.choice([1, 2, 3, 4, 8, 16]))
    b = int(rng.integers(1, T + 1))
    dense = bool(rng.integers(2))
    M = torch.tril(torch.randn(T, T, generator=torch.Generator().manual_seed(seed), dtype=torch.float64))
    if not dense:
        M = M - torch.tril(M, -b)
    op = ops.MOperator(M, DEV)
    k0 = Tl * int(rng.integers(0, G))
    K = ops.kernels
    X = torch.randn(T, N, F, generator=torch.Generator().manual_seed(seed + 1))
    dY = torch.randn(Tl, N, F, generator=torch.Generator().manual_seed(seed + 2))
    pos = torch.tensor([(k % Tl) * (T // Tl) + k // Tl for k in range(T)], device=DEV)   # grouped storage, group = Tl rows
    cuts = sorted(set([0, N] + [int(c) for c in rng.integers(1, N, size=int(rng.integers(0, 4)))]))
    for k in (0, int(rng.integers(1, 4))):
        Xd, dYd = off(X, k), off(dY, k)
        whole = K.mtransform(op, Xd, row_off=k0, col_off=0, T_out=Tl)
        whole_T = K.mtransform(op, dYd, transpose=True, row_off=0, col_off=k0, T_out=T)
        assert max_rel_err(whole, torch.einsum("kj,jnf->knf", M[k0:k0 + Tl], X.double())) <= REL_TOL
        Xg = torch.empty_like(Xd)
        Xg[pos] = Xd
        out = off(torch.full((Tl, N, F), 7.0), k)
        out_T = torch.full((T, N, F), 7.0, device=DEV)
        for c0, c1 in zip(cuts[:-1], cuts[1:]):
            K.mtransform_out(op, off(Xg[:, c0:c1].contiguous().cpu(), k), out[:, c0:c1], row_off=k0, col_off=0, x_group_rows=Tl)
            buf = off(torch.zeros(T, c1 - c0, F), k)
            K.mtransform_out(op, dYd[:, c0:c1], buf, transpose=True, row_off=0, col_off=k0, y_group_rows=Tl)
            out_T[:, c0:c1] = buf[pos]
        # the band kernel's float4 and scalar forms do the same arithmetic per element: bit-equal always.  A
        # dense M takes the bf16-split kernel only for 16-byte aligned operands with C % 4 == 0 and the
        # exact-f32 one otherwise, so windows and whole agree bit for bit when F % 4 == 0 and the operands
        # are aligned (the chunked all-gather's case) and to fp32 accuracy otherwise.
        same_kernel = (op.band_lo + op.band_hi + 1 <= 20) or (F % 4 == 0 and k == 0)
        if same_kernel:
            assert torch.equal(out, whole), ("window fwd", T, Tl, N, F, b, dense, k, cuts)
            assert torch.equal(out_T, whole_T), ("window adj", T, Tl, N, F, b, dense, k, cuts)
        else:
            assert max_rel_err(out, whole) <= REL_TOL and max_rel_err(out_T, whole_T) <= REL_TOL, (T, Tl, N, F, b, dense, k)


@pytest.mark.parametrize("seed", range(40))
def test_edge_head_and_loss_random_shapes_misaligned(seed):
    rng = np.random.default_rng(300 + seed)
    T, N = int(rng.integers(1, 6)), int(rng.integers(2, 500))
    F = int(rng.choice([1, 2, 3, 4, 6, 8, 10, 16, 20, 64, 100, 128, 200]))
    C = int(rng.choice([1, 2, 3, 4, 5, 8]))
    E = int(rng.integers(1, 6000))
    R = T * N
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(0, T, (E,), generator=g)
    edges = torch.stack([t, torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)])
    eidx = ops.EdgeIndex(edges, N, DEV, T=T)
    Z = torch.randn(R, F, generator=g)
    U = torch.randn(2 * F, C, generator=g)
    dout = torch.randn(E, C, generator=g)
    src, dst = (edges[0] * N + edges[1]), (edges[0] * N + edges[2])
    cat = torch.cat([Z[src], Z[dst]], 1).double()
    ref = cat @ U.double()
    ref_du = cat.t() @ dout.double()
    dcat = dout.double() @ U.double().t()
    ref_dz = torch.zeros(R, F, dtype=torch.float64)
    ref_dz.index_add_(0, src, dcat[:, :F])
    ref_dz.index_add_(0, dst, dcat[:, F:])
    for k in (0, int(rng.integers(1, 4))):
        Zd, Ud, dd = off(Z, k), off(U, k), off(dout, k)
        assert max_rel_err(ops.kernels.edge_head_fwd(Zd, eidx, Ud), ref) <= REL_TOL, ("head", F, C, E, k)
        dZ, dU = ops.kernels.edge_head_bwd(Zd, eidx, Ud, dd)
        assert max_rel_err(dZ, ref_dz) <= REL_TOL, ("dZ", F, C, E, R, k)
        assert max_rel_err(dU, ref_du) <= REL_TOL, ("dU", F, C, E, k)
    if C >= 2:
        tgt = torch.randint(0, C, (E,), generator=g)
        w = torch.rand(C, generator=g) + 0.1
        lref = torch.nn.functional.cross_entropy(ref.clone().requires_grad_(True), tgt, weight=w.double())
        zr = ref.clone().requires_grad_(True)
        torch.nn.functional.cross_entropy(zr, tgt, weight=w.double()).backward()
        for k in (0, 1):
            z = off(ref.float(), k).requires_grad_(True)
            loss = weighted_ce(z, tgt.to(DEV), off(w, k), -100)
            loss.backward()
            assert abs(float(loss.detach()) - float(lref.detach())) <= 2e-6 * max(1.0, abs(float(lref))), ("wce", C, E, k)
            assert max_rel_err(z.grad, zr.grad) <= REL_TOL, ("wce grad", C, E, k)

"""Autograd operators of the TM-GCN layer over the C-ABI kernels (include/tmgcn.h).

    m_transform   P1  Xt = M ×₁ X                 (ehf:204, 308, 346, 404; Minv ehf:224)
    spmm          P2  AtXt[k] = Â_k · Xt[k]        (ehf:206-207, 303-304, 310-311, 471-472)
    feature_gemm  P3  Y = act(AtXt · W)            (ehf:222, 330, 344, 349, 486-489)
    activation    P5                               (ehf:284-289)

Every operator runs the hand-written HIP kernels on the current torch stream; there is no
CPU path — tensors that are not fp32 ROCm tensors raise.  The device code is reached through
the PyTorch-ROCm extension layer ``torch.ops.tmgcn.*`` (csrc/torch_ops.cpp: TORCH_LIBRARY
operators with registered C++ autograd over the C-ABI of include/tmgcn.h).  ``kernels`` is the
one object through which the operators reach it (tests of the sharding logic substitute it; the
product never does).
"""
from __future__ import annotations

import functools
import os

import ctypes as C
from typing import Optional

import numpy as np
import torch

from . import _lib
from .csr import BatchedCSR


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None and t.numel() else C.c_void_p(0)


def _stream(t: torch.Tensor):
    return C.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)


def _want(t: torch.Tensor, name: str, dtype=torch.float32):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name}: expected a torch.Tensor, got {type(t).__name__}")
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a ROCm (cuda) tensor, got device {t.device}; "
                           "the TM-GCN layer has no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")
    return t


class EdgeIndexError(IndexError, RuntimeError):
    """An edge (slice, src, dst) outside the embedding tensor.  The reference's
    ``Y.reshape(-1,F)[idx]`` raises IndexError there; the kernels gather unchecked, so the
    check is made once per edge set, before anything is launched."""


class EdgeIndex:
    """Flat row indices t*N+node of every labelled edge (ehf:196-198), plus — built lazily, once
    per edge set — the inverted index the atomic-free head backward walks.  With T given, every
    edge is checked against 0 <= slice < T and 0 <= node < N at construction (one pass where the
    edges live, one host sync per edge set): the forward kernel has no row count to check against."""

    def __init__(self, edges: torch.Tensor, N: int, device, T: Optional[int] = None):
        e = edges.detach()
        if e.dim() != 2 or e.shape[0] != 3:
            raise RuntimeError(f"edges must be [3, E] (slice, src, dst), got {tuple(e.shape)}")
        if e.numel():
            bad = (e[1:] < 0) | (e[1:] >= N)
            bad = bad[0] | bad[1] | (e[0] < 0)
            if T is not None:
                bad = bad | (e[0] >= T)
            if bool(bad.any()):
                j = int(torch.nonzero(bad)[0])
                raise EdgeIndexError(f"edge {j} = (slice {int(e[0, j])}, src {int(e[1, j])}, dst {int(e[2, j])}) is outside "
                                     f"the embedding tensor (T={T}, N={N})")
        e = e.to(device=device, dtype=torch.int64)
        src, dst = e[0] * N + e[1], e[0] * N + e[2]
        self.E = int(src.numel())
        self.T = T
        # The kernels are gather-bound and the index arrays are a third of what they read: on the
        # device they are kept in 32 bits whenever every row index and 2·E fit (always, for the
        # reference's experiments), in the reference's 64 bits otherwise.
        small = T is not None and T * N < 2 ** 31 - 1 and 2 * self.E < 2 ** 31 - 1
        self.index_dtype = torch.int32 if (small and e.device.type == "cuda") else torch.int64
        self.src = src.to(self.index_dtype).contiguous()
        self.dst = dst.to(self.index_dtype).contiguous()
        self._inv = None

    def inverted(self, R: int):
        """(eptr[R+1], eidx[2E]): entries of row r = 2*edge + role, ascending (fixed sum order)."""
        if self._inv is None or self._inv[0] != R:
            if self.E and (int(self.src.max()) >= R or int(self.dst.max()) >= R or
                           int(self.src.min()) < 0 or int(self.dst.min()) < 0):
                raise EdgeIndexError("edge index out of range for the embedding matrix")
            ids = torch.arange(self.E, device=self.src.device, dtype=torch.int64) * 2
            rows = torch.cat((self.src, self.dst)).long()
            ent = torch.cat((ids, ids + 1))
            order = torch.sort(rows, stable=True).indices
            eptr = torch.zeros(R + 1, dtype=torch.int64, device=self.src.device)
            torch.cumsum(torch.bincount(rows, minlength=R), 0, out=eptr[1:])
            it = self.index_dtype if R < 2 ** 31 - 1 else torch.int64
            if it != self.index_dtype:     # the four arrays go to one entry point: one width
                self.index_dtype, self.src, self.dst = it, self.src.to(it), self.dst.to(it)
            self._inv = (R, eptr.to(it), ent[order].to(it).contiguous())
        return self._inv[1], self._inv[2]


class HeadLossPlan:
    """What the one-pass head + loss kernel (csrc/head_loss.hip) walks, built once per (edge set, target):
    the inverted edge index of the edge set, and in the same entry order the row of each entry's OTHER
    endpoint and one byte role << 7 | target class (127 = the criterion's ignore_index), the list of rows that have
    entries at all, plus the
    number of labelled edges per class (Σ_e w[t_e] = Σ_c count[c]·w[c] for whatever class weights the
    call brings).  A label outside [0, C) other than ignore_index raises here (one host sync per target
    tensor; nn.CrossEntropyLoss device-asserts on it).  `sync` is the launch's hand-off word: zero
    between launches, so launches that share a plan must not overlap (one training loop does not)."""

    SPLIT = 0            # entries per part of a split row; 0 = from the lanes per row (see __init__)
    WORTH = 6            # … and rows are only split when the longest holds more than WORTH parts

    def __init__(self, edges: "EdgeIndex", R: int, target: torch.Tensor, C: int, ignore_index: int = -100):
        if edges.index_dtype != torch.int32 or R >= 2 ** 31 - 1:
            raise RuntimeError("head_loss: the edge set needs 64-bit indices; use edge_head + weighted_ce")
        dev = edges.src.device
        t = target.detach().to(device=dev, dtype=torch.int64)
        if t.dim() != 1 or t.numel() != edges.E:
            raise RuntimeError(f"head_loss: target must be [E={edges.E}], got {tuple(target.shape)}")
        if 0 <= ignore_index < C:
            raise RuntimeError(f"head_loss: ignore_index {ignore_index} names a real class (C={C})")
        ignored = t == ignore_index
        bad = ((t < 0) | (t >= C)) & ~ignored
        if bool(bad.any()):
            j = int(torch.nonzero(bad)[0])
            raise RuntimeError(f"head_loss: target[{j}] = {int(t[j])} is outside [0, {C}) and is not ignore_index")
        self.eptr, self.ent = edges.inverted(R)
        e = (self.ent >> 1).long()
        role = (self.ent & 1).bool()
        self.other = torch.where(role, edges.src[e], edges.dst[e]).to(torch.int32).contiguous()
        t8 = torch.where(ignored, torch.full_like(t, 127), t)                     # 7 bits of class, bit 7 = role
        self.meta = (t8[e] | (role.long() << 7)).to(torch.uint8).contiguous()
        active = torch.nonzero(self.eptr[1:] > self.eptr[:-1]).flatten()            # rows with at least one entry
        beg, end = self.eptr[:-1][active].long(), self.eptr[1:][active].long()
        # Hubs of the labelled edges: a group of G <= 16 lanes walks a row's entries, so one node incident to 20 000 of them
        # (real interaction graphs; the negatives of link prediction multiply them by 20) costs the launch 1.5 ms where the
        # uniform case takes 0.1 — and in the sparse regime (G = 1) already a row of 600 costs 0.19 ms instead of 0.045.  Rows
        # that would take their group more than about eight trips are cut into parts of that size (include/tmgcn.h: every
        # sum of the kernel is linear in the entries; dZ shares are added up by tmgcn_head_loss_combine_f32).  The part size
        # follows the lanes the launcher will use (tmgcn_head_loss_lanes), which depend on the number of arow entries:
        # settled in at most two rounds (more parts can only lower the lane count).
        lanes = _lib.load().tmgcn_head_loss_lanes
        n_arow = int(active.numel())
        longest = int((end - beg).max()) if active.numel() else 0
        for _ in range(3):
            G = int(lanes(edges.E, max(1, n_arow)))
            split = self.SPLIT if self.SPLIT else 8 * (2 if G == 1 else 8) * G          # 16 / 256 / 1 024 entries
            n_part = torch.clamp((end - beg + split - 1) // split, min=1)
            if not self.SPLIT and longest <= self.WORTH * split:
                # the 2-layer models pay a launch for adding the parts' dZ shares (about 5 us): not for a few trips more (the
                # reference's chess data, longest row a few dozen entries: head + loss 22.9 us whole, 20.9 + 4.8 split)
                n_part = torch.ones_like(n_part)
            n_new = int(n_part.sum()) if active.numel() else 0
            if n_new == n_arow or int(lanes(edges.E, max(1, n_new))) == G:
                break
            n_arow = n_new
        self.arow, self.srow, self.n_parts = self.split_rows(active, beg, end, n_part, split)
        self.counts = torch.bincount(t[~ignored], minlength=C)[:C].to(torch.int64).contiguous()
        self.sync = torch.zeros(_lib.SYNC_INTS, dtype=torch.int32, device=dev)   # include/tmgcn.h: TMGCN_SYNC_INTS
        self.R, self.C, self.ignore_index = R, C, ignore_index
        self._target, self._version = target, target._version

    @staticmethod
    def split_rows(active: torch.Tensor, beg: torch.Tensor, end: torch.Tensor, n_part: torch.Tensor, split: int):
        """The kernel's row list for active rows `active` with entry ranges [beg, end), row i cut into n_part[i] parts of
        `split` entries (the last one shorter): (arow int32 [n, 4] = (row, first entry, end entry, part id — 0 for a whole
        row, else 1, 2, … over all parts of all split rows in list order), srow int32 [m, 4] = (row, its first part id − 1,
        its number of parts, 0) for the split rows or None, the number of parts).  Pure index arithmetic (tests/test_abi_and_host.py)."""
        dev = active.device
        if not active.numel() or int(n_part.max()) <= 1:
            return torch.stack((active, beg, end, torch.zeros_like(active)), dim=1).to(torch.int32).contiguous(), None, 0
        own = torch.repeat_interleave(torch.arange(active.numel(), device=dev), n_part)     # arow entry -> active row
        k = torch.arange(own.numel(), device=dev) - (torch.cumsum(n_part, 0) - n_part)[own]    # part index inside its row
        is_split = n_part[own] > 1
        pid = torch.cumsum(is_split.long(), 0) * is_split                                    # 1, 2, … over all split rows' parts
        pbeg = beg[own] + k * split
        pend = torch.minimum(pbeg + split, end[own])
        arow = torch.stack((active[own], pbeg, pend, pid), dim=1).to(torch.int32).contiguous()
        sp = n_part > 1
        first = (torch.cumsum(n_part * sp, 0) - n_part * sp)[sp]                             # first part - 1 of every split row
        srow = torch.stack((active[sp], first, n_part[sp], torch.zeros_like(first)), dim=1).to(torch.int32).contiguous()
        return arow, srow, int((n_part * sp).sum())

    def matches(self, target: torch.Tensor, R: int, C: int, ignore_index: int) -> bool:
        return (self._target is target and self._version == target._version and self.R == R and self.C == C
                and self.ignore_index == ignore_index)


def head_loss_plan(edges: "EdgeIndex", R: int, target: torch.Tensor, C: int, ignore_index: int = -100) -> HeadLossPlan:
    """The plan of (edges, target), cached on the edge set (the scripts pass the same target tensor every epoch)."""
    plans = edges.__dict__.setdefault("_loss_plans", [])
    for p in plans:
        if p.matches(target, R, C, ignore_index):
            return p
    p = HeadLossPlan(edges, R, target, C, ignore_index)
    del plans[2:]                      # train / val / test targets at most; older ones are rebuilt on demand
    plans.insert(0, p)
    return p


class MOperator:
    """The T×T mixing matrix M of the M-product, resident on the device in fp32, with the
    band structure the kernels exploit (read_data.m:116-124 builds a lower band of 20)."""

    def __init__(self, M, device):
        M64 = torch.as_tensor(M).detach().to("cpu", torch.float64).contiguous()
        if M64.dim() != 2 or M64.shape[0] != M64.shape[1]:
            raise RuntimeError(f"M must be square, got {tuple(M64.shape)}")
        self.T = int(M64.shape[0])
        nz = torch.nonzero(M64)
        if nz.numel():
            d = nz[:, 1] - nz[:, 0]  # column - row
            self.band_lo = int(max(0, -int(d.min())))
            self.band_hi = int(max(0, int(d.max())))
        else:
            self.band_lo = self.band_hi = 0
        self.M64 = M64
        self.M = M64.to(torch.float32).to(device).contiguous()
        self._inv: Optional["MOperator"] = None

    def inverse(self) -> "MOperator":
        """M⁻¹ computed on the host in fp64 as the reference does (ehf:184, 273)."""
        if self._inv is None:
            self._inv = MOperator(torch.from_numpy(np.linalg.inv(self.M64.numpy())), self.M.device)
        return self._inv

    def window(self, k0: int, k1: int) -> "MOperator":
        """The [k0,k1) x [k0,k1) principal block (the reference's ``M[:-1,:-1]`` idiom)."""
        return MOperator(self.M64[k0:k1, k0:k1], self.M.device)


class KernelTimer:
    """Optional per-launch timing with events recorded on the stream the kernels run on
    (bench.py's roofline leg).  Off by default: `kernels.timer = None`."""

    def __init__(self):
        self.spans = {}

    def launch(self, tag, dev, fn):
        s = torch.cuda.Event(enable_timing=True)
        e = torch.cuda.Event(enable_timing=True)
        st = torch.cuda.current_stream(dev)
        s.record(st)
        rc = fn()
        e.record(st)
        self.spans.setdefault(tag, []).append((s, e))
        return rc

    def summary(self):
        """Per tag: launches, avg_ms, total_ms — each launch from its start event to its end event on the
        stream it ran on.  Where launches of one tag are enqueued on two streams at once (the pipelined
        sharded layer's one-slice launches) a launch's start event precedes its residency, so its time
        includes waiting for the previous launch's blocks to leave: an UPPER bound on the kernel's own
        time there.  (Charging only end-to-end gaps across streams was tried and is not usable: event
        timestamps of different HIP streams did not reproduce rocprofv3's kernel intervals.)"""
        torch.cuda.synchronize()
        out = {}
        for tag, spans in self.spans.items():
            ms = [a.elapsed_time(b) for a, b in spans]
            out[tag] = {"launches": len(ms), "avg_ms": sum(ms) / len(ms), "total_ms": sum(ms)}
        return out


class HipKernels:
    """Kernel-level launchers: one ``torch.ops.tmgcn.*`` call each (tm-gcn_amd/csrc/torch_ops.cpp — the
    TORCH_LIBRARY layer over the C-ABI; it validates, allocates the outputs and launches on torch's
    current stream).  Loading fails loudly if either shared library is missing."""

    name = "hip"

    def __init__(self):
        self.timer: Optional[KernelTimer] = None
        self._ops = None

    @property
    def ops(self):
        if self._ops is None:
            self._ops = _lib.load_torch_ops()
        return self._ops

    def _run(self, tag, dev, fn):
        return self.timer.launch(tag, dev, fn) if self.timer is not None else fn()

    # P1 ---------------------------------------------------------------------------------
    def mtransform(self, op: MOperator, X: torch.Tensor, transpose=False, row_off=0, col_off=0,
                   T_out: Optional[int] = None, x_group_rows=0, y_group_rows=0) -> torch.Tensor:
        lo, hi = (op.band_hi, op.band_lo) if transpose else (op.band_lo, op.band_hi)
        return self._run("mtransform_T" if transpose else "mtransform", X.device, lambda: self.ops.mtransform(
            op.M, X, bool(transpose), row_off, col_off, -1 if T_out is None else T_out, lo, hi, x_group_rows, y_group_rows))

    def mtransform_out(self, op: MOperator, X: torch.Tensor, Y: torch.Tensor, transpose=False, row_off=0, col_off=0,
                       x_group_rows=0, y_group_rows=0, tag=None) -> torch.Tensor:
        """The same product on a column window: X [T_in, n, F] and Y [T_out, n, F] may be views whose
        slices lie further apart than n*F (Xt[:, c0:c1, :] of a resident tensor); written in place
        into Y (tmgcn_mtransform_ld_f32).  The node-chunked all-gather of dist.py is built on it."""
        lo, hi = (op.band_hi, op.band_lo) if transpose else (op.band_lo, op.band_hi)
        tag = tag or ("mtransform_T" if transpose else "mtransform")
        self._run(tag, X.device, lambda: self.ops.mtransform_out(
            op.M, X, Y, bool(transpose), row_off, col_off, lo, hi, x_group_rows, y_group_rows))
        return Y

    # P2 ---------------------------------------------------------------------------------
    def spmm(self, A: BatchedCSR, X: torch.Tensor, tag="spmm") -> torch.Tensor:
        return self._run(tag, X.device, lambda: self.ops.spmm_csr_batched(A.rowptr, A.col, A.val, X, A.N, A.avg_nnz_per_row,
                                                                          *A.giant_plan()))

    # P2+P3 fused ----------------------------------------------------------------------
    def spmm_gemm_supported(self, K: int, Nf: int) -> bool:
        return bool(_lib.load().tmgcn_spmm_gemm_supported(K, Nf))

    def spmm_gemm(self, A: BatchedCSR, X: torch.Tensor, W: torch.Tensor, trans_w=False, act=None,
                  want_ax=False, want_pre=False, tag="spmm_gemm", out=None, grid_reserve=0):
        """act((Â ⋆ X) · Wop) in one launch.  Returns (Y, AX or None, pre or None).
        out = (Y, AX, pre) writes into caller-provided (views of) tensors instead of allocating.
        grid_reserve: block slots this launch leaves free (the pipelined sharded layer's RCCL
        kernels need them); a per-call argument, nothing process-wide."""
        act_id = _lib.ACT_IDS[act]
        if out is not None:
            Y, AX, pre = out
            self._run(tag, X.device, lambda: self.ops.spmm_gemm_out(
                A.rowptr, A.col, A.val, X, A.N, W, bool(trans_w), act_id, Y, AX, pre, int(grid_reserve),
                float(A.avg_nnz_per_row), *A.giant_plan()))
            return Y, AX, pre
        Y, AX, pre = self._run(tag, X.device, lambda: self.ops.spmm_gemm(
            A.rowptr, A.col, A.val, X, A.N, W, bool(trans_w), act_id, bool(want_ax), bool(want_pre), int(grid_reserve),
            float(A.avg_nnz_per_row), *A.giant_plan()))
        return Y, (AX if AX.numel() else None), (pre if pre.numel() else None)

    # P3 ---------------------------------------------------------------------------------
    def gemm(self, A: torch.Tensor, W: torch.Tensor, trans_w=False, act=None, want_pre=False, algo=None):
        """A [T,N,K] · W ([K,Nf] shared or [T,K,Nf] per slice; transposed if trans_w).  W may be stored
        in bf16 (tmgcn_gemm_bf16w_f32: half the matrix-core work, same bits as its fp32-widened copy).
        algo: None / "auto" (bf16x3 split on the bf16 matrix cores for K a multiple of 4 in [16,128])
        or "f32mfma" (exact-f32 MFMA: bitwise an fmaf chain) — per call."""
        Y, pre = self._run("gemm_dA" if trans_w else "gemm", A.device, lambda: self.ops.bgemm(
            A, W, bool(trans_w), _lib.ACT_IDS[act], bool(want_pre), _lib.GEMM_ALGOS[algo]))
        return (Y, pre if pre.numel() else None) if want_pre else Y

    def gemm_dw(self, A: torch.Tensor, dY: torch.Tensor, per_slice: bool, algo=None) -> torch.Tensor:
        """dW = Σ_r A[r]ᵀ dY[r].  algo: None / "auto" (bf16x3 split on the bf16 matrix cores where the
        shapes allow) or "f32mfma" (exact-f32 MFMA kernel) — per call."""
        return self._run("gemm_dW", A.device, lambda: self.ops.bgemm_dW(A, dY, bool(per_slice), _lib.DW_ALGOS[algo]))

    # P4 ---------------------------------------------------------------------------------
    def edge_head_supported(self, F: int, Cn: int) -> bool:
        return bool(_lib.load().tmgcn_edge_head_supported(F, Cn))

    def edge_head_fwd(self, Z2: torch.Tensor, edges: "EdgeIndex", U: torch.Tensor) -> torch.Tensor:
        return self._run("edge_head", Z2.device, lambda: self.ops.edge_head_fwd(Z2, edges.src, edges.dst, U))

    def edge_head_bwd(self, Z2, edges: "EdgeIndex", U, dout, need_dz=True, need_du=True):
        eptr, eidx = edges.inverted(Z2.shape[0])
        dZ, dU = self._run("edge_head_bwd", Z2.device, lambda: self.ops.edge_head_bwd(
            Z2, edges.src, edges.dst, U, dout, eptr, eidx, bool(need_dz), bool(need_du)))
        return (dZ if need_dz else None), (dU if need_du else None)

    # P5 ---------------------------------------------------------------------------------
    def act_fwd(self, x: torch.Tensor, act) -> torch.Tensor:
        return self.ops.act_fwd(x, _lib.ACT_IDS[act])

    def act_bwd(self, x: torch.Tensor, dy: torch.Tensor, act) -> torch.Tensor:
        return self.ops.act_bwd(x, dy, _lib.ACT_IDS[act])


kernels = HipKernels()


# ---------------------------------------------------------------------------------------
# autograd
# ---------------------------------------------------------------------------------------
class _MTransform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, op, row_off, col_off, T_out, x_group_rows, y_group_rows):
        ctx.op, ctx.row_off, ctx.col_off, ctx.T_in = op, row_off, col_off, X.shape[0]
        ctx.xg, ctx.yg = x_group_rows, y_group_rows
        return kernels.mtransform(op, X, False, row_off, col_off, T_out, x_group_rows, y_group_rows)

    @staticmethod
    def backward(ctx, dY):
        # Y[k] = Σ_j M[ro+k][co+j] X[j]   =>   dX[j] = Σ_k Mᵀ[co+j][ro+k] dY[k]
        dX = kernels.mtransform(ctx.op, dY.contiguous(), True, ctx.col_off, ctx.row_off, ctx.T_in,
                                x_group_rows=ctx.yg, y_group_rows=ctx.xg)
        return dX, None, None, None, None, None, None


class _Spmm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X, A):
        ctx.A = A
        return kernels.spmm(A, X)

    @staticmethod
    def backward(ctx, dY):
        # sparse.mm backward: dX_k = Â_kᵀ dY_k (Â is a constant: no gradient, as in the reference)
        return kernels.spmm(ctx.A.transpose(), dY.contiguous(), tag="spmm_T"), None


class _FeatureGemm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, A, W, act):
        ctx.act = act if _lib.ACT_IDS[act] else None
        if ctx.act is not None:
            Y, pre = kernels.gemm(A, W, act=act, want_pre=True)
            ctx.save_for_backward(A, W, pre)
        else:
            Y = kernels.gemm(A, W)
            ctx.save_for_backward(A, W)
        return Y

    @staticmethod
    def backward(ctx, dY):
        dY = dY.contiguous()
        if ctx.act is not None:
            A, W, pre = ctx.saved_tensors
            dY = kernels.act_bwd(pre, dY, ctx.act)
        else:
            A, W = ctx.saved_tensors
        dA = kernels.gemm(dY, W, trans_w=True) if ctx.needs_input_grad[0] else None
        dW = kernels.gemm_dw(A, dY, per_slice=W.dim() == 3) if ctx.needs_input_grad[1] else None
        if dW is not None and dW.dtype != W.dtype:
            dW = dW.to(W.dtype)              # a bf16-stored parameter: summed in fp32, rounded once
        return dA, dW, None


class _SpmmGemm(torch.autograd.Function):
    """Fused P2+P3.  Backward uses Âᵀ(dY·Wᵀ) = (Âᵀ·dY)·Wᵀ: the same fused kernel on dY."""

    @staticmethod
    def forward(ctx, X, W, A, act):
        ctx.A = A
        ctx.act = act if _lib.ACT_IDS[act] else None
        need_w = ctx.needs_input_grad[1]
        Y, AX, pre = kernels.spmm_gemm(A, X, W, act=act, want_ax=need_w, want_pre=True)
        ctx.save_for_backward(W, AX if AX is not None else torch.empty(0, device=X.device),
                              pre if pre is not None else torch.empty(0, device=X.device))
        return Y

    @staticmethod
    def backward(ctx, dY):
        W, AX, pre = ctx.saved_tensors
        dY = dY.contiguous()
        if ctx.act is not None:
            dY = kernels.act_bwd(pre, dY, ctx.act)
        dX = dW = None
        if ctx.needs_input_grad[0]:
            if kernels.spmm_gemm_supported(dY.shape[-1], W.shape[-2]):
                dX, _, _ = kernels.spmm_gemm(ctx.A.transpose(), dY, W, trans_w=True, tag="spmm_gemm_T")
            else:  # the transposed widths have no fused kernel: dA = dY·Wᵀ, then Âᵀ·dA
                dX = kernels.spmm(ctx.A.transpose(), kernels.gemm(dY, W, trans_w=True), tag="spmm_T")
        if ctx.needs_input_grad[1]:
            dW = kernels.gemm_dw(AX, dY, per_slice=W.dim() == 3)
        return dX, dW, None, None


class _EdgeHead(torch.autograd.Function):
    @staticmethod
    def forward(ctx, Z, U, edges):
        Z2 = Z.reshape(-1, Z.shape[-1])
        ctx.edges, ctx.zshape = edges, Z.shape
        ctx.save_for_backward(Z2, U)
        return kernels.edge_head_fwd(Z2, edges, U)

    @staticmethod
    def backward(ctx, dout):
        Z2, U = ctx.saved_tensors
        dZ, dU = kernels.edge_head_bwd(Z2, ctx.edges, U, dout.contiguous(), ctx.needs_input_grad[0],
                                       ctx.needs_input_grad[1])
        return (dZ.reshape(ctx.zshape) if dZ is not None else None), dU, None


class _Activation(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, act):
        ctx.act = act
        ctx.save_for_backward(x)
        return kernels.act_fwd(x, act)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return kernels.act_bwd(x, dy.contiguous(), ctx.act), None


def _registered() -> bool:
    """True when the operators should go through the registered differentiable ops of the torch
    extension (torch.ops.tmgcn.m_transform / spmm / feature_gemm / spmm_feature_gemm / edge_head /
    activation: C++ autograd, one dispatcher call per operator).  The Python autograd functions
    above remain for two cases: a per-launch KernelTimer is attached (bench.py's roofline leg times
    every launch, backward ones included), or `kernels` has been substituted (gloo tests of the
    sharding logic)."""
    return kernels.name == "hip" and kernels.timer is None


def _csr_t(A: BatchedCSR, needed: bool):
    """(t_rowptr, t_col, t_val) of the transposed adjacency (built once per adjacency) when a
    gradient with respect to the dense operand will be asked for, else three Nones."""
    if not needed:
        return None, None, None
    At = A.transpose()
    return At.rowptr, At.col, At.val


def _giant(A: BatchedCSR, needed: bool):
    """(rows, chunks) of A's giant-row plan and of its transpose's (the backward operand; None, None when not needed)."""
    return (*A.giant_plan(), *(A.transpose().giant_plan() if needed else (None, None)))


def m_transform(X: torch.Tensor, op: MOperator, row_off=0, col_off=0, T_out=None, x_group_rows=0,
                y_group_rows=0) -> torch.Tensor:
    """P1: Y[k] = Σ_j M[row_off+k][col_off+j] · X[j]  along the first (time) mode.
    x_group_rows / y_group_rows: group-interleaved row storage of X / Y (include/tmgcn.h)."""
    if _registered():
        return kernels.ops.m_transform(X, op.M, op.band_lo, op.band_hi, row_off, col_off,
                                       -1 if T_out is None else T_out, x_group_rows, y_group_rows)
    return _MTransform.apply(X, op, row_off, col_off, T_out, x_group_rows, y_group_rows)


def spmm(A: BatchedCSR, X: torch.Tensor) -> torch.Tensor:
    """P2: Y[k] = Â_k · X[k] for all frontal slices in one launch."""
    if _registered():
        need = X.requires_grad and torch.is_grad_enabled()
        return kernels.ops.spmm(X, A.rowptr, A.col, A.val, *_csr_t(A, need), A.N, A.avg_nnz_per_row, *_giant(A, need))
    return _Spmm.apply(X, A)


def feature_gemm(A: torch.Tensor, W: torch.Tensor, act=None) -> torch.Tensor:
    """P3 (+ fused P5): act(A · W), W shared ([K,Nf]) or per slice ([T,K,Nf])."""
    if _registered():
        return kernels.ops.feature_gemm(A, W, _lib.ACT_IDS[act])
    return _FeatureGemm.apply(A, W, act)


def spmm_feature_gemm(A: BatchedCSR, X: torch.Tensor, W: torch.Tensor, act=None, fuse: Optional[bool] = None):
    """P2 then P3 (+P5): act((Â ⋆ X) · W).  One fused launch when the kernel supports the
    widths (K a multiple of 8 in [16,128] with Nf <= 128, or K in {1,2,3,4,6,8} with Nf <= 16), else the two
    kernels back to back."""
    K, Nf = X.shape[-1], W.shape[-1]
    can = hasattr(kernels, "spmm_gemm_supported") and kernels.spmm_gemm_supported(K, Nf)
    if fuse is None:
        fuse = can
    if fuse and not can:
        raise RuntimeError(f"fused SpMM+GEMM does not support K={K}, Nf={Nf}")
    if fuse:
        if _registered():
            need = X.requires_grad and torch.is_grad_enabled()
            return kernels.ops.spmm_feature_gemm(X, W, A.rowptr, A.col, A.val, *_csr_t(A, need), A.N,
                                                 A.avg_nnz_per_row, _lib.ACT_IDS[act], 0, *_giant(A, need))
        return _SpmmGemm.apply(X, W, A, act)
    return feature_gemm(spmm(A, X), W, act=act)


@functools.lru_cache(maxsize=None)          # asked once per forward call: a shape's answer never changes (1-2 us per ctypes call)
def _layer12_widths_ok(K0: int, F: int, Nf: int) -> bool:
    return bool(_lib.load().tmgcn_layer12_supported(K0, F, Nf))


def layer12_supported(K0: int, F: int, Nf: int) -> bool:
    return kernels.name == "hip" and _layer12_widths_ok(int(K0), int(F), int(Nf))


# A/B switch of the entry-balanced row blocks of the entry-major layer kernels (tools/: TMGCN_L12_ROW_BLOCKS=0 gives the
# kernels' own 256-row blocks; the forward's results do not depend on it, the backward's dW1 to fp32 summation order)
L12_ROW_BLOCKS = os.environ.get("TMGCN_L12_ROW_BLOCKS", "1") != "0"


def layer12(H: torch.Tensor, W1: torch.Tensor, act1, A: BatchedCSR, W2: torch.Tensor, act2=None, fuse: Optional[bool] = None):
    """act2((Â ⋆ act1(H·W1))·W2): layers 1 and 2 of the narrow 2-layer models (ehf:330-335 + 348-349; 486-487) with H the
    model's cached constant (AtXt / AX).  One forward and one backward launch (csrc/layer12.hip) when H carries no
    gradient, the weights are shared ([K,F] / [F,Nf]) and the widths are 2 -> even F <= 8 -> even Nf <= 8 — the layer-1
    output, its pre-activation and their gradients are never stored; otherwise the two operators back to back."""
    can = (W1.dim() == 2 and W2.dim() == 2 and H.dim() == 3 and not (H.requires_grad and torch.is_grad_enabled())
           and W1.dtype == torch.float32 and W2.dtype == torch.float32 and _registered()
           and layer12_supported(H.shape[-1], W1.shape[-1], W2.shape[-1]) and H.data_ptr() % 8 == 0)
    if fuse is None:
        fuse = can
    if fuse and not can:
        raise RuntimeError("layer12: operands do not allow the fused kernels")
    if not fuse:
        return spmm_feature_gemm(A, feature_gemm(H, W1, act=act1), W2, act=act2)
    need = torch.is_grad_enabled() and W1.requires_grad
    # row blocks cut by entries where 256-row blocks would hold several tiles (real, skewed data): csr.BatchedCSR.row_blocks
    blk = A.row_blocks() if (A.N >= 256 and L12_ROW_BLOCKS) else None
    t_blk = None
    if need and A.N >= 256 and L12_ROW_BLOCKS:
        # the backward takes its entry-major kernel whenever it is handed a partition (include/tmgcn.h): sparse rows take that
        # kernel anyway, skewed ones (hub rows) are 1.7x faster on it, evenly filled denser ones are not — no partition there.
        At = A.transpose()
        if (At.avg_nnz_per_row < 4 or At.is_skewed()) and (At.row_blocks() is not None or At.avg_nnz_per_row >= 4):
            t_blk = At.row_block_runs()
    return kernels.ops.layer12(H, W1.contiguous(), W2.contiguous(), A.rowptr, A.col, A.val, *_csr_t(A, need), A.N, A.avg_nnz_per_row,
                               _lib.ACT_IDS[act1], _lib.ACT_IDS[act2], blk, t_blk)


def head_loss_sgd(AtXt: torch.Tensor, edges: EdgeIndex, W: torch.Tensor, U: torch.Tensor, target: torch.Tensor, weight: torch.Tensor,
                  ignore_index: int, buf_W: Optional[torch.Tensor], buf_U: Optional[torch.Tensor], lr: float, momentum: float = 0.0,
                  dampening: float = 0.0, weight_decay: float = 0.0, nesterov: bool = False, maximize: bool = False,
                  first_step: bool = False):
    """The folded 1-layer model's WHOLE training step in one launch (csrc/head_loss.hip, tmgcn_head_loss_sgd_f32): the
    class-weighted mean cross entropy of logits = [AtXt·W][src] ‖ [AtXt·W][dst] · U, its gradients dW, dU, and torch.optim.SGD's
    update of W and U IN PLACE (momentum buffers updated as well).  Returns (loss, dW, dU), detached.  No autograd: for an
    optimizer-aware step (graphs.GraphedTrainStep(fold_optimizer=True))."""
    R = AtXt.shape[0] * AtXt.shape[1]
    plan = head_loss_plan(edges, R, target, U.shape[-1], ignore_index)
    w = weight.to(device=AtXt.device, dtype=torch.float32).contiguous()
    with torch.no_grad():
        return kernels.ops.head_loss_sgd(AtXt.reshape(R, AtXt.shape[-1]), W, U, plan.eptr, plan.arow, plan.other, plan.meta, plan.counts, w,
                                         plan.sync, buf_W, buf_U, float(lr), float(momentum), float(dampening), float(weight_decay),
                                         bool(nesterov), bool(maximize), bool(first_step))


def edge_head(Z: torch.Tensor, edges: EdgeIndex, U: torch.Tensor, fuse: Optional[bool] = None) -> torch.Tensor:
    """P4: logits[e] = [Z[src[e]], Z[dst[e]]] · U  (ehf:228-232).  One fused gather-and-dot kernel
    (atomic-free backward) when F <= 256 and C <= 8, else stock gather + matmul."""
    F, Cn = Z.shape[-1], U.shape[-1]
    can = hasattr(kernels, "edge_head_supported") and kernels.edge_head_supported(F, Cn)
    if fuse is None:
        fuse = can
    if fuse and not can:
        raise RuntimeError(f"fused edge head does not support F={F}, C={Cn}")
    if fuse:
        if _registered():
            need = torch.is_grad_enabled() and (Z.requires_grad or U.requires_grad)
            eptr, eidx = edges.inverted(Z.numel() // F) if need else (None, None)
            return kernels.ops.edge_head(Z, U.contiguous(), edges.src, edges.dst, eptr, eidx)
        return _EdgeHead.apply(Z.contiguous(), U.contiguous(), edges)
    Zf = Z.reshape(-1, F)
    return torch.matmul(torch.cat((Zf[edges.src.long()], Zf[edges.dst.long()]), dim=1), U)


@functools.lru_cache(maxsize=None)
def _head_loss_widths_ok(F: int, Cn: int, K: int) -> bool:
    return bool(_lib.load().tmgcn_head_loss_supported(F, Cn, K))


def head_loss_supported(F: int, Cn: int, K: int = 0) -> bool:
    return kernels.name == "hip" and _head_loss_widths_ok(int(F), int(Cn), int(K))


def widen_params(params):
    """fp32 copies of bf16-stored parameters in one launch (registered autograd: the gradients are rounded back to
    bf16 in one launch too)."""
    return list(kernels.ops.widen_params(list(params))) if params else []


def unit_gradient(device) -> torch.Tensor:
    """The constant 1.0 (one 0-dim fp32 tensor per device) to pass as ``loss.backward(gradient=...)``: autograd then
    does not fill a fresh ones_like(loss) every step, and the fused head + loss recognises it by address and skips
    the launch that multiplies its gradients by the upstream gradient."""
    return kernels.ops.unit_gradient(torch.empty(0, device=device))


def head_loss(Z: torch.Tensor, edges: EdgeIndex, U: torch.Tensor, target: torch.Tensor, weight: torch.Tensor,
              ignore_index: int = -100, want_logits: bool = False, fold_W: Optional[torch.Tensor] = None,
              unit_grad: bool = False):
    """``nn.CrossEntropyLoss(weight)(edge_head(Z, edges, U), target)`` — P4 and the criterion of every experiment
    script (ehf:228-232 + experiment_reddit_our_link_prediction.py:69, 79) — with every gradient, in ONE launch
    (csrc/head_loss.hip) where the widths allow (even F <= 8, C <= 4, 32-bit indices); otherwise the edge head
    followed by the fused weighted CE.  Returns the loss, or (loss, logits) with ``want_logits`` (the logits are
    a non-differentiable by-product there: differentiate the loss).
    fold_W: Z is the 1-layer model's cached AtXt [T,N,2] and fold_W its shared weight [2,F]: Z = AtXt·W (ehf:222)
    is recomputed inside the kernel and dW returned through autograd — nothing of size [T,N,F] is stored.
    unit_grad: a promise that backward will be run as ``loss.backward(gradient=ops.unit_gradient(device))``: loss and
    gradients then take one launch for every shape (a different upstream gradient is still honoured, at the price of
    one scaling launch)."""
    F = (fold_W if fold_W is not None else Z).shape[-1]
    K = Z.shape[-1] if fold_W is not None else 0
    Cn = U.shape[-1]
    R = Z.numel() // Z.shape[-1]
    fused = (head_loss_supported(F, Cn, K) and edges.E > 0 and edges.index_dtype == torch.int32 and R < 2 ** 31 - 1
             and Z.dtype == torch.float32)
    if not fused:
        from .losses import weighted_ce
        Zf = feature_gemm(Z, fold_W) if fold_W is not None else Z
        logits = edge_head(Zf, edges, U)
        loss = weighted_ce(logits, target.to(logits.device), weight.to(logits.device), ignore_index)
        return (loss, logits) if want_logits else loss
    plan = head_loss_plan(edges, R, target, Cn, ignore_index)
    w = weight.detach().to(device=Z.device, dtype=torch.float32).contiguous()
    loss, logits = kernels.ops.head_loss(Z, fold_W, U.contiguous(), plan.eptr, plan.arow, plan.ent, plan.other, plan.meta,
                                         plan.counts, w, plan.sync, bool(want_logits), bool(unit_grad), plan.srow, plan.n_parts)
    return (loss, logits) if want_logits else loss


def activation(x: torch.Tensor, act) -> torch.Tensor:
    if _registered():
        return kernels.ops.activation(x, _lib.ACT_IDS[act])
    return _Activation.apply(x, act)

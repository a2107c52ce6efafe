"""Device-side adjacency pipeline: raw dynamic-graph edges -> Ĉ and Â = M ×₁ Ĉ
as batched CSR, all on the MI355X through the C-ABI (tm-gcn_amd/csrc/adjacency.hip).

Reference (offline, per-slice / per-nnz Python loops, minutes on real data):
    read_data.py:88-111   func_make_symmetric          B = (A + Aᵀ)/2
    read_data.py:116-125  func_edge_life               B'[t] = Σ_{s>t-L} B[s]
    read_data.py:130-169  func_laplacian_transformation   C = D^-1/2 (B' + I) D^-1/2
    read_data.py:204-223  func_MProduct                Ct[k] = Σ_j M[k,j] C[j]
    (MATLAB originals: read_data.m:172-209)
Here symmetrise / edge life / + I are expand -> rocPRIM sort -> reduce-by-key on 64-bit keys (slice,
row, col); the M-product — the step that multiplies the stored non-zeros by the band width — is a
hand-written segmented merge of the column-sorted CSR rows (`m_product_csr`: two passes, no expansion,
no sort, memory = the output), with the expand + sort form kept for bands wider than 64 slices.
A T=95, N=6000 graph takes milliseconds.  Checked against the reference's own functions run on
the chess data it ships (tests/golden/g5_chess_gcn2.npz).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from .csr import BatchedCSR
from .ops import MOperator, _ptr, _stream

SENTINEL = -1  # ~0 as int64
MERGE_MAX_BAND = 64   # slices the band of M may reach from one output slice in the merge kernel


class DeviceCOO:
    """A batched COO on the device: int64-viewed uint64 keys (slice*N+row)*N+col, fp32 values."""

    def __init__(self, key: torch.Tensor, val: torch.Tensor, T: int, N: int, sorted_reduced: bool = False):
        self.key, self.val, self.T, self.N = key, val, T, N
        # keys ascending and distinct (the state sort_reduce leaves): what to_csr and the merge kernels assume.
        # Steps that need it sort first when the flag is not set, instead of silently building a wrong CSR.
        self.sorted_reduced = sorted_reduced

    @property
    def n(self) -> int:
        return int(self.key.numel())

    @staticmethod
    def from_edges(t, i, j, w, T: int, N: int, device="cuda") -> "DeviceCOO":
        lib = _lib.load()
        t = torch.as_tensor(t).to(device=device, dtype=torch.int64).contiguous()
        i = torch.as_tensor(i).to(device=device, dtype=torch.int64).contiguous()
        j = torch.as_tensor(j).to(device=device, dtype=torch.int64).contiguous()
        w = torch.as_tensor(w).to(device=device, dtype=torch.float32).contiguous()
        if t.numel():
            if int(t.min()) < 0 or int(t.max()) >= T or int(min(i.min(), j.min())) < 0 or int(max(i.max(), j.max())) >= N:
                raise RuntimeError("edge list: index out of range")
        if T * N * N >= 2 ** 63:
            raise RuntimeError("T*N*N does not fit the 64-bit key")
        key = torch.empty(t.numel(), dtype=torch.int64, device=device)
        _lib.check(lib.tmgcn_adj_make_keys(_ptr(t), _ptr(i), _ptr(j), t.numel(), N, _ptr(key), _stream(w)),
                   "tmgcn_adj_make_keys")
        return DeviceCOO(key, w, T, N)

    # -- the one primitive ---------------------------------------------------------------------
    def sort_reduce(self) -> "DeviceCOO":
        """Sort by key, sum equal keys (sorted order), drop the sentinel run."""
        lib = _lib.load()
        n = self.n
        dev = self.val.device
        ko = torch.empty(n, dtype=torch.int64, device=dev)
        vo = torch.empty(n, dtype=torch.float32, device=dev)
        cnt = torch.zeros(1, dtype=torch.int64, device=dev)
        need = int(lib.tmgcn_coo_sort_reduce_workspace_bytes(n))
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        _lib.check(lib.tmgcn_coo_sort_reduce(_ptr(self.key), _ptr(self.val), n, 0, _ptr(ko), _ptr(vo), _ptr(cnt),
                                             _ptr(ws), ws.numel(), _stream(vo)), "tmgcn_coo_sort_reduce")
        m = int(cnt.item())  # plan-time sync: the number of distinct entries is data dependent
        if m and int(ko[m - 1].item()) == SENTINEL:
            m -= 1
        return DeviceCOO(ko[:m].contiguous(), vo[:m].contiguous(), self.T, self.N, sorted_reduced=True)

    def _expand(self, fn_name: str, fan: int, *args) -> "DeviceCOO":
        lib = _lib.load()
        dev = self.val.device
        ko = torch.empty(self.n * fan, dtype=torch.int64, device=dev)
        vo = torch.empty(self.n * fan, dtype=torch.float32, device=dev)
        fn = getattr(lib, fn_name)
        _lib.check(fn(_ptr(self.key), _ptr(self.val), self.n, self.N, *args, _ptr(ko), _ptr(vo), _stream(vo)), fn_name)
        return DeviceCOO(ko, vo, self.T, self.N)

    # -- the reference's steps -------------------------------------------------------------------
    def symmetrise(self) -> "DeviceCOO":
        """(A + Aᵀ)/2 per slice — read_data.py:88-111."""
        return self._expand("tmgcn_adj_symmetrise", 2).sort_reduce()

    @staticmethod
    def from_csr(A: BatchedCSR) -> "DeviceCOO":
        """Batched CSR -> sorted, reduced COO keys (slice*N + row)*N + col."""
        return DeviceCOO((A.row_ids() * A.N + A.col.long()).contiguous(), A.val, A.T, A.N, sorted_reduced=True)

    def edge_life(self, window: int, algo: str = "auto") -> "DeviceCOO":
        """B'[t] = B[t] + B[t-1] + … + B[t-window+1] — read_data.py:116-125.  That is the mode-1 product
        with a lower band of ones, so it runs as the same segmented merge as the M-product (no
        `window`-fold expansion, no sort) whenever the window fits the merge kernel; `algo="expand"`
        keeps the expand + sort + reduce form.  An input that is not sorted and reduced yet (a raw
        `from_edges` list) is sorted first."""
        if window <= 1:
            return self
        if not self.sorted_reduced:
            return self.sort_reduce().edge_life(window, algo)
        if algo == "auto":
            algo = "merge" if window <= MERGE_MAX_BAND else "expand"
        if algo == "merge":
            import numpy as np
            ones = np.tril(np.ones((self.T, self.T))) - np.tril(np.ones((self.T, self.T)), -min(window, self.T))
            return DeviceCOO.from_csr(m_product_csr(self.to_csr(), ones, algo="merge"))
        return self._expand("tmgcn_adj_edge_life", window, self.T, window).sort_reduce()

    def add_identity_and_normalise(self) -> "DeviceCOO":
        """C = D^-1/2 (B + I) D^-1/2, D = row sums of B + I — read_data.py:130-169."""
        lib = _lib.load()
        if not self.sorted_reduced:
            return self.sort_reduce().add_identity_and_normalise()
        dev = self.val.device
        TN = self.T * self.N
        ik = torch.empty(TN, dtype=torch.int64, device=dev)
        iv = torch.empty(TN, dtype=torch.float32, device=dev)
        _lib.check(lib.tmgcn_adj_identity(TN, self.N, _ptr(ik), _ptr(iv), _stream(iv)), "tmgcn_adj_identity")
        c = DeviceCOO(torch.cat((self.key, ik)), torch.cat((self.val, iv)), self.T, self.N).sort_reduce()
        rowptr = torch.empty(TN + 1, dtype=torch.int64, device=dev)
        dinv = torch.empty(TN, dtype=torch.float32, device=dev)
        _lib.check(lib.tmgcn_adj_normalise(_ptr(c.key), _ptr(c.val), c.n, self.N, TN, _ptr(rowptr), _ptr(dinv),
                                           _stream(dinv)), "tmgcn_adj_normalise")
        return c

    def m_product(self, M) -> "DeviceCOO":
        """Mode-1 product of the sparse tensor with M: Ct[k] = Σ_j M[k,j] C[j] — read_data.py:204-223."""
        op = M if isinstance(M, MOperator) else MOperator(M, self.val.device)
        if op.T != self.T:
            raise RuntimeError(f"M is {op.T}x{op.T} but the tensor has {self.T} slices")
        lo, hi = min(op.band_lo, self.T - 1), min(op.band_hi, self.T - 1)
        return self._expand("tmgcn_adj_mproduct_expand", lo + hi + 1, self.T, _ptr(op.M), op.T, lo, hi).sort_reduce()

    def to_csr(self) -> BatchedCSR:
        """Sorted, reduced COO -> batched CSR (rowptr by binary search of the keys, col = key mod N); an unsorted
        list is sorted and reduced first."""
        if not self.sorted_reduced:
            return self.sort_reduce().to_csr()
        lib = _lib.load()
        dev = self.val.device
        TN = self.T * self.N
        rowptr = torch.empty(TN + 1, dtype=torch.int64, device=dev)
        col = torch.empty(self.n, dtype=torch.int32, device=dev)
        _lib.check(lib.tmgcn_adj_keys_to_csr(_ptr(self.key), self.n, self.N, TN, _ptr(rowptr), _ptr(col),
                                             _stream(self.val)), "tmgcn_adj_keys_to_csr")
        return BatchedCSR(rowptr, col, self.val, self.T, self.N)


def m_product_csr(A: BatchedCSR, M, algo: str = "auto") -> BatchedCSR:
    """Ct[k] = Σ_j M[k,j] C[j] on a batched CSR (read_data.py:204-223, SBM_our.py:78-86).
    algo: "merge" — segmented merge of the CSR rows (tmgcn_adj_mproduct_merge_count / _fill: count,
    prefix sum, fill; contributions to one entry summed in fp64 in a fixed order), "expand" — fan
    every entry out to the slices it reaches, sort, reduce by key (any band width; W x the memory),
    "auto" — merge whenever the band fits (it always does for the reference's 20 diagonals)."""
    lib = _lib.load()
    op = M if isinstance(M, MOperator) else MOperator(M, A.device)
    if op.T != A.T:
        raise RuntimeError(f"M is {op.T}x{op.T} but the tensor has {A.T} slices")
    lo, hi = min(op.band_lo, A.T - 1), min(op.band_hi, A.T - 1)
    if algo not in ("auto", "merge", "expand"):
        raise RuntimeError(f"unknown algo {algo!r}")
    if algo == "auto":
        algo = "merge" if lo + hi + 1 <= MERGE_MAX_BAND else "expand"
    if algo == "expand":
        rows = A.row_ids()
        key = (rows * A.N + A.col.long()).contiguous()          # (slice*N + row)*N + col
        return DeviceCOO(key, A.val, A.T, A.N).m_product(op).to_csr()
    TN = A.n_rows
    cnt = torch.empty(TN + 1, dtype=torch.int64, device=A.device)
    _lib.check(lib.tmgcn_adj_mproduct_merge_count(_ptr(A.rowptr), _ptr(A.col), TN, A.N, A.T, _ptr(op.M), op.T, lo, hi,
                                                  _ptr(cnt), _stream(A.val)), "tmgcn_adj_mproduct_merge_count")
    rowptr = torch.cumsum(cnt, 0)
    nnz = int(rowptr[-1].item())    # plan-time sync: the size of the result is data dependent
    col = torch.empty(nnz, dtype=torch.int32, device=A.device)
    val = torch.empty(nnz, dtype=torch.float32, device=A.device)
    _lib.check(lib.tmgcn_adj_mproduct_merge_fill(_ptr(A.rowptr), _ptr(A.col), _ptr(A.val), TN, A.N, A.T, _ptr(op.M), op.T,
                                                 lo, hi, _ptr(rowptr), _ptr(col), _ptr(val), _stream(A.val)),
               "tmgcn_adj_mproduct_merge_fill")
    return BatchedCSR(rowptr, col, val, A.T, A.N)


def csr_transpose(A: BatchedCSR) -> BatchedCSR:
    """Per-slice transpose of a batched CSR through the native sort (no torch compute ops)."""
    lib = _lib.load()
    key = torch.empty(A.nnz, dtype=torch.int64, device=A.device)
    _lib.check(lib.tmgcn_adj_transpose_keys(_ptr(A.rowptr), _ptr(A.col), A.n_rows, A.N, _ptr(key), _stream(A.val)),
               "tmgcn_adj_transpose_keys")
    return DeviceCOO(key, A.val, A.T, A.N).sort_reduce().to_csr()


def build_adjacency(t, i, j, w, T: int, N: int, M=None, window: int = 10, symmetric: bool = True,
                    device="cuda") -> Tuple[BatchedCSR, Optional[BatchedCSR]]:
    """Raw edges (slice, src, dst, weight) -> (Ĉ, Â): the normalised adjacency (EmbeddingKWGCN's
    input) and its M-product (EmbeddingGCN / EmbeddingGCN2's input; None when M is None).
    Duplicate edges are summed, as `sptensor(...)` / `.coalesce()` do in the reference."""
    coo = DeviceCOO.from_edges(t, i, j, w, T, N, device).sort_reduce()
    if symmetric:
        coo = coo.symmetrise()
    c = coo.edge_life(window).add_identity_and_normalise()
    Chat = c.to_csr()
    Ahat = m_product_csr(Chat, M) if M is not None else None
    return Chat, Ahat

"""Batched CSR: the HBM format of the adjacency tensor Â (T frontal slices, N×N each).

The reference keeps Â as a Python list of T 2-D COO tensors with int64 indices and fp64
values (embedding_help_functions.py:560-574, experiment_bitcoin_our.py:52-64) and
multiplies them one ``t.sparse.mm`` at a time.  Here all T slices live in ONE block-diagonal
CSR so that a single kernel launch covers the whole tensor:

    rowptr  int64 [T*N + 1]   global offsets into col/val; row r = k*N + i  (slice k, node i)
    col     int32 [nnz]       column inside the slice, 0..N-1
    val     fp32  [nnz]

Ingest on a ROCm device goes through the native sort/reduce of csrc/adjacency.hip (rocPRIM
radix sort on 64-bit (slice,row,col) keys; duplicate entries are summed, which is what
``sparse.mm`` does with uncoalesced COO).  On the CPU (oracle/tests only) the same layout is
built with torch ops and duplicates are kept as separate entries.  Within a row, entries are
ordered by column, so every row sum has a fixed order.
"""
from __future__ import annotations

import os
from typing import List, Optional, Sequence

import torch


_GIANT_PLAN = os.environ.get("TMGCN_GIANT_PLAN", "1") != "0"      # A/B switch (tools/): 0 = never build a giant-row plan


def _header_constants(*names):
    """Integer #defines of include/tmgcn.h (the header the kernels were compiled against ships with the package)."""
    import re
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "tmgcn.h")
    text = open(path).read()
    out = []
    for n in names:
        m = re.search(r"^#define\s+" + n + r"\s+(\d+)\b", text, re.M)
        if m is None:
            raise RuntimeError(f"{n} is not defined in {path}")
        out.append(int(m.group(1)))
    return tuple(out)


class BatchedCSR:
    def __init__(self, rowptr: torch.Tensor, col: torch.Tensor, val: torch.Tensor, T: int, N: int):
        assert rowptr.dtype == torch.int64 and col.dtype == torch.int32 and val.dtype == torch.float32
        assert rowptr.numel() == T * N + 1, (rowptr.numel(), T, N)
        assert col.numel() == val.numel()
        self.rowptr = rowptr.contiguous()
        self.col = col.contiguous()
        self.val = val.contiguous()
        self.T = int(T)
        self.N = int(N)
        self.nnz = int(col.numel())
        self._t: Optional["BatchedCSR"] = None
        self._blocks = {}

    # ------------------------------------------------------------------ properties
    @property
    def device(self):
        return self.val.device

    @property
    def n_rows(self) -> int:
        return self.T * self.N

    @property
    def avg_nnz_per_row(self) -> float:
        return self.nnz / max(1, self.n_rows)

    def __repr__(self):
        return f"BatchedCSR(T={self.T}, N={self.N}, nnz={self.nnz}, device={self.device})"

    # ------------------------------------------------------------------ constructors
    @staticmethod
    def from_coo(slice_idx, row, col, val, T: int, N: int, device=None) -> "BatchedCSR":
        """Build from one batched COO (any order).  Index tensors are integer, val any float."""
        val = torch.as_tensor(val)
        device = torch.device(device) if device is not None else val.device
        if device.type == "cuda":
            from .adjacency import DeviceCOO  # native path: no torch compute ops
            return DeviceCOO.from_edges(slice_idx, row, col, val, T, N, device).sort_reduce().to_csr()
        slice_idx = torch.as_tensor(slice_idx).to(device=device, dtype=torch.int64)
        row = torch.as_tensor(row).to(device=device, dtype=torch.int64)
        col = torch.as_tensor(col).to(device=device, dtype=torch.int64)
        val = torch.as_tensor(val).to(device=device)
        if slice_idx.numel():
            if int(slice_idx.min()) < 0 or int(slice_idx.max()) >= T:
                raise RuntimeError(f"slice index out of range [0,{T})")
            if int(min(row.min(), col.min())) < 0 or int(max(row.max(), col.max())) >= N:
                raise RuntimeError(f"node index out of range [0,{N}) — the adjacency does not match X.shape[1]")
        key = (slice_idx * N + row) * N + col  # T*N*N < 2^63 for every config in scope
        order = torch.sort(key, stable=True).indices
        rkey = (slice_idx * N + row)[order]
        counts = torch.bincount(rkey, minlength=T * N)
        rowptr = torch.zeros(T * N + 1, dtype=torch.int64, device=device)
        torch.cumsum(counts, 0, out=rowptr[1:])
        return BatchedCSR(rowptr, col[order].to(torch.int32), val[order].to(torch.float32), T, N)

    @staticmethod
    def from_coo_list(slices: Sequence[torch.Tensor], N: Optional[int] = None, device=None) -> "BatchedCSR":
        """Build from the reference's list of T sparse COO matrices (ehf:560-574).

        The reference builds each slice without an explicit size (ehf:564); N therefore has
        to come from the feature tensor (``X.shape[1]``) and is validated against the indices.
        """
        T = len(slices)
        if T == 0:
            raise RuntimeError("empty adjacency list")
        ks, rs, cs, vs = [], [], [], []
        n_seen = 0
        for k, a in enumerate(slices):
            if not a.is_sparse:
                raise RuntimeError(f"slice {k} is not a sparse COO tensor")
            idx = a._indices()
            v = a._values()
            if idx.shape[0] != 2:
                raise RuntimeError(f"slice {k} is not 2-D")
            ks.append(torch.full((idx.shape[1],), k, dtype=torch.int64, device=idx.device))
            rs.append(idx[0])
            cs.append(idx[1])
            vs.append(v)
            n_seen = max(n_seen, int(a.shape[0]), int(a.shape[1]))
        if N is None:
            N = n_seen
        return BatchedCSR.from_coo(torch.cat(ks), torch.cat(rs), torch.cat(cs), torch.cat(vs), T, int(N),
                                   device=device if device is not None else vs[0].device)

    @staticmethod
    def from_scipy_list(mats, device=None) -> "BatchedCSR":
        """Build from a list of scipy.sparse matrices (harness convenience)."""
        import numpy as np

        T = len(mats)
        N = mats[0].shape[0]
        ks, rs, cs, vs = [], [], [], []
        for k, m in enumerate(mats):
            m = m.tocoo()
            ks.append(np.full(m.nnz, k, dtype=np.int64))
            rs.append(m.row.astype(np.int64))
            cs.append(m.col.astype(np.int64))
            vs.append(m.data)
        return BatchedCSR.from_coo(torch.from_numpy(np.concatenate(ks)), torch.from_numpy(np.concatenate(rs)),
                                   torch.from_numpy(np.concatenate(cs)), torch.from_numpy(np.concatenate(vs)),
                                   T, N, device=device)

    # ------------------------------------------------------------------ views / conversions
    def to(self, device) -> "BatchedCSR":
        device = torch.device(device)
        if device == self.device:
            return self
        out = BatchedCSR(self.rowptr.to(device), self.col.to(device), self.val.to(device), self.T, self.N)
        return out

    def row_ids(self) -> torch.Tensor:
        """Global row index (k*N + i) of every stored entry."""
        counts = self.rowptr[1:] - self.rowptr[:-1]
        return torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), counts)

    def transpose(self) -> "BatchedCSR":
        """Per-slice transpose (Â_kᵀ for every k), cached.  Used by the backward SpMM.
        Built slice by slice (one stable sort of nnz_k keys each) so the peak scratch is a few
        times one slice, not the whole tensor; inside a transposed row the entries keep the
        order of the original rows, so the backward sums are reproducible too."""
        if self._t is None and self.device.type == "cuda":
            from .adjacency import csr_transpose  # native sort of the transposed keys
            t = csr_transpose(self)
            t._t = self
            self._t = t
        if self._t is None:
            N, T, dev = self.N, self.T, self.device
            counts = torch.zeros(T * N, dtype=torch.int64, device=dev)
            col_t = torch.empty_like(self.col)
            val_t = torch.empty_like(self.val)
            bounds = self.rowptr[::N].tolist()  # nnz offset of every slice (T+1 host ints)
            for k in range(T):
                a, b = bounds[k], bounds[k + 1]
                if a == b:
                    continue
                rp = self.rowptr[k * N:(k + 1) * N + 1] - a
                rows = torch.repeat_interleave(torch.arange(N, device=dev, dtype=torch.int32),
                                               rp[1:] - rp[:-1])
                c = self.col[a:b]
                order = torch.sort(c, stable=True).indices
                counts[k * N:(k + 1) * N] = torch.bincount(c, minlength=N)
                col_t[a:b] = rows[order]
                val_t[a:b] = self.val[a:b][order]
                del rows, order, c
            rowptr_t = torch.zeros(T * N + 1, dtype=torch.int64, device=dev)
            torch.cumsum(counts, 0, out=rowptr_t[1:])
            t = BatchedCSR(rowptr_t, col_t, val_t, T, N)
            t._t = self
            self._t = t
        return self._t

    def max_row_length(self) -> int:
        """Stored entries of the longest row (cached; one host sync per CSR)."""
        if "max_row" not in self._blocks:
            self._blocks["max_row"] = int((self.rowptr[1:] - self.rowptr[:-1]).max()) if self.n_rows else 0
        return self._blocks["max_row"]

    def is_skewed(self) -> bool:
        """Hub rows: the longest row holds more than 64 entries and more than 8 times the mean."""
        m = self.max_row_length()
        return m > 64 and m > 8 * self.avg_nnz_per_row

    def row_blocks(self, max_rows: int = 256, max_entries: int = 1024) -> Optional[torch.Tensor]:
        """Partition of the rows for the entry-major layer kernels (tmgcn_layer12_fwd/bwd_f32's `row_blocks`), cached:
        blocks of `max_rows` consecutive rows, and every such block that holds more than `max_entries` stored entries
        cut further — at row boundaries, into ceil(entries / max_entries) parts of about equal entry counts — so that
        no block holds more than max_entries + its longest row.  int64 [n_blocks, 2] pairs (first row, rows) on the
        adjacency's device, the blocks with the most entries FIRST (the forward starts them in this order: heavy blocks
        started last would leave the chip idle behind them; equal blocks keep their ascending order); None when no block
        needs cutting (balanced data: the kernels' own 256-row blocks).  The backward's form: row_block_runs."""
        key = (int(max_rows), int(max_entries))
        if key not in self._blocks:
            R, dev = self.n_rows, self.device
            starts = torch.arange(0, R, max_rows, device=dev, dtype=torch.int64)
            ends = torch.clamp(starts + max_rows, max=R)
            lo, hi = self.rowptr[starts], self.rowptr[ends]
            parts = torch.clamp((hi - lo + max_entries - 1) // max_entries, min=1)
            if R == 0 or int(parts.max()) <= 1:
                self._blocks[key] = None
            else:
                owner = torch.repeat_interleave(torch.arange(starts.numel(), device=dev), parts)
                first_part = torch.cumsum(parts, 0) - parts
                i = torch.arange(owner.numel(), device=dev) - first_part[owner]              # part index inside its block
                target = lo[owner] + (hi[owner] - lo[owner]) * i // parts[owner]             # entry position the part starts at
                cut = torch.searchsorted(self.rowptr, target.contiguous(), right=False)      # first row starting at or after it
                cut = torch.minimum(torch.maximum(cut, starts[owner]), ends[owner])
                cut = torch.where(i == 0, starts[owner], cut)
                blk = torch.unique(torch.cat([cut, torch.tensor([R], device=dev, dtype=torch.int64)]))   # sorted; empty parts vanish
                self._blocks[key] = self._heaviest_first(blk)
        return self._blocks[key]

    RUNS = 16          # include/tmgcn.h: TMGCN_L12_RUNS

    def _heaviest_first(self, bounds: torch.Tensor) -> torch.Tensor:
        """Ascending block boundaries [n + 1] -> the (first row, rows) pairs [n, 2], most entries first (stable): the order
        the forward starts its blocks in."""
        first, rows = bounds[:-1], bounds[1:] - bounds[:-1]
        entries = self.rowptr[bounds[1:]] - self.rowptr[first]
        order = torch.sort(entries, descending=True, stable=True).indices
        return torch.stack((first[order], rows[order]), dim=1).contiguous()

    def row_block_runs(self, max_rows: int = 256, max_entries: int = 1024) -> torch.Tensor:
        """The backward's form of the partition (tmgcn_layer12_bwd_f32's row_blocks), cached: the same row blocks as
        row_blocks() (the trivial 256-row ones when nothing needs cutting), listed for a kernel that reads the list as
        min(RUNS, n) runs of equal length — run g = list entries [n·g / runs, n·(g + 1) / runs), worked on by the thread blocks
        of one XCD: the ascending order cut into those runs (neighbouring row blocks gather the same rows: the XCD's L2 then
        fetches them once), and inside a run the row blocks with the most entries first (stable)."""
        key = ("runs", int(max_rows), int(max_entries))
        if key not in self._blocks:
            pairs = self.row_blocks(max_rows, max_entries)
            if pairs is None:
                pairs = self.trivial_row_blocks(max_rows)
            pairs = pairs[torch.argsort(pairs[:, 0])]
            n = int(pairs.shape[0])
            entries = self.rowptr[pairs[:, 0] + pairs[:, 1]] - self.rowptr[pairs[:, 0]]
            runs = min(self.RUNS, n)
            ends = (torch.arange(1, runs + 1, device=pairs.device) * n) // runs                # run g ends at n·(g+1)/runs
            run = torch.searchsorted(ends, torch.arange(n, device=pairs.device), right=True)
            order = torch.sort(entries, descending=True, stable=True).indices
            order = order[torch.sort(run[order], stable=True).indices]                          # by (run, entries descending)
            self._blocks[key] = pairs[order].contiguous()
        return self._blocks[key]

    def trivial_row_blocks(self, max_rows: int = 256) -> torch.Tensor:
        """The kernels' own blocks of `max_rows` consecutive rows as an explicit partition (heaviest first)."""
        key = ("trivial", int(max_rows))
        if key not in self._blocks:
            b = torch.arange(0, self.n_rows + max_rows, max_rows, device=self.device, dtype=torch.int64).clamp_(max=self.n_rows)
            self._blocks[key] = self._heaviest_first(torch.unique(b))
        return self._blocks[key]

    GIANT_ROW, GIANT_CHUNK = _header_constants("TMGCN_GIANT_ROW", "TMGCN_GIANT_CHUNK")   # include/tmgcn.h: what the kernels compiled with

    def giant_plan(self):
        """The giant-row plan of the SpMM launchers (include/tmgcn.h "Giant rows"), cached: rows of more than GIANT_ROW
        stored entries are summed chunk by chunk (GIANT_CHUNK entries, one block each) by a small launch in front of the main
        kernel instead of on the four waves of one block.  Returns (rows int64 [n], chunks int32 [n + 1 + m]) on the
        adjacency's device — ascending row indices; first chunk of every giant row, then every chunk's giant row — or
        (None, None) when no row is that long (every adjacency in the reference's scope: one host sync per CSR to find out)."""
        if "giant" not in self._blocks:
            cnt = self.rowptr[1:] - self.rowptr[:-1]
            plan = (None, None)
            if self.n_rows and self.device.type == "cuda" and _GIANT_PLAN and int(cnt.max()) > self.GIANT_ROW:
                rows = torch.nonzero(cnt > self.GIANT_ROW).reshape(-1)
                n_chunks = (cnt[rows] + self.GIANT_CHUNK - 1) // self.GIANT_CHUNK
                chunk_ptr = torch.zeros(rows.numel() + 1, dtype=torch.int64, device=self.device)
                torch.cumsum(n_chunks, 0, out=chunk_ptr[1:])
                chunk_giant = torch.repeat_interleave(torch.arange(rows.numel(), device=self.device), n_chunks)
                plan = (rows.contiguous(), torch.cat([chunk_ptr, chunk_giant]).to(torch.int32).contiguous())
            self._blocks["giant"] = plan
        return self._blocks["giant"]

    def slices(self, k0: int, k1: int) -> "BatchedCSR":
        """Slices [k0, k1) as their own batched CSR (the shard one rank owns)."""
        assert 0 <= k0 <= k1 <= self.T
        lo, hi = k0 * self.N, k1 * self.N
        base = int(self.rowptr[lo])
        end = int(self.rowptr[hi])
        return BatchedCSR((self.rowptr[lo:hi + 1] - base).clone(), self.col[base:end].clone(),
                          self.val[base:end].clone(), k1 - k0, self.N)

    def slice_views(self) -> List["BatchedCSR"]:
        """One single-slice BatchedCSR per slice, sharing this object's storage (no copies): the
        rowptr window of slice k keeps its global offsets into the full col/val arrays.  Used to
        launch slice by slice so that compute on slice k overlaps the exchange of slice k+1."""
        out = []
        bounds = self.rowptr[::self.N].tolist()
        for k in range(self.T):
            v = BatchedCSR.__new__(BatchedCSR)
            v.rowptr = self.rowptr[k * self.N:(k + 1) * self.N + 1]
            v.col, v.val = self.col, self.val
            v.T, v.N = 1, self.N
            v.nnz = bounds[k + 1] - bounds[k]
            v._t = None
            v._blocks = {}
            out.append(v)
        return out

    def to_coo_list(self, dtype=torch.float64) -> List[torch.Tensor]:
        """Back to the reference's list-of-COO form (CPU), for the oracle / CPU baseline."""
        if int(self.rowptr[0]) != 0 or int(self.rowptr[-1]) != self.col.numel():
            raise RuntimeError("to_coo_list: call on the owning BatchedCSR, not on a slice view")
        rid = self.row_ids().cpu()
        col = self.col.cpu().to(torch.int64)
        val = self.val.cpu().to(dtype)
        rp = self.rowptr.cpu()
        out = []
        for k in range(self.T):
            a, b = int(rp[k * self.N]), int(rp[(k + 1) * self.N])
            idx = torch.stack([rid[a:b] - k * self.N, col[a:b]])
            out.append(torch.sparse_coo_tensor(idx, val[a:b], (self.N, self.N)))
        return out

    def to_dense(self) -> torch.Tensor:
        """[T, N, N] dense tensor (tests only; small sizes)."""
        rid = self.row_ids()
        k = rid // self.N
        i = rid - k * self.N
        d = torch.zeros(self.T, self.N, self.N, dtype=self.val.dtype, device=self.device)
        d.index_put_((k, i, self.col.to(torch.int64)), self.val, accumulate=True)
        return d

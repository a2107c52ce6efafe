"""Seeded synthetic inputs for the TM-GCN layer: the configurations of BASELINE.json / SURVEY §8d.

There are no datasets in the container and no network, so the harness builds dynamic graphs of
the reference scripts' shapes with the reference's preprocessing steps restated on scipy.sparse
(vectorised — the reference's versions are per-slice / per-nnz Python loops):

    symmetrise          read_data.m:172-180, read_data.py:88-111
    edge-life window    read_data.m:183-187, read_data.py:116-125
    add I, D^-1/2 C D^-1/2   read_data.m:190-199, read_data.py:130-169
    band M              read_data.m:116-127 ("matlab": weight 1/d), read_data.py:55-62
                        ("python": ones, row-normalised), SBM_our.py:88-96 ("sbm" = matlab)
    M-product of Â      read_data.m:207-209, read_data.py:204-223, SBM_our.py:78-86
    node features       ehf.create_node_features:597-609 (in/out degree -> F0 = 2)

S4 (T=128, N=2M, deg 32, F=128) is generated directly on the device, slice by slice.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional

import numpy as np
import scipy.sparse as sp
import torch

from .csr import BatchedCSR


def band_M(T: int, no_diag: int = 20, kind: str = "matlab") -> np.ndarray:
    """Lower-triangular band mixing matrix, fp64 [T,T]."""
    M = np.zeros((T, T))
    for d in range(min(no_diag, T)):
        w = 1.0 if kind == "python" else 1.0 / (d + 1)
        M[np.arange(d, T), np.arange(0, T - d)] = w
    if kind == "python":
        M = M / M.sum(axis=1, keepdims=True)
    return M


def random_slices(T: int, N: int, edges_per_slice: int, rng: np.random.Generator, zipf: float = 0.0) -> List[sp.csr_matrix]:
    """T raw directed adjacency slices with unit weights (duplicates merged).  zipf > 0: source nodes drawn with
    probability ∝ rank^-zipf (the same hubs in every slice), as interaction graphs have them; 0: uniform."""
    out = []
    p = None
    if zipf > 0:
        p = np.arange(1, N + 1, dtype=np.float64) ** (-zipf)
        p /= p.sum()
    for _ in range(T):
        r = rng.choice(N, edges_per_slice, p=p) if p is not None else rng.integers(0, N, edges_per_slice)
        c = rng.integers(0, N, edges_per_slice)
        a = sp.coo_matrix((np.ones(edges_per_slice), (r, c)), shape=(N, N)).tocsr()
        a.data[:] = 1.0
        out.append(a)
    return out


def symmetrise(A: List[sp.spmatrix]) -> List[sp.csr_matrix]:
    return [((a + a.T) / 2).tocsr() for a in A]


def edge_life(A: List[sp.spmatrix], window: int = 10) -> List[sp.csr_matrix]:
    """B[t] = sum of A[t-window+1 .. t]."""
    out = []
    for t in range(len(A)):
        acc = A[t].copy()
        for s in range(max(0, t - window + 1), t):
            acc = acc + A[s]
        out.append(acc.tocsr())
    return out


def normalise(B: List[sp.spmatrix]) -> List[sp.csr_matrix]:
    """C = D^-1/2 (B + I) D^-1/2 with D = row sums of B + I."""
    out = []
    for b in B:
        n = b.shape[0]
        c = (b + sp.identity(n, format="csr")).tocsr()
        d = 1.0 / np.sqrt(np.asarray(c.sum(axis=1)).ravel())
        out.append((sp.diags(d) @ c @ sp.diags(d)).tocsr())
    return out


def m_product(C: List[sp.spmatrix], M: np.ndarray) -> List[sp.csr_matrix]:
    """Mode-1 product of the sparse adjacency tensor: Ct[k] = sum_j M[k,j] C[j]."""
    T = len(C)
    out = []
    for k in range(T):
        acc = None
        for j in np.nonzero(M[k])[0]:
            term = C[j] * M[k, j]
            acc = term if acc is None else acc + term
        if acc is None:
            acc = sp.csr_matrix(C[0].shape)
        out.append(acc.tocsr())
    return out


def node_features(A: List[sp.spmatrix]) -> np.ndarray:
    """X[t,:,0] = column sums, X[t,:,1] = row sums of the unweighted adjacency (fp64 [T,N,2])."""
    T, N = len(A), A[0].shape[0]
    X = np.zeros((T, N, 2))
    for t, a in enumerate(A):
        u = a.copy()
        u.data[:] = 1.0
        X[t, :, 0] = np.asarray(u.sum(axis=0)).ravel()
        X[t, :, 1] = np.asarray(u.sum(axis=1)).ravel()
    return X


def to_coo_list(mats: List[sp.spmatrix], dtype=torch.float64) -> List[torch.Tensor]:
    """scipy slices -> the reference's list of torch sparse COO matrices (explicit N×N size)."""
    out = []
    for m in mats:
        m = m.tocoo()
        idx = torch.from_numpy(np.stack([m.row, m.col]).astype(np.int64))
        out.append(torch.sparse_coo_tensor(idx, torch.from_numpy(m.data).to(dtype), m.shape))
    return out


@dataclass
class DynamicGraph:
    T: int
    N: int
    A_raw: List[sp.csr_matrix]      # unweighted raw slices
    C: List[sp.csr_matrix]          # normalised adjacency Ĉ (KWGCN input)
    Ct: List[sp.csr_matrix]         # M-transformed adjacency Â (TM-GCN input)
    M: np.ndarray                   # [T,T] fp64
    X: np.ndarray                   # [T,N,F0] fp64
    edges: np.ndarray               # [3,E] int64 (slice, src, dst)
    labels: np.ndarray              # [E] int64

    def At_list(self):
        return to_coo_list(self.Ct)

    def A_list(self):
        return to_coo_list(self.C)


def dynamic_graph(T: int, N: int, edges_per_slice: int, seed: int = 0, window: int = 10,
                  no_diag: int = 20, m_kind: str = "matlab", F0: Optional[int] = None,
                  neg_per_pos: int = 0, zipf: float = 0.0) -> DynamicGraph:
    """Reference-shaped dynamic graph: raw random slices -> Ĉ -> Â = M ×₁ Ĉ, features, labelled edges.

    neg_per_pos = 0 : edge classification (labels random in {0,1}, the Bitcoin scripts' shape)
    neg_per_pos > 0 : link prediction (positives label 0, ``neg_per_pos`` sampled non-edges label 1;
                      ehf.augment_edges:500-526 with a seeded vectorised sampler)
    zipf > 0        : hub source nodes (random_slices): skewed rows in Â and skewed endpoints of the labelled edges
    """
    rng = np.random.default_rng(seed)
    A = random_slices(T, N, edges_per_slice, rng, zipf)
    C = normalise(edge_life(symmetrise(A), window))
    M = band_M(T, no_diag, m_kind)
    Ct = m_product(C, M)
    X = node_features(A) if F0 is None else rng.standard_normal((T, N, F0))
    es, ls = [], []
    for t, a in enumerate(A):
        a = a.tocoo()
        pos = np.stack([np.full(a.nnz, t), a.row, a.col]).astype(np.int64)
        if neg_per_pos:
            n_neg = neg_per_pos * a.nnz
            r = rng.integers(0, N, n_neg)
            c = rng.integers(0, N, n_neg)
            keep = np.asarray(a.tocsr()[r, c]).ravel() == 0
            neg = np.stack([np.full(int(keep.sum()), t), r[keep], c[keep]]).astype(np.int64)
            es += [pos, neg]
            ls += [np.zeros(pos.shape[1], np.int64), np.ones(neg.shape[1], np.int64)]
        else:
            es.append(pos)
            ls.append(rng.integers(0, 2, pos.shape[1]).astype(np.int64))
    return DynamicGraph(T, N, A, C, Ct, M, X, np.concatenate(es, axis=1), np.concatenate(ls))


def sbm_dynamic_graph(T: int = 10, N: int = 500, p_in: float = 0.1, p_out: float = 0.01, migrate: int = 10,
                      F0: int = 16, no_diag: int = 20, seed: int = 0) -> DynamicGraph:
    """S0, the plumbing config of BASELINE.json: a 2-community stochastic block model in which
    `migrate` nodes move from community 1 to community 0 at every step (SBM_our.py:98-109 uses
    dynamicgem's get_community_diminish_series_v2 for this), RAW symmetric adjacency without
    normalisation or self loops (SBM_our.py:111-131), M = 1/(d+1) on `no_diag` diagonals
    (SBM_our.py:88-96), random-normal features, every edge labelled."""
    import networkx as nx

    rng = np.random.default_rng(seed)
    member = np.zeros(N, dtype=int)
    member[N // 2:] = 1
    A = []
    for t in range(T):
        if t:
            ones = np.nonzero(member == 1)[0]
            member[rng.choice(ones, size=min(migrate, len(ones)), replace=False)] = 0
        order = np.argsort(member, kind="stable")
        sizes = [int((member == 0).sum()), int((member == 1).sum())]
        g = nx.stochastic_block_model(sizes, [[p_in, p_out], [p_out, p_in]], seed=int(rng.integers(1 << 31)))
        a = nx.to_scipy_sparse_array(g, format="coo")
        A.append(sp.coo_matrix((np.ones(a.nnz), (order[a.row], order[a.col])), shape=(N, N)).tocsr())
    M = band_M(T, no_diag, "matlab")
    Ct = m_product(A, M)
    X = rng.standard_normal((T, N, F0))
    es = [np.stack([np.full(a.nnz, t), a.tocoo().row, a.tocoo().col]).astype(np.int64) for t, a in enumerate(A)]
    edges = np.concatenate(es, axis=1)
    labels = rng.integers(0, 2, edges.shape[1]).astype(np.int64)
    return DynamicGraph(T, N, A, A, Ct, M, X, edges, labels)


# Named stand-ins for the BASELINE configs (N and edges/slice are assumptions,
# T / feature / hidden sizes are the scripts').  S0 (SBM plumbing) is sbm_dynamic_graph().
CONFIGS = {
    "S1": dict(T=95, N=6000, edges_per_slice=250),                        # Bitcoin-OTC-shaped
    "S2": dict(T=65, N=3800, edges_per_slice=2500, neg_per_pos=19),       # Reddit-LP-shaped
    "S3": dict(T=150, N=1000, edges_per_slice=500),                       # AMLSim-shaped
    # the Reddit-LP shape with HUB source nodes (one node is the source of a third of every slice's edges): skewed rows in Â
    # (up to ~3 800 entries, mean 12) and in the inverted index of the labelled edges — not a BASELINE config; parity:
    # tests/test_gpu_configs.py::test_S2_shape_with_hub_nodes, epoch times: tools/epoch_bench.py S2z S2z2
    "S2z": dict(T=65, N=3800, edges_per_slice=1500, neg_per_pos=4, zipf=1.5),
    "S2z2": dict(T=65, N=3800, edges_per_slice=1500, neg_per_pos=4, zipf=1.5),
    # the two model-level probes of BASELINE.md §2 (uniform random graphs, wide features)
    "P128": dict(T=32, N=20000, edges_per_slice=16000, window=1, F0=128), # F = 128 -> 128 -> 128
}


# ---------------------------------------------------------------------------------------
# S4: large synthetic layer input generated on the device
# ---------------------------------------------------------------------------------------
def device_er_csr(T: int, N: int, deg: int, device, first_slice: int = 0) -> BatchedCSR:
    """T slices of a directed Erdős–Rényi-style graph: `deg` uniformly random out-neighbours per
    row plus the self loop, values 1/(deg+1) (row-normalised), columns sorted inside each row.
    Slice k is seeded with first_slice + k, so any rank can generate exactly its own slices."""
    per = deg + 1
    nnz_slice = N * per
    col = torch.empty(T * nnz_slice, dtype=torch.int32, device=device)
    g = torch.Generator(device=device)
    for k in range(T):
        g.manual_seed(1000003 * (first_slice + k) + 17)
        c = torch.randint(0, N, (N, per), generator=g, device=device, dtype=torch.int32)
        c[:, 0] = torch.arange(N, device=device, dtype=torch.int32)
        c = torch.sort(c, dim=1).values
        col[k * nnz_slice:(k + 1) * nnz_slice] = c.reshape(-1)
        del c
    val = torch.full((T * nnz_slice,), 1.0 / per, dtype=torch.float32, device=device)
    rowptr = torch.arange(0, T * N + 1, dtype=torch.int64, device=device) * per
    return BatchedCSR(rowptr, col, val, T, N)


def device_features(T: int, N: int, F: int, device, first_slice: int = 0) -> torch.Tensor:
    """X ~ U(0,1) fp32 [T,N,F], slice k seeded with first_slice + k."""
    X = torch.empty(T, N, F, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for k in range(T):
        g.manual_seed(7919 * (first_slice + k) + 3)
        X[k].uniform_(0.0, 1.0, generator=g)
    return X


def device_normal(T: int, N: int, F: int, device, first_slice: int = 0, salt: int = 99) -> torch.Tensor:
    """~ N(0,1) fp32 [T,N,F], slice k seeded with (salt, first_slice + k): any rank can regenerate any
    slice of it (bench.py's upstream gradient dY, which its verify leg rebuilds slice by slice)."""
    Y = torch.empty(T, N, F, dtype=torch.float32, device=device)
    g = torch.Generator(device=device)
    for k in range(T):
        g.manual_seed(104729 * (first_slice + k) + 31 * salt + 5)
        Y[k].normal_(0.0, 1.0, generator=g)
    return Y


# ---------------------------------------------------------------------------------------
# S4 with skewed degrees: the shape of the reference's REAL operand
# ---------------------------------------------------------------------------------------
# The reference's adjacency is the M-product of up to 20 symmetrised, edge-life-windowed slices of a
# real graph (read_data.py:88-127, 204-223): its one shipped data set (chess) has half its rows at one
# entry and 13 % of the rows holding 59 % of the entries.  device_er_csr gives every row exactly deg+1
# entries — the kernels' best case.  This generator keeps N, the mean row length and the uniform random
# columns of S4 and draws the row lengths from a capped Zipf law instead.
_POWERLAW_CACHE: dict = {}


def powerlaw_degrees(N: int, mean_deg: float, alpha: float = 0.8, hub_cap: int = 100_000) -> np.ndarray:
    """Out-degrees (self loop not counted) of N rows, descending: floor(min(c · rank^-alpha, cap)) with c
    solved by bisection so that they sum to about mean_deg · N; cap = min(hub_cap, N).  With the defaults
    at N = 2 M / mean 32: a handful of rows at the 100 000 cap, ~10 % of the rows holding ~60 % of the
    entries, the shortest rows at 7."""
    key = (N, float(mean_deg), float(alpha), int(hub_cap))
    if key not in _POWERLAW_CACHE:
        cap = float(min(hub_cap, N))
        w = np.arange(1, N + 1, dtype=np.float64) ** (-alpha)
        target = mean_deg * N
        lo, hi = 0.0, max(1.0, target / w[-1])
        for _ in range(80):
            c = 0.5 * (lo + hi)
            if np.minimum(c * w, cap).sum() < target:
                lo = c
            else:
                hi = c
        _POWERLAW_CACHE[key] = np.floor(np.minimum(hi * w, cap)).astype(np.int64)
    return _POWERLAW_CACHE[key]


def device_powerlaw_csr(T: int, N: int, deg: int, device, first_slice: int = 0, alpha: float = 0.8,
                        hub_cap: int = 100_000, symmetric: bool = False) -> BatchedCSR:
    """T slices with the mean row length of device_er_csr (deg random neighbours + the self loop) but
    capped-Zipf row lengths (powerlaw_degrees), the long rows at random positions of every slice, columns
    uniform, duplicates kept, values 1/row length, columns sorted inside each row.  Slice k is seeded
    with first_slice + k.
    symmetric=False  skewed OUT-degree only: the transposed operand (backward) has Poisson row lengths
    symmetric=True   each random pair stored both ways, as the reference's symmetrised slices are
                     (read_data.py:88-111): hub rows are hub columns, forward and backward both skewed"""
    base = powerlaw_degrees(N, deg / 2 if symmetric else deg, alpha, hub_cap)
    base_d = torch.from_numpy(base).to(device)
    g = torch.Generator(device=device)
    ar = torch.arange(N, device=device, dtype=torch.int64)
    rowptrs, cols, vals = [], [], []
    off = 0
    for k in range(T):
        g.manual_seed(2000003 * (first_slice + k) + 29)
        d = base_d[torch.randperm(N, generator=g, device=device)]
        r = torch.repeat_interleave(ar, d)
        c = torch.randint(0, N, (int(r.numel()),), generator=g, device=device, dtype=torch.int64)
        if symmetric:
            r, c = torch.cat([r, c, ar]), torch.cat([c, r, ar])
        else:
            r, c = torch.cat([r, ar]), torch.cat([c, ar])
        key = torch.sort(r * N + c).values
        del r, c
        r = key // N
        cols.append((key - r * N).to(torch.int32))
        cnt = torch.bincount(r, minlength=N)
        del key, r
        vals.append(torch.repeat_interleave(1.0 / cnt.to(torch.float32), cnt))
        rp = torch.cumsum(cnt, 0) + off
        rowptrs.append(rp)
        off = int(rp[-1])
    rowptr = torch.cat([torch.zeros(1, dtype=torch.int64, device=device)] + rowptrs)
    return BatchedCSR(rowptr, torch.cat(cols), torch.cat(vals), T, N)


# ---------------------------------------------------------------------------------------
# S4 on the reference's REAL operand structure: a shipped adjacency replicated on the block diagonal
# ---------------------------------------------------------------------------------------
# The generators above draw columns uniformly and give no row fewer than 7 entries.  The one operand the
# reference ships — the M-product of its symmetrised, windowed chess slices (read_data.py:116-127, 204-223;
# N = 7 301, T = 80) — has 3.97 entries per row on average, half of its rows holding the self loop only, and
# columns that stay inside communities.  `tile_block_diagonal` scales such an adjacency to bench size WITHOUT
# changing either property: `reps` copies of every chosen slice on the block diagonal (a REPLICATION, not a
# larger real graph: copy q's rows reference only copy q's columns).
def tile_block_diagonal(A: BatchedCSR, reps: int, slices: Optional[List[int]] = None) -> BatchedCSR:
    """Slices `slices` (default: all) of A, each as `reps` copies of itself on the block diagonal:
    a BatchedCSR with T = len(slices), N = reps · A.N; row q·A.N + i of new slice s is row i of A's slice
    slices[s] with its columns shifted by q·A.N.  Row lengths, values and the order inside a row are kept."""
    ks = list(range(A.T)) if slices is None else [int(k) for k in slices]
    N0, dev = A.N, A.device
    bounds = A.rowptr[::N0].tolist()
    shift = (torch.arange(reps, device=dev, dtype=torch.int32) * N0)[:, None]
    cnt = A.rowptr[1:] - A.rowptr[:-1]
    cols, vals, cnts = [], [], []
    for k in ks:
        a, b = bounds[k], bounds[k + 1]
        cols.append((A.col[a:b][None, :] + shift).reshape(-1))
        vals.append(A.val[a:b].repeat(reps))
        cnts.append(cnt[k * N0:(k + 1) * N0].repeat(reps))
    rowptr = torch.zeros(len(ks) * reps * N0 + 1, dtype=torch.int64, device=dev)
    torch.cumsum(torch.cat(cnts), 0, out=rowptr[1:])
    return BatchedCSR(rowptr, torch.cat(cols), torch.cat(vals), len(ks), reps * N0)


_REAL_OPERAND: dict = {}


def chess_operand(device, fixture: Optional[str] = None) -> BatchedCSR:
    """The reference's own operand: Ât of its chess data (experiment_chess_our.py's 80 training slices of 7 301
    players; read_data.py's symmetrise / edge-life 10 / normalise / M-product with its 20-diagonal band), built
    by the device adjacency pipeline from the raw edge list of fixture G10 (tests/golden/g10_chess_full.npz: the
    reference's data/chess/out.chess.csv as (slice, i, j) rows).  Cached per device."""
    import os
    key = str(device)
    if key not in _REAL_OPERAND:
        from . import adjacency
        if fixture is None:
            fixture = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "g10_chess_full.npz")
        d = np.load(fixture)
        TT, N, T = int(d["TT"]), int(d["N"]), int(d["S_train"])
        k, i, j = (d[n].astype(np.int64) for n in ("raw_k", "raw_i", "raw_j"))
        Chat, _ = adjacency.build_adjacency(k, i, j, np.ones(len(k), np.float32), TT, N, M=None, window=10, device=device)
        _REAL_OPERAND[key] = adjacency.m_product_csr(Chat.slices(0, T), d["M"])
    return _REAL_OPERAND[key]


def device_chess_tiled_csr(T: int, N: int, device, first_slice: int = 0) -> BatchedCSR:
    """T slices of the chess operand tiled to about N nodes (reps = round(N / 7 301) copies on the block
    diagonal).  Global slice g of the tiled tensor is chess slice (5·g + 4) mod 80 — every fifth slice, so that
    16 slices span the sparse early months and the dense late ones alike; any rank can build its own slices."""
    A = chess_operand(device)
    reps = max(1, int(round(N / A.N)))
    return tile_block_diagonal(A, reps, [(5 * (first_slice + s) + 4) % A.T for s in range(T)])


def device_csr(kind: str, T: int, N: int, deg: int, device, first_slice: int = 0) -> BatchedCSR:
    """The S4 adjacency by name: "er" (the headline), "powerlaw", "powerlaw_sym", "chess_tiled"
    (the reference's real operand replicated on the block diagonal; `deg` unused, N rounded to a multiple of 7 301)."""
    if kind == "er":
        return device_er_csr(T, N, deg, device, first_slice)
    if kind in ("powerlaw", "powerlaw_sym"):
        return device_powerlaw_csr(T, N, deg, device, first_slice, symmetric=kind == "powerlaw_sym")
    if kind == "chess_tiled":
        return device_chess_tiled_csr(T, N, device, first_slice)
    raise ValueError(f"unknown graph kind {kind!r}")

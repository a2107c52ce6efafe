"""The data side of the reference's ``embedding_help_functions``: what its experiment scripts call
before and around the layer ("callers and data formats either side of the path").

    load_data             ehf:542-593   MATLAB ``saved_content_*.mat`` -> A, A_labels, slice lists, N (, M)
    create_node_features  ehf:597-609   in/out-degree features X [T,N,2] and their train/val/test blocks
    augment_edges         ehf:500-526   negative edges for link prediction (beta per positive)
    split_data            ehf:612-655   train/val/test edge sets, targets, "previous slice" edge sets
    print_f1              ehf:658-666   the scripts' result lines
    compute_At            ehf:28-153    normalise + M-product of a sparse [T,N,N] tensor, pickled cache

Same names, argument meaning and return tuples as the reference, so a script keeps its data
section unchanged.  Everything is index arithmetic on whole tensors: the reference's per-slice
boolean masks (O(T·nnz), ehf:562-572), per-candidate tensor compare (ehf:513) and per-nnz Python
loops (ehf:51-58, 114-128) are replaced by one sort / searchsorted / hash lookup each.  Sparse
inputs and outputs stay torch COO tensors on the host, as the scripts expect; the layer
constructors (layers.py) move them to the device and build the batched CSR there.
"""
from __future__ import annotations

import os
import pickle
import random
from typing import List, Optional

import numpy as np
import scipy.io as sio
import torch


# ---------------------------------------------------------------------------------------
# .mat ingest
# ---------------------------------------------------------------------------------------
def _subs(a) -> torch.Tensor:
    """MATLAB subscripts [nnz, d], 1-based, any numeric dtype -> int64 [d, nnz], 0-based."""
    a = np.asarray(a)
    if a.ndim != 2:
        raise RuntimeError(f"subscript array must be 2-D [nnz, d], got shape {a.shape}")
    return torch.from_numpy(np.ascontiguousarray(a.T).astype(np.int64) - 1)


def _vals(a) -> torch.Tensor:
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a)).reshape(-1))


def _coo3(subs, vals, size) -> torch.Tensor:
    return torch.sparse_coo_tensor(_subs(subs), _vals(vals), size).coalesce()


def _slice_list(A3: torch.Tensor, first: int, count: int) -> List[torch.Tensor]:
    """Frontal slices first..first+count-1 of a coalesced sparse [T,N,N] tensor as 2-D COO
    matrices.  Coalesced indices are sorted by (slice, row, col), so every slice is one
    contiguous range found by a single searchsorted."""
    idx, val = A3.indices(), A3.values()
    N = A3.shape[1]
    bounds = torch.searchsorted(idx[0].contiguous(), torch.arange(first, first + count + 1)).tolist()
    return [torch.sparse_coo_tensor(idx[1:3, b0:b1], val[b0:b1], (N, N), is_coalesced=True)
            for b0, b1 in zip(bounds[:-1], bounds[1:])]


def load_data(data_loc, mat_f_name, S_train, S_val, S_test, transformed):
    """ehf:542-593.  Returns ``A, A_labels, Ct_train, Ct_val, Ct_test, N, M`` (transformed) or
    ``A, A_labels, C_train, C_val, C_test, N``.  A_labels: sparse [T,N,N] of the edge labels, A: its
    pattern with fp32 unit values; the three lists hold 2-D sparse matrices, one per slice.  Values
    keep the file's precision (fp64 from MATLAB — the reference's ``t.sparse.FloatTensor`` legacy
    constructor does not cast either).  As in the reference the transformed val/test tensors
    are blocks of S_train slices (ehf:550, 566, 570) and the untransformed lists are the
    consecutive S_train / S_val / S_test slices of C (ehf:581-591)."""
    content = sio.loadmat(str(data_loc) + mat_f_name)        # plain concatenation, as the scripts rely on
    lab = np.asarray(content["A_labels_subs"])
    T = int(lab[:, 0].max())
    N = int(max(lab[:, 1].max(), lab[:, 2].max()))
    A_labels = _coo3(lab, content["A_labels_vals"], (T, N, N))
    A = torch.sparse_coo_tensor(A_labels.indices(), torch.ones(A_labels._nnz()), (T, N, N), is_coalesced=True)
    if transformed:
        lists = [_slice_list(_coo3(content[f"Ct_{p}_subs"], content[f"Ct_{p}_vals"], (S_train, N, N)), 0, S_train)
                 for p in ("train", "val", "test")]
        M = torch.tensor(np.asarray(content["M"]), dtype=torch.float64)
        return A, A_labels, lists[0], lists[1], lists[2], N, M
    C = _coo3(content["C_subs"], content["C_vals"], (T, N, N))
    return (A, A_labels, _slice_list(C, 0, S_train), _slice_list(C, S_train, S_val),
            _slice_list(C, S_train + S_val, S_test), N)


# ---------------------------------------------------------------------------------------
# node features
# ---------------------------------------------------------------------------------------
def create_node_features(A: torch.Tensor, S_train, S_val, S_test, same_block_size):
    """ehf:597-609.  X[t,n,0] = Σ_i A[t,i,n] (column sums), X[t,n,1] = Σ_j A[t,n,j] (row sums),
    accumulated in fp32 like ``t.sparse.sum`` on the fp32 A, returned as fp64 blocks.
    same_block_size=True (TM-GCN): val / test blocks are S_train slices long, shifted by S_val and
    S_val+S_test; False (baselines): consecutive S_train / S_val / rest."""
    A = A.coalesce()
    T, N = int(A.shape[0]), int(A.shape[1])
    idx, v = A.indices(), A.values().to(torch.float32)
    X = torch.zeros(T * N, 2, dtype=torch.float32, device=v.device)
    X[:, 0].index_add_(0, idx[0] * N + idx[2], v)
    X[:, 1].index_add_(0, idx[0] * N + idx[1], v)
    X = X.view(T, N, 2).double()
    if same_block_size:
        return X[0:S_train], X[S_val:S_train + S_val], X[S_val + S_test:]
    return X[0:S_train], X[S_train:S_train + S_val], X[S_train + S_val:]


# ---------------------------------------------------------------------------------------
# negative edges
# ---------------------------------------------------------------------------------------
def augment_edges(edges: torch.Tensor, N, beta1, beta2, cutoff, generator: Optional[torch.Generator] = None):
    """ehf:500-526.  For every slice j (0..max) add beta·(#edges of j) node pairs drawn uniformly
    from [0,N)² that are not edges of that slice (beta = beta1 for j < cutoff, else beta2).  A pair
    may repeat among the negatives and may be a self pair, as in the reference.  Returns
    ``edges_aug`` [3, E'] sorted by slice and ``labels`` (0 = real, 1 = added).

    generator=None   draws with Python's global ``random.randint`` in the reference's order (two
                     draws per candidate, slice by slice), so ``random.seed(s)`` gives exactly the
                     reference's edge set; membership is one hash lookup per candidate instead of a
                     tensor compare over the slice.
    generator=g      bulk sampler on g's device: candidates for all slices at once, rejection by
                     sorted-key lookup, repeated for the few rejected ones.  Same distribution,
                     different stream."""
    N = int(N)
    s = edges[0]
    n_slices = int(s.max()) + 1 if s.numel() else 0
    counts = torch.bincount(s, minlength=n_slices)
    beta = torch.where(torch.arange(n_slices, device=s.device) < cutoff, int(beta1), int(beta2))
    need = beta * counts
    if generator is None:
        order = torch.argsort(s, stable=True)
        pair = (edges[1][order] * N + edges[2][order]).tolist()
        start = (torch.cumsum(counts, 0) - counts).tolist()
        cnt, need_l = counts.tolist(), need.tolist()
        new = []
        for j in range(n_slices):
            real = set(pair[start[j]:start[j] + cnt[j]])
            added = 0
            while added < need_l[j]:
                a = random.randint(0, N - 1)
                b = random.randint(0, N - 1)
                if a * N + b not in real:
                    new.append((j, a, b))
                    added += 1
        new_t = torch.tensor(new, dtype=edges.dtype, device=edges.device).reshape(-1, 3).t()
    else:
        dev = generator.device
        real = torch.sort((edges[0].to(dev) * N + edges[1].to(dev)) * N + edges[2].to(dev)).values
        want = torch.repeat_interleave(torch.arange(n_slices, device=dev), need.to(dev))
        got = []
        while want.numel():
            ab = torch.randint(0, N, (2, want.numel()), generator=generator, device=dev)
            key = (want * N + ab[0]) * N + ab[1]
            pos = torch.searchsorted(real, key).clamp_(max=max(real.numel() - 1, 0))
            ok = real[pos] != key if real.numel() else torch.ones_like(key, dtype=torch.bool)
            got.append(torch.stack([want[ok], ab[0][ok], ab[1][ok]]))
            want = want[~ok]
        new_t = (torch.cat(got, dim=1) if got else torch.zeros(3, 0, dtype=torch.int64, device=dev)).to(edges.device, edges.dtype)
    edges_aug = torch.cat((edges, new_t), 1)
    # The reference sorts with torch's default (unstable) sort (ehf:519); the exact mode makes the
    # same call so that the order inside a slice is the reference's too on the same torch build.
    sort_id = torch.sort(edges_aug[0], stable=generator is not None).indices
    labels = torch.cat((torch.zeros(edges.shape[1], dtype=torch.long, device=edges.device),
                        torch.ones(new_t.shape[1], dtype=torch.long, device=edges.device)))
    return edges_aug[:, sort_id], labels[sort_id]


# ---------------------------------------------------------------------------------------
# train / val / test split
# ---------------------------------------------------------------------------------------
def _block(edges_aug, labels, mask, shift):
    """Edges of one block with the slice index rebased to the block, their targets, and the edges
    of slices ≥ 1 re-indexed to the previous slice (a model fed slices 0..S-2 predicts the edges
    of slices 1..S-1, ehf:617-618)."""
    e = edges_aug[:, mask].clone()
    e[0] -= shift
    later = e[:, e[0] != 0].clone()
    later[0] -= 1
    return e, labels[mask], later


def split_data(edges_aug: torch.Tensor, labels: torch.Tensor, S_train, S_val, S_test, same_block_size):
    """ehf:612-655.  same_block_size=True returns ``edges_train, target_train, e_train, edges_val,
    target_val, e_val, K_val, edges_test, target_test, e_test, K_test`` — val/test blocks are
    S_train slices long, starting at S_val and S_val+S_test, and K_* counts the edges in a
    block's last S_val (S_test) slices, the only ones evaluated (scripts use ``[-K_val:]``).
    same_block_size=False returns the nine tensors without K_*, blocks consecutive."""
    s = edges_aug[0]
    edges_train, target_train, e_train = _block(edges_aug, labels, s < S_train, 0)
    if same_block_size:
        v0, t0 = S_val, S_test + S_val
        m_val, m_test = (s >= v0) & (s < S_train + S_val), s >= t0
    else:
        v0, t0 = S_train, S_train + S_val
        m_val, m_test = (s >= v0) & (s < S_train + S_val), s >= t0
    edges_val, target_val, e_val = _block(edges_aug, labels, m_val, v0)
    edges_test, target_test, e_test = _block(edges_aug, labels, m_test, t0)
    if not same_block_size:
        return edges_train, target_train, e_train, edges_val, target_val, e_val, edges_test, target_test, e_test
    K_val = (edges_val[0] > S_train - S_val - 1).sum()
    K_test = (edges_test[0] > S_train - S_test - 1).sum()
    return (edges_train, target_train, e_train, edges_val, target_val, e_val, K_val,
            edges_test, target_test, e_test, K_test)


# ---------------------------------------------------------------------------------------
# result lines
# ---------------------------------------------------------------------------------------
def print_f1(precision_train, recall_train, f1_train, loss_train, precision_val, recall_val, f1_val, loss_val,
             precision_test, recall_test, f1_test, loss_test, alpha=None, tr=None, ep=None, is_final=False):
    """ehf:658-666 — three lines (train / val / test), prefixed ``FINAL:`` or ``alpha/Tr/Ep``."""
    head = "FINAL:" if is_final else "alpha/Tr/Ep %.2f/%d/%d." % (alpha, tr, ep)
    rows = (("Train", precision_train, recall_train, f1_train, loss_train),
            ("Val", precision_val, recall_val, f1_val, loss_val),
            ("Test", precision_test, recall_test, f1_test, loss_test))
    for i, (name, p, r, f, l) in enumerate(rows):
        print("%s %s precision/recall/f1 %.16f/%.16f/%.16f. %s loss %.16f.%s"
              % (head, name, p, r, f, name, l, "\n" if i == 2 else ""))


# ---------------------------------------------------------------------------------------
# compute_At (unused by the reference's scripts, kept for surface completeness)
# ---------------------------------------------------------------------------------------
def _normalise3(A: torch.Tensor, normalization_type: int) -> torch.Tensor:
    T, N = int(A.shape[0]), int(A.shape[1])
    if normalization_type == 1:
        A = (A + A.transpose(1, 2)) / 2                                            # ehf:63-64
    k = torch.arange(T).repeat_interleave(N)
    n = torch.arange(N).repeat(T)
    eye = torch.sparse_coo_tensor(torch.stack([k, n, n]), torch.ones(T * N, dtype=A.dtype), (T, N, N))
    A = (A + eye).coalesce()                                                       # ehf:39-45 / 65-71
    idx, v = A.indices(), A.values()
    if normalization_type == 0:
        col = torch.zeros(T * N, dtype=v.dtype).index_add_(0, idx[0] * N + idx[2], v)
        v = v / col[idx[0] * N + idx[2]]                                           # columns sum to 1 (ehf:46-59)
    else:
        row = torch.zeros(T * N, dtype=v.dtype).index_add_(0, idx[0] * N + idx[1], v)
        d = torch.sqrt(row)
        v = v / d[idx[0] * N + idx[2]] / d[idx[0] * N + idx[1]]                    # D^-1/2 (A+I) D^-1/2 (ehf:72-100)
    return torch.sparse_coo_tensor(idx, v, (T, N, N), is_coalesced=True)


def compute_At(fname_At_mat, fname_ij_matT, A: torch.Tensor, M: torch.Tensor, normalization_type=0):
    """ehf:28-153.  A: sparse [T,N,N]; normalise (0: add I, columns to unit 1-norm; 1: symmetrise,
    add I, D^-1/2·D^-1/2), then the M-product along the slice mode.  Every (i,j) tube with a
    non-zero in any slice becomes one column of ``At_mat`` [T, nnz_ij] = M · tube, and the result
    is the list of T sparse matrices sharing the index pattern ``ij_matT`` [2, nnz_ij] (ordered by
    (j, i), the reference's transpose(0,2) coalesce order).  Both arrays are pickled to / loaded
    from the two file names exactly as the reference does."""
    if not os.path.isfile(fname_At_mat) and not os.path.isfile(fname_ij_matT):
        T, N = int(A.shape[0]), int(A.shape[1])
        if normalization_type in (0, 1):
            A = _normalise3(A.coalesce(), normalization_type)
        A = A.coalesce()
        idx, v = A.indices(), A.values()
        tube, inv = torch.unique(idx[2] * N + idx[1], return_inverse=True)         # sorted by (j, i)
        dense = torch.zeros(tube.numel(), T, dtype=v.dtype)
        dense[inv, idx[0]] = v
        At_mat = (dense @ M.to(v.dtype).t()).t().contiguous()                      # ehf:113-130: vec += val·Mᵀ[k]
        ij_matT = torch.stack([tube % N, tube // N])
        with open(fname_At_mat, "wb") as f:
            pickle.dump(At_mat, f)
        with open(fname_ij_matT, "wb") as f:
            pickle.dump(ij_matT, f)
    else:
        with open(fname_ij_matT, "rb") as f:
            ij_matT = pickle.load(f)
        with open(fname_At_mat, "rb") as f:
            At_mat = pickle.load(f)
    return [torch.sparse_coo_tensor(ij_matT, vl) for vl in At_mat]

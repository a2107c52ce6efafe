"""Raw dynamic-graph edge list -> the ``saved_content_*.mat`` the experiment scripts load: the
reference's offline preprocessing (read_data.m, read_data.py top to bottom) with every tensor step
on the MI355X through the adjacency pipeline (adjacency.py / csrc/adjacency.hip).

    slice_by_time   read_data.m:105-111, 130-145   read_data.py:44-49, 65-83
    band_matrix     read_data.m:116-127            read_data.py:55-62
    read_data       read_data.m:129-209            read_data.py:64-227
    save_content    read_data.m:211-232  (the MATLAB layout: 1-based [nnz,3] subscripts, [nnz,1]
                    values — what ehf.load_data:551-577 expects.  read_data.py:248-270 writes
                    0-based [3,nnz] arrays that load_data cannot read; that is not reproduced.)

The reference takes minutes to hours here (a Python loop per slice, per row and per non-zero:
read_data.py:94-107, 118-124, 135-163, 210-222); each step below is one expand + sort +
reduce-by-key launch sequence over all slices.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import numpy as np
import scipy.io as sio
import torch

from .adjacency import DeviceCOO


def slice_by_time(times, time_delta: Optional[float] = None) -> Tuple[np.ndarray, np.ndarray, int]:
    """Time stamp of every edge -> (slice index, keep mask, number of slices TT).

    time_delta=None : one slice per distinct time stamp, in increasing order (the 'Chess' branch,
                      read_data.m:106-108 / 140).
    time_delta=d    : TT = floor((max-min)/d) slices of width d starting at the first time stamp;
                      edges at or after min + TT·d are dropped (read_data.m:110, 131, 142-144)."""
    times = np.asarray(times, dtype=np.float64)
    if time_delta is None:
        dates, k = np.unique(times, return_inverse=True)
        return k.astype(np.int64), np.ones(times.shape, bool), int(dates.size)
    t0 = times.min()
    TT = int(np.floor((times.max() - t0) / time_delta))
    keep = times < t0 + TT * time_delta
    k = np.floor((times - t0) / time_delta).astype(np.int64)
    return np.where(keep, k, 0), keep, TT


def band_matrix(T: int, no_diag: int = 20, weights: str = "ones", rownorm: bool = True) -> np.ndarray:
    """Lower-triangular band M [T,T], fp64.  weights "ones" (read_data.m M_choice 1, read_data.py:55-59)
    or "harmonic" = 1/d on the d-th diagonal (M_choice 2, SBM_our.py:88-96); rownorm divides every
    row by its absolute sum (read_data.m:125-127, read_data.py:60-61)."""
    if weights not in ("ones", "harmonic"):
        raise RuntimeError('weights must be "ones" or "harmonic"')
    M = np.zeros((T, T))
    for d in range(min(no_diag, T)):
        M[np.arange(d, T), np.arange(0, T - d)] = 1.0 if weights == "ones" else 1.0 / (d + 1)
    return M / np.abs(M).sum(axis=1, keepdims=True) if rownorm else M


def _block(c: DeviceCOO, start: int, count: int) -> DeviceCOO:
    """Slices start..start+count-1 of a sorted COO, re-based to slice 0 (func_create_sparse,
    read_data.py:174-183).  Sorted keys make a block one contiguous key range."""
    if not c.sorted_reduced:
        c = c.sort_reduce()
    NN = c.N * c.N
    lo, hi = torch.searchsorted(c.key, torch.tensor([start * NN, (start + count) * NN], device=c.key.device)).tolist()
    return DeviceCOO((c.key[lo:hi] - start * NN).contiguous(), c.val[lo:hi].contiguous(), count, c.N,
                     sorted_reduced=c.sorted_reduced)


def _mat_arrays(c: DeviceCOO) -> Tuple[np.ndarray, np.ndarray]:
    key = c.key.cpu().numpy()
    N = c.N
    subs = np.stack([key // (N * N), (key // N) % N, key % N], axis=1) + 1
    return subs.astype(np.float64), c.val.double().cpu().numpy()[:, None]


def read_data(data, no_train_samples: int, no_val_samples: int, no_test_samples: int, *,
              time_delta: Optional[float] = None, edge_life: bool = True, edge_life_window: int = 10,
              no_diag: int = 20, make_symmetric: bool = True, m_weights: str = "ones", m_rownorm: bool = True,
              device="cuda") -> Dict[str, np.ndarray]:
    """``data``: [n,4] rows (src, dst, label, time), node ids 1-based as in the reference's csv
    files.  Returns the dict read_data.m:232 saves: tensor_idx / tensor_labels (the raw rows), A,
    A_labels, the normalised adjacency C and its train / val / test blocks C_* of
    ``no_train_samples`` slices each (starting at 0, no_val_samples, no_val_samples +
    no_test_samples), their M-products Ct_*, and M.

    Steps: slice by time; A = edge multiplicities, A_labels = summed labels (sptensor / coalesce
    semantics); B = (A + Aᵀ)/2; edge life; C = D^-1/2 (B + I) D^-1/2; blocks; Ct = M ×₁ block.
    Values pass through the device pipeline in fp32 (the layer consumes fp32)."""
    data = np.asarray(data, dtype=np.float64)
    k, keep, TT = slice_by_time(data[:, 3], time_delta)
    data, k = data[keep], k[keep]
    N = int(max(data[:, 0].max(), data[:, 1].max()))
    T = int(no_train_samples)
    i, j = data[:, 0].astype(np.int64) - 1, data[:, 1].astype(np.int64) - 1
    M = band_matrix(T, no_diag, m_weights, m_rownorm)

    A = DeviceCOO.from_edges(k, i, j, np.ones(len(k), np.float32), TT, N, device).sort_reduce()
    A_labels = DeviceCOO.from_edges(k, i, j, data[:, 2].astype(np.float32), TT, N, device).sort_reduce()
    B = A.symmetrise() if make_symmetric else A
    if edge_life:
        B = B.edge_life(edge_life_window)
    C = B.add_identity_and_normalise()
    out = {"tensor_idx": np.stack([k + 1, i + 1, j + 1], axis=1).astype(np.float64),
           "tensor_labels": data[:, 2:3].copy(), "M": M}
    for name, coo in (("A", A), ("A_labels", A_labels), ("C", C)):
        out[name + "_subs"], out[name + "_vals"] = _mat_arrays(coo)
    starts = {"train": 0, "val": int(no_val_samples), "test": int(no_val_samples) + int(no_test_samples)}
    for name, start in starts.items():
        blk = _block(C, start, T)        # slices past TT, if any, are empty (read_data.py:168-172)
        out[f"C_{name}_subs"], out[f"C_{name}_vals"] = _mat_arrays(blk)
        out[f"Ct_{name}_subs"], out[f"Ct_{name}_vals"] = _mat_arrays(blk.m_product(M))
    return out


def save_content(path: str, content: Dict[str, np.ndarray]) -> None:
    """Write the dict of read_data() as a MATLAB v5 .mat that ehf.load_data reads.  Subscript
    arrays are stored in the narrowest unsigned integer type that holds them, as MATLAB's own
    ``save`` does for integer-valued doubles."""
    packed = {}
    for name, a in content.items():
        if name.endswith("_subs") or name == "tensor_idx":
            top = int(a.max()) if a.size else 0
            a = a.astype(np.uint8 if top < 2 ** 8 else np.uint16 if top < 2 ** 16 else np.uint32)
        packed[name] = a
    sio.savemat(path, packed, do_compression=True)

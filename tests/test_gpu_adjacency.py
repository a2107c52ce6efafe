"""GPU: the device-side adjacency pipeline (tm-gcn_amd/adjacency.py, csrc/adjacency.hip) against
(a) the reference's own preprocessing functions run on the chess data it ships (fixture G5) and
(b) the scipy restatement in synth.py on random weighted multigraphs."""
import numpy as np
import pytest
import torch

from _util import assert_close, golden
import tmgcn_amd.layers as ehf
from tmgcn_amd import adjacency, synth
from tmgcn_amd.csr import BatchedCSR

pytestmark = pytest.mark.gpu


def _dense(T, N, k, i, j, v):
    out = np.zeros((T, N, N))
    np.add.at(out, (k, i, j), v)
    return out


def test_chess_pipeline_matches_reference_functions():
    d = golden("g5_chess_gcn2")
    T, N = int(d["T"]), int(d["N"])
    ones = np.ones(len(d["raw_k"]), np.float32)
    Chat, Ahat = adjacency.build_adjacency(d["raw_k"], d["raw_i"], d["raw_j"], ones, T, N, M=d["M"], window=10)
    assert_close(Chat.to_dense(), _dense(T, N, d["C_k"], d["C_i"], d["C_j"], d["C_v"]), 1e-6, "normalised adjacency")
    assert_close(Ahat.to_dense(), _dense(T, N, d["At_k"], d["At_i"], d["At_j"], d["At_v"]), 1e-6, "M-product of A")
    # same sparsity pattern as the reference's coalesced tensors
    assert Chat.nnz == len(d["C_v"]) and Ahat.nnz == len(d["At_v"])
    # and the model on top of the device-built adjacency reproduces the reference's logits
    torch.manual_seed(int(d["seed"]))
    m = ehf.EmbeddingGCN2(Ahat, torch.from_numpy(d["X"]), torch.from_numpy(d["edges"]), torch.from_numpy(d["M"]),
                          hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")
    assert_close(m(), d["logits"], 1e-5, "chess logits from the device-built adjacency")


@pytest.mark.parametrize("T,N,E,window,b,kind", [(12, 60, 200, 10, 5, "matlab"), (7, 33, 50, 3, 20, "python"),
                                                  (5, 10, 0, 4, 2, "matlab"), (9, 200, 3000, 1, 4, "matlab")])
def test_random_multigraph_matches_scipy_pipeline(T, N, E, window, b, kind):
    import scipy.sparse as sp
    rng = np.random.default_rng(T * 100 + N)
    t = rng.integers(0, T, E)
    i = rng.integers(0, N, E)
    j = rng.integers(0, N, E)
    w = rng.uniform(0.5, 2.0, E).astype(np.float32)  # weighted, with duplicate (t,i,j) entries
    A = [sp.coo_matrix((w[t == k].astype(np.float64), (i[t == k], j[t == k])), shape=(N, N)).tocsr() for k in range(T)]
    C = synth.normalise(synth.edge_life(synth.symmetrise(A), window))
    M = synth.band_M(T, b, kind)
    Ct = synth.m_product(C, M)
    Chat, Ahat = adjacency.build_adjacency(t, i, j, w, T, N, M=M, window=window)
    assert_close(Chat.to_dense(), np.stack([c.toarray() for c in C]), 2e-6, "C")
    assert_close(Ahat.to_dense(), np.stack([c.toarray() for c in Ct]), 2e-6, "Ct")
    # asymmetric variant (make_symmetric = False in read_data.py:10)
    Cn = synth.normalise(synth.edge_life(A, window))
    Chat2, none = adjacency.build_adjacency(t, i, j, w, T, N, M=None, window=window, symmetric=False)
    assert none is None
    assert_close(Chat2.to_dense(), np.stack([c.toarray() for c in Cn]), 2e-6, "C asymmetric")


def test_native_transpose_matches_host_side_transpose():
    g = torch.Generator().manual_seed(0)
    T, N, nnz = 4, 300, 5000
    k = torch.randint(0, T, (nnz,), generator=g)
    i = torch.randint(0, N, (nnz,), generator=g)
    j = torch.randint(0, N, (nnz,), generator=g)
    A = adjacency.DeviceCOO.from_edges(k, i, j, torch.randn(nnz, generator=g), T, N).sort_reduce().to_csr()
    At = adjacency.csr_transpose(A)
    ref = A.transpose()
    assert torch.equal(At.rowptr, ref.rowptr) and torch.equal(At.col, ref.col) and torch.equal(At.val, ref.val)
    assert torch.equal(At.to_dense(), A.to_dense().transpose(1, 2))


def test_bad_edges_raise():
    with pytest.raises(RuntimeError, match="out of range"):
        adjacency.build_adjacency([0, 5], [0, 1], [1, 2], [1.0, 1.0], T=3, N=4)


@pytest.mark.parametrize("T,N,deg,lo,hi", [(12, 70, 4.0, 5, 0),      # lower band, half-wave groups
                                           (40, 33, 2.0, 19, 0),     # the reference's 20 diagonals
                                           (40, 25, 3.0, 0, 7),      # upper band (Mᵀ-shaped operator)
                                           (50, 20, 2.0, 30, 9),     # 40 slices reached: full-wave groups
                                           (6, 15, 0.3, 5, 5),       # band wider than T, mostly empty rows
                                           (3, 8, 0.0, 1, 1)])       # empty tensor
def test_mproduct_segmented_merge_matches_expand_sort_and_scipy(T, N, deg, lo, hi):
    """The hand-written segmented merge (tmgcn_adj_mproduct_merge_count / _fill) against the expand + sort
    + reduce form it replaces and against scipy: same sparsity pattern entry for entry, values within
    fp32 rounding of each other (the merge sums in fp64 and rounds once), bit-reproducible."""
    import scipy.sparse as sp
    rng = np.random.default_rng(T * 1000 + N)
    nnz = int(T * N * deg)
    k, i, j = rng.integers(0, T, nnz), rng.integers(0, N, nnz), rng.integers(0, N, nnz)
    v = rng.standard_normal(nnz).astype(np.float32)
    A = adjacency.DeviceCOO.from_edges(k, i, j, v, T, N).sort_reduce().to_csr()
    M = np.zeros((T, T))
    for a in range(T):
        for b in range(max(0, a - lo), min(T, a + hi + 1)):
            M[a, b] = rng.uniform(0.2, 1.0) * (1 if rng.random() < 0.9 else 0)    # a few exact zeros inside the band
    merged = adjacency.m_product_csr(A, M, algo="merge")
    expanded = adjacency.m_product_csr(A, M, algo="expand")
    assert torch.equal(merged.rowptr, expanded.rowptr) and torch.equal(merged.col, expanded.col)
    if merged.nnz:
        scale = float(expanded.val.abs().max())
        assert float((merged.val - expanded.val).abs().max()) <= 2e-6 * scale
    again = adjacency.m_product_csr(A, M, algo="merge")
    assert torch.equal(again.val, merged.val) and torch.equal(again.col, merged.col)
    dense = A.to_dense().cpu().double().numpy()
    ref = np.einsum("kj,jab->kab", M, dense)
    assert_close(merged.to_dense(), ref, 2e-6, "merge vs dense einsum")
    # columns ascending inside every row (the CSR contract the SpMM's fixed summation order rests on)
    if merged.nnz > 1:
        rid = merged.row_ids()
        same_row = rid[1:] == rid[:-1]
        assert bool((merged.col[1:][same_row] > merged.col[:-1][same_row]).all())


def test_mproduct_merge_sums_duplicate_columns_and_refuses_wide_bands():
    """Input rows with repeated columns (a CSR built without coalescing) are summed, not emitted twice;
    a band reaching more than 64 slices is refused by the merge entry point and taken by the expand form."""
    T, N = 4, 6
    rowptr = torch.zeros(T * N + 1, dtype=torch.int64)
    rowptr[1:] = 3                                                   # row 0 of slice 0 holds everything
    A = BatchedCSR(rowptr.cuda(), torch.tensor([2, 2, 5], dtype=torch.int32).cuda(), torch.tensor([1.0, 2.0, 4.0]).cuda(), T, N)
    M = np.eye(T) * 2.0
    out = adjacency.m_product_csr(A, M, algo="merge")
    assert out.nnz == 2 and out.col.tolist() == [2, 5] and out.val.tolist() == [6.0, 8.0]
    Tw = 80
    Aw = adjacency.DeviceCOO.from_edges([0, 79], [1, 2], [3, 4], [1.0, 1.0], Tw, 5).sort_reduce().to_csr()
    Mw = np.tril(np.ones((Tw, Tw)))
    with pytest.raises(RuntimeError, match="wider than 64"):
        adjacency.m_product_csr(Aw, Mw, algo="merge")
    auto = adjacency.m_product_csr(Aw, Mw)                            # auto: falls back to expand + sort
    assert_close(auto.to_dense(), np.einsum("kj,jab->kab", Mw, Aw.to_dense().cpu().double().numpy()), 1e-6, "wide band")


@pytest.mark.parametrize("T,N,E,window", [(12, 40, 300, 10), (7, 25, 60, 3), (5, 9, 0, 4), (30, 15, 200, 40)])
def test_edge_life_as_a_merge_matches_the_expand_form(T, N, E, window):
    """B'[t] = B[t] + … + B[t-window+1] is the mode-1 product with a lower band of ones: the segmented
    merge gives the pattern of the expand + sort form entry for entry and its values to fp32 rounding."""
    rng = np.random.default_rng(7 * T + N)
    coo = adjacency.DeviceCOO.from_edges(rng.integers(0, T, E), rng.integers(0, N, E), rng.integers(0, N, E),
                                         rng.uniform(0.5, 2.0, E).astype(np.float32), T, N).sort_reduce()
    a, b = coo.edge_life(window, algo="merge"), coo.edge_life(window, algo="expand")
    assert torch.equal(a.key, b.key)
    if a.n:
        assert float((a.val - b.val).abs().max()) <= 2e-6 * float(b.val.abs().max())

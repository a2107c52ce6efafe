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

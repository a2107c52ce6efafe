"""Host logic of tm-gcn_amd/preprocess.py (time slicing, M recipes, the .mat layout) on the CPU,
and — on the GPU — the whole raw-edge-list -> .mat -> ehf.load_data chain against the scipy
restatement of the reference's preprocessing (synth.py, itself pinned to the reference's own
functions by fixture G5)."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from tmgcn_amd import data as ehf_data
from tmgcn_amd import preprocess, synth


def test_slice_by_time_fixed_width_matches_the_running_window_loop():
    rng = np.random.default_rng(0)
    times = rng.integers(1_000_000, 1_000_000 + 37 * 3600, 500).astype(np.float64)
    delta = 3600.0
    k, keep, TT = preprocess.slice_by_time(times, delta)
    # read_data.m:131-145 restated literally
    TT_ref = int(np.floor((times.max() - times.min()) / delta))
    kept = times < times.min() + TT_ref * delta
    start = times[kept].min()
    k_ref = np.full(times.shape, -1)
    for t in range(TT_ref):
        end = start + delta
        k_ref[(times >= start) & (times < end) & kept] = t
        start = end
    assert TT == TT_ref and np.array_equal(keep, kept) and np.array_equal(k[keep], k_ref[kept])
    assert k[keep].max() == TT - 1


def test_slice_by_time_distinct_dates():
    times = np.array([5.5, 1.0, 5.5, 3.25, 1.0])
    k, keep, TT = preprocess.slice_by_time(times)
    assert TT == 3 and keep.all() and k.tolist() == [2, 0, 2, 1, 0]


def test_band_matrix_recipes():
    assert np.array_equal(preprocess.band_matrix(9, 4, "ones", True), synth.band_M(9, 4, "python"))
    assert np.array_equal(preprocess.band_matrix(9, 4, "harmonic", False), synth.band_M(9, 4, "matlab"))
    assert np.array_equal(preprocess.band_matrix(3, 20, "harmonic", False), synth.band_M(3, 20, "matlab"))
    M = preprocess.band_matrix(6, 3, "harmonic", True)
    assert np.allclose(M.sum(1), 1) and np.all(np.triu(M, 1) == 0) and M[5, 2] == 0 and M[5, 3] > 0
    with pytest.raises(RuntimeError):
        preprocess.band_matrix(4, 2, "gaussian")


def _content(T, S, N, rng):
    """A saved_content dict built on the host (scipy pipeline), in read_data()'s output format."""
    A = synth.random_slices(T, N, 3 * N, rng)
    A[-1] = sp.csr_matrix(A[-1] + sp.coo_matrix(([1.0], ([N - 1], [0])), shape=(N, N)))
    A[-1].data[:] = 1.0
    C = synth.normalise(synth.edge_life(synth.symmetrise(A), 3))
    M = preprocess.band_matrix(S[0], 3)

    def arrays(mats):
        rows = [np.stack([np.full(m.nnz, k), m.tocoo().row, m.tocoo().col], 1) for k, m in enumerate(mats)]
        return (np.concatenate(rows) + 1).astype(np.float64), np.concatenate([m.tocoo().data for m in mats])[:, None]

    out = {"M": M}
    out["A_labels_subs"], out["A_labels_vals"] = arrays(A)
    out["C_subs"], out["C_vals"] = arrays(C)
    blocks = {"train": C[:S[0]], "val": C[S[1]:S[0] + S[1]], "test": C[S[1] + S[2]:]}
    for name, blk in blocks.items():
        out[f"Ct_{name}_subs"], out[f"Ct_{name}_vals"] = arrays(synth.m_product(blk, M))
    return out, C, blocks, M


def test_saved_content_round_trips_through_load_data(tmp_path):
    S, N = (6, 2, 1), 300                       # N > 255 -> uint16 subscripts
    content, C, blocks, M = _content(sum(S), S, N, np.random.default_rng(3))
    preprocess.save_content(str(tmp_path / "saved_content_x.mat"), content)
    A, A_labels, Ct_train, Ct_val, Ct_test, N2, M2 = ehf_data.load_data(str(tmp_path) + "/", "saved_content_x.mat", *S, transformed=True)
    assert N2 == N and np.array_equal(M2.numpy(), M)
    for lst, blk in ((Ct_train, blocks["train"]), (Ct_val, blocks["val"]), (Ct_test, blocks["test"])):
        ref = synth.m_product(blk, M)
        assert len(lst) == S[0]
        for got, want in zip(lst, ref):
            assert np.allclose(got.to_dense().numpy(), want.toarray(), rtol=1e-15, atol=0)
    _, _, C_train, C_val, C_test, _ = ehf_data.load_data(str(tmp_path) + "/", "saved_content_x.mat", *S, transformed=False)
    for got, want in zip(C_train + C_val + C_test, C):
        assert np.allclose(got.to_dense().numpy(), want.toarray(), rtol=1e-15, atol=0)


@pytest.mark.gpu
@pytest.mark.parametrize("time_delta,symmetric,life", [(None, True, 3), (50.0, True, 10), (50.0, False, 1)])
def test_read_data_matches_scipy_pipeline_and_feeds_the_model(tmp_path, time_delta, symmetric, life):
    import tmgcn_amd.ehf as ehf
    from _util import assert_close
    rng = np.random.default_rng(11)
    S, N, n = (8, 2, 2), 120, 4000
    TT = sum(S)
    if time_delta is None:
        times = rng.choice(np.sort(rng.uniform(0, 1e6, TT)), n)           # TT distinct dates
    else:
        times = np.concatenate(([1000.0, 1000.0 + TT * time_delta + 1], rng.uniform(1000.0, 1000.0 + TT * time_delta, n - 2)))
    raw = np.stack([rng.integers(1, N + 1, n), rng.integers(1, N + 1, n), rng.choice([-1.0, 1.0, 2.0], n), times], 1)
    raw[0, :2] = (N, 1)                                                   # the last node id is used
    content = preprocess.read_data(raw, *S, time_delta=time_delta, edge_life=life > 1, edge_life_window=life,
                                   no_diag=4, make_symmetric=symmetric)
    # host restatement on scipy
    k, keep, TT2 = preprocess.slice_by_time(raw[:, 3], time_delta)
    assert TT2 == TT
    r, k = raw[keep], k[keep]
    A = [sp.coo_matrix((np.ones((k == t).sum()), (r[k == t, 0].astype(int) - 1, r[k == t, 1].astype(int) - 1)), shape=(N, N)).tocsr()
         for t in range(TT)]
    B = synth.symmetrise(A) if symmetric else A
    C = synth.normalise(synth.edge_life(B, life))
    M = preprocess.band_matrix(S[0], 4)

    def dense(subs, vals, T):
        out = np.zeros((T, N, N))
        s = subs.astype(int) - 1
        np.add.at(out, (s[:, 0], s[:, 1], s[:, 2]), vals[:, 0])
        return out

    assert_close(dense(content["C_subs"], content["C_vals"], TT), np.stack([c.toarray() for c in C]), 2e-6, "C")
    assert np.array_equal(dense(content["A_subs"], content["A_vals"], TT), np.stack([a.toarray() for a in A]))
    lab = np.zeros((TT, N, N))
    np.add.at(lab, (k, r[:, 0].astype(int) - 1, r[:, 1].astype(int) - 1), r[:, 2])
    assert np.array_equal(dense(content["A_labels_subs"], content["A_labels_vals"], TT), lab)
    for name, start in (("train", 0), ("val", S[1]), ("test", S[1] + S[2])):
        blk = C[start:start + S[0]]
        assert_close(dense(content[f"C_{name}_subs"], content[f"C_{name}_vals"], S[0]), np.stack([c.toarray() for c in blk]), 2e-6, name)
        assert_close(dense(content[f"Ct_{name}_subs"], content[f"Ct_{name}_vals"], S[0]),
                     np.stack([c.toarray() for c in synth.m_product(blk, M)]), 2e-6, "Ct_" + name)
    assert np.array_equal(content["M"], M) and content["tensor_idx"].shape == (len(r), 3)

    # raw -> .mat -> ehf.load_data -> model, the reference's two-stage workflow
    preprocess.save_content(str(tmp_path / "saved_content_t.mat"), content)
    A3, A_labels, Ct_train, Ct_val, Ct_test, N2, M2 = ehf.load_data(str(tmp_path) + "/", "saved_content_t.mat", *S, transformed=True)
    assert N2 == N and tuple(A3.shape) == (TT, N, N)
    X_train, X_val, X_test = ehf.create_node_features(A3, *S, same_block_size=True)
    e = A_labels._indices()
    e_train = e[:, e[0] < S[0]]
    torch.manual_seed(1)
    gcn = ehf.EmbeddingGCN2(Ct_train, X_train, e_train, M2, hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")
    out = gcn()
    assert out.shape == (e_train.shape[1], 2) and torch.isfinite(out).all()
    out_val = gcn(Ct_val, X_val, e_train)
    assert torch.isfinite(out_val).all()

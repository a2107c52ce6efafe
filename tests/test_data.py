"""The data side of the ehf surface (tm-gcn_amd/data.py) against G9: what the real reference's
load_data / create_node_features / augment_edges / split_data / compute_At returned for the
committed synthetic ``g9_saved_content.mat`` (tests/golden/make_golden_data.py).  Integer results
are compared bit-exactly, values to fp64/fp32 round-off."""
import io
import os
import random
from contextlib import redirect_stdout

import numpy as np
import pytest
import torch

from _util import GOLDEN as GOLDEN_DIR, golden
from tmgcn_amd import data

MAT = "g9_saved_content.mat"


@pytest.fixture(scope="module")
def g9():
    return golden("g9_data")


def _load(g9, transformed):
    S = [int(s) for s in g9["S"]]
    return data.load_data(GOLDEN_DIR + "/", MAT, S[0], S[1], S[2], transformed=transformed), S


def _same_sparse(x, idx, val, rtol=0.0):
    x = x.coalesce()
    assert np.array_equal(x.indices().numpy(), idx)
    assert x.values().numpy().dtype == val.dtype
    np.testing.assert_allclose(x.values().numpy(), val, rtol=rtol, atol=0)


def _same_list(lst, g9, prefix):
    assert len(lst) == int(g9[prefix + "_n"])
    for k, m in enumerate(lst):
        _same_sparse(m, g9[f"{prefix}_{k}_idx"], g9[f"{prefix}_{k}_val"])


def test_load_data_transformed(g9):
    (A, A_labels, Ct_train, Ct_val, Ct_test, N, M), S = _load(g9, True)
    assert N == int(g9["N"]) and tuple(A.shape) == (sum(S), N, N)
    _same_sparse(A, g9["A_idx"], g9["A_val"])
    _same_sparse(A_labels, g9["A_labels_idx"], g9["A_labels_val"])
    assert M.dtype == torch.float64 and np.array_equal(M.numpy(), g9["M"])
    for name, lst in (("Ct_train", Ct_train), ("Ct_val", Ct_val), ("Ct_test", Ct_test)):
        _same_list(lst, g9, name)
        assert all(tuple(m.shape) == (N, N) for m in lst)


def test_load_data_untransformed(g9):
    (A, A_labels, C_train, C_val, C_test, N), S = _load(g9, False)
    assert [len(C_train), len(C_val), len(C_test)] == S
    for name, lst in (("C_train", C_train), ("C_val", C_val), ("C_test", C_test)):
        _same_list(lst, g9, name)


@pytest.mark.parametrize("sbs", [True, False])
def test_create_node_features(g9, sbs):
    (A, *_), S = _load(g9, True)
    for name, x in zip(("train", "val", "test"), data.create_node_features(A, *S, same_block_size=sbs)):
        ref = g9[f"X_{name}_sbs{int(sbs)}"]
        assert x.dtype == torch.float64 and np.array_equal(x.numpy(), ref), name


def test_augment_edges_reproduces_the_seeded_reference_stream(g9):
    (_, A_labels, *_), _ = _load(g9, True)
    random.seed(7)
    N = int(g9["N"])
    edges_aug, labels = data.augment_edges(A_labels.indices(), N, 3, 2, 6)
    canon = lambda e, l: np.sort(((e[0] * N + e[1]) * N + e[2]) * 2 + l)
    assert np.array_equal(canon(edges_aug.numpy(), labels.numpy()), canon(g9["edges_aug"], g9["labels"]))
    assert np.array_equal(edges_aug[0].numpy(), g9["edges_aug"][0])
    # same torch build as the one that wrote the fixture: the order inside the slices matches too
    assert np.array_equal(edges_aug.numpy(), g9["edges_aug"]) and np.array_equal(labels.numpy(), g9["labels"])


def test_augment_edges_bulk_sampler_properties(g9):
    (_, A_labels, *_), _ = _load(g9, True)
    edges, N = A_labels.indices(), int(g9["N"])
    gen = torch.Generator().manual_seed(11)
    edges_aug, labels = data.augment_edges(edges, N, 3, 2, 6, generator=gen)
    s = edges_aug[0]
    assert bool((s[1:] >= s[:-1]).all())                                   # sorted by slice
    counts = torch.bincount(edges[0])
    beta = torch.where(torch.arange(counts.numel()) < 6, 3, 2)
    assert np.array_equal(torch.bincount(s[labels == 1], minlength=counts.numel()).numpy(), (beta * counts).numpy())
    key = lambda e: (e[0] * N + e[1]) * N + e[2]
    assert not torch.isin(key(edges_aug[:, labels == 1]), key(edges)).any()   # negatives are non-edges of their slice
    assert np.array_equal(np.sort(key(edges_aug[:, labels == 0]).numpy()), np.sort(key(edges).numpy()))
    assert int(edges_aug[1:].min()) >= 0 and int(edges_aug[1:].max()) < N
    # deterministic in the generator
    again = data.augment_edges(edges, N, 3, 2, 6, generator=torch.Generator().manual_seed(11))
    assert torch.equal(again[0], edges_aug) and torch.equal(again[1], labels)


def test_augment_edges_without_any_edge_to_add():
    edges = torch.tensor([[0, 0, 1], [1, 2, 0], [2, 0, 1]])
    edges_aug, labels = data.augment_edges(edges, 3, 0, 0, 5)
    assert torch.equal(edges_aug, edges) and labels.tolist() == [0, 0, 0]


@pytest.mark.parametrize("sbs", [True, False])
def test_split_data(g9, sbs):
    S = [int(s) for s in g9["S"]]
    edges_aug, labels = torch.from_numpy(g9["edges_aug"]), torch.from_numpy(g9["labels"])
    keep = edges_aug.clone()
    got = data.split_data(edges_aug, labels, *S, same_block_size=sbs)
    assert torch.equal(edges_aug, keep)                                    # the caller's tensor is not modified
    names = ("edges_train", "target_train", "e_train", "edges_val", "target_val", "e_val", "K_val",
             "edges_test", "target_test", "e_test", "K_test")
    if not sbs:
        names = tuple(n for n in names if not n.startswith("K_"))
    assert len(got) == len(names)
    for n, v in zip(names, got):
        assert np.array_equal(np.asarray(v), g9[f"split{int(sbs)}_{n}"]), n
    if sbs:                                                                # usable as the scripts use it: x[-K_val:]
        assert got[4][-got[6]:].numel() == int(got[6])


@pytest.mark.parametrize("nt", [0, 1])
def test_compute_At(g9, nt, tmp_path):
    A = torch.sparse_coo_tensor(torch.from_numpy(g9["cAt_idx"]), torch.from_numpy(g9["cAt_val"]), (4, 6, 6))
    M = torch.from_numpy(g9["cAt_M"])
    f_at, f_ij = str(tmp_path / "At"), str(tmp_path / "ij")
    for attempt in range(2):                                               # second call loads the pickles
        At = data.compute_At(f_at, f_ij, A, M, normalization_type=nt)
        assert os.path.isfile(f_at) and os.path.isfile(f_ij)
        assert len(At) == 4
        assert np.array_equal(At[0]._indices().numpy(), g9[f"cAt{nt}_ij"])
        got = np.stack([a._values().numpy() for a in At])
        np.testing.assert_allclose(got, g9[f"cAt{nt}_vals"], rtol=2e-6, atol=1e-7)


def test_print_f1_lines():
    buf = io.StringIO()
    with redirect_stdout(buf):
        data.print_f1(*[0.1 * i for i in range(1, 13)], alpha=0.9, tr=0, ep=100)
        data.print_f1(*[0.1 * i for i in range(1, 13)], is_final=True)
    lines = buf.getvalue().split("\n")
    assert lines[0] == "alpha/Tr/Ep 0.90/0/100. Train precision/recall/f1 %.16f/%.16f/%.16f. Train loss %.16f." % (0.1, 0.2, 0.1 * 3, 0.4)
    assert lines[2].startswith("alpha/Tr/Ep 0.90/0/100. Test precision/recall/f1 ") and lines[3] == ""
    assert lines[4].startswith("FINAL: Train precision/recall/f1 ") and lines[5].startswith("FINAL: Val ")
    assert lines[6].endswith("Test loss %.16f." % (0.1 * 12)) and lines[7] == ""

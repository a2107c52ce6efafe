"""CPU: the oracle (oracle/tmgcn_oracle.py, oracle/tmgcn_ref.c) against the golden fixtures
captured from the real reference (tests/golden/make_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest
import torch

from _util import REL_TOL, assert_close, coo_list, cptr, golden, golden_names, load_c_oracle, unpack_sym
from oracle import tmgcn_oracle as orc
import tmgcn_amd
from tmgcn_amd import synth
from tmgcn_amd.csr import BatchedCSR

TIGHT = 2e-7  # same ATen ops in the same order: only thread-count reduction-order noise


def _inputs(d, T=None, N=None, prefix=""):
    X = torch.from_numpy(d[prefix + "X"])
    T, N = X.shape[0], X.shape[1]
    return dict(T=T, N=N, X=X, M=torch.from_numpy(d[prefix + "M"]), edges=torch.from_numpy(d[prefix + "edges"]),
                labels=torch.from_numpy(d[prefix + "labels"]),
                At=coo_list(d, "At", T, N, prefix=prefix), A=coo_list(d, "A", T, N, prefix=prefix) if prefix + "A_k" in d else None)


def _loss_grads(out_fn, params, target, alpha=0.9):
    ps = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    out = out_fn(ps)
    loss = torch.nn.CrossEntropyLoss(weight=torch.tensor([alpha, 1 - alpha]))(out, target)
    loss.backward()
    return out.detach(), float(loss), {k: v.grad for k, v in ps.items()}


@pytest.mark.parametrize("name", golden_names("g1_"))
def test_g1_compute_AtXt(name):
    d = golden(name)
    i = _inputs(d)
    assert_close(orc.compute_AtXt(i["M"], i["At"], i["X"]), d["AtXt"], TIGHT, name)


@pytest.mark.parametrize("name", golden_names("g1_"))
def test_g1_c_oracle_and_dense_identity(name):
    """The plain-C restatement and the einsum identity agree with the reference's AtXt."""
    d = golden(name)
    i = _inputs(d)
    lib = load_c_oracle()
    T, N, F = i["X"].shape
    csr = BatchedCSR.from_coo_list(i["At"], N=N)
    X32 = i["X"].float().contiguous()
    Xt = torch.empty_like(X32)
    M64 = i["M"].contiguous()
    lib.ref_mtransform(cptr(M64), T, 0, cptr(X32), cptr(Xt), N * F)
    Y = torch.empty_like(X32)
    lib.ref_spmm(cptr(csr.rowptr), cptr(csr.col), cptr(csr.val), cptr(Xt), cptr(Y), T * N, N, F)
    assert_close(Y, d["AtXt"], REL_TOL, name + " C oracle")
    dense = torch.einsum("knm,kmf->knf", csr.to_dense().double(), torch.einsum("kj,jnf->knf", i["M"], i["X"]))
    assert_close(dense, d["AtXt"], REL_TOL, name + " einsum")


@pytest.mark.parametrize("name", ["g2_gcn_condensed1", "g2_gcn_condensed0"])
def test_g2_c_oracle_gemm_and_gemm_dw(name):
    """Pins the GEMM half of the C oracle (ref_gemm, ref_gemm_dw: the checker of the P3 / dW kernel
    tests and of the Y / dW legs of the bench-size test and of bench.py's verify block) on the
    reference's own numbers: ehf:222 `t.matmul(AtXt, W)` through the edge head reproduces the
    fixture's logits, and its autograd `dW = Σ AtXtᵀ·dY` reproduces the fixture's dW — for the
    shared weight (condensed_W) and for one weight per slice."""
    d = golden(name)
    i = _inputs(d)
    lib = load_c_oracle()
    T, N, F0 = i["X"].shape
    W0 = torch.from_numpy(d["W0"]).contiguous()
    per_slice = W0.dim() == 3
    F1 = W0.shape[-1]
    AtXt = orc.compute_AtXt(i["M"], i["At"], i["X"]).contiguous()           # fp32 [T,N,F0], pinned by G1
    assert AtXt.dtype == torch.float32
    Y = torch.empty(T, N, F1)
    lib.ref_gemm(cptr(AtXt), cptr(W0), cptr(Y), T * N, F0, F1, 0, N if per_slice else 0, F0 * F1 if per_slice else 0)
    src, dst = orc.flat_edge_index(i["edges"], N)
    U = torch.from_numpy(d["U0"]).clone().requires_grad_(True)
    Yl = Y.clone().requires_grad_(True)
    logits = orc.edge_head(Yl, src, dst, U)
    assert_close(logits.detach(), d["logits"], 1e-6, name + " logits through ref_gemm")
    loss = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))(logits, i["labels"])
    assert abs(float(loss) - float(d["loss"])) <= 1e-6 * max(1.0, abs(float(d["loss"])))
    loss.backward()
    assert_close(U.grad, d["dU"], 1e-6, name + " dU")
    dY = Yl.grad.contiguous()
    dW = torch.empty_like(W0)
    lib.ref_gemm_dw(cptr(AtXt), cptr(dY), cptr(dW), T * N, F0, F1, N if per_slice else 0)
    assert_close(dW, d["dW"], 1e-6, name + " dW through ref_gemm_dw")
    # the transposed-weight form (dA = dY·Wᵀ, what the backward GEMM tests use) against torch fp64
    dA = torch.empty(T, N, F0)
    lib.ref_gemm(cptr(dY), cptr(W0), cptr(dA), T * N, F1, F0, 1, N if per_slice else 0, F0 * F1 if per_slice else 0)
    ref = torch.matmul(dY.double(), W0.double().transpose(-1, -2)) if not per_slice else \
        torch.einsum("tnf,tkf->tnk", dY.double(), W0.double())
    assert_close(dA, ref, 1e-6, name + " dA through ref_gemm(trans_w)")


def test_head_loss_fp64_restatement_is_pinned_on_g2():
    """tests/_util.head_loss_fp64 — the checker of the one-pass head + loss kernel's GPU tests — against the real reference:
    on G2's inputs (the condensed-W model: Z = AtXt, folded weight W0, head U0) it reproduces the fixture's logits, loss, dW
    and dU.  A second oracle is only worth something pinned."""
    from _util import head_loss_fp64
    d = golden("g2_gcn_condensed1")
    i = _inputs(d)
    T, N, _ = i["X"].shape
    AtXt = orc.compute_AtXt(i["M"], i["At"], i["X"]).contiguous()
    logits, loss, _, dU, dW = head_loss_fp64(AtXt, torch.from_numpy(d["W0"]), torch.from_numpy(d["U0"]), i["edges"], i["labels"],
                                             torch.tensor([0.9, 0.1]), N)
    assert_close(logits, d["logits"], 1e-6, "fp64 head restatement: logits")
    assert abs(float(loss) - float(d["loss"])) <= 1e-6 * max(1.0, abs(float(d["loss"])))
    assert_close(dU, d["dU"], 1e-6, "fp64 head restatement: dU")
    assert_close(dW, d["dW"], 1e-6, "fp64 head restatement: dW")


def test_c_oracle_row_window_is_the_full_product():
    """ref_mtransform_rows (one window of output rows; used by bench.py's verify block) returns the
    rows ref_mtransform returns, bit for bit, forward and transposed."""
    lib = load_c_oracle()
    T, C = 12, 37
    g = torch.Generator().manual_seed(0)
    M = torch.from_numpy(synth.band_M(T, 5, "matlab")).contiguous()
    X = torch.randn(T, C, generator=g)
    for tr in (0, 1):
        full = torch.empty(T, C)
        lib.ref_mtransform(cptr(M), T, tr, cptr(X), cptr(full), C)
        for r0, n in ((0, T), (3, 1), (7, 5)):
            part = torch.empty(n, C)
            lib.ref_mtransform_rows(cptr(M), T, tr, r0, n, cptr(X), cptr(part), C)
            assert torch.equal(part, full[r0:r0 + n])


@pytest.mark.parametrize("name", golden_names("g2_"))
def test_g2_gcn(name):
    d = golden(name)
    i = _inputs(d)
    fp32 = name.endswith("minv_fp32")
    if fp32:
        M, At, X = i["M"].float(), [a.float() for a in i["At"]], i["X"].float()
        Minv = torch.tensor(np.linalg.inv(M))
    else:
        M, At, X, Minv = i["M"], i["At"], i["X"], None
    # parameter draw order and values
    torch.manual_seed(int(d["seed"]))
    p = orc.draw_params("gcn", i["T"], [X.shape[-1], 6, 2], condensed_W=d["W0"].ndim == 2)
    assert np.array_equal(p["W"].numpy(), d["W0"]) and np.array_equal(p["U"].numpy(), d["U0"])
    AtXt = orc.compute_AtXt(M, At, X)
    src, dst = orc.flat_edge_index(i["edges"], i["N"])
    out, loss, g = _loss_grads(lambda q: orc.gcn_forward(AtXt, q["W"], q["U"], src, dst, Minv), p, i["labels"])
    assert_close(out, d["logits"], 1e-6, name + " logits")
    assert abs(loss - float(d["loss"])) <= 1e-6 * max(1.0, abs(float(d["loss"])))
    assert_close(g["W"], d["dW"], 1e-6, name + " dW")
    assert_close(g["U"], d["dU"], 1e-6, name + " dU")


@pytest.mark.parametrize("name", golden_names("g3_"))
def test_g3_gcn2(name):
    d = golden(name)
    i = _inputs(d)
    v = _inputs(d, prefix="val_")
    _, _, branch, nl, cond = name.split("_")
    kw = dict(apply_M_twice=branch in ("twice", "three"), apply_M_three_times=branch == "three", nonlin=nl)
    torch.manual_seed(int(d["seed"]))
    p = orc.draw_params("gcn2", i["T"], [2, 6, 6, 2], condensed_W=cond.endswith("1"))
    for k in ("W1", "W2", "U"):
        assert np.array_equal(p[k].numpy(), d[k + "0"]), k
    AtXt = orc.compute_AtXt(i["M"], i["At"], i["X"])
    src, dst = orc.flat_edge_index(i["edges"], i["N"])
    fwd = lambda q, a=AtXt, s=src, t=dst: orc.gcn2_forward(a, i["At"], i["M"], q["W1"], q["W2"], q["U"], s, t, **kw)
    out, loss, g = _loss_grads(fwd, p, i["labels"])
    assert_close(out, d["logits"], 1e-6, name + " logits")
    for k in ("W1", "W2", "U"):
        assert_close(g[k], d["d" + k], 2e-6, name + " d" + k)
    # validation-style call: layer 1 uses the passed At/X, layer 2 the TRAINING At (ehf:339-348)
    vs, vd = orc.flat_edge_index(v["edges"], v["N"])
    with torch.no_grad():
        out_val = fwd(p, orc.compute_AtXt(i["M"], v["At"], v["X"]), vs, vd)
    assert_close(out_val, d["logits_val"], 1e-6, name + " val logits")


@pytest.mark.parametrize("name", golden_names("g4_"))
def test_g4_kwgcn(name):
    d = golden(name)
    i = _inputs(d)
    two = "2layer" in name
    nl = name.split("_")[-1]
    torch.manual_seed(int(d["seed"]))
    p = orc.draw_params("kw", i["T"], [2, 6, 5, 2] if two else [2, 6, 2])
    for k in p:
        assert np.array_equal(p[k].numpy(), d[k + "0"]), k
    AX = orc.slice_spmm(i["A"], i["X"])
    src, dst = orc.flat_edge_index(i["edges"], i["N"])
    out, loss, g = _loss_grads(lambda q: orc.kwgcn_forward(AX, i["A"], q["W1"], q["U"], src, dst, q.get("W2"), nl), p, i["labels"])
    assert_close(out, d["logits"], 1e-6, name)
    for k in p:
        assert_close(g[k], d["d" + k], 2e-6, name + " d" + k)
    v = _inputs(d, prefix="val_")                               # shorter validation window, zero-padded (ehf:470)
    assert v["T"] < i["T"]
    vs, vd = orc.flat_edge_index(v["edges"], v["N"])
    with torch.no_grad():
        out_val = orc.kwgcn_forward(orc.slice_spmm(v["A"], v["X"]), i["A"], p["W1"], p["U"], vs, vd, p.get("W2"), nl)
    assert_close(out_val, d["logits_val"], 1e-6, name + " val")


def test_g5_chess_pipeline_and_model():
    """The reference's own preprocessing functions (read_data.py, run on the chess data it ships)
    pin synth's restated pipeline; the 2-layer model on top pins the oracle on a real graph."""
    import scipy.sparse as sp
    d = golden("g5_chess_gcn2")
    T, N = int(d["T"]), int(d["N"])
    raw = [sp.coo_matrix((np.ones((d["raw_k"] == t).sum()), (d["raw_i"][d["raw_k"] == t], d["raw_j"][d["raw_k"] == t])),
                         shape=(N, N)).tocsr() for t in range(T)]
    C = synth.normalise(synth.edge_life(synth.symmetrise(raw), 10))
    M = synth.band_M(T, 20, "python")
    assert_close(M, d["M"], 1e-15, "band M (read_data.py:55-62)")
    Ct = synth.m_product(C, M)

    def dense(k, i, j, v):
        out = np.zeros((T, N, N))
        np.add.at(out, (k, i, j), v)
        return out

    assert_close(np.stack([c.toarray() for c in C]), dense(d["C_k"], d["C_i"], d["C_j"], d["C_v"]), 1e-12, "normalised adjacency")
    assert_close(np.stack([c.toarray() for c in Ct]), dense(d["At_k"], d["At_i"], d["At_j"], d["At_v"]), 1e-12, "M-product of A")
    assert_close(synth.node_features(raw), d["X"], 0.0, "node features")

    At = coo_list(d, "At", T, N)
    X, Mt = torch.from_numpy(d["X"]), torch.from_numpy(d["M"])
    torch.manual_seed(int(d["seed"]))
    p = orc.draw_params("gcn2", T, [2, 6, 6, 2])
    AtXt = orc.compute_AtXt(Mt, At, X)
    src, dst = orc.flat_edge_index(torch.from_numpy(d["edges"]), N)
    out, loss, g = _loss_grads(lambda q: orc.gcn2_forward(AtXt, At, Mt, q["W1"], q["W2"], q["U"], src, dst, nonlin="selu"),
                               p, torch.from_numpy(d["labels"]))
    assert_close(out, d["logits"], 1e-6, "chess logits")
    for k in ("W1", "W2", "U"):
        assert_close(g[k], d["d" + k], 2e-6, "chess d" + k)


@pytest.mark.parametrize("kind", ["gcn", "gcn2"])
def test_g6_sgd_trajectory(kind):
    d = golden("g6_sgd_" + kind)
    i = _inputs(d)
    torch.manual_seed(int(d["seed"]))
    p = orc.draw_params(kind, i["T"], [2, 6, 2] if kind == "gcn" else [2, 6, 6, 2])
    ps = {k: torch.nn.Parameter(v) for k, v in p.items()}
    AtXt = orc.compute_AtXt(i["M"], i["At"], i["X"])
    src, dst = orc.flat_edge_index(i["edges"], i["N"])
    opt = torch.optim.SGD(list(ps.values()), lr=0.01, momentum=0.9)
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))
    losses = []
    for _ in range(10):
        opt.zero_grad()
        if kind == "gcn":
            out = orc.gcn_forward(AtXt, ps["W"], ps["U"], src, dst)
        else:
            out = orc.gcn2_forward(AtXt, i["At"], i["M"], ps["W1"], ps["W2"], ps["U"], src, dst, nonlin="selu")
        loss = crit(out, i["labels"])
        loss.backward()
        opt.step()
        losses.append(float(loss))
    assert_close(np.array(losses), d["losses"], 1e-5, kind + " losses")
    for k, v in ps.items():
        assert_close(v.detach(), d[k + "_final"], 1e-5, kind + " final " + k)


def test_S0_sbm_plumbing_config_on_cpu():
    """BASELINE config 0 (SBM, T=10, N=500, F=16, CPU only): the oracle's 2-layer model on the SBM
    stand-in agrees with the dense fp64 einsum form of the same math (no GPU involved)."""
    g = synth.sbm_dynamic_graph()
    At, X, M = g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.M)
    edges, labels = torch.from_numpy(g.edges), torch.from_numpy(g.labels)
    torch.manual_seed(0)
    p = {k: v * 0.1 for k, v in orc.draw_params("gcn2", g.T, [16, 6, 6, 2]).items()}
    src, dst = orc.flat_edge_index(edges, g.N)
    out = orc.gcn2_forward(orc.compute_AtXt(M, At, X), At, M, p["W1"], p["W2"], p["U"], src, dst, nonlin="selu",
                           apply_M_twice=True)
    A = torch.stack([a.to_dense() for a in At])
    mt = lambda Z: torch.einsum("kj,jnf->knf", M, Z)
    spm = lambda Z: torch.einsum("knm,kmf->knf", A, Z)
    Y = torch.nn.functional.selu(spm(mt(X)) @ p["W1"].double())
    Z = (spm(mt(Y)) @ p["W2"].double()).reshape(-1, 6)
    ref = torch.cat((Z[src], Z[dst]), 1) @ p["U"].double()
    assert_close(out, ref, 1e-5, "S0 SBM 2-layer logits")


@pytest.mark.parametrize("name", golden_names("g8_"))
def test_g8_gcn_reg(name):
    d = golden(name)
    i = _inputs(d)
    W = torch.from_numpy(d["W0"]).requires_grad_(True)
    lw = torch.from_numpy(d["lin1_weight0"]).requires_grad_(True)
    lb = torch.from_numpy(d["lin1_bias0"]).requires_grad_(True)
    out = orc.gcn_reg_forward(orc.compute_AtXt(i["M"], i["At"], i["X"]), W, lw, lb)
    loss = torch.nn.MSELoss()(out, torch.from_numpy(d["y"]))
    loss.backward()
    assert_close(out.detach(), d["out"], 1e-6, name)
    assert_close(W.grad, d["dW"], 2e-6, name + " dW")
    assert_close(lw.grad, d["dlin1_weight"], 2e-6, name + " dlin1.weight")
    assert_close(lb.grad, d["dlin1_bias"], 2e-6, name + " dlin1.bias")


def test_g10_full_chess_pipeline_and_models():
    """Fixture G10: the reference's own read_data.py functions on the WHOLE chess data set (7 301 players, 100
    slices, T = 80 training slices > the 20 diagonals of M, so the band is truncated — G5 at T = 16 never was),
    then experiment_chess_our.py's models.  Pins (a) synth's restated preprocessing — pattern entry for entry,
    values to fp32 rounding of the stored fp64->fp32 values, fp64 slice sums — and (b) the oracle's 1- and
    2-layer models incl. the `apply_M_twice` branch and the script's validation call, at real scale."""
    import scipy.sparse as sp
    from _g10 import G10
    g = G10()
    d, TT, T, N = g.d, g.TT, g.T, g.N
    k, i, j = g.raw
    raw = [sp.coo_matrix((np.ones(int((k == t).sum())), (i[k == t], j[k == t])), shape=(N, N)).tocsr() for t in range(TT)]
    C = synth.normalise(synth.edge_life(synth.symmetrise(raw), 10))
    M = synth.band_M(T, 20, "python")
    assert_close(M, g.M, 1e-15, "band M (read_data.py:55-62)")
    assert (np.count_nonzero(M, axis=1) == np.minimum(np.arange(T) + 1, 20)).all() and T > 20   # truncated band

    def check(mats, name, Tn):
        rk, ri, rj, rv = unpack_sym(d, name, Tn, N)
        coo = [m.tocsr().sorted_indices().tocoo() for m in mats]
        mk = np.concatenate([np.full(c.nnz, t, np.int64) for t, c in enumerate(coo)])
        mi, mj = np.concatenate([c.row for c in coo]).astype(np.int64), np.concatenate([c.col for c in coo]).astype(np.int64)
        mv = np.concatenate([c.data for c in coo])
        assert len(mv) == len(rv) == int(d[name + "_nnz"]), name + ": number of stored entries"
        assert np.array_equal(mk, rk) and np.array_equal(mi, ri) and np.array_equal(mj, rj), name + ": pattern"
        assert float(np.abs(mv - rv).max()) <= 1e-7 * float(np.abs(rv).max()), name + ": values (fixture holds fp32)"
        sums = np.bincount(mk, weights=mv, minlength=Tn)
        assert np.allclose(sums, d[name + "_slice_sum"], rtol=1e-12, atol=0), name + ": fp64 slice sums"

    check(C, "C", TT)
    Ct = synth.m_product(C[:T], M)                                    # func_create_sparse(0, T) then func_MProduct
    check(Ct, "Ct", T)
    Ct_val = synth.m_product(C[g.S_val:g.S_val + T], M)               # the validation block (read_data.py:187, 226)
    assert sum(c.nnz for c in Ct_val) == int(d["Ct_val_nnz"])
    assert np.allclose([c.sum() for c in Ct_val], d["Ct_val_slice_sum"], rtol=1e-12, atol=0)

    # the models (experiment_chess_our.py:92-103): 3 classes, weights .33
    At, At_val = synth.to_coo_list(Ct), synth.to_coo_list(Ct_val)
    Mt = torch.from_numpy(g.M)
    X, Xv = torch.from_numpy(g.X_train), torch.from_numpy(g.X_val)
    src, dst = orc.flat_edge_index(torch.from_numpy(g.edges_train), N)
    vs, vd = orc.flat_edge_index(torch.from_numpy(g.edges_val), N)
    tgt = torch.from_numpy(g.target_train)
    crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g.class_weights))
    AtXt = orc.compute_AtXt(Mt, At, X)
    AtXt_val = orc.compute_AtXt(Mt, At_val, Xv)
    ev = torch.from_numpy(g.eval_val)
    # the C oracle's P2 / P3 halves at real scale (G2 pins them at 60 nodes): ref_spmm on the reference's Ât and ref_gemm
    # (fp64 accumulation, where ehf:222 is an fp32 matmul) reproduce the 1-layer model's logits on all 584 k rows
    lib = load_c_oracle()
    csr = BatchedCSR.from_coo_list(At, N=N)
    Xt = orc.m_transform(Mt, X).float().contiguous()
    AtXt_c = torch.empty(T, N, 2)
    lib.ref_spmm(cptr(csr.rowptr), cptr(csr.col), cptr(csr.val), cptr(Xt), cptr(AtXt_c), T * N, N, 2)
    assert_close(AtXt_c, AtXt, 2e-6, "C oracle P2 at N = 7301")
    W0 = torch.from_numpy(d["gcn_W0"]).contiguous()
    Yc = torch.empty(T, N, W0.shape[1])
    lib.ref_gemm(cptr(AtXt.contiguous()), cptr(W0), cptr(Yc), T * N, 2, W0.shape[1], 0, 0, 0)
    with torch.no_grad():
        assert_close(orc.edge_head(Yc, src, dst, torch.from_numpy(d["gcn_U0"])), d["gcn_logits"], 1e-6, "logits through ref_gemm at N = 7301")
    for name, kind, F, kw in (("gcn", "gcn", [2, 6, 3], {}), ("gcn2", "gcn2", [2, 6, 6, 3], dict(nonlin="selu")),
                              ("gcn2_twice", "gcn2", [2, 6, 6, 3], dict(nonlin="selu", apply_M_twice=True))):
        torch.manual_seed(int(d["seed"]))
        p = orc.draw_params(kind, T, F)
        for n in p:
            assert np.array_equal(p[n].numpy(), d[f"{name}_{n}0"]), (name, n)
        if kind == "gcn":
            fwd = lambda q, a=AtXt, s=src, t=dst: orc.gcn_forward(a, q["W"], q["U"], s, t)
        else:
            fwd = lambda q, a=AtXt, s=src, t=dst: orc.gcn2_forward(a, At, Mt, q["W1"], q["W2"], q["U"], s, t, **kw)
        ps = {n: v.clone().requires_grad_(True) for n, v in p.items()}
        out = fwd(ps)
        loss = crit(out, tgt)
        loss.backward()
        loss = loss.detach()
        assert_close(out.detach(), d[name + "_logits"], 1e-6, name + " logits")
        assert abs(float(loss) - float(d[name + "_loss"])) <= 1e-6 * max(1.0, abs(float(d[name + "_loss"])))
        for n in p:
            assert_close(ps[n].grad, d[f"{name}_d{n}"], 2e-6, f"{name} d{n}")
        if name != "gcn2_twice":
            with torch.no_grad():
                out_val = fwd(p, AtXt_val, vs, vd)
            assert_close(out_val[ev], d[name + "_logits_val_eval"], 1e-6, name + " validation logits (last S_val slices)")


def test_g10_full_chess_baseline_kwgcn():
    """G10, the baseline of experiment_chess_baseline.py: EmbeddingKWGCN (1 and 2 layers) on the reference's un-transformed
    C of the whole chess data set, training on slices 0..79, the validation call on the SHORTER window 80..89
    (compute_AX zero-pads to the training T, ehf:469-473) — row a5 at real scale."""
    from _g10 import G10
    g = G10()
    d, T, N = g.d, g.T, g.N
    rk, ri, rj, rv = g.C()
    bounds = np.searchsorted(rk, np.arange(g.TT + 1))

    def slice_list(t0, t1):
        out = []
        for t in range(t0, t1):
            sl = slice(bounds[t], bounds[t + 1])
            out.append(torch.sparse_coo_tensor(torch.from_numpy(np.stack([ri[sl], rj[sl]])), torch.from_numpy(rv[sl].astype(np.float64)), (N, N)))
        return out

    A, A_val = slice_list(0, T), slice_list(T, T + g.S_val)
    X, Xv = torch.from_numpy(g.X_train), torch.from_numpy(g.X_val_b)
    src, dst = orc.flat_edge_index(torch.from_numpy(g.edges_train), N)
    vs, vd = orc.flat_edge_index(torch.from_numpy(g.edges_val_b), N)
    tgt = torch.from_numpy(g.target_train)
    crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g.class_weights))
    AX = orc.slice_spmm(A, X)
    AX_val = torch.cat((orc.slice_spmm(A_val, Xv), torch.zeros(T - g.S_val, N, 2)))      # ehf:469-473
    for name, F in (("kw1", [2, 6, 3]), ("kw2", [2, 6, 6, 3])):
        torch.manual_seed(int(d["seed"]))
        p = orc.draw_params("kw", T, F)
        for n in p:
            assert np.array_equal(p[n].numpy(), d[f"{name}_{n}0"]), (name, n)
        ps = {n: v.clone().requires_grad_(True) for n, v in p.items()}
        out = orc.kwgcn_forward(AX, A, ps["W1"], ps["U"], src, dst, ps.get("W2"), "selu")
        loss = crit(out, tgt)
        loss.backward()
        assert_close(out.detach(), d[name + "_logits"], 2e-6, name + " logits")      # the fixture's C values are fp32-rounded
        assert abs(float(loss.detach()) - float(d[name + "_loss"])) <= 2e-6 * max(1.0, abs(float(d[name + "_loss"])))
        for n in p:
            assert_close(ps[n].grad, d[f"{name}_d{n}"], 5e-6, f"{name} d{n}")
        with torch.no_grad():
            out_val = orc.kwgcn_forward(AX_val, A, p["W1"], p["U"], vs, vd, p.get("W2"), "selu")
        assert_close(out_val, d[name + "_logits_val"], 2e-6, name + " validation logits (shorter window)")


def test_g11_oracle_follows_the_reference_training_run_early_and_late():
    """Fixture G11: experiment_chess_our.py's loop (:108-123 — SGD lr .01 momentum .9, class-weighted CE, 2-layer model)
    run for 300 epochs with the REAL ehf.EmbeddingGCN2 on the full chess data.  The oracle is pinned on the run where the
    weights have moved, not only at initialisation: the first 12 epochs from the seed, and the last 20 epochs resumed from
    the fixture's state in front of epoch 280 (parameters + momentum buffers) — every loss, the script's validation
    numbers of epoch 299 and the final parameters.  (The whole run is walked on the device: tests/test_gpu_g11_*.)"""
    import scipy.sparse as sp
    from _g10 import G10
    g = G10()
    d = golden("g11_chess_train300")
    T, N = g.T, g.N
    k, i, j = g.raw
    raw = [sp.coo_matrix((np.ones(int((k == t).sum())), (i[k == t], j[k == t])), shape=(N, N)).tocsr() for t in range(g.TT)]
    C = synth.normalise(synth.edge_life(synth.symmetrise(raw), 10))
    M = synth.band_M(T, 20, "python")
    At = synth.to_coo_list(synth.m_product(C[:T], M))
    At_val = synth.to_coo_list(synth.m_product(C[g.S_val:g.S_val + T], M))
    Mt = torch.from_numpy(g.M)
    AtXt = orc.compute_AtXt(Mt, At, torch.from_numpy(g.X_train))
    AtXt_val = orc.compute_AtXt(Mt, At_val, torch.from_numpy(g.X_val))
    src, dst = orc.flat_edge_index(torch.from_numpy(g.edges_train), N)
    vs, vd = orc.flat_edge_index(torch.from_numpy(g.edges_val), N)
    tgt, tgt_val, ev = torch.from_numpy(g.target_train), torch.from_numpy(g.target_val), torch.from_numpy(g.eval_val)
    crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g.class_weights))
    names = ("W1", "W2", "U")

    def run(params, epochs, momentum=None):
        ps = [torch.nn.Parameter(params[n].clone()) for n in names]
        opt = torch.optim.SGD(ps, lr=float(d["lr"]), momentum=float(d["momentum"]))
        if momentum is not None:
            for q, n in zip(ps, names):
                opt.state[q]["momentum_buffer"] = momentum[n].clone()
        losses = []
        for _ in range(epochs):
            opt.zero_grad()
            out = orc.gcn2_forward(AtXt, At, Mt, ps[0], ps[1], ps[2], src, dst, nonlin="selu")
            loss = crit(out, tgt)
            loss.backward()
            opt.step()
            losses.append(float(loss))
        return ps, np.array(losses), out.detach()

    torch.manual_seed(int(d["seed"]))
    p0 = orc.draw_params("gcn2", T, [2, 6, 6, 3])
    _, early, _ = run(p0, 12)
    assert_close(early, d["losses"][:12], 1e-6, "G11 losses of epochs 0-11 from the seed")
    ck = {n: torch.from_numpy(d[f"ckpt280_{n}"]) for n in names}
    mom = {n: torch.from_numpy(d[f"ckpt280_mom_{n}"]) for n in names}
    ps, late, out = run(ck, 20, mom)
    assert_close(late, d["losses"][280:], 1e-6, "G11 losses of epochs 280-299 resumed from the fixture's state")
    for q, n in zip(ps, names):
        assert_close(q.detach(), d[f"{n}_final"], 1e-6, f"G11 {n} after epoch 299")
    mark = {c: v for c, v in zip(d["marks_columns"], d["marks"][-1])}
    assert int(mark["epoch"]) == 299
    with torch.no_grad():
        guess = out.argmax(1)
        assert abs(int((guess == tgt).sum()) / len(tgt) - mark["acc_train"]) < 1e-12
        out_val = orc.gcn2_forward(AtXt_val, At, Mt, ps[0], ps[1], ps[2], vs, vd, nonlin="selu")    # layer 2 on the TRAINING adjacency (ehf:348)
        gv = out_val.argmax(1)
        assert abs(int((gv[ev] == tgt_val[ev]).sum()) / int(ev.sum()) - mark["acc_val"]) < 1e-12
        assert abs(float(crit(out_val[ev], tgt_val[ev])) - mark["loss_val"]) <= 1e-6 * mark["loss_val"]
        assert np.array_equal(np.bincount(gv[ev].numpy(), minlength=3), [mark[f"val_argmax_{c}"] for c in range(3)])

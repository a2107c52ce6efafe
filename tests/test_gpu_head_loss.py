"""GPU: the one-pass edge head + class-weighted cross entropy + gradients (csrc/head_loss.hip, ops.head_loss,
`model.loss(criterion, target)`) against (a) a torch fp64 restatement of the scripts' statements
(ehf:228-232 + nn.CrossEntropyLoss(weight=...), autograd: tests/_util.head_loss_fp64, which the CPU suite pins on the real
reference's G2 numbers), (b) the unfused kernels it replaces, (c) the
reference's own numbers (fixtures G2, G3, G10)."""
import numpy as np
import pytest
import torch

from _util import REL_TOL, assert_close, coo_list, golden, head_loss_fp64
import tmgcn_amd.layers as ehf
from tmgcn_amd import ops
from tmgcn_amd.losses import WeightedCrossEntropy, weighted_ce

pytestmark = pytest.mark.gpu


def _problem(T, N, F, C, E, seed, ignore_frac=0.0, K=0):
    g = torch.Generator().manual_seed(seed)
    Z = torch.randn(T, N, K if K else F, generator=g).cuda()
    W = (torch.randn(K, F, generator=g) * 0.7).cuda() if K else None
    U = torch.randn(2 * F, C, generator=g).cuda()
    edges = torch.stack([torch.randint(0, T, (E,), generator=g), torch.randint(0, N, (E,), generator=g),
                         torch.randint(0, N, (E,), generator=g)])
    target = torch.randint(0, C, (E,), generator=g)
    if ignore_frac:
        target[torch.rand(E, generator=g) < ignore_frac] = -100
    weight = torch.rand(C, generator=g) + 0.1
    return Z, W, U, edges, target.cuda(), weight.cuda()


_fp64 = head_loss_fp64          # tests/_util.py: the fp64 restatement, pinned on fixture G2 by the CPU suite


@pytest.mark.parametrize("T,N,F,C,E,ign", [(5, 40, 6, 2, 3000, 0.0),     # the reference's head, ~30 entries per row
                                           (3, 500, 6, 2, 700, 0.1),     # most rows without an entry; ignored labels
                                           (7, 33, 2, 1, 900, 0.0), (4, 64, 4, 3, 5000, 0.05), (2, 17, 8, 4, 2500, 0.0),
                                           (1, 9, 6, 3, 1, 0.0),         # one edge
                                           (6, 300, 6, 2, 200000, 0.0)])  # hub-free, 220 entries per row: 32-lane groups
def test_head_loss_vs_fp64_and_unfused(T, N, F, C, E, ign):
    Z, _, U, edges, target, weight = _problem(T, N, F, C, E, seed=T * 100 + F * 10 + C, ignore_frac=ign)
    idx = ops.EdgeIndex(edges, N, "cuda", T=T)
    Zr, Ur = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    loss, logits = ops.head_loss(Zr, idx, Ur, target, weight, want_logits=True)
    loss.backward()
    ref_logits, ref_loss, ref_dZ, ref_dU, _ = _fp64(Z, None, U, edges, target, weight, N)
    assert_close(logits, ref_logits, 1e-6, "logits")
    assert abs(float(loss) - float(ref_loss)) <= 1e-6 * max(1.0, abs(float(ref_loss)))
    assert_close(Zr.grad, ref_dZ, 2e-6, "dZ")
    assert_close(Ur.grad, ref_dU, 2e-6, "dU")
    # the kernels it replaces: same logits bit for bit (same fmaf chain), same loss / gradients to rounding
    Z2, U2 = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    lg2 = ops.edge_head(Z2, idx, U2)
    l2 = weighted_ce(lg2, target, weight)
    l2.backward()
    assert torch.equal(logits, lg2.detach())
    assert abs(float(loss) - float(l2)) <= 2e-6 * max(1.0, abs(float(l2)))
    assert_close(Zr.grad, Z2.grad, 5e-6, "dZ vs unfused")
    assert_close(Ur.grad, U2.grad, 5e-6, "dU vs unfused")
    # reproducible bit for bit, and the hand-off word is left zero
    Z3, U3 = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    l3 = ops.head_loss(Z3, idx, U3, target, weight)
    l3.backward()
    assert torch.equal(l3.detach(), loss.detach()) and torch.equal(Z3.grad, Zr.grad) and torch.equal(U3.grad, Ur.grad)
    assert int(ops.head_loss_plan(idx, T * N, target, C).sync.abs().sum()) == 0      # every hand-off counter back at zero
    # loss only (no_grad): the same value
    with torch.no_grad():
        assert torch.equal(ops.head_loss(Z, idx, U, target, weight), loss.detach())


@pytest.mark.parametrize("T,N,F,C,E", [(5, 40, 6, 2, 3000), (3, 200, 6, 3, 900), (4, 31, 8, 4, 4000), (2, 50, 2, 2, 100)])
def test_head_loss_fold_layer1_gemm(T, N, F, C, E):
    """K = 2: the kernel gathers AtXt rows and sees the head W·U; dW and dU come from the per-row sums; nothing of
    size [T,N,F] is read or stored."""
    X, W, U, edges, target, weight = _problem(T, N, F, C, E, seed=7 * T + F, K=2)
    idx = ops.EdgeIndex(edges, N, "cuda", T=T)
    Wr, Ur = W.clone().requires_grad_(True), U.clone().requires_grad_(True)
    loss, logits = ops.head_loss(X, idx, Ur, target, weight, want_logits=True, fold_W=Wr)
    loss.backward()
    ref_logits, ref_loss, _, ref_dU, ref_dW = _fp64(X, W, U, edges, target, weight, N)
    assert_close(logits, ref_logits, 2e-6, "logits")
    assert abs(float(loss) - float(ref_loss)) <= 2e-6 * max(1.0, abs(float(ref_loss)))
    assert_close(Wr.grad, ref_dW, 3e-6, "dW")
    assert_close(Ur.grad, ref_dU, 3e-6, "dU")
    # the unfused route: standalone small GEMM, head, loss (the folded kernel re-associates (x·W)·U = x·(W·U))
    W2, U2 = W.clone().requires_grad_(True), U.clone().requires_grad_(True)
    lg2 = ops.edge_head(ops.feature_gemm(X, W2), idx, U2)
    weighted_ce(lg2, target, weight).backward()
    assert_close(logits, lg2.detach(), 2e-6, "logits vs unfused")
    assert_close(Wr.grad, W2.grad, 5e-6, "dW vs unfused")


def test_upstream_gradient_scales_and_bad_labels_raise():
    Z, _, U, edges, target, weight = _problem(4, 30, 6, 2, 800, seed=3)
    idx = ops.EdgeIndex(edges, 30, "cuda", T=4)
    Za, Ua = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    ops.head_loss(Za, idx, Ua, target, weight).backward()
    Zb, Ub = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    (ops.head_loss(Zb, idx, Ub, target, weight) * -2.5).backward()
    assert_close(Zb.grad, -2.5 * Za.grad, 1e-7, "scaled dZ")
    assert_close(Ub.grad, -2.5 * Ua.grad, 1e-7, "scaled dU")
    bad = target.clone()
    bad[5] = 7
    with pytest.raises(RuntimeError, match="outside"):
        ops.head_loss(Z, idx, U, bad, weight)
    # a target edited in place is noticed (version counter), not served from the cached plan
    t2 = target.clone()
    l1 = float(ops.head_loss(Z, idx, U, t2, weight))
    t2[:400] = 1 - t2[:400]
    l2 = float(ops.head_loss(Z, idx, U, t2, weight))
    assert abs(l2 - float(weighted_ce(ops.edge_head(Z, idx, U), t2, weight))) <= 2e-6 * max(1.0, abs(l2)) and l1 != l2


def test_wide_head_falls_back_to_the_unfused_kernels():
    Z, _, U, edges, target, weight = _problem(3, 50, 16, 2, 600, seed=5)
    idx = ops.EdgeIndex(edges, 50, "cuda", T=3)
    assert not ops.head_loss_supported(16, 2)
    Zr = Z.clone().requires_grad_(True)
    loss, logits = ops.head_loss(Zr, idx, U, target, weight, want_logits=True)
    loss.backward()
    ref_logits, ref_loss, ref_dZ, _, _ = _fp64(Z, None, U, edges, target, weight, 50)
    assert_close(logits, ref_logits, 1e-5, "logits")
    assert_close(Zr.grad, ref_dZ, 1e-5, "dZ")


def _grads(model, loss):
    model.zero_grad()
    loss.backward()
    return {n: p.grad.clone() for n, p in model.named_parameters()}


@pytest.mark.parametrize("name", ["g2_gcn_condensed1", "g2_gcn_condensed0", "g3_gcn2_default_selu_condensed1",
                                  "g3_gcn2_twice_relu_condensed1"])
@pytest.mark.parametrize("crit_kind", ["torch", "fused", "tensor"])
def test_model_loss_reproduces_the_reference(name, crit_kind):
    """`gcn.loss(criterion, target)` == the reference's `criterion(gcn(), target)` and its gradients (G2: the
    folded 1-layer form and the per-slice-W form that cannot fold; G3: 2-layer)."""
    d = golden(name)
    X = torch.from_numpy(d["X"])
    T, N = X.shape[0], X.shape[1]
    At, M, edges, labels = coo_list(d, "At", T, N), torch.from_numpy(d["M"]), torch.from_numpy(d["edges"]), torch.from_numpy(d["labels"])
    torch.manual_seed(int(d["seed"]))
    if name.startswith("g2"):
        m = ehf.EmbeddingGCN(At, X, edges, M, hidden_feat=[6, 2], condensed_W=d["W0"].ndim == 2, use_Minv=False)
    else:
        _, _, branch, nl, cond = name.split("_")
        m = ehf.EmbeddingGCN2(At, X, edges, M, hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False,
                              apply_M_twice=branch == "twice", nonlin2=nl)
    w = torch.tensor([0.9, 0.1])
    crit = {"torch": torch.nn.CrossEntropyLoss(weight=w), "fused": WeightedCrossEntropy(w), "tensor": w}[crit_kind]
    loss, logits = m.loss(crit, labels.cuda(), want_logits=True)
    g = _grads(m, loss)
    assert_close(logits, d["logits"], REL_TOL, name + " logits")
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    for n in g:
        assert_close(g[n], d["d" + n], REL_TOL, f"{name} d{n}")


def test_model_loss_on_full_chess_three_classes():
    """G10 (C = 3, E = 52 k, R = 584 k rows — most without a labelled edge): the reference's loss and gradients."""
    from _g10 import G10
    from tmgcn_amd import adjacency
    g = G10()
    k, i, j = g.raw
    Chat, _ = adjacency.build_adjacency(k, i, j, np.ones(len(k), np.float32), g.TT, g.N, M=None, window=10)
    A = adjacency.m_product_csr(Chat.slices(0, g.T), g.M)
    for name, ctor in (("gcn", lambda: ehf.EmbeddingGCN(A, torch.from_numpy(g.X_train), torch.from_numpy(g.edges_train),
                                                        torch.from_numpy(g.M), hidden_feat=[6, 3], condensed_W=True, use_Minv=False)),
                       ("gcn2", lambda: ehf.EmbeddingGCN2(A, torch.from_numpy(g.X_train), torch.from_numpy(g.edges_train),
                                                          torch.from_numpy(g.M), hidden_feat=[6, 6, 3], condensed_W=True,
                                                          use_Minv=False, nonlin2="selu"))):
        torch.manual_seed(int(g.d["seed"]))
        m = ctor()
        crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g.class_weights))
        loss, logits = m.loss(crit, torch.from_numpy(g.target_train).cuda(), want_logits=True)
        gr = _grads(m, loss)
        assert_close(logits, g.d[name + "_logits"], REL_TOL, name + " logits")
        assert abs(float(loss) - float(g.d[name + "_loss"])) <= 1e-5 * max(1.0, abs(float(g.d[name + "_loss"])))
        for n in gr:
            assert_close(gr[n], g.d[f"{name}_d{n}"], REL_TOL, f"{name} d{n}")


def test_graphed_step_with_fused_loss_follows_the_eager_trajectory():
    from tmgcn_amd.graphs import GraphedTrainStep
    d = golden("g6_sgd_gcn2")
    X = torch.from_numpy(d["X"])
    T, N = X.shape[0], X.shape[1]
    At, M, edges = coo_list(d, "At", T, N), torch.from_numpy(d["M"]), torch.from_numpy(d["edges"])
    tgt = torch.from_numpy(d["labels"]).cuda()
    w = torch.tensor([0.9, 0.1])

    def make():
        torch.manual_seed(int(d["seed"]))
        m = ehf.EmbeddingGCN2(At, X, edges, M, hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")
        return m, torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)

    m1, o1 = make()
    crit = torch.nn.CrossEntropyLoss(weight=w.cuda())
    losses = []
    for _ in range(10):
        o1.zero_grad()
        l = crit(m1(), tgt)
        l.backward()
        o1.step()
        losses.append(float(l))
    assert_close(np.array(losses), d["losses"], REL_TOL, "eager trajectory vs the reference")
    m2, o2 = make()
    step = GraphedTrainStep(m2, WeightedCrossEntropy(w).cuda(), o2, tgt, warmup=3)
    assert step.fused
    got = [float(step()) for _ in range(7)]
    assert_close(np.array(got), d["losses"][3:], REL_TOL, "graph + fused loss, steps 4-10")
    # several steps per replay (one launch latency for all of them): the same trajectory, every loss kept; both the
    # fused one-launch SGD and torch's own optimizer
    from tmgcn_amd.optim import FusedSGD
    for fused_opt in (False, True):
        m3, o3 = make()
        if fused_opt:
            o3 = FusedSGD(m3.parameters(), lr=0.01, momentum=0.9)
        step3 = GraphedTrainStep(m3, WeightedCrossEntropy(w).cuda(), o3, tgt, warmup=1, steps_per_replay=3)
        got3 = []
        for _ in range(3):
            last = step3()
            assert last is step3.losses[-1] and len(step3.losses) == 3
            got3 += [float(l) for l in step3.losses]
        assert_close(np.array(got3), d["losses"][1:], REL_TOL, "three steps per replay, steps 2-10")
        for q2, q3 in zip(m2.parameters(), m3.parameters()):
            assert_close(q3.detach(), q2.detach(), REL_TOL, "parameters after ten steps")
    with pytest.raises(ValueError):
        GraphedTrainStep(m2, WeightedCrossEntropy(w).cuda(), o2, tgt, steps_per_replay=0)
    # the one-pass route keeps no logits: asking for them is an error that says so, never a silent None
    with pytest.raises(RuntimeError, match="keep_logits"):
        step.output
    m4, o4 = make()
    step4 = GraphedTrainStep(m4, WeightedCrossEntropy(w).cuda(), o4, tgt, warmup=3, keep_logits=True)
    got4 = [float(step4()) for _ in range(7)]
    assert_close(np.array(got4), d["losses"][3:], REL_TOL, "graph + fused loss + kept logits, steps 4-10")
    assert step4.output.shape == (tgt.numel(), 2) and step4.output.device.type == "cuda"
    # capturing the momentum-buffer-creating first step would reset the momentum on every replay: refused
    m5, o5 = make()
    with pytest.raises(ValueError, match="momentum"):
        GraphedTrainStep(m5, WeightedCrossEntropy(w).cuda(), o5, tgt, warmup=0)


@pytest.mark.parametrize("T,N,E", [(4, 30, 800), (6, 2000, 300)])       # dense entries / most rows without an entry
def test_unit_gradient_fast_path(T, N, E):
    """unit_grad=True + loss.backward(gradient=ops.unit_gradient(dev)): loss and gradients from ONE launch, bit-equal to
    the default schedules; the promise is only a hint — another upstream gradient is still multiplied in."""
    Z, _, U, edges, target, weight = _problem(T, N, 6, 2, E, seed=11)
    idx = ops.EdgeIndex(edges, N, "cuda", T=T)
    Za, Ua = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    la = ops.head_loss(Za, idx, Ua, target, weight)
    la.backward()
    Zb, Ub = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    lb = ops.head_loss(Zb, idx, Ub, target, weight, unit_grad=True)
    lb.backward(gradient=ops.unit_gradient(Z.device))
    assert torch.equal(la.detach(), lb.detach())
    assert_close(Zb.grad, Za.grad, 1e-7, "dZ")          # deferred vs speculative schedule: same sums, one rounding apart at most
    assert_close(Ub.grad, Ua.grad, 1e-7, "dU")
    Zc, Uc = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    (ops.head_loss(Zc, idx, Uc, target, weight, unit_grad=True) * 3.0).backward()
    assert_close(Zc.grad, 3.0 * Za.grad, 1e-6, "dZ x 3")
    assert_close(Uc.grad, 3.0 * Ua.grad, 1e-6, "dU x 3")


def test_hub_rows_and_skewed_degrees():
    """One node is an endpoint of a third of all labelled edges in its slice (a row with ~10 k entries next to rows with
    one or none): the lanes of its group walk thousands of entries each; sums stay in fp64, result within 2e-6."""
    T, N, F, C, E = 3, 400, 6, 2, 30000
    g = torch.Generator().manual_seed(21)
    Z = torch.randn(T, N, F, generator=g).cuda()
    U = torch.randn(2 * F, C, generator=g).cuda()
    t = torch.randint(0, T, (E,), generator=g)
    src = torch.randint(0, N, (E,), generator=g)
    dst = torch.randint(0, N, (E,), generator=g)
    hub = torch.rand(E, generator=g) < 0.33
    src[hub] = 7                                           # node 7 of every slice is a hub on the src side
    dst[torch.rand(E, generator=g) < 0.1] = 7              # ... and sometimes on the dst side (self-pairs included)
    edges = torch.stack([t, src, dst])
    target = torch.randint(0, C, (E,), generator=g).cuda()
    weight = torch.tensor([0.9, 0.1]).cuda()
    idx = ops.EdgeIndex(edges, N, "cuda", T=T)
    Zr, Ur = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
    loss, logits = ops.head_loss(Zr, idx, Ur, target, weight, want_logits=True)
    loss.backward()
    ref_logits, ref_loss, ref_dZ, ref_dU, _ = _fp64(Z, None, U, edges, target, weight, N)
    assert_close(logits, ref_logits, 1e-6, "logits")
    assert abs(float(loss.detach()) - float(ref_loss)) <= 1e-6 * max(1.0, abs(float(ref_loss)))
    assert_close(Zr.grad, ref_dZ, 2e-6, "dZ")
    assert_close(Ur.grad, ref_dU, 2e-6, "dU")


@pytest.mark.parametrize("T,N,E", [(2, 300, 60_000), (50, 2000, 60_000)])       # gradients in the loss launch / in the backward's launch
@pytest.mark.parametrize("K,C", [(0, 2), (0, 3), (2, 2)])
def test_rows_of_hub_nodes_are_split_into_parts(T, N, E, K, C):
    """Hubs of the LABELLED edges (one node incident to 40 % of a slice's edges: rows of ~12 000 entries next to rows of one):
    the plan cuts rows that would take their lanes more than about eight trips into parts (ops.HeadLossPlan -> arow's 4th column, tmgcn_head_loss_combine_f32),
    every sum of the kernel being linear in the entries.  Loss, logits, dZ (or the folded model's dW) and dU against the
    scripts' statements in fp64; reproducible to the bit."""
    F = 6
    g = torch.Generator().manual_seed(T * 31 + K + C)
    Z = torch.randn(T, N, K if K else F, generator=g).cuda()
    W = (torch.randn(K, F, generator=g) * 0.7).cuda() if K else None
    U = torch.randn(2 * F, C, generator=g).cuda()
    t = torch.randint(0, T, (E,), generator=g)
    src, dst = torch.randint(0, N, (E,), generator=g), torch.randint(0, N, (E,), generator=g)
    src[torch.rand(E, generator=g) < 0.4] = 11                     # node 11 of every slice: a hub on the src side …
    dst[torch.rand(E, generator=g) < 0.05] = 11                    # … sometimes on the dst side, self-pairs included
    dst[torch.rand(E, generator=g) < 0.02] = 5                     # a second, smaller hub (rows of 300-600 entries: two or three parts)
    edges = torch.stack([t, src, dst])
    target = torch.randint(0, C, (E,), generator=g)
    target[torch.rand(E, generator=g) < 0.03] = -100
    target = target.cuda()
    weight = (torch.rand(C, generator=g) + 0.1).cuda()
    idx = ops.EdgeIndex(edges, N, "cuda", T=T)
    plan = ops.head_loss_plan(idx, T * N, target, C)
    assert plan.srow is not None and plan.n_parts >= 2 * T and int(plan.arow[:, 3].max()) == plan.n_parts
    assert int((plan.arow[:, 2] - plan.arow[:, 1]).max()) <= 1024

    def run():
        Zr, Ur = Z.clone().requires_grad_(True), U.clone().requires_grad_(True)
        Wr = W.clone().requires_grad_(True) if K else None
        loss, logits = ops.head_loss(Zr if not K else Z, idx, Ur, target, weight, want_logits=True, fold_W=Wr)
        loss.backward()
        return loss.detach(), logits, (Wr.grad if K else Zr.grad), Ur.grad

    loss, logits, dA, dU = run()
    ref_logits, ref_loss, ref_dZ, ref_dU, ref_dW = _fp64(Z, W, U, edges, target, weight, N)
    assert_close(logits, ref_logits, 1e-6, "logits")
    assert abs(float(loss) - float(ref_loss)) <= 1e-6 * max(1.0, abs(float(ref_loss)))
    assert_close(dA, ref_dW if K else ref_dZ, 2e-6, "dW" if K else "dZ")
    assert_close(dU, ref_dU, 2e-6, "dU")
    again = run()
    assert all(torch.equal(x, y) for x, y in zip((loss, logits, dA, dU), again))


@pytest.mark.parametrize("kw", [dict(lr=0.01, momentum=0.9), dict(lr=0.05), dict(lr=0.02, momentum=0.8, dampening=0.1, weight_decay=0.01, nesterov=False),
                                dict(lr=0.01, momentum=0.9, nesterov=True)])
def test_whole_training_step_of_the_folded_model_in_one_launch(kw):
    """layers.fused_train_step / GraphedTrainStep(fold_optimizer=True): loss, gradients and the SGD update of the folded
    1-layer model (G2: ehf.EmbeddingGCN, condensed W) in ONE launch — the same losses, parameters, momentum buffers and
    .grad as criterion(gcn(), target) / backward / FusedSGD.step(), eagerly and as a captured graph (several steps per replay
    included); combinations the kernel does not cover fall back (None / the unfolded graph)."""
    from tmgcn_amd.graphs import GraphedTrainStep
    from tmgcn_amd.layers import fused_train_step
    from tmgcn_amd.optim import FusedSGD
    d = golden("g2_gcn_condensed1")
    X = torch.from_numpy(d["X"])
    T, N = X.shape[0], X.shape[1]
    At, M, edges, labels = coo_list(d, "At", T, N), torch.from_numpy(d["M"]), torch.from_numpy(d["edges"]), torch.from_numpy(d["labels"]).cuda()
    w = torch.tensor([0.9, 0.1])

    def make():
        torch.manual_seed(int(d["seed"]))
        m = ehf.EmbeddingGCN(At, X, edges, M, hidden_feat=[6, 2], condensed_W=True, use_Minv=False)
        return m, FusedSGD(m.parameters(), **kw)

    m1, o1 = make()
    crit = WeightedCrossEntropy(w).cuda()
    ref = []
    for _ in range(9):
        o1.zero_grad()
        l = m1.loss(crit, labels)
        l.backward()
        o1.step()
        ref.append(float(l))
    m2, o2 = make()
    got = []
    for _ in range(9):
        l = fused_train_step(m2, crit, labels, o2)
        assert l is not None
        got.append(float(l))
    assert_close(np.array(got), np.array(ref), 2e-6, "losses, one launch per step")
    for (n, p), q in zip(m1.named_parameters(), m2.parameters()):
        assert_close(q.detach(), p.detach(), 2e-6, f"{n} after nine steps")
        assert_close(q.grad, p.grad, 5e-6, f"{n}.grad of the last step")
        if kw.get("momentum"):
            assert_close(o2.state[q]["momentum_buffer"], o1.state[p]["momentum_buffer"], 5e-6, f"momentum buffer of {n}")
    for k in (1, 3):
        m3, o3 = make()
        step = GraphedTrainStep(m3, crit, o3, labels, warmup=3, steps_per_replay=k, fold_optimizer=True)
        assert step.folded
        tail = []
        for _ in range(6 // k):
            step()
            tail += [float(x) for x in step.losses]
        assert_close(np.array(tail), np.array(ref[3:]), 2e-6, f"captured, {k} steps per replay")
        for p, q in zip(m1.parameters(), m3.parameters()):
            assert_close(q.detach(), p.detach(), 2e-6, "parameters after 3 + 6 steps")
    # not covered: torch's own optimizer, or a 2-layer model -> the unfolded route
    m4, _ = make()
    assert fused_train_step(m4, crit, labels, torch.optim.SGD(m4.parameters(), lr=0.01)) is None
    d3 = golden("g6_sgd_gcn2")
    X3 = torch.from_numpy(d3["X"])
    m5 = ehf.EmbeddingGCN2(coo_list(d3, "At", X3.shape[0], X3.shape[1]), X3, torch.from_numpy(d3["edges"]), torch.from_numpy(d3["M"]),
                           hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")
    o5 = FusedSGD(m5.parameters(), lr=0.01, momentum=0.9)
    assert fused_train_step(m5, crit, torch.from_numpy(d3["labels"]).cuda(), o5) is None
    assert not GraphedTrainStep(m5, crit, o5, torch.from_numpy(d3["labels"]).cuda(), fold_optimizer=True).folded

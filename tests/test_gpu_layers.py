"""GPU: the drop-in modules (tmgcn_amd.layers) against the golden fixtures captured from the
real reference — logits, loss, parameter gradients, SGD trajectories — and against the oracle."""
import numpy as np
import pytest
import torch

from _util import REL_TOL, assert_close, coo_list, golden, golden_names
import tmgcn_amd.layers as ehf

pytestmark = pytest.mark.gpu
TOL = REL_TOL  # 1e-5 · max|ref|


def _inputs(d, prefix=""):
    X = torch.from_numpy(d[prefix + "X"])
    T, N = X.shape[0], X.shape[1]
    out = dict(T=T, N=N, X=X, M=torch.from_numpy(d[prefix + "M"]), edges=torch.from_numpy(d[prefix + "edges"]),
               labels=torch.from_numpy(d[prefix + "labels"]), At=coo_list(d, "At", T, N, prefix=prefix))
    if prefix + "A_k" in d:
        out["A"] = coo_list(d, "A", T, N, prefix=prefix)
    return out


def _loss_grads(model, target, alpha=0.9):
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([alpha, 1 - alpha], device="cuda"))
    out = model()
    loss = crit(out, target.cuda())
    model.zero_grad()
    loss.backward()
    return out.detach(), float(loss.detach()), {n: p.grad for n, p in model.named_parameters()}


@pytest.mark.parametrize("name", golden_names("g1_"))
def test_g1_AtXt(name):
    d = golden(name)
    i = _inputs(d)
    m = ehf.EmbeddingGCN(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[3, 2], condensed_W=True, use_Minv=False)
    assert m.AtXt.dtype == torch.float32 and m.AtXt.is_cuda
    assert_close(m.AtXt, d["AtXt"], TOL, name)


@pytest.mark.parametrize("name", golden_names("g2_"))
def test_g2_gcn(name):
    d = golden(name)
    i = _inputs(d)
    torch.manual_seed(int(d["seed"]))
    m = ehf.EmbeddingGCN(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 2], condensed_W=d["W0"].ndim == 2,
                         use_Minv=name.endswith("minv_fp32"))
    # same seed => same initial weights as the reference (CPU generator, same draw order)
    assert np.array_equal(m.W.detach().cpu().numpy(), d["W0"]) and np.array_equal(m.U.detach().cpu().numpy(), d["U0"])
    out, loss, g = _loss_grads(m, i["labels"])
    assert out.dtype == torch.float32 and tuple(out.shape) == d["logits"].shape
    tol = TOL   # also for the all-fp32 fixture (measured 5.6e-7)
    assert_close(out, d["logits"], tol, name + " logits")
    assert abs(loss - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    assert_close(g["W"], d["dW"], tol, name + " dW")
    assert_close(g["U"], d["dU"], tol, name + " dU")


@pytest.mark.parametrize("name", golden_names("g3_"))
def test_g3_gcn2(name):
    d = golden(name)
    i, v = _inputs(d), _inputs(d, prefix="val_")
    _, _, branch, nl, cond = name.split("_")
    torch.manual_seed(int(d["seed"]))
    m = ehf.EmbeddingGCN2(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 6, 2], condensed_W=cond.endswith("1"),
                          use_Minv=False, apply_M_twice=branch in ("twice", "three"),
                          apply_M_three_times=branch == "three", nonlin2=nl)
    for k in ("W1", "W2", "U"):
        assert np.array_equal(getattr(m, k).detach().cpu().numpy(), d[k + "0"]), k
    out, loss, g = _loss_grads(m, i["labels"])
    assert_close(out, d["logits"], TOL, name + " logits")
    for k in ("W1", "W2", "U"):
        assert_close(g[k], d["d" + k], TOL, name + " d" + k)
    with torch.no_grad():  # validation call: layer 2 keeps the training adjacency (ehf:339-348)
        out_val = m(v["At"], v["X"], v["edges"])
    assert_close(out_val, d["logits_val"], TOL, name + " val logits")
    # anything that is not (list, Tensor, Tensor) falls back to the cached tensors (ehf:316)
    with torch.no_grad():
        assert torch.equal(m(None, v["X"], v["edges"]), m())


@pytest.mark.parametrize("name", golden_names("g4_"))
def test_g4_kwgcn(name):
    d = golden(name)
    i = _inputs(d)
    two = "2layer" in name
    torch.manual_seed(int(d["seed"]))
    m = ehf.EmbeddingKWGCN(i["A"], i["X"], i["edges"], hidden_feat=[6, 5, 2] if two else [6, 2], nonlin2=name.split("_")[-1])
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), d[n + "0"]), n
    out, loss, g = _loss_grads(m, i["labels"])
    assert_close(out, d["logits"], TOL, name)
    for n in g:
        assert_close(g[n], d["d" + n], TOL, name + " d" + n)
    # validation-style call on a SHORTER slice list (the baseline scripts: 25 vs 150 slices):
    # AX zero-padded to the training T, layer 2 over all of self.A (ehf:469-473, 486-487)
    v = _inputs(d, "val_")
    assert v["T"] < i["T"]
    with torch.no_grad():
        out_val = m(v["A"], v["X"], v["edges"])
    assert_close(out_val, d["logits_val"], TOL, name + " shorter validation window")
    with pytest.raises(RuntimeError):                       # more slices than the model holds: ehf raises too
        m(i["A"] + v["A"], torch.cat((i["X"], v["X"])), i["edges"])


def test_g5_chess():
    d = golden("g5_chess_gcn2")
    T, N = int(d["T"]), int(d["N"])
    At = coo_list(d, "At", T, N)
    torch.manual_seed(int(d["seed"]))
    m = ehf.EmbeddingGCN2(At, torch.from_numpy(d["X"]), torch.from_numpy(d["edges"]), torch.from_numpy(d["M"]),
                          hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")
    out, loss, g = _loss_grads(m, torch.from_numpy(d["labels"]))
    assert_close(out, d["logits"], TOL, "chess logits")
    for k in ("W1", "W2", "U"):
        assert_close(g[k], d["d" + k], TOL, "chess d" + k)


@pytest.mark.parametrize("kind", ["gcn", "gcn2"])
def test_g6_sgd_trajectory(kind):
    """The reference's training loop (SGD lr .01 mom .9, weighted CE) runs unchanged on the module."""
    d = golden("g6_sgd_" + kind)
    i = _inputs(d)
    torch.manual_seed(int(d["seed"]))
    if kind == "gcn":
        m = ehf.EmbeddingGCN(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 2], condensed_W=True, use_Minv=False)
    else:
        m = ehf.EmbeddingGCN2(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 6, 2], condensed_W=True,
                              use_Minv=False, nonlin2="selu")
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1], device="cuda"))
    tgt = i["labels"].cuda()
    losses = []
    for _ in range(10):
        opt.zero_grad()
        loss = crit(m(), tgt)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert_close(np.array(losses), d["losses"], TOL, kind + " loss trajectory")     # 10 SGD steps: measured 9e-7
    for n, p in m.named_parameters():
        assert_close(p.detach(), d[n + "_final"], TOL, kind + " final " + n)


def test_gcn2_use_minv_runs_and_matches_dense_math():
    """EmbeddingGCN2(use_Minv=True) raises a dtype error in the reference for every input dtype
    (SURVEY fact 3); here the math the code describes is computed.  Checked against fp64 einsum."""
    d = golden("g3_gcn2_default_relu_condensed1")
    i = _inputs(d)
    torch.manual_seed(5)
    m = ehf.EmbeddingGCN2(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 6, 2], condensed_W=True,
                          use_Minv=True, nonlin2="relu")
    out = m().detach().cpu().double()
    A = m.At.to_dense().cpu().double()
    M = i["M"]
    Minv = torch.linalg.inv(M)
    mt = lambda Q, Z: torch.einsum("kj,jnf->knf", Q, Z)
    sp = lambda Z: torch.einsum("knm,kmf->knf", A, Z)
    W1, W2, U = (p.detach().cpu().double() for p in (m.W1, m.W2, m.U))
    Y = torch.relu(mt(Minv, sp(mt(M, i["X"])) @ W1))
    Z = mt(Minv, sp(mt(M, Y)) @ W2).reshape(-1, 6)
    e = i["edges"]
    ref = torch.cat((Z[e[0] * i["N"] + e[1]], Z[e[0] * i["N"] + e[2]]), 1) @ U
    assert_close(out, ref, TOL, "use_Minv 2-layer")


def test_accepts_prebuilt_batched_csr_and_gpu_inputs():
    from tmgcn_amd.csr import BatchedCSR
    d = golden("g2_gcn_condensed1")
    i = _inputs(d)
    csr = BatchedCSR.from_coo_list(i["At"], N=i["N"], device="cuda")
    torch.manual_seed(1)
    a = ehf.EmbeddingGCN(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 2], condensed_W=True, use_Minv=False)
    torch.manual_seed(1)
    b = ehf.EmbeddingGCN(csr, i["X"].cuda().float(), i["edges"].cuda(), i["M"], hidden_feat=[6, 2], condensed_W=True, use_Minv=False)
    assert torch.equal(a(), b())
    assert torch.equal(a(i["At"], i["X"], i["edges"]), a())  # recompute branch == cached branch


@pytest.mark.parametrize("branch", ["default", "twice"])
def test_wide_features_use_fused_kernel_and_match_oracle(branch):
    """F = 16 -> 32 -> 16: layer 2 runs the fused P2+P3 kernel; compare with the oracle executed
    the reference's way (list of COO fp64, sparse.mm per slice) incl. all parameter gradients."""
    from oracle import tmgcn_oracle as orc
    from tmgcn_amd import synth
    g = synth.dynamic_graph(T=8, N=150, edges_per_slice=300, seed=5, no_diag=4, F0=16)
    At, X, M = g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.M)
    edges, labels = torch.from_numpy(g.edges), torch.from_numpy(g.labels)
    torch.manual_seed(3)
    m = ehf.EmbeddingGCN2(At, X, edges, M, hidden_feat=[32, 16, 2], condensed_W=True, use_Minv=False,
                          apply_M_twice=branch == "twice", nonlin2="leaky")
    with torch.no_grad():  # N(0,1) weights at these widths blow the activations up; scale down
        m.W1.mul_(0.2); m.W2.mul_(0.2)
    out, loss, grads = _loss_grads(m, labels)
    p = {n: q.detach().cpu().clone().requires_grad_(True) for n, q in m.named_parameters()}
    src, dst = orc.flat_edge_index(edges, g.N)
    ref = orc.gcn2_forward(orc.compute_AtXt(M, At, X), At, M, p["W1"], p["W2"], p["U"], src, dst,
                           nonlin="leaky", apply_M_twice=branch == "twice")
    torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))(ref, labels).backward()
    assert_close(out, ref.detach(), TOL, "logits")
    for n in p:
        assert_close(grads[n], p[n].grad, TOL, "d" + n)


def test_graphed_train_step_matches_eager():
    """The whole epoch (forward, weighted CE, backward, SGD) captured into one hipGraph gives the
    same trajectory as eager execution (every C-ABI launcher is capture-safe)."""
    from tmgcn_amd.graphs import GraphedTrainStep
    from tmgcn_amd.losses import WeightedCrossEntropy
    d = golden("g6_sgd_gcn2")
    i = _inputs(d)
    tgt = i["labels"].cuda()

    def build():
        torch.manual_seed(int(d["seed"]))
        m = ehf.EmbeddingGCN2(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 6, 2], condensed_W=True,
                              use_Minv=False, nonlin2="selu")
        return m, torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9), WeightedCrossEntropy(torch.tensor([0.9, 0.1])).cuda()

    m1, o1, c1 = build()
    eager = []
    for _ in range(3 + 10):  # GraphedTrainStep runs 3 warm-up steps before capturing
        o1.zero_grad()
        l = c1(m1(), tgt)
        l.backward()
        o1.step()
        eager.append(float(l.detach()))
    m2, o2, c2 = build()
    step = GraphedTrainStep(m2, c2, o2, tgt, warmup=3)
    graphed = [float(step()) for _ in range(10)]
    assert_close(np.array(graphed), np.array(eager[3:]), 1e-5, "graphed vs eager losses")
    for (n, p), (_, q) in zip(m1.named_parameters(), m2.named_parameters()):
        assert_close(q.detach(), p.detach(), 1e-5, "graphed vs eager " + n)
    # and the fixture's first 10 losses are what the first 10 eager steps gave (reference trajectory)
    assert_close(np.array(eager[:10]), d["losses"], TOL, "reference trajectory")


def test_tiny_and_degenerate_inputs():
    """T = 1, a slice with no edges at all, isolated nodes, a single labelled edge."""
    from oracle import tmgcn_oracle as orc
    N = 5
    a0 = torch.sparse_coo_tensor(torch.tensor([[0, 1, 4], [1, 0, 4]]), torch.tensor([0.5, 0.5, 1.0], dtype=torch.float64), (N, N))
    empty = torch.sparse_coo_tensor(torch.zeros(2, 0, dtype=torch.long), torch.zeros(0, dtype=torch.float64), (N, N))
    X = torch.arange(2 * N * 2, dtype=torch.float64).reshape(2, N, 2)
    M = torch.tensor([[1.0, 0.0], [0.5, 1.0]], dtype=torch.float64)
    edges = torch.tensor([[1], [2], [3]])
    torch.manual_seed(0)
    m = ehf.EmbeddingGCN([a0, empty], X, edges, M, hidden_feat=[3, 2], condensed_W=True, use_Minv=False)
    assert_close(m.AtXt, orc.compute_AtXt(M, [a0, empty], X), TOL, "AtXt with an empty slice")
    assert float(m.AtXt[1].abs().max()) == 0.0 and tuple(m().shape) == (1, 2)
    m().sum().backward()
    assert torch.isfinite(m.W.grad).all() and torch.isfinite(m.U.grad).all()
    # T = 1
    torch.manual_seed(0)
    m1 = ehf.EmbeddingGCN2([a0], X[:1], torch.tensor([[0, 0], [0, 4], [1, 4]]), M[:1, :1], hidden_feat=[4, 3, 2],
                           condensed_W=False, use_Minv=False, apply_M_twice=True, nonlin2="relu")
    p = {n: q.detach().cpu() for n, q in m1.named_parameters()}
    src, dst = orc.flat_edge_index(torch.tensor([[0, 0], [0, 4], [1, 4]]), N)
    ref = orc.gcn2_forward(orc.compute_AtXt(M[:1, :1], [a0], X[:1]), [a0], M[:1, :1], p["W1"], p["W2"], p["U"], src, dst,
                           nonlin="relu", apply_M_twice=True)
    assert_close(m1(), ref, TOL, "T = 1 two-layer model")
    # mismatched adjacency / features fail loudly, as RuntimeError (the reference's convention)
    with pytest.raises(RuntimeError):
        ehf.EmbeddingGCN([a0], X, edges, M, hidden_feat=[3, 2], condensed_W=True, use_Minv=False)
    with pytest.raises(RuntimeError):
        ehf.EmbeddingGCN([a0, empty], X[:, :3], edges, M, hidden_feat=[3, 2], condensed_W=True, use_Minv=False)


def test_graphed_train_step_refuses_a_parameter_the_model_does_not_own():
    """an optimizer parameter that is not a parameter of one of the model's modules cannot be aliased for the
    captured forward; it would silently never be updated.  Refused at construction, with a message that says why."""
    from tmgcn_amd.graphs import GraphedTrainStep
    from tmgcn_amd.losses import WeightedCrossEntropy
    d = golden("g6_sgd_gcn2")
    i = _inputs(d)
    torch.manual_seed(0)
    m = ehf.EmbeddingGCN2(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")
    stray = torch.nn.Parameter(torch.zeros(3, device="cuda"))
    opt = torch.optim.SGD(list(m.parameters()) + [stray], lr=0.01, momentum=0.9)
    with pytest.raises(RuntimeError, match="not parameters of"):
        GraphedTrainStep(m, WeightedCrossEntropy(torch.tensor([0.9, 0.1])).cuda(), opt, i["labels"].cuda(), warmup=3)


def test_graphed_train_step_wide_features():
    """Graph capture also covers the persistent MFMA kernels (their tile counters are zeroed by a
    memset node that is captured with the launch)."""
    from tmgcn_amd import synth
    from tmgcn_amd.graphs import GraphedTrainStep
    from tmgcn_amd.losses import WeightedCrossEntropy
    g = synth.dynamic_graph(T=6, N=300, edges_per_slice=600, seed=9, no_diag=4, F0=32)
    At, X, M = g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.M)
    edges, tgt = torch.from_numpy(g.edges), torch.from_numpy(g.labels).cuda()

    def build():
        torch.manual_seed(2)
        m = ehf.EmbeddingGCN2(At, X, edges, M, hidden_feat=[64, 32, 2], condensed_W=True, use_Minv=False,
                              apply_M_twice=True, nonlin2="relu")
        with torch.no_grad():
            for q in m.parameters():
                q.mul_(0.1)
        return m, torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9), WeightedCrossEntropy(torch.tensor([0.9, 0.1])).cuda()

    m1, o1, c1 = build()
    eager = []
    for _ in range(3 + 5):
        o1.zero_grad()
        l = c1(m1(), tgt)
        l.backward()
        o1.step()
        eager.append(float(l.detach()))
    m2, o2, c2 = build()
    step = GraphedTrainStep(m2, c2, o2, tgt, warmup=3)
    graphed = [float(step().detach()) for _ in range(5)]
    assert_close(np.array(graphed), np.array(eager[3:]), 1e-5, "graphed vs eager losses (wide)")


@pytest.mark.parametrize("name", golden_names("g8_"))
def test_g8_gcn_reg(name):
    """EmbeddingGCN_reg (ehf:359-423): output [T,N], parameters incl. the nn.Linear head drawn as in
    the reference, gradients of an MSE loss."""
    d = golden(name)
    i = _inputs(d)
    torch.manual_seed(int(d["seed"]))
    m = ehf.EmbeddingGCN_reg(i["At"], i["X"], i["M"], hidden_feat=[6], condensed_W=d["W0"].ndim == 2, use_Minv=False)
    for n, p in m.named_parameters():
        assert np.array_equal(p.detach().cpu().numpy(), d[n.replace(".", "_") + "0"]), n
    out = m()
    assert tuple(out.shape) == d["out"].shape
    loss = torch.nn.MSELoss()(out, torch.from_numpy(d["y"]).cuda())
    loss.backward()
    assert_close(out, d["out"], TOL, name + " output")
    assert abs(float(loss) - float(d["loss"])) <= 1e-5 * max(1.0, abs(float(d["loss"])))
    for n, p in m.named_parameters():
        assert_close(p.grad, d["d" + n.replace(".", "_")], TOL, name + " d" + n)
    assert torch.equal(m(i["At"], i["X"]), m())  # forward ignores its arguments (ehf:410-412)


def test_registered_cpp_autograd_and_python_autograd_paths_are_bit_identical():
    """ops.py reaches the kernels two ways: the registered C++ autograd operators of the torch
    extension (torch.ops.tmgcn.*, the product default) and — when a per-launch KernelTimer is attached,
    as bench.py's roofline leg does — Python autograd functions over the kernel-level operators.  Both
    must launch the same kernels with the same arguments: outputs and every gradient bit for bit."""
    from tmgcn_amd import ops
    d = golden("g3_gcn2_three_selu_condensed1")
    i = _inputs(d)

    def run(timed):
        torch.manual_seed(int(d["seed"]))
        m = ehf.EmbeddingGCN2(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 6, 2], condensed_W=True,
                              use_Minv=False, apply_M_twice=True, apply_M_three_times=True, nonlin2="selu")
        ops.kernels.timer = ops.KernelTimer() if timed else None
        try:
            out, loss, g = _loss_grads(m, i["labels"])
            tags = set(ops.kernels.timer.summary()) if timed else set()
        finally:
            ops.kernels.timer = None
        return out, loss, g, tags

    out_c, loss_c, g_c, _ = run(False)
    out_p, loss_p, g_p, tags = run(True)
    assert {"mtransform", "mtransform_T", "spmm_gemm", "spmm_gemm_T", "gemm", "gemm_dW", "edge_head", "edge_head_bwd"} <= tags, tags
    assert torch.equal(out_c, out_p) and loss_c == loss_p
    for n in g_c:
        assert torch.equal(g_c[n], g_p[n]), n
    assert_close(out_c, d["logits"], TOL, "logits")

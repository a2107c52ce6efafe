"""A whole link-prediction experiment in the shape of the reference's
experiment_*_our_link_prediction.py, run with ``import tmgcn_amd.ehf as ehf`` as the only change and
everything else (host-side targets, class weights, criterion, optimiser, metric calls, [-K_val:]
slicing) as the scripts have it, against fixture G9: what the real reference produced for the
committed ``g9_saved_content.mat`` (tests/golden/make_golden_data.py)."""
import random

import numpy as np
import pytest
import torch
import torch.nn as nn

import tmgcn_amd.ehf as ehf
from _util import GOLDEN, assert_close, golden, record_tolerance

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("no_layers", [1, 2])
def test_link_prediction_script_flow(no_layers):
    g9 = golden("g9_data")
    S_train, S_val, S_test = (int(s) for s in g9["S"])
    beta1, beta2, cutoff = 3, 2, 6
    lr, momentum, alpha = 0.01, 0.9, 0.9

    A, A_labels, Ct_train_2, Ct_val_2, Ct_test_2, N, M = ehf.load_data(GOLDEN + "/", "g9_saved_content.mat",
                                                                      S_train, S_val, S_test, transformed=True)
    X_train, X_val, X_test = ehf.create_node_features(A, S_train, S_val, S_test, same_block_size=True)
    edges = A_labels._indices()
    random.seed(7)
    edges_aug, labels = ehf.augment_edges(edges, N, beta1, beta2, cutoff)
    assert np.array_equal(edges_aug.numpy(), g9["edges_aug"])
    (edges_train, target_train, e_train, edges_val, target_val, e_val, K_val,
     edges_test, target_test, e_test, K_test) = ehf.split_data(edges_aug, labels, S_train, S_val, S_test, same_block_size=True)

    class_weights = torch.tensor([alpha, 1.0 - alpha])
    torch.manual_seed(100 + no_layers)
    if no_layers == 2:
        gcn = ehf.EmbeddingGCN2(Ct_train_2[:-1], X_train[:-1], e_train, M[:-1, :-1], hidden_feat=[6, 6, 2],
                                condensed_W=True, use_Minv=False, nonlin2="selu")
    else:
        gcn = ehf.EmbeddingGCN(Ct_train_2[:-1], X_train[:-1], e_train, M[:-1, :-1], hidden_feat=[6, 2],
                               condensed_W=True, use_Minv=False)
    assert all(p.is_cuda for p in gcn.parameters())          # the model lives on the MI355X ...
    optimizer = torch.optim.SGD(gcn.parameters(), lr=lr, momentum=momentum)
    criterion = nn.CrossEntropyLoss(weight=class_weights)     # ... the criterion is the script's own, built on the host

    losses = []
    for ep in range(6):
        optimizer.zero_grad()
        output_train = gcn()
        assert output_train.is_cuda                           # ... and so does the output (hosted.DeviceResult)
        loss_train = criterion(output_train, target_train[edges_train[0] != 0])
        loss_train.backward()
        optimizer.step()
        losses.append(float(loss_train.detach()))
    pre = f"exp{no_layers}_"
    assert_close(np.array(losses), g9[pre + "loss"], 1e-5, "loss trajectory")

    with torch.no_grad():
        output_train = gcn()
        output_val = gcn(Ct_val_2[:-1], X_val[:-1], e_val)
        output_test = gcn(Ct_test_2[:-1], X_test[:-1], e_test)
        assert_close(output_train, g9[pre + "out_train"], 1e-5, "train logits")
        assert_close(output_val, g9[pre + "out_val"], 1e-5, "val logits")
        assert_close(output_test, g9[pre + "out_test"], 1e-5, "test logits")
        loss_val = criterion(output_val[-K_val:], target_val[-K_val:])
        loss_test = criterion(output_test[-K_test:], target_test[-K_test:])
        for nm, got_l in (("loss_val", loss_val), ("loss_test", loss_test)):
            e = abs(float(got_l) - float(g9[pre + nm])) / abs(float(g9[pre + nm]))
            record_tolerance(f"G9 {pre}{nm}", e, 1e-5, kind="relative")
            assert e <= 1e-5, (nm, float(got_l), float(g9[pre + nm]))

        guess_val = torch.argmax(output_val, dim=1)
        f1_val = ehf.compute_f1(guess_val[-K_val:], target_val[-K_val:])
        ref_guess = torch.from_numpy(g9[pre + "out_val"]).argmax(1)
        if torch.equal(guess_val, ref_guess):                  # no logit pair closer than the tolerance
            np.testing.assert_allclose([float(v) for v in f1_val], g9[pre + "f1_val"], rtol=1e-12)
        for name, (out, tgt, e) in {
            "train": (output_train, target_train[edges_train[0] != 0], edges_train[:, edges_train[0] != 0]),
            "val": (output_val[-K_val:], target_val[-K_val:], edges_val[:, -K_val:]),
            "test": (output_test[-K_test:], target_test[-K_test:], edges_test[:, -K_test:]),
        }.items():
            MAP, MRR = ehf.compute_MAP_MRR(out, tgt, e)
            ref = g9[pre + "mapmrr_" + name]
            # rank metrics move by 1/E-sized steps when two nearly equal scores swap
            for nm, got_m, want_m in (("MAP", MAP, ref[0]), ("MRR", MRR, ref[1])):
                record_tolerance(f"G9 {pre}{name} {nm}", abs(float(got_m) - want_m), 5e-4, kind="absolute")
            assert abs(float(MAP) - ref[0]) <= 5e-4 and abs(float(MRR) - ref[1]) <= 5e-4, (name, float(MAP), float(MRR), ref)


def _inputs():
    g9 = golden("g9_data")
    S = [int(s) for s in g9["S"]]
    A, A_labels, Ct_train, _, _, N, M = ehf.load_data(GOLDEN + "/", "g9_saved_content.mat", *S, transformed=True)
    X_train, _, _ = ehf.create_node_features(A, *S, same_block_size=True)
    e = A_labels._indices()
    e = e[:, e[0] < S[0]]
    return Ct_train, X_train, e, M


def _small_model(cls_module, **attrs):
    Ct_train, X_train, e, M = _inputs()
    torch.manual_seed(0)
    m = cls_module.EmbeddingGCN(Ct_train, X_train, e, M, hidden_feat=[6, 2], condensed_W=True, use_Minv=False)
    for k, v in attrs.items():
        setattr(m, k, v)
    return m, e


def test_three_ways_to_hand_over_the_logits_agree():
    import tmgcn_amd.layers as layers
    from tmgcn_amd.hosted import DeviceResult
    dev, _ = _small_model(layers)                                           # plain device tensor
    hosted, _ = _small_model(ehf)                                           # DeviceResult on the device
    host, _ = _small_model(ehf, output_device="cpu", host_operands=False)  # plain host tensor
    a, b, c = dev(), hosted(), host()
    assert a.is_cuda and type(a) is torch.Tensor
    assert b.is_cuda and isinstance(b, DeviceResult)
    assert not c.is_cuda and type(c) is torch.Tensor
    assert torch.equal(a, b.as_subclass(torch.Tensor)) and torch.equal(a.cpu(), c)
    for out in (a, b, c):
        out.sum().backward()
    assert hosted.W.grad.is_cuda and torch.equal(hosted.W.grad, dev.W.grad) and torch.equal(host.W.grad, dev.W.grad)
    assert type(hosted.W.grad) is torch.Tensor


def test_device_result_takes_host_operands_the_way_the_scripts_use_them():
    from tmgcn_amd.hosted import DeviceResult
    m, e = _small_model(ehf)
    E = e.shape[1]
    g = torch.Generator().manual_seed(5)
    target = torch.randint(0, 2, (E,), generator=g)                         # host tensors, as in the scripts
    class_weights = torch.tensor([0.8, 0.2])
    out = m()
    ref = out.detach().as_subclass(torch.Tensor).cpu().double().requires_grad_(True)

    # criterion: the fused kernel's value == torch's on the host in fp64; gradients too
    loss = nn.CrossEntropyLoss(weight=class_weights)(out, target)
    assert isinstance(loss, DeviceResult) and loss.is_cuda
    loss_ref = nn.CrossEntropyLoss(weight=class_weights.double())(ref, target)
    assert abs(float(loss) - float(loss_ref)) <= 1e-6 * abs(float(loss_ref))
    # that loss came from the one-pass head + loss kernel (hosted.FUSE_HEAD_LOSS): its parameter gradients are what
    # differentiating through the logits gives — and d loss / d logits itself is there when the switch is off
    m.zero_grad()
    loss.backward(retain_graph=True)
    fused_grads = {n: p.grad.clone() for n, p in m.named_parameters()}
    from tmgcn_amd import hosted as hosted_mod
    hosted_mod.FUSE_HEAD_LOSS = False
    try:
        out = m()                                             # with the switch off gcn() hands back formed logits, never a placeholder
        assert type(out) is DeviceResult
        loss2 = nn.CrossEntropyLoss(weight=class_weights)(out, target)
        assert abs(float(loss2) - float(loss)) <= 1e-6 * abs(float(loss))
        (g_out,) = torch.autograd.grad(loss2, out, retain_graph=True)
        (g_ref,) = torch.autograd.grad(loss_ref, ref)
        assert_close(g_out, g_ref, 1e-6, "dloss/dlogits")
        m.zero_grad()
        loss2.backward(retain_graph=True)
        for n, p in m.named_parameters():
            assert_close(fused_grads[n], p.grad, 2e-6, "fused vs through-the-logits d" + n)
    finally:
        hosted_mod.FUSE_HEAD_LOSS = True
    # unweighted, ignore_index targets, and forms the kernel does not cover (torch runs them on the device)
    tgt_ign = target.clone()
    tgt_ign[::7] = -100
    for crit, crit_ref in ((nn.CrossEntropyLoss(), nn.CrossEntropyLoss()),
                           (nn.CrossEntropyLoss(weight=class_weights, reduction="sum"), nn.CrossEntropyLoss(weight=class_weights.double(), reduction="sum")),
                           (nn.CrossEntropyLoss(label_smoothing=0.1), nn.CrossEntropyLoss(label_smoothing=0.1))):
        for tg in (target, tgt_ign):
            got, want = crit(out, tg), crit_ref(ref, tg)
            record_tolerance(f"hosted criterion {crit}", abs(float(got) - float(want)) / abs(float(want)), 1e-5, kind="relative")
            assert got.is_cuda and abs(float(got) - float(want)) <= 1e-5 * abs(float(want)), (crit, float(got), float(want))

    # metrics and bookkeeping idioms of the scripts
    guess = torch.argmax(out, dim=1)
    p, r, f1 = ehf.compute_f1(guess, target)
    p2, r2, f2 = ehf.compute_f1(guess.cpu().as_subclass(torch.Tensor), target)
    assert float(p) == float(p2) and float(r) == float(r2) and float(f1) == float(f2)
    K = torch.tensor(10)
    assert out[-K:].shape == (10, 2) and (target[-K:] == guess[-K:]).shape == (10,)
    row = np.zeros((2, 4))
    row[0] = [p, r, f1, loss]                                                # ep_acc_loss[ep] = [...]
    assert row[0, 3] == float(loss) and row[0, 0] == float(p)
    print("alpha/Tr/Ep %.2f/%d/%d. Train precision/recall/f1 %.16f/%.16f/%.16f. Train loss %.16f." % (0.9, 0, 0, p, r, f1, loss))
    buf = torch.empty(E, 2)
    buf.copy_(out.detach())                                                  # a host destination stays a host tensor
    assert not buf.is_cuda and torch.equal(buf, out.detach().as_subclass(torch.Tensor).cpu())
    both = torch.cat((torch.zeros(1, 2), out.detach()))
    assert both.is_cuda and both.shape == (E + 1, 2)
    MAP, MRR = ehf.compute_MAP_MRR(out, target, e)
    MAP2, MRR2 = ehf.compute_MAP_MRR(ref.detach().float(), target, e)
    assert abs(float(MAP) - float(MAP2)) <= 1e-12 and abs(float(MRR) - float(MRR2)) <= 1e-12


def test_host_operands_cross_pcie_once():
    """hosted.DeviceResult keeps the device copy of a host operand on the host tensor: the scripts' targets and class
    weights are uploaded once, not every epoch; a tensor that is modified in between is uploaded again."""
    from tmgcn_amd import hosted
    out = torch.zeros(300_000, 2, device="cuda").as_subclass(hosted.DeviceResult)
    big = torch.ones(300_000, 2)                                     # 2.4 MB on the host
    small = torch.ones(2)
    for _ in range(6):
        r = out + big
        r = r * small
    assert r.is_cuda and float(r.sum()) == 600_000.0
    assert big._tmgcn_uploads == 1 and small._tmgcn_uploads == 1
    big[0, 0] = 5.0                                                  # written to: the stale copy is not served
    assert float((out + big).sum()) == 600_004.0 and big._tmgcn_uploads == 2


def _plain_logits(m):
    """The logits of the model's CURRENT parameters through the plain route (no placeholder), detached."""
    from tmgcn_amd import hosted as hosted_mod
    hosted_mod.LAZY_LOGITS = False
    try:
        with torch.no_grad():
            return m().detach().as_subclass(torch.Tensor).clone()
    finally:
        hosted_mod.LAZY_LOGITS = True


def test_lazy_logits_after_a_step_behind_a_custom_loss_and_second_loss_terms():
    """(i) `out = gcn(); <a loss F.cross_entropy never sees>; backward; optimizer.step(); out.argmax()` — a path the reference
    supports — returns the logits of the pre-step parameters; (ii) a second loss term on the output is differentiated
    (after the fused criterion call the placeholder held DETACHED logits and the term contributed no gradient):
    every parameter gradient equals the LAZY_LOGITS = False run's; (iii) nn.CrossEntropyLoss on the placeholder still
    reaches the one-pass kernel and on a formed DeviceResult the fused weighted-CE kernel (the interception rides on
    `func is F.cross_entropy` inside __torch_function__: a torch upgrade that changes it fails HERE)."""
    import tmgcn_amd.layers as layers
    from tmgcn_amd import hosted as hosted_mod
    from tmgcn_amd.hosted import LazyLogits
    g = torch.Generator().manual_seed(8)
    for no_layers in (1, 2):
        def make():
            torch.manual_seed(13)
            Ct_train, X_train, e, M = _inputs()
            if no_layers == 1:
                return ehf.EmbeddingGCN(Ct_train, X_train, e, M, hidden_feat=[6, 2], condensed_W=True, use_Minv=False), e
            return ehf.EmbeddingGCN2(Ct_train, X_train, e, M, hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu"), e
        # (i) custom loss, step, late read
        m, e = make()
        opt = torch.optim.SGD(m.parameters(), lr=0.5)
        target = torch.randint(0, 2, (e.shape[1],), generator=g)
        out = m()
        assert type(out) is LazyLogits
        pre = _plain_logits(m)
        custom = (out.softmax(1)[:, 0] - target.float()).pow(2).mean()          # never passes F.cross_entropy
        custom.backward()
        opt.step()
        assert_close(out.detach().as_subclass(torch.Tensor), pre, 1e-6, "logits read after a step behind a custom loss")
        opt.zero_grad()
        out = m()                                                                # no loss at all, parameters moved, first read late
        pre = _plain_logits(m)
        with torch.no_grad():
            m.U.mul_(1.5)
        assert torch.equal(out.argmax(1).as_subclass(torch.Tensor), pre.argmax(1))
        # (ii) criterion + a second term on the same output
        grads = {}
        for lazy in (True, False):
            hosted_mod.LAZY_LOGITS = lazy
            try:
                m, e = make()
                crit = nn.CrossEntropyLoss(weight=torch.tensor([0.7, 0.3]))
                out = m()
                assert (type(out) is LazyLogits) == lazy
                loss = crit(out, target) + 0.1 * out.pow(2).mean() + 0.05 * crit(out, 1 - target)
                loss.backward()
                grads[lazy] = {n: q.grad.detach().clone() for n, q in m.named_parameters()}
            finally:
                hosted_mod.LAZY_LOGITS = True
        for n in grads[True]:
            assert_close(grads[True][n], grads[False][n], 1e-5, f"{no_layers}-layer d{n} with a second loss term")
        # (iii) the interception itself
        calls = {"head": 0, "ce": 0}
        real_head, real_ce = hosted_mod._fused_head_loss, hosted_mod._fused_cross_entropy

        def count_head(*a, **k):
            calls["head"] += 1
            return real_head(*a, **k)

        def count_ce(*a, **k):
            calls["ce"] += 1
            return real_ce(*a, **k)
        hosted_mod._fused_head_loss, hosted_mod._fused_cross_entropy = count_head, count_ce
        try:
            m, e = make()
            crit = nn.CrossEntropyLoss(weight=torch.tensor([0.7, 0.3]))
            crit(m(), target)                                                   # placeholder -> one-pass head + loss
            assert calls == {"head": 1, "ce": 0}
            with torch.no_grad():
                crit(m(), target)                                               # evaluation: formed logits -> fused weighted CE
            assert calls["ce"] == 1
        finally:
            hosted_mod._fused_head_loss, hosted_mod._fused_cross_entropy = real_head, real_ce


def test_lazy_logits_are_the_values_before_the_step_and_form_on_demand():
    """Script mode, training epoch: gcn() returns a placeholder (hosted.LazyLogits); the criterion's one launch forms loss,
    gradients AND the logits — the values before optimizer.step(), which is what the scripts' accuracy lines read afterwards
    (experiment_reddit_our_link_prediction.py:76-87); anything else that touches the output first forms it from the
    embedding; reading it for the first time after a step, without the criterion in between, is refused."""
    import tmgcn_amd.layers as layers
    from tmgcn_amd import hosted as hosted_mod
    from tmgcn_amd.hosted import DeviceResult, LazyLogits
    g = torch.Generator().manual_seed(5)
    crit_w = torch.tensor([0.8, 0.2])
    for no_layers in (1, 2):
        def make(mod):
            torch.manual_seed(11)
            Ct_train, X_train, e, M = _inputs()
            if no_layers == 1:
                return mod.EmbeddingGCN(Ct_train, X_train, e, M, hidden_feat=[6, 2], condensed_W=True, use_Minv=False), e
            return mod.EmbeddingGCN2(Ct_train, X_train, e, M, hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu"), e
        m, e = make(ehf)
        ref, _ = make(layers)                                   # plain device tensors, logits always formed
        target = torch.randint(0, 2, (e.shape[1],), generator=g)
        crit = nn.CrossEntropyLoss(weight=crit_w)
        crit_dev = nn.CrossEntropyLoss(weight=crit_w.cuda())
        o1 = torch.optim.SGD(m.parameters(), lr=0.05, momentum=0.9)
        o2 = torch.optim.SGD(ref.parameters(), lr=0.05, momentum=0.9)
        for step in range(4):
            o1.zero_grad(); o2.zero_grad()
            out = m()
            assert type(out) is LazyLogits and out._tmgcn_value is None
            assert out.shape == (e.shape[1], 2) and out.is_cuda and len(out) == e.shape[1] and out._tmgcn_value is None
            if step == 2:                                       # something reads the output BEFORE the criterion: formed from the embedding
                guess_before = torch.argmax(out, dim=1)
                assert out._tmgcn_value is not None
            loss = crit(out, target)
            assert isinstance(loss, DeviceResult) and out._tmgcn_value is not None
            want = ref()
            loss_ref = crit_dev(want, target.cuda())
            loss.backward(); loss_ref.backward()
            o1.step(); o2.step()
            # read AFTER the step, as the scripts do: the logits of the parameters before it
            got = out.detach().as_subclass(torch.Tensor)
            assert_close(got, want.detach(), 1e-6, f"{no_layers}-layer logits read after step {step}")
            loss64 = nn.CrossEntropyLoss(weight=crit_w.double())(want.detach().double().cpu(), target)   # (torch-ROCm's fp32 NLL mean is itself ~4e-6 off)
            assert abs(float(loss) - float(loss64)) <= 1e-5 * abs(float(loss64))     # the stated bar (two models' fp32 trajectories)
            if step == 2:
                assert torch.equal(guess_before, want.detach().argmax(1))
            for (n, p), q in zip(m.named_parameters(), ref.parameters()):
                assert_close(p.detach(), q.detach(), 2e-6, f"{no_layers}-layer {n} after step {step}")
        # never through the criterion, first read after a parameter changed (a custom loss, or none, then a step): the
        # reference's output_train is an ordinary tensor and still holds the logits of the parameters gcn() ran with
        # (experiment_reddit_our_link_prediction.py:78-87) — formed here from the snapshot gcn() took
        out = m()
        want_pre = _plain_logits(m)
        with torch.no_grad():
            for q in m.parameters():
                q.add_(0.37)
        got = out.argmax(1)
        assert torch.equal(got.as_subclass(torch.Tensor), want_pre.argmax(1))
        assert_close(out.detach().as_subclass(torch.Tensor), want_pre, 1e-6, f"{no_layers}-layer logits read first after the parameters moved")
        assert not out.detach().requires_grad and not _plain_logits(m).allclose(want_pre)    # (the live parameters give other logits)
        # evaluation calls and the opt-out form the logits at once
        with torch.no_grad():
            assert type(m()) is DeviceResult
        hosted_mod.FUSE_HEAD_LOSS = False
        try:
            assert type(m()) is DeviceResult
        finally:
            hosted_mod.FUSE_HEAD_LOSS = True

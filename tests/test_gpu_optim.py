"""GPU: tmgcn_amd.optim.FusedSGD (one launch for the step of all parameters) against torch.optim.SGD — the
optimizer of every reference script (experiment_reddit_our_link_prediction.py:68)."""
import pytest
import torch

from _util import assert_close
from tmgcn_amd.optim import FusedSGD

pytestmark = pytest.mark.gpu


def _params(dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    return [torch.nn.Parameter(torch.randn(*s, generator=g).to("cuda", dtype)) for s in ((2, 6), (6, 6), (12, 2), (95, 6, 6), (1000,))]


@pytest.mark.parametrize("kw", [dict(lr=0.01, momentum=0.9), dict(lr=0.05), dict(lr=0.02, momentum=0.8, dampening=0.1, weight_decay=0.01),
                                dict(lr=0.01, momentum=0.9, nesterov=True), dict(lr=0.03, momentum=0.5, maximize=True)])
def test_fused_sgd_follows_torch_sgd(kw):
    a, b = _params(torch.float32), _params(torch.float32)
    oa, ob = torch.optim.SGD(a, **kw), FusedSGD(b, **kw)
    g = torch.Generator().manual_seed(1)
    for step in range(6):
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g).cuda()
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        for i, (p, q) in enumerate(zip(a, b)):
            assert_close(q.detach(), p.detach(), 1e-6, f"step {step} parameter {i}")
    if kw.get("momentum"):
        for p, q in zip(a, b):
            assert_close(ob.state[q]["momentum_buffer"], oa.state[p]["momentum_buffer"], 1e-6, "momentum buffer")
        # the state layouts match: a torch.optim.SGD resumes from a FusedSGD checkpoint
        oc = torch.optim.SGD(b, **kw)
        oc.load_state_dict(ob.state_dict())
        assert torch.equal(oc.state[b[0]]["momentum_buffer"], ob.state[b[0]]["momentum_buffer"])


def test_fused_sgd_bf16_rounds_once_per_step():
    a, b = _params(torch.bfloat16), _params(torch.bfloat16)
    ref = [p.detach().float().clone() for p in a]            # fp32 shadow of the same trajectory
    bufs = [None] * len(ref)
    ob = FusedSGD(b, lr=0.01, momentum=0.9)
    g = torch.Generator().manual_seed(2)
    for step in range(4):
        for i, q in enumerate(b):
            gr = torch.randn(q.shape, generator=g).cuda().bfloat16()
            q.grad = gr.clone()
            bufs[i] = gr.float() if bufs[i] is None else (0.9 * bufs[i] + gr.float()).bfloat16().float()
            ref[i] = (ref[i] - 0.01 * bufs[i]).bfloat16().float()
        ob.step()
    for i, q in enumerate(b):
        assert q.dtype == torch.bfloat16
        assert_close(q.detach().float(), ref[i], 1e-2, f"bf16 parameter {i}")   # a few bf16 ulps over 4 steps


def test_fused_sgd_skips_parameters_without_gradient_and_captures_into_a_graph():
    ps = _params(torch.float32)[:3]
    opt = FusedSGD(ps, lr=0.1, momentum=0.9)
    ps[0].grad = torch.ones_like(ps[0])
    before = [p.detach().clone() for p in ps]
    opt.step()
    assert torch.equal(ps[1].detach(), before[1]) and not torch.equal(ps[0].detach(), before[0])
    for p in ps:
        p.grad = torch.ones_like(p)
    opt.step()                                                 # every buffer exists now
    static_grads = [p.grad for p in ps]
    gph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gph):
        opt.step()
    want = [p.detach().clone() for p in ps]
    ref = torch.optim.SGD([torch.nn.Parameter(w.clone()) for w in want], lr=0.1, momentum=0.9)
    import copy
    ref.load_state_dict(copy.deepcopy(opt.state_dict()))        # its own momentum buffers, not views of opt's
    for q, gr in zip(ref.param_groups[0]["params"], static_grads):
        q.grad = gr.clone()
    gph.replay()
    ref.step()
    for p, q in zip(ps, ref.param_groups[0]["params"]):
        assert_close(p.detach(), q.detach(), 1e-6, "replayed step")


def test_fused_sgd_step_hooks_scheduler_and_zero_grad():
    """FusedSGD.step() skips torch's profiler-range wrapper unless a step hook is registered: hooks fire exactly as on
    torch.optim.SGD, an LR scheduler drives it, zero_grad() behaves in both modes."""
    a, b = _params(torch.float32), _params(torch.float32)
    oa, ob = torch.optim.SGD(a, lr=0.1, momentum=0.9), FusedSGD(b, lr=0.1, momentum=0.9)
    sa, sb = torch.optim.lr_scheduler.StepLR(oa, step_size=2, gamma=0.5), torch.optim.lr_scheduler.StepLR(ob, step_size=2, gamma=0.5)
    calls = {"pre": 0, "post": 0}
    h1 = ob.register_step_pre_hook(lambda opt, args, kwargs: calls.__setitem__("pre", calls["pre"] + 1))
    h2 = ob.register_step_post_hook(lambda opt, args, kwargs: calls.__setitem__("post", calls["post"] + 1))
    g = torch.Generator().manual_seed(4)
    for step in range(5):
        if step == 3:                       # from here on without hooks: the lean path
            h1.remove()
            h2.remove()
        for p, q in zip(a, b):
            gr = torch.randn(p.shape, generator=g).cuda()
            p.grad, q.grad = gr.clone(), gr.clone()
        oa.step()
        ob.step()
        sa.step()
        sb.step()
        assert oa.param_groups[0]["lr"] == ob.param_groups[0]["lr"]
        for i, (p, q) in enumerate(zip(a, b)):
            assert_close(q.detach(), p.detach(), 1e-6, f"step {step} parameter {i}")
    assert calls == {"pre": 3, "post": 3}
    ob.zero_grad()
    assert all(q.grad is None for q in b)
    for q in b:
        q.grad = torch.ones_like(q)
    ob.zero_grad(set_to_none=False)
    assert all(q.grad is not None and float(q.grad.abs().max()) == 0.0 for q in b)
    loss = ob.step(closure=lambda: torch.tensor(3.0))
    assert float(loss) == 3.0

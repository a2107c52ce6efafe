"""``FusedSGD``: ``torch.optim.SGD`` with the step of ALL parameters in one launch.

The reference scripts train with ``t.optim.SGD(gcn.parameters(), lr=0.01, momentum=0.9)``
(experiment_reddit_our_link_prediction.py:68, 80).  The models have two or three parameters of a few dozen
floats each; torch's foreach implementation spends three to four launches on them (``mul``, ``add``, ``add``
per list), which is a third of the launches of a captured training step.  This class keeps SGD's semantics,
hyper-parameters and state layout (``state[p]["momentum_buffer"]``, so a ``state_dict`` moves between the two)
and runs ``torch.ops.tmgcn.sgd_step`` (csrc/pointwise.hip: tmgcn_sgd_step) — fp32 arithmetic; parameters stored
in bf16 (the "bf16 weights" configuration) are widened on load and rounded once per step, where the unfused
optimizer rounds after each of its three operations.

    opt = tmgcn_amd.optim.FusedSGD(gcn.parameters(), lr=0.01, momentum=0.9)
"""
from __future__ import annotations

import torch

from . import _lib

_MAX = 16    # tensors per launch (kSgdMaxTensors in csrc/pointwise.hip)


class FusedSGD(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, momentum=0.0, dampening=0.0, weight_decay=0.0, nesterov=False, maximize=False):
        if lr < 0.0 or momentum < 0.0 or weight_decay < 0.0:
            raise ValueError("FusedSGD: lr, momentum and weight_decay must be non-negative")
        if nesterov and (momentum <= 0 or dampening != 0):
            raise ValueError("Nesterov momentum requires a momentum and zero dampening")
        super().__init__(params, dict(lr=lr, momentum=momentum, dampening=dampening, weight_decay=weight_decay,
                                      nesterov=nesterov, maximize=maximize))

    # torch wraps every optimizer's step() in a profiler range plus a walk over the step hooks, and zero_grad() in another
    # range: ≈ 35 us of host time per epoch (tools/host_profile_epoch.py), a sixth of an eager epoch of the reference-shaped
    # configs.  Here step() takes that wrapper only when a step hook is registered (same hook semantics), and zero_grad()
    # with set_to_none drops the gradients directly.
    def _patch_step_function(self) -> None:
        self._zero_grad_profile_name = f"Optimizer.zero_grad#{self.__class__.__name__}.zero_grad"

    def step(self, closure=None):
        from torch.optim import optimizer as _o
        if (self._optimizer_step_pre_hooks or self._optimizer_step_post_hooks or _o._global_optimizer_pre_hooks
                or _o._global_optimizer_post_hooks):
            return self._hooked_step(closure)
        return self._step(closure)

    def zero_grad(self, set_to_none: bool = True) -> None:
        if not set_to_none:
            return super().zero_grad(set_to_none=False)
        for group in self.param_groups:
            for p in group["params"]:
                p.grad = None

    @torch.no_grad()
    def _step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ops = _lib.load_torch_ops()
        for group in self.param_groups:
            mom = float(group["momentum"])
            # parameters that share a launch: same device, same storage type, same "first step" status
            buckets = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse or not p.is_cuda or p.dtype not in (torch.float32, torch.bfloat16):
                    raise RuntimeError("FusedSGD: dense fp32 / bf16 parameters on a ROCm device only")
                st = self.state[p]
                first = mom != 0.0 and st.get("momentum_buffer") is None
                if first:
                    st["momentum_buffer"] = torch.empty_like(p, memory_format=torch.contiguous_format)
                buckets.setdefault((p.device, p.dtype, first), []).append(p)
            for (_, _, first), ps in buckets.items():
                for i in range(0, len(ps), _MAX):
                    chunk = ps[i:i + _MAX]
                    grads = [q.grad if q.grad.is_contiguous() else q.grad.contiguous() for q in chunk]
                    bufs = [self.state[q]["momentum_buffer"] for q in chunk] if mom != 0.0 else []
                    ops.sgd_step([q.data for q in chunk], grads, bufs, float(group["lr"]), mom, float(group["dampening"]),
                                 float(group["weight_decay"]), bool(group["nesterov"]), bool(group["maximize"]), bool(first))
        return loss


FusedSGD._hooked_step = torch.optim.Optimizer.profile_hook_step(FusedSGD._step)

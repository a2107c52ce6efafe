"""hipGraph capture of a whole training step (forward, loss, backward, optimizer step).

The reference's real configs are tiny (N ≈ 1k–7k, F = 2→6→6): a step is a dozen kernels of a few µs
each, so eager execution is bound by Python and launch overhead.  Every launcher of the C-ABI is
asynchronous, allocation-free and sync-free, so the whole step captures into one graph:

    step = GraphedTrainStep(gcn, criterion, optimizer, target)
    for ep in range(no_epochs):
        loss = step()            # replays; with keep_logits=True `step.output` holds the logits of that step

What keeps the launch count down (rocprofv3 --kernel-trace of an S1 step: profiles/):
  * `fused_loss=True` (default when the model offers it): `gcn.loss(criterion, target)` — the edge head,
    the class-weighted cross entropy and every gradient of both in ONE launch (csrc/head_loss.hip)
    instead of head / loss / loss-backward / dU / dZ + three reduction tails;
  * gradients are released before the capture (`zero_grad(set_to_none=True)`), so that the captured
    backward WRITES each `.grad` into graph-private memory instead of zero-filling and then adding
    into a persistent one (two launches per parameter saved) — PyTorch's whole-network capture recipe;
  * with `tmgcn_amd.optim.FusedSGD` the optimizer step of all parameters is one launch;
  * `steps_per_replay=k` captures k consecutive steps into the one graph: a replay costs ≈ 8 µs of launch latency
    whatever it holds (the gap in front of the first kernel in every rocprofv3 trace of a replayed step), a tenth of
    a Bitcoin-OTC-sized step and a sixth of a Reddit-LP-sized one.  `step()` then advances the model by k epochs
    and `step.losses` holds the k losses in order.
"""
from __future__ import annotations

import torch


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, criterion, optimizer: torch.optim.Optimizer,
                 target: torch.Tensor, warmup: int = 3, fused_loss: bool = True, keep_logits: bool = False,
                 steps_per_replay: int = 1, fold_optimizer: bool = False):
        self.model, self.criterion, self.optimizer, self.target = model, criterion, optimizer, target
        if steps_per_replay < 1:
            raise ValueError("steps_per_replay must be at least 1")
        self.steps_per_replay = int(steps_per_replay)
        dev = target.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs ROCm tensors")
        from . import ops
        self.fused = bool(fused_loss) and hasattr(model, "loss")
        self.keep_logits = keep_logits or not self.fused
        # The first step of a momentum optimizer CREATES its buffers (buf = g): captured, every replay would restart the
        # momentum.  So at least one eager step must precede the capture whenever such state is still missing.
        if warmup < 1 and any(g.get("momentum", 0) != 0 and any(p.requires_grad and "momentum_buffer" not in optimizer.state.get(p, {})
                                                                 for p in g["params"]) for g in optimizer.param_groups):
            raise ValueError("GraphedTrainStep: warmup=0 with a momentum optimizer whose buffers do not exist yet — the capture "
                             "would record the buffer-creating first step and every replay would reset the momentum")
        # fold_optimizer: loss, gradients AND the SGD update in one launch where the model / optimizer pair allows it (the
        # folded 1-layer model with FusedSGD: layers.fused_train_step) — the whole step of the Reddit-LP config is one kernel
        self.folded = False
        if fold_optimizer and self.fused and not self.keep_logits:
            from .layers import fused_train_step
            if warmup < 1:
                raise ValueError("fold_optimizer needs at least one warm-up step (the first SGD step creates the momentum buffers)")
            probe = fused_train_step(model, criterion, target, optimizer)         # one real step: also the applicability test
            self.folded = probe is not None

        def forward_loss():
            if self.fused:
                if self.keep_logits:
                    return model.loss(criterion, target, want_logits=True, unit_grad=True)
                return model.loss(criterion, target, unit_grad=True), None
            out = model()
            return criterion(out, target), out

        params = [p for g in optimizer.param_groups for p in g["params"] if p.requires_grad]
        # where the model keeps each of them: (module, attribute name), so that a step can run on fresh leaf aliases
        slots = {}
        for mod in model.modules():
            for name, q in mod._parameters.items():
                if q is not None:
                    slots.setdefault(id(q), []).append((mod, name))

        # an optimizer parameter the model does not OWN (no module slot) cannot be aliased below: the forward would reach it
        # by another route, autograd.grad(…, allow_unused=True) would hand back None for it and it would silently never be
        # updated — refuse at construction
        homeless = [tuple(p.shape) for p in params if id(p) not in slots]
        if homeless and not self.folded:
            raise RuntimeError(f"GraphedTrainStep: {len(homeless)} optimizer parameter(s) of shape(s) {homeless} are not parameters of "
                               "a module of the model; the captured step can only differentiate parameters the model owns")

        class _FreshLeaves:
            """The optimizer's parameters replaced, inside the model, by fresh leaf ALIASES (`p.detach().requires_grad_()`:
            same storage, no launch) for the duration of a forward.  Why: autograd binds a leaf's AccumulateGrad node to
            the stream it was created on and keeps it alive as long as ANY graph references it — e.g. the `loss` of an
            eager epoch the caller still holds.  A backward on another stream then synchronises with that stream, and
            inside a capture this cross-stream dependency invalidates the capture (hipStreamEndCapture crashes:
            tools/debug_capture.py; neither loss.backward() nor autograd.grad avoids the node's input buffer).  The
            aliases' nodes are created by the very forward that uses them, on its stream."""

            def __enter__(self):
                self.alias = [p.detach().requires_grad_(True) for p in params]
                for p, a in zip(params, self.alias):
                    for mod, name in slots.get(id(p), ()):
                        mod._parameters[name] = a
                return self.alias

            def __exit__(self, *exc):
                for p in params:
                    for mod, name in slots.get(id(p), ()):
                        mod._parameters[name] = p
                return False

        def forward_backward():
            with _FreshLeaves() as alias:
                loss, out = forward_loss()
                grads = torch.autograd.grad(loss, alias, grad_outputs=ops.unit_gradient(dev), allow_unused=True)
            for p, g in zip(params, grads):          # what a backward into an empty .grad leaves behind
                p.grad = g
            return loss, out

        def one_step():
            if self.folded:
                from .layers import fused_train_step
                return fused_train_step(model, criterion, target, optimizer), None
            optimizer.zero_grad(set_to_none=True)
            loss, out = forward_backward()
            optimizer.step()
            return loss, out

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up off the capture stream (allocator, lazy init, plans)
            if not self.folded:
                # the aliased forward must reach exactly the parameters an ordinary backward reaches (a cached attribute
                # reference to a parameter would bypass the aliases): compare once, refuse to capture on a mismatch
                optimizer.zero_grad(set_to_none=True)
                l0, o0 = forward_loss()
                l0.backward()
                reached = [p.grad is not None for p in params]
                del l0, o0
                optimizer.zero_grad(set_to_none=True)
                forward_backward()
                got = [p.grad is not None for p in params]
                optimizer.zero_grad(set_to_none=True)
                if reached != got:
                    raise RuntimeError("GraphedTrainStep: the captured step's forward reaches other parameters than loss.backward() does "
                                       f"(ordinary backward: {reached}, captured route: {got}); a parameter is being used through a "
                                       "reference the model's modules do not own")
            for _ in range(warmup - (1 if self.folded else 0)):          # (the applicability probe above was a step)
                one_step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        # the upstream gradient of the loss, made once (ops.unit_gradient): a bare `loss.backward()` has autograd fill a
        # fresh ones_like(loss) on every replay (one more launch per step), and the fused head + loss skips its scaling
        # launch when it meets this very tensor
        from . import ops
        self._one = ops.unit_gradient(dev)
        self.losses = []
        with torch.cuda.graph(self.graph):
            for i in range(self.steps_per_replay):
                if self.folded:
                    self.loss, self._output = one_step()
                else:
                    if i:
                        optimizer.zero_grad(set_to_none=True)      # host side only: the next backward writes fresh gradients
                    self.loss, self._output = forward_backward()
                    optimizer.step()
                self.losses.append(self.loss)

    @property
    def output(self) -> torch.Tensor:
        """Logits of the last captured step (overwritten by every replay).  Only with keep_logits=True: the one-pass
        head + loss route never materialises them."""
        if self._output is None:
            raise RuntimeError("GraphedTrainStep.output: the logits were not kept (the fused head + loss launch does not "
                               "materialise them); construct the step with keep_logits=True")
        return self._output

    def __call__(self) -> torch.Tensor:
        """One replay = `steps_per_replay` epochs.  Returns the loss of the last of them (a tensor the next replay
        overwrites); `self.losses` lists all of them, `self.output` is the last step's logits when kept."""
        self.graph.replay()
        return self.loss

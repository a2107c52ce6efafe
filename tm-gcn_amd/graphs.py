"""hipGraph capture of a whole training step (forward, loss, backward, optimizer step).

The reference's real configs are tiny (N ≈ 1k–7k, F = 2→6→6): a step is ~25 kernels of a few µs
each, so eager execution is bound by Python and launch overhead (0.5 ms per step at the
Bitcoin-shaped size vs 0.2 ms replayed).  Every launcher of the C-ABI is asynchronous,
allocation-free and sync-free, so the whole step captures into one graph:

    step = GraphedTrainStep(gcn, criterion, optimizer, target)
    for ep in range(no_epochs):
        loss = step()            # replays; `step.output` holds the logits of that step
"""
from __future__ import annotations

import torch


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, criterion, optimizer: torch.optim.Optimizer,
                 target: torch.Tensor, warmup: int = 3):
        self.model, self.criterion, self.optimizer, self.target = model, criterion, optimizer, target
        dev = target.device
        if dev.type != "cuda":
            raise RuntimeError("GraphedTrainStep needs ROCm tensors")
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):  # warm-up off the capture stream (allocator, lazy init)
            for _ in range(warmup):
                optimizer.zero_grad(set_to_none=False)
                criterion(model(), target).backward()
                optimizer.step()
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=False)
        with torch.cuda.graph(self.graph):
            self.output = model()
            self.loss = criterion(self.output, target)
            self.loss.backward()
            optimizer.step()
            optimizer.zero_grad(set_to_none=False)

    def __call__(self) -> torch.Tensor:
        self.graph.replay()
        return self.loss

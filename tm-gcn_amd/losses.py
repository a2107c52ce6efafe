"""Opt-in fused class-weighted cross entropy (mean reduction) — the criterion of every reference
experiment script (`nn.CrossEntropyLoss(weight=class_weights)`, e.g.
experiment_reddit_our_link_prediction.py:69, 79).  The scripts' own criterion keeps working on
the module's logits; swap it for this one when E is in the millions: torch-ROCm's NLL reduction
kernels take 4.4 ms at E = 3.2 M (and normalise in fp32), this takes two streaming passes."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib


MAX_CLASSES = 8   # kLossMaxC in csrc/loss.hip


def weighted_ce(logits: torch.Tensor, target: torch.Tensor, weight: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """Σ w[t]·nll / Σ w[t] through the registered operator ``torch.ops.tmgcn.weighted_ce`` (C++ autograd
    over tmgcn_wce_fwd_f32 / tmgcn_wce_bwd_f32)."""
    return _lib.load_torch_ops().weighted_ce(logits, target, weight, int(ignore_index))


class WeightedCrossEntropy(nn.Module):
    """Drop-in for ``nn.CrossEntropyLoss(weight=class_weights)`` (mean reduction), C <= 8.
    Targets equal to ``ignore_index`` are skipped; any other label outside [0, C) makes the loss and
    the gradients NaN (torch device-asserts there) instead of silently shrinking the training set."""

    def __init__(self, weight: torch.Tensor, ignore_index: int = -100):
        super().__init__()
        self.ignore_index = int(ignore_index)
        self.register_buffer("weight", weight.detach().float().contiguous())

    def forward(self, output: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        w = self.weight if self.weight.device == output.device else self.weight.to(output.device)
        return weighted_ce(output.contiguous(), target.contiguous(), w, self.ignore_index)

"""Opt-in fused class-weighted cross entropy (mean reduction) — the criterion of every reference
experiment script (`nn.CrossEntropyLoss(weight=class_weights)`, e.g.
experiment_reddit_our_link_prediction.py:69, 79).  The scripts' own criterion keeps working on
the module's logits; swap it for this one when E is in the millions: torch-ROCm's NLL reduction
kernels take 4.4 ms at E = 3.2 M (and normalise in fp32), this takes two streaming passes."""
from __future__ import annotations

import torch
import torch.nn as nn

from . import _lib
from .ops import _ptr, _stream, _want


MAX_CLASSES = 8   # kLossMaxC in csrc/loss.hip


class _WCE(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, target, weight, ignore_index=-100):
        lib = _lib.load()
        _want(logits, "wce logits")
        _want(weight, "wce weight")
        _want(target, "wce target", torch.int64)
        E, Cn = logits.shape
        if weight.numel() != Cn or target.numel() != E:
            raise RuntimeError(f"wce: shapes logits {tuple(logits.shape)} target {tuple(target.shape)} weight {tuple(weight.shape)}")
        loss = torch.empty((), dtype=torch.float32, device=logits.device)
        stats = torch.empty(2, dtype=torch.float64, device=logits.device)
        ws = torch.empty(int(lib.tmgcn_wce_workspace_bytes(E)), dtype=torch.uint8, device=logits.device)
        _lib.check(lib.tmgcn_wce_fwd_f32(_ptr(logits), _ptr(target), _ptr(weight), E, Cn, int(ignore_index), _ptr(loss),
                                         _ptr(stats), _ptr(ws), ws.numel(), _stream(logits)), "tmgcn_wce_fwd_f32")
        ctx.ignore_index = int(ignore_index)
        ctx.save_for_backward(logits, target, weight, stats)
        return loss

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        logits, target, weight, stats = ctx.saved_tensors
        E, Cn = logits.shape
        dz = torch.empty_like(logits)
        g = g.contiguous().float()
        _lib.check(lib.tmgcn_wce_bwd_f32(_ptr(logits), _ptr(target), _ptr(weight), _ptr(stats), _ptr(g), E, Cn,
                                         ctx.ignore_index, _ptr(dz), _stream(logits)), "tmgcn_wce_bwd_f32")
        return dz, None, None, None


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
        return _WCE.apply(output.contiguous(), target.contiguous(), w, self.ignore_index)

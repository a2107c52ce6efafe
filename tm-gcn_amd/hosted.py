"""Device-resident results for host-resident scripts.

The reference's experiment scripts keep everything on the CPU: targets, class weights, the
criterion and the metric inputs are host tensors that get combined with the model's output
(``criterion(gcn(), target_train[...])``, ``ehf.compute_f1(guess, target)``,
experiment_reddit_our_link_prediction.py:77-105).  ``DeviceResult`` lets such a script run
unchanged while the output — and therefore the loss, its backward and the metrics — stays on the
MI355X: it is a ``torch.Tensor`` subclass whose ``__torch_function__`` sends the host tensors an
operation combines it with to its own device first.  Results of those operations are
``DeviceResult``s again, so the property carries through ``argmax``, slicing, the loss, ….

Two more conveniences for the scripts' idioms:
  * ``F.cross_entropy(input, target, weight)`` in its plain form (mean reduction, class-index
    targets, no label smoothing, C <= 8) — what ``nn.CrossEntropyLoss(weight=class_weights)``
    calls — is computed by the fused weighted-CE kernel (losses.py, csrc/loss.hip; same value,
    fp64 sums) instead of torch-ROCm's NLL reduction kernels, which take 4.4 of the 4.9 ms of a
    Reddit-sized epoch;
  * ``numpy`` conversion (``ep_acc_loss[ep] = [precision_train, …, loss_train, …]``) copies to the
    host instead of raising.
"""
from __future__ import annotations

import warnings

import torch
import torch.nn.functional as F
from torch.utils._pytree import tree_leaves, tree_map


_UPLOADS: dict = {}          # (data_ptr, nbytes) -> times a host tensor was sent to the device
_WARN_BYTES = 1 << 20
_warned = False


def _note_upload(x: torch.Tensor) -> None:
    """The uploads are a convenience, not free: a host tensor that takes part in every epoch (the scripts'
    targets) crosses PCIe every epoch.  Say so once instead of staying silent."""
    global _warned
    nbytes = x.numel() * x.element_size()
    if _warned or nbytes < _WARN_BYTES:
        return
    key = (x.data_ptr(), nbytes)
    n = _UPLOADS[key] = _UPLOADS.get(key, 0) + 1
    if len(_UPLOADS) > 64:
        _UPLOADS.clear()
    if n == 3:
        _warned = True
        warnings.warn(f"tmgcn_amd: a host tensor of {nbytes / 1e6:.0f} MB is combined with a device-resident result on every "
                      "call and is uploaded each time (about 20 us per MB); keep it on the device (`.cuda()`) to avoid that",
                      RuntimeWarning, stacklevel=4)


def _is_inplace_or_out(func, kwargs) -> bool:
    name = getattr(func, "__name__", "")
    return kwargs.get("out") is not None or (name.endswith("_") and not name.endswith("__"))


class DeviceResult(torch.Tensor):
    """See the module docstring.  Create with ``tensor.as_subclass(DeviceResult)`` (differentiable)."""

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        # DisableTorchFunctionSubclass is the guard PyTorch's own __torch_function__ documentation uses
        with torch._C.DisableTorchFunctionSubclass():      # attribute access below must not re-enter
            dev = None
            for a in tree_leaves((args, kwargs)):               # also inside lists: torch.cat((host, result))
                if isinstance(a, DeviceResult) and a.device.type != "cpu":
                    dev = a.device
                    break
            if dev is not None:
                def follow(x):
                    if isinstance(x, torch.Tensor) and x.device.type == "cpu":
                        _note_upload(x)
                        return x.to(dev)
                    return x

                if _is_inplace_or_out(func, kwargs):
                    # a host tensor that is written to stays where it is (cpu_buffer.copy_(result))
                    args = (args[0],) + tuple(tree_map(follow, a) for a in args[1:])
                    kwargs = {k: (v if k == "out" else tree_map(follow, v)) for k, v in kwargs.items()}
                else:
                    args = tuple(tree_map(follow, a) for a in args)
                    kwargs = {k: tree_map(follow, v) for k, v in kwargs.items()}
                if func is F.cross_entropy:
                    fused = _fused_cross_entropy(*args, **kwargs)
                    if fused is not None:
                        return fused
        return super().__torch_function__(func, types, args, kwargs)

    def __array__(self, dtype=None, copy=None):
        with torch._C.DisableTorchFunctionSubclass():
            a = self.detach().as_subclass(torch.Tensor).cpu().numpy()
        return a if dtype is None else a.astype(dtype, copy=False)


def _fused_cross_entropy(input, target, weight=None, size_average=None, ignore_index=-100, reduce=None,
                         reduction="mean", label_smoothing=0.0):
    """The fused kernel's result for the plain weighted / unweighted mean cross entropy, or None
    when the call uses anything it does not cover (torch's own implementation runs then)."""
    from .losses import weighted_ce, MAX_CLASSES
    if (size_average is not None or reduce is not None or reduction != "mean" or label_smoothing != 0.0
            or not isinstance(input, torch.Tensor) or input.dim() != 2 or input.dtype != torch.float32
            or not input.is_cuda or input.shape[0] == 0 or input.shape[1] > MAX_CLASSES
            or not isinstance(target, torch.Tensor) or target.dtype != torch.int64 or target.dim() != 1
            or target.shape[0] != input.shape[0]):
        return None
    if ignore_index >= 0 and ignore_index < input.shape[1]:
        return None                       # an in-range class is being ignored: not what the kernel does
    if weight is None:
        weight = torch.ones(input.shape[1], dtype=torch.float32, device=input.device)
    elif weight.dtype != torch.float32 or weight.numel() != input.shape[1]:
        return None
    plain = lambda x: x.as_subclass(torch.Tensor) if isinstance(x, DeviceResult) else x
    # labels outside [0, C) other than ignore_index come back as a NaN loss / NaN gradients (loss.hip),
    # where torch would device-assert: corrupt targets are loud either way
    out = weighted_ce(plain(input).contiguous(), plain(target).contiguous(), plain(weight).contiguous(), ignore_index)
    return out.as_subclass(DeviceResult)

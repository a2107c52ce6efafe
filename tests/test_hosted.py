"""hosted.DeviceResult mechanics that do not need a device (the cross-device behaviour is covered
on the MI355X in test_gpu_experiment.py)."""
import warnings

import numpy as np
import pytest
import torch
import torch.nn as nn

from tmgcn_amd import hosted
from tmgcn_amd.hosted import DeviceResult, _fused_cross_entropy


@pytest.fixture(params=["documented guard", "public API only"], autouse=True)
def scope(request):
    """Every test runs twice: with torch._C.DisableTorchFunctionSubclass (the guard PyTorch's own
    extension note uses) and with the module's flag scope, which needs public API only."""
    hosted._FORCE_FLAG_SCOPE = request.param == "public API only"
    yield
    hosted._FORCE_FLAG_SCOPE = False


def test_subclass_carries_through_the_scripts_operations_and_autograd():
    x = torch.randn(7, 2, requires_grad=True)
    out = (x * 2).as_subclass(DeviceResult)
    target = torch.tensor([0, 1, 0, 1, 1, 0, 0])
    w = torch.tensor([0.9, 0.1])
    loss = nn.CrossEntropyLoss(weight=w)(out, target)
    assert isinstance(loss, DeviceResult)
    want = nn.CrossEntropyLoss(weight=w)(x * 2, target)
    assert torch.equal(loss.as_subclass(torch.Tensor), want)              # host tensors: torch's own path, untouched
    loss.backward()
    assert type(x.grad) is torch.Tensor and x.grad.shape == x.shape
    guess = torch.argmax(out, dim=1)
    assert isinstance(guess, DeviceResult) and isinstance((guess == 0) & (target == 0), DeviceResult)
    K = torch.tensor(3)
    assert out[-K:].shape == (3, 2)


def test_numpy_and_python_scalar_conversions():
    out = (torch.arange(6.0).reshape(3, 2).requires_grad_(True) * 1).as_subclass(DeviceResult)
    s = out.sum()
    row = np.zeros(3)
    row[:] = [s, out[0, 1], 2.5]                                          # ep_acc_loss[ep] = [...] in the scripts
    assert row.tolist() == [15.0, 1.0, 2.5]
    assert np.asarray(out).shape == (3, 2) and np.asarray(out, dtype=np.float64).dtype == np.float64
    assert "%.3f" % s == "15.000" and float(s) == 15.0 and out.tolist()[2] == [4.0, 5.0]


def test_fused_cross_entropy_declines_what_the_kernel_does_not_cover():
    z, t = torch.randn(4, 2), torch.tensor([0, 1, 1, 0])
    assert _fused_cross_entropy(z, t) is None                               # host tensor: not ours
    # (shape / dtype / option screening happens before any device work)
    for kw in (dict(reduction="sum"), dict(label_smoothing=0.1), dict(ignore_index=1), dict(size_average=True)):
        assert _fused_cross_entropy(z, t, **kw) is None
    assert _fused_cross_entropy(z, t.float()) is None and _fused_cross_entropy(torch.randn(4, 9), t) is None


def test_host_operands_are_uploaded_once_per_content_and_keep_no_module_state():
    """The device copy and its version stamp live on the host tensor; an unchanged tensor is uploaded once, a tensor
    written to in between again; a large one that keeps changing warns once."""
    assert not any(n in vars(hosted) for n in ("_UPLOADS", "_warned"))
    dev = torch.device("meta")                                              # a device that needs no hardware
    big = torch.zeros(300_000)                                              # 1.2 MB > the 1 MB threshold
    a = hosted._device_copy(big, dev)
    assert a.device.type == "meta" and hosted._device_copy(big, dev) is a and big._tmgcn_uploads == 1
    with warnings.catch_warnings(record=True) as got:
        warnings.simplefilter("always")
        for _ in range(4):
            big.add_(1.0)                                                   # the version counter moves: stale copy
            assert hosted._device_copy(big, dev) is not a
    assert big._tmgcn_uploads == 5
    assert len([w for w in got if issubclass(w.category, RuntimeWarning)]) == 1
    small = torch.zeros(10)
    with warnings.catch_warnings(record=True) as got:
        warnings.simplefilter("always")
        for _ in range(5):
            small.add_(1.0)
            hosted._device_copy(small, dev)
    assert not got


def test_cross_entropy_reaches_torch_function_as_F_cross_entropy():
    """The script-mode criterion path (hosted.DeviceResult.__torch_function__) recognises `criterion(gcn(), target)` by
    `func is F.cross_entropy`.  That nn.CrossEntropyLoss dispatches exactly that function object through
    __torch_function__ is a property of the installed torch: this test is the guard for a torch upgrade."""
    import torch.nn.functional as F
    seen = []

    class Probe(torch.Tensor):
        @classmethod
        def __torch_function__(cls, func, types, args=(), kwargs=None):
            seen.append(func)
            return super().__torch_function__(func, types, args, kwargs or {})

    x = torch.randn(5, 3).as_subclass(Probe)
    torch.nn.CrossEntropyLoss(weight=torch.tensor([0.2, 0.3, 0.5]))(x, torch.tensor([0, 1, 2, 1, 0]))
    assert any(f is F.cross_entropy for f in seen), [getattr(f, "__name__", f) for f in seen]

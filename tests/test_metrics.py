"""Metrics against golden vectors captured from the real reference
(tests/golden/make_golden_metrics.py: ehf.compute_f1, ehf.compute_MAP_MRR)."""
import numpy as np
import pytest
import torch

from _util import golden
from tmgcn_amd import metrics


def _case(d, c, device):
    return (torch.from_numpy(d[f"c{c}_logits"]).to(device), torch.from_numpy(d[f"c{c}_target"]).to(device),
            torch.from_numpy(d[f"c{c}_edges"]).to(device))


def _check(device):
    d = golden("g7_metrics")
    for c in range(int(d["n_cases"])):
        logits, target, edges = _case(d, c, device)
        p, r, f1 = metrics.compute_f1(logits.argmax(1), target)
        assert np.allclose([float(p), float(r), float(f1)], d[f"c{c}_f1"], rtol=1e-12)
        MAP, MRR = metrics.compute_MAP_MRR(logits, target, edges)
        assert abs(float(MAP) - float(d[f"c{c}_map"])) <= 1e-9, (c, float(MAP), float(d[f"c{c}_map"]))
        assert abs(float(MRR) - float(d[f"c{c}_mrr"])) <= 1e-9, (c, float(MRR), float(d[f"c{c}_mrr"]))


def test_metrics_match_reference_cpu():
    _check("cpu")


@pytest.mark.gpu
def test_metrics_match_reference_on_device():
    _check("cuda")


def test_public_per_slice_helpers():
    """get_MAP / get_MRR / get_row_MRR (ehf:669-711) are part of the surface too."""
    d = golden("g7_metrics")
    logits, target, edges = _case(d, 2, "cpu")                      # the single-slice case
    MAP, MRR = metrics.compute_MAP_MRR(logits, target, edges)
    assert abs(float(metrics.get_MAP(logits, target, True)) - float(MAP)) <= 1e-12
    assert abs(float(metrics.get_MRR(logits, target, edges[1:3], False)) - float(MRR)) <= 1e-12
    assert abs(float(metrics.get_MAP(torch.softmax(logits, 1)[:, 0], target, False)) - float(MAP)) <= 1e-12
    probs = np.array([0.1, 0.9, 0.5, 0.7, 0.3])
    true = np.array([0, 1, 0, 0, 1])                                  # class-0 entries rank 5, 3, 2
    assert abs(float(metrics.get_row_MRR(probs, true)) - (1 / 5 + 1 / 3 + 1 / 2) / 3) <= 1e-15

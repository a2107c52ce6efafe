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

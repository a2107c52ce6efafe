"""The example scripts run end to end on the MI355X (as child processes, like a user would)."""
import os
import subprocess
import sys

import pytest

from _util import ROOT

pytestmark = pytest.mark.gpu


def _run(*argv):
    r = subprocess.run([sys.executable, *argv], cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


@pytest.mark.parametrize("layers", [1, 2])
def test_mat_link_prediction_example(layers):
    out = _run(os.path.join("examples", "experiment_mat_link_prediction.py"), "--epochs", "40", "--layers", str(layers),
               "--nodes", "300", "--edges-per-slice", "400", "--eval-every", "20")
    assert "summary: {" in out and '"logits_device": "cuda' in out and "test: MAP" in out


@pytest.mark.parametrize("extra", [(), ("--graph",)], ids=["eager", "hipgraph"])
def test_synthetic_example(extra):
    out = _run(os.path.join("examples", "experiment_synthetic_our.py"), "--epochs", "30", *extra)
    assert "adjacency pipeline on the device" in out
    assert ("hipGraph replay" if extra else "eager") in out and "precision/recall/f1" in out

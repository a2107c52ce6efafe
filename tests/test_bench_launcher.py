"""bench.py's own multi-rank launcher and its deadline, exercised without a GPU:
`python bench.py --gpus N` with no torchrun around it must start N fresh ranks itself, relay their
failure as a non-zero exit code, and a stalled job must end — stacks dumped, non-zero — inside the
deadline instead of hanging until somebody's timeout ("What's missing" #1)."""
import os
import subprocess
import sys
import time

from _util import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_self_launch_starts_the_ranks_and_relays_their_exit_code():
    """No GPU here: every rank stops at `bench.py needs MI355X GPUs`; the parent (which never imports
    torch) must have launched 2 ranks through torch.distributed.run and must come back non-zero."""
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--deadline", "120", "--watchdog", "0"], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "launching 2 ranks" in r.stderr and "torch.distributed.run" in r.stderr
    assert "[bench r0" in r.stderr and "[bench r1" in r.stderr            # both ranks started
    assert "needs MI355X GPUs" in r.stderr
    assert r.stdout.strip() == ""                                          # no JSON line from a failed job
    assert time.time() - t0 < 200


def test_a_stalled_job_ends_non_zero_inside_the_deadline():
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--deadline", "8", "--watchdog", "3", "--selftest-stall"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert r.returncode != 0
    assert "selftest: stalling on purpose" in r.stderr
    assert "watchdog: still running after 3 s" in r.stderr                 # soft dump first
    assert "Timeout (0:00:08)" in r.stderr                                 # faulthandler's hard deadline fired, with stacks
    assert took < 280, took                                                # generous: a cold `import torch` in the launcher alone can take 1-2 min


def test_single_rank_form_rejects_a_world_size_mismatch():
    env = _env()
    env.update(RANK="0", WORLD_SIZE="4", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in (r.stderr + r.stdout)

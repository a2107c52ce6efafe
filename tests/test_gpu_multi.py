"""GPU, MORE THAN ONE DEVICE: the slice-sharded layer and the slice-sharded drop-in models over RCCL
(`nccl` backend), one fresh process per GPU — turns itself on wherever torch.cuda.device_count() >= 2
and reports "skipped" on a single-GPU box.  World sizes 2, 4 and 8 as the node allows.

  * ShardedTMGCNLayer in both exchange modes ("a2a": pipelined per-slice all-to-all beside the fused
    kernel; "allgather": node-chunked all-gather fused with P1, against the literal unchunked form and
    ragged chunkings, bit for bit) against the unsharded layer computed by the same kernels on every
    rank — forward, dX, dW                                              (ehf:204, 206-207, 222)
  * sharded EmbeddingGCN2 / EmbeddingGCN / EmbeddingKWGCN against the reference's fixtures (G2-G4):
    logits in the caller's edge order on every rank, loss, every parameter gradient, the
    validation-style call, SGD trajectories with bit-identical replicas.

The workers are the ones the single-GPU emulation runs (tests/test_gpu_dist2.py and
tests/test_gpu_dist_models.py: same processes sharing cuda:0 over gloo), started with backend="nccl"."""
import pytest
import torch
import torch.multiprocessing as mp

NGPU = torch.cuda.device_count()          # counting devices does not initialise the GPU
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(NGPU < 2, reason=f"needs >= 2 GPUs for RCCL ranks on distinct devices (found {NGPU})")]
# world 2 and the largest power of two the node offers (each case starts fresh ranks and a fresh RCCL
# communicator: ~10-15 s apiece, so the matrix is kept to what distinguishes code paths)
WORLDS = sorted({2, max([w for w in (2, 4, 8) if w <= NGPU] or [2])})


CASE_DEADLINE_S = 420      # a stalled collective must end as a failed test, not as a hung suite


def _run(worker, world, args):
    """Start `world` fresh ranks and wait for them WITH A DEADLINE: ranks still alive after it are
    killed and the case fails with what every rank had reported so far."""
    import time
    from _util import free_port
    ret = mp.Manager().dict()
    ctx = mp.spawn(worker, args=(world, free_port(), *args, ret, "nccl"), nprocs=world, join=False)
    t_end = time.monotonic() + CASE_DEADLINE_S
    done = False
    try:
        while not done and time.monotonic() < t_end:
            done = ctx.join(timeout=5)
    finally:
        if not done:
            for p in ctx.processes:
                if p.is_alive():
                    p.kill()
    assert done, f"ranks still running after {CASE_DEADLINE_S} s (killed); reports so far: {dict(ret)}"
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}:\n{ret.get(r)}"


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("exchange,F0,F1,pipeline,condensed,act", [
    ("a2a", 16, 32, True, True, None),          # fused MFMA kernel, per-slice all-to-all pipelined beside it
    ("a2a", 16, 32, False, False, "selu"),      # same kernels, exchange first; one weight per slice
    ("a2a", 6, 6, True, False, "selu"),         # fused small-F kernel
    ("a2a", 5, 7, True, True, None),            # widths without a fused kernel
    ("allgather", 16, 32, True, True, None),    # node-chunked all-gather (+ literal form and ragged chunkings, bit for bit)
    ("allgather", 16, 32, True, False, "selu")])
def test_sharded_layer_over_rccl(world, exchange, F0, F1, pipeline, condensed, act):
    from test_gpu_dist2 import _worker
    _run(_worker, world, (exchange, F0, F1, condensed, act, pipeline))


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("name", ["g3_gcn2_twice_selu_condensed1", "g3_gcn2_three_relu_condensed1", "g3_gcn2_default_leaky_condensed1",
                                  "g2_gcn_minv_fp32", "g4_kwgcn_2layer_selu"])
def test_sharded_models_over_rccl(world, name):
    from test_gpu_dist_models import CASES, _worker
    assert name in CASES
    _run(_worker, world, (name,))


@pytest.mark.parametrize("world", WORLDS)
@pytest.mark.parametrize("exchange", ["a2a", "allgather"])
def test_bench_over_rccl_verifies_itself(world, exchange, tmp_path):
    """bench.py as the driver runs it (self-launched ranks, RCCL, one GPU each) at a reduced N: rc 0, the
    JSON line reports the world size RCCL itself counted, and the verify leg accepted the sharded step."""
    import json
    import os
    import subprocess
    import sys
    from _util import ROOT
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--exchange", exchange,
                        "--nodes", "200000", "--steps", "2", "--warmup", "1", "--gather-chunk-nodes", "60000",
                        "--no-compare-exchange", "--deadline", "600"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads(r.stdout.strip().split("\n")[-1])
    assert line["n_gpus"] == world and line["ranks"]["rccl_ranks"] == world
    assert len({(d["pci_bus_id"], d["uuid"], d["device"]) for d in line["ranks"]["devices"]}) == world   # distinct devices
    assert line["verify"]["ok"], line["verify"]
    assert line["config"]["exchange"] == exchange
    if exchange == "allgather":
        assert line["config"]["gather_chunks"] == 4

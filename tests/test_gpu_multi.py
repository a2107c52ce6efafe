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


def _run(worker, world, args):
    from _util import free_port
    ret = mp.Manager().dict()
    mp.spawn(worker, args=(world, free_port(), *args, ret, "nccl"), nprocs=world, join=True)
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

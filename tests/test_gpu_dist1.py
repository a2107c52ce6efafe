"""GPU: the sharded layer's exchange code path through RCCL at world size 1 (the GPU box has one
GPU; world_size-2 logic is covered on CPU by test_dist_gloo.py).  Checks that the pipelined
per-slice all-to-all + fused kernel path, the plain a2a path and the all-gather path all
reproduce the unsharded layer on the device."""
import os

import pytest
import torch
import torch.distributed as dist

from _util import REL_TOL, assert_close, free_port
from tmgcn_amd import synth
from tmgcn_amd.csr import BatchedCSR
from tmgcn_amd.dist import ShardedTMGCNLayer

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pg():
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange,pipeline,F0,chunk", [("a2a", True, 16, None), ("a2a", False, 16, None), ("a2a", True, 6, None),
                                                        ("allgather", True, 16, None),   # node-chunked gather, one chunk
                                                        ("allgather", True, 16, 50),     # three chunks, ragged tail
                                                        ("allgather", True, 16, 0)])     # the literal unchunked form
@pytest.mark.parametrize("condensed,act", [(True, None), (False, "leaky")])
def test_world1_rccl_paths_match_unsharded(pg, exchange, pipeline, F0, chunk, condensed, act):
    T, N, F1 = 6, 120, 32
    g = synth.dynamic_graph(T, N, edges_per_slice=300, seed=2, no_diag=4, F0=F0)
    A = BatchedCSR.from_scipy_list(g.Ct, device="cuda")
    gen = torch.Generator().manual_seed(4)
    X0 = torch.from_numpy(g.X).float().cuda()
    W0 = (torch.randn(*(() if condensed else (T,)), F0, F1, generator=gen) * 0.3).cuda()
    dY = torch.randn(T, N, F1, generator=gen).cuda()
    res = []
    for force in (False, True):
        layer = ShardedTMGCNLayer(A, g.M, T, exchange=exchange, pipeline=pipeline, force_collectives=force,
                                  gather_chunk_nodes=chunk)
        assert layer.collective == force
        X = X0.clone().requires_grad_(True)
        W = W0.clone().requires_grad_(True)
        Y = layer(X, W, act=act)
        Y.backward(dY)
        torch.cuda.synchronize()
        res.append((Y.detach(), X.grad, W.grad))
        if force and exchange == "a2a":
            assert_close(layer.to_node_sharded(Y.detach()), Y.detach(), 0.0, "to_node_sharded at G=1")
    for a, b, what in zip(res[1], res[0], ("Y", "dX", "dW")):
        assert_close(a, b, REL_TOL, f"{exchange} pipeline={pipeline} {what}")


def test_cu_masked_streams_run_the_pipelined_path(pg):
    """The pipelined a2a path on CU-masked compute streams (what every real multi-GPU run uses: 32 CUs
    left to RCCL's kernels): same bits as on ordinary streams, and the stream really is a distinct,
    usable HIP stream."""
    from tmgcn_amd.dist import cu_masked_stream
    T, N, F0, F1 = 6, 120, 16, 32
    g = synth.dynamic_graph(T, N, edges_per_slice=300, seed=2, no_diag=4, F0=F0)
    A = BatchedCSR.from_scipy_list(g.Ct, device="cuda")
    gen = torch.Generator().manual_seed(4)
    X0 = torch.from_numpy(g.X).float().cuda()
    W0 = (torch.randn(F0, F1, generator=gen) * 0.3).cuda()
    dY = torch.randn(T, N, F1, generator=gen).cuda()
    res = []
    for cu in (0, 32):
        layer = ShardedTMGCNLayer(A, g.M, T, exchange="a2a", force_collectives=True, cu_reserve=cu)
        assert layer.cu_reserve == cu
        X, W = X0.clone().requires_grad_(True), W0.clone().requires_grad_(True)
        for _ in range(2):
            X.grad = W.grad = None
            Y = layer(X, W, act="relu")
            Y.backward(dY)
        torch.cuda.synchronize()
        if cu:
            lanes = layer.compute_lanes(torch.cuda.current_stream())
            assert len(lanes) == 2 and all(s.cuda_stream != torch.cuda.current_stream().cuda_stream for s in lanes)
        res.append((Y.detach(), X.grad, W.grad))
    for a, b in zip(*res):
        assert torch.equal(a, b)
    s = cu_masked_stream("cuda", 250)                    # nearly everything masked away still runs (6 CUs)
    with torch.cuda.stream(s):
        z = torch.arange(1 << 20, device="cuda").float().sum()
    s.synchronize()
    assert float(z) == float(sum(range(1 << 20)))
    # the multi-GPU defaults: CU mask on, slot reserve off
    import inspect
    src = inspect.getsource(ShardedTMGCNLayer.__init__)
    assert '"32" if real_exchange else "0"' in src and "grid_reserve = 0" in src


MODEL_CASES = {
    "gcn2_twice_selu": ("gcn2", dict(condensed_W=True, use_Minv=False, apply_M_twice=True, nonlin2="selu")),
    "gcn2_three_relu_per_slice_W": ("gcn2", dict(condensed_W=False, use_Minv=False, apply_M_twice=True,
                                                  apply_M_three_times=True, nonlin2="relu")),
    "gcn2_default_leaky": ("gcn2", dict(condensed_W=True, use_Minv=False, nonlin2="leaky")),
    "gcn_minv": ("gcn", dict(condensed_W=True, use_Minv=True)),
    "gcn_per_slice_W": ("gcn", dict(condensed_W=False, use_Minv=False)),
    "kwgcn_2layer_selu": ("kw", dict(nonlin2="selu")),
}


@pytest.mark.parametrize("name", sorted(MODEL_CASES))
def test_world1_rccl_sharded_models_match_unsharded(pg, name):
    """The drop-in models with ``group=`` — SliceShard's padded all-gather / reduce-scatter, the
    all-reduce of shared weights' gradients and the rank-major logits gather — through RCCL itself
    (world size 1: the collectives are issued and take RCCL's dtype / contiguity checks; the
    world-size-2/3 logic is pinned on CPU by test_dist_models_gloo.py and on the device over gloo by
    test_gpu_dist_models.py).  Same seed => same weights; logits and every gradient must agree."""
    import tmgcn_amd.layers as ehf
    kind, kw = MODEL_CASES[name]
    T, N = 7, 90
    g = synth.dynamic_graph(T, N, edges_per_slice=200, seed=5, no_diag=3, F0=3)
    At, X, M = g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.M)
    edges = torch.from_numpy(g.edges)
    gen = torch.Generator().manual_seed(11)
    out = []
    for group in (None, dist.group.WORLD):
        torch.manual_seed(3)
        if kind == "gcn":
            m = ehf.EmbeddingGCN(At, X, edges, M, hidden_feat=[6, 2], group=group, **kw)
        elif kind == "gcn2":
            m = ehf.EmbeddingGCN2(At, X, edges, M, hidden_feat=[6, 6, 2], group=group, **kw)
        else:
            m = ehf.EmbeddingKWGCN(At, X, edges, hidden_feat=[6, 6, 2], group=group, **kw)
        logits = m()
        if not out:
            dlogits = torch.randn(logits.shape, generator=gen).cuda()
        logits.backward(dlogits)
        torch.cuda.synchronize()
        out.append((logits.detach(), {n: q.grad.clone() for n, q in m.named_parameters()}))
    assert_close(out[1][0], out[0][0], REL_TOL, f"{name} logits")
    for n in out[0][1]:
        assert_close(out[1][1][n], out[0][1][n], REL_TOL, f"{name} d{n}")

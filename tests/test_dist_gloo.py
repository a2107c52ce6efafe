"""CPU, world_size 2 and 4, gloo: the slice-sharded layer (tmgcn_amd.dist) reproduces the unsharded
layer — forward, dX, dW — in both exchange modes.  The device kernels are substituted by the
oracle here (no GPU in this environment); the collectives and the sharding arithmetic are the
product's."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _setup(rank, world, port):
    for p in (ROOT, HERE):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from tmgcn_amd import ops
    from _oracle_kernels import OracleKernels
    ops.kernels = OracleKernels()


def _problem(T, N, F0, F1, b, condensed):
    from tmgcn_amd import synth
    g = synth.dynamic_graph(T, N, edges_per_slice=3 * N, seed=7, no_diag=b, F0=F0)
    gen = torch.Generator().manual_seed(3)
    W = torch.randn(F0, F1, generator=gen) if condensed else torch.randn(T, F0, F1, generator=gen)
    dY = torch.randn(T, N, F1, generator=gen)
    return g, torch.from_numpy(g.X).float(), W, dY


def _worker(rank, world, port, exchange, condensed, act, F0, ret):
    try:
        _setup(rank, world, port)
        from tmgcn_amd.csr import BatchedCSR
        from tmgcn_amd.dist import ShardedTMGCNLayer, even_bounds
        T, N, F1, b = 8, (30 if world == 2 else 8 * world), 6, 5
        g, X, W, dY = _problem(T, N, F0, F1, b, condensed)
        k0, k1 = even_bounds(T, world)[rank]
        n0, n1 = even_bounds(N, world)[rank]
        A_local = BatchedCSR.from_scipy_list(g.Ct).slices(k0, k1)
        layer = ShardedTMGCNLayer(A_local, g.M, T, group=None, exchange=exchange)
        assert layer.G == world and layer.k0 == k0
        # multi-GPU defaults: the pipelined a2a leaves 32 CUs to RCCL through CU-masked streams (device
        # tensors only; nothing of it is touched on the CPU), no block-slot reserve; the all-gather form needs neither
        assert layer.grid_reserve == 0 and layer.cu_reserve == (32 if exchange == "a2a" else 0)
        Xin = (X[:, n0:n1] if exchange == "a2a" else X[k0:k1]).contiguous().clone().requires_grad_(True)
        Wl = (W if condensed else W[k0:k1]).contiguous().clone().requires_grad_(True)
        Y = layer(Xin, Wl, act=act)
        Y.backward(dY[k0:k1].contiguous())

        # unsharded answer, computed redundantly on every rank
        dist.barrier()
        Yr, dXr, dWr = _reference_local(g, X, W, dY, act)
        tol = 1e-5
        def close(a, b, what):
            err = float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))
            assert err <= tol, f"{what}: {err:.2e}"
        close(Y.detach(), Yr[k0:k1], "Y")
        close(Xin.grad, dXr[:, n0:n1] if exchange == "a2a" else dXr[k0:k1], "dX")
        close(Wl.grad, dWr if condensed else dWr[k0:k1], "dW")
        if exchange == "a2a":  # slice-sharded output back to node-sharded (input of a next layer)
            Yn = layer.to_node_sharded(Y.detach())
            close(Yn, Yr[:, n0:n1], "to_node_sharded")
        else:
            # the all-gather ran node-chunked (automatic chunk size: one chunk at this N).  The literal
            # unchunked form and a ragged multi-chunk split must give the same bits: per output
            # element the arithmetic is the same, only the grouping of the collectives differs.
            assert layer.gather_chunk_nodes == N and len(layer.gather_chunks()) == 1
            for chunk in (0, 7, 1):
                l2 = ShardedTMGCNLayer(A_local, g.M, T, group=None, exchange="allgather", gather_chunk_nodes=chunk)
                X2 = Xin.detach().clone().requires_grad_(True)
                W2 = Wl.detach().clone().requires_grad_(True)
                Y2 = l2(X2, W2, act=act)
                Y2.backward(dY[k0:k1].contiguous())
                if chunk:
                    assert len(l2.gather_chunks()) == -(-N // chunk) and l2._gbufs[0].numel() == T * chunk * F0
                else:
                    assert l2._gbufs is None                    # the literal form materialises [T,N,F] instead
                assert torch.equal(Y2.detach(), Y.detach()), f"Y differs, chunk={chunk}"
                assert torch.equal(X2.grad, Xin.grad), f"dX differs, chunk={chunk}"
                assert torch.equal(W2.grad, Wl.grad), f"dW differs, chunk={chunk}"
        dist.barrier()
        ret[rank] = "ok"
    except Exception as e:  # surface the failure in the parent
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _reference_local(g, X, W, dY, act):
    """Unsharded layer on this process only (no collectives): dense fp64 einsum + autograd."""
    from oracle import tmgcn_oracle as orc
    from tmgcn_amd.csr import BatchedCSR
    A = BatchedCSR.from_scipy_list(g.Ct).to_dense().double()
    M = torch.from_numpy(g.M)
    X = X.double().clone().requires_grad_(True)
    W = W.double().clone().requires_grad_(True)
    Xt = torch.einsum("kj,jnf->knf", M, X)
    AX = torch.einsum("knm,kmf->knf", A, Xt)
    pre = AX @ W if W.dim() == 2 else torch.einsum("knf,kfg->kng", AX, W)
    Y = orc.ACTS[act](pre) if act else pre
    Y.backward(dY.double())
    return Y.detach(), X.grad, W.grad


@pytest.mark.parametrize("exchange", ["a2a", "allgather"])
@pytest.mark.parametrize("condensed,act", [(True, None), (False, "selu")])
@pytest.mark.parametrize("F0", [4, 16])  # 16: widths the fused kernel takes -> pipelined per-slice exchange
def test_sharded_layer_matches_unsharded(exchange, condensed, act, F0):
    world = 2
    from _util import free_port
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, exchange, condensed, act, F0, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}:\n{ret.get(r)}"


@pytest.mark.parametrize("exchange,condensed,act,F0", [("a2a", True, "relu", 16), ("a2a", False, None, 4),
                                                       ("allgather", True, None, 16)])
def test_sharded_layer_world_size_4(exchange, condensed, act, F0):
    """Four ranks: two slices and eight nodes per rank — the group-interleaved send / receive layouts
    with more than one peer on either side."""
    world = 4
    from _util import free_port
    port = free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, exchange, condensed, act, F0, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}:\n{ret.get(r)}"


def test_even_bounds():
    from tmgcn_amd.dist import even_bounds
    assert even_bounds(128, 8)[3] == (48, 64)
    b = even_bounds(10, 4)
    assert b == [(0, 3), (3, 6), (6, 8), (8, 10)]


def test_memory_plan_of_the_headline_config():
    """S4 at G = 8 (T = 128, N = 2 M, F = 128, 66 M stored non-zeros per slice): north_star's literal
    all-gather needs [T,N,F] twice (it cannot fit a 288 GB device); node-chunked it needs two 8 GB
    buffers and fits with room to spare, as the all-to-all form does."""
    from tmgcn_amd.dist import chunk_nodes_for, memory_plan
    T, G, N, F, nnz = 128, 8, 2_000_000, 128, 16 * 66_000_000
    literal = memory_plan("allgather", T, G, N, F, F, nnz, gather_chunk_nodes=0)
    chunked = memory_plan("allgather", T, G, N, F, F, nnz)
    a2a = memory_plan("a2a", T, G, N, F, F, nnz)
    assert literal["exchange"] == 2 * T * N * F * 4 and literal["total"] > 288e9
    assert chunked["total"] < 200e9 and a2a["total"] < 200e9
    nc = chunk_nodes_for(T, N, F)
    assert chunked["exchange"] == 2 * T * nc * F * 4 <= 2 * (8 << 30) and -(-N // nc) == 16
    assert memory_plan("none", 16, 1, N, F, F, nnz)["exchange"] == 0
    # equal-sized chunks, never more than the target, never more than N
    assert chunk_nodes_for(8, 30, 4) == 30 and chunk_nodes_for(128, 2_000_000, 128, target=1 << 30) * 128 * 128 * 4 <= 1 << 30


def _fuzz_worker(rank, world, port, n_cases, ret):
    try:
        _setup(rank, world, port)
        import numpy as np
        from tmgcn_amd import synth
        from tmgcn_amd.csr import BatchedCSR
        from tmgcn_amd.dist import ShardedTMGCNLayer, even_bounds
        rng = np.random.default_rng(1234)                       # the same stream on every rank: same cases
        for case in range(n_cases):
            Tl = int(rng.integers(1, 4))
            T = Tl * world
            N = int(rng.integers(world, 6)) * world             # divisible by the world size (a2a)
            F0, F1 = int(rng.choice([1, 3, 4, 8])), int(rng.choice([1, 2, 5]))
            b = int(rng.integers(1, T + 1))
            dense_m = bool(rng.integers(0, 2))
            chunk = int(rng.integers(1, N + 1))
            g = synth.dynamic_graph(T, N, edges_per_slice=3 * N, seed=int(rng.integers(1 << 30)), no_diag=b, F0=F0)
            M = g.M
            if dense_m:                                          # a dense lower-triangular mixing matrix (the Minv shape)
                M = np.tril(rng.standard_normal((T, T))) + 2.0 * np.eye(T)
            gen = torch.Generator().manual_seed(case)
            X = torch.from_numpy(g.X).float()
            W = torch.randn(F0, F1, generator=gen)
            dY = torch.randn(T, N, F1, generator=gen)
            k0, k1 = even_bounds(T, world)[rank]
            n0, n1 = even_bounds(N, world)[rank]
            A_local = BatchedCSR.from_scipy_list(g.Ct).slices(k0, k1)
            g.M = M
            Yr, dXr, dWr = _reference_local(g, X, W, dY, None)
            outs = {}
            for name, kw, xin in (("a2a", dict(exchange="a2a"), X[:, n0:n1]),
                                  ("literal", dict(exchange="allgather", gather_chunk_nodes=0), X[k0:k1]),
                                  ("chunked", dict(exchange="allgather", gather_chunk_nodes=chunk), X[k0:k1])):
                layer = ShardedTMGCNLayer(A_local, M, T, group=None, **kw)
                Xi = xin.contiguous().clone().requires_grad_(True)
                Wi = W.clone().requires_grad_(True)
                Y = layer(Xi, Wi)
                Y.backward(dY[k0:k1].contiguous())
                outs[name] = (Y.detach(), Xi.grad, Wi.grad)
                what = f"case {case} (T={T} N={N} F={F0}->{F1} b={b} dense={dense_m} chunk={chunk}) {name}"
                for got, ref, q in ((Y.detach(), Yr[k0:k1], "Y"), (Xi.grad, dXr[:, n0:n1] if name == "a2a" else dXr[k0:k1], "dX"),
                                    (Wi.grad, dWr, "dW")):
                    err = float((got.double() - ref).abs().max() / max(float(ref.abs().max()), 1e-30))
                    assert err <= 1e-5, f"{what} {q}: {err:.2e}"
            for a, c in zip(outs["literal"], outs["chunked"]):
                assert torch.equal(a, c), f"case {case}: chunked all-gather differs from the literal form (chunk={chunk})"
        dist.barrier()
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_layer_fuzz_shapes_bands_and_chunkings(world):
    """Seeded random cases (slices per rank, nodes, widths, band width or a dense triangular M, chunk
    size) through all three exchange forms: each reproduces the unsharded fp64 layer — forward, dX, dW —
    and the node-chunked all-gather reproduces the literal one bit for bit."""
    from _util import free_port
    ret = mp.Manager().dict()
    mp.spawn(_fuzz_worker, args=(world, free_port(), 12, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}:\n{ret.get(r)}"


def test_cu_mask_words():
    """The CU mask handed to hipExtStreamCreateWithCUMask: every CU but the last `cus_free`."""
    from tmgcn_amd.dist import cu_mask_words
    assert cu_mask_words(256, 0) == [0xFFFFFFFF] * 8
    m = cu_mask_words(256, 32)
    assert m == [0xFFFFFFFF] * 7 + [0] and sum(bin(w).count("1") for w in m) == 224
    m = cu_mask_words(256, 16)
    assert m[-1] == 0x0000FFFF and sum(bin(w).count("1") for w in m) == 240
    m = cu_mask_words(304, 40)                           # a CU count that is not a multiple of 32 (MI300X)
    assert len(m) == 10 and sum(bin(w).count("1") for w in m) == 264 and m[-1] == 0
    assert sum(bin(w).count("1") for w in cu_mask_words(8, 100)) == 1      # never the empty mask

"""CPU, world_size 2 (and 3: uneven shards), gloo: the slice-sharded drop-in models
(EmbeddingGCN / EmbeddingGCN2 / EmbeddingKWGCN with group=…) reproduce the REFERENCE's fixtures
rank-wise — logits in the caller's edge order on every rank, every parameter gradient after the
all-reduce, and the validation-style call (layer 2 on the training adjacency, ehf:339-348).
Fixtures G2 / G3 / G4 come from the real embedding_help_functions (tests/golden/make_golden.py).
The device kernels are substituted by the oracle here (no GPU in this environment); the
sharding arithmetic and the collectives are the product's (tm-gcn_amd/dist.py: SliceShard)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CASES = {
    # fixture -> (class, kwargs)
    "g3_gcn2_twice_selu_condensed1": ("gcn2", dict(condensed_W=True, use_Minv=False, apply_M_twice=True, nonlin2="selu")),
    "g3_gcn2_twice_selu_condensed0": ("gcn2", dict(condensed_W=False, use_Minv=False, apply_M_twice=True, nonlin2="selu")),
    "g3_gcn2_three_relu_condensed1": ("gcn2", dict(condensed_W=True, use_Minv=False, apply_M_twice=True,
                                                    apply_M_three_times=True, nonlin2="relu")),
    "g3_gcn2_default_leaky_condensed1": ("gcn2", dict(condensed_W=True, use_Minv=False, nonlin2="leaky")),
    "g2_gcn_condensed0": ("gcn", dict(condensed_W=False, use_Minv=False)),
    "g2_gcn_minv_fp32": ("gcn", dict(condensed_W=True, use_Minv=True)),
    "g4_kwgcn_2layer_selu": ("kw", dict(nonlin2="selu")),
}


def _worker(rank, world, port, name, ret):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        from tmgcn_amd import ops
        from _oracle_kernels import OracleKernels
        from _util import coo_list, golden
        import tmgcn_amd.layers as ehf
        ops.kernels = OracleKernels()

        d = golden(name)
        kind, kw = CASES[name]

        def inputs(prefix=""):
            X = torch.from_numpy(d[prefix + "X"])
            T, N = X.shape[0], X.shape[1]
            return dict(T=T, N=N, X=X, M=torch.from_numpy(d[prefix + "M"]), edges=torch.from_numpy(d[prefix + "edges"]),
                        labels=torch.from_numpy(d[prefix + "labels"]), At=coo_list(d, "At", T, N, prefix=prefix),
                        A=coo_list(d, "A", T, N, prefix=prefix) if prefix + "A_k" in d else None)

        i = inputs()
        torch.manual_seed(int(d["seed"]))
        group = dist.group.WORLD
        if kind == "gcn":
            m = ehf.EmbeddingGCN(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 2], device="cpu", group=group, **kw)
        elif kind == "gcn2":
            m = ehf.EmbeddingGCN2(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 6, 2], device="cpu", group=group, **kw)
        else:
            m = ehf.EmbeddingKWGCN(i["A"], i["X"], i["edges"], hidden_feat=[6, 5, 2], device="cpu", group=group, **kw)
        sh = m._shard
        assert sh.G == world and sh.Tl in (i["T"] // world, i["T"] // world + 1)
        for n, q in m.named_parameters():                                   # replicated, bit-identical to the reference's draws
            assert torch.equal(q.detach(), torch.from_numpy(d[n + "0"])), n
        # only this rank's slices are held
        held = m.AtXt if kind != "kw" else m.AX
        assert held.shape[0] == sh.Tl

        def close(a, b, what, tol=1e-5):
            a, b = torch.as_tensor(a).double(), torch.as_tensor(b).double()
            err = float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))
            assert err <= tol, f"{what}: {err:.2e}"

        out = m()
        assert out.shape == tuple(d["logits"].shape)
        close(out.detach(), d["logits"], "logits")
        loss = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))(out, i["labels"])
        close(loss.detach(), float(d["loss"]), "loss")
        m.zero_grad()
        loss.backward()
        for n, q in m.named_parameters():
            close(q.grad, d["d" + n], "d" + n)
        # gcn.loss(criterion, target) on a sharded model: the unfused statements (every rank evaluates the same loss on
        # the gathered logits), same value and gradients as criterion(gcn(), target)
        m.zero_grad()
        loss_f = m.loss(torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1])), i["labels"])
        close(loss_f.detach(), float(d["loss"]), "loss through gcn.loss()")
        loss_f.backward()
        for n, q in m.named_parameters():
            close(q.grad, d["d" + n], "d" + n + " through gcn.loss()")
        if "logits_val" in d.files:                                         # validation-style call on another window
            v = inputs("val_")
            with torch.no_grad():
                ov = m(v["A"] if kind == "kw" else v["At"], v["X"], v["edges"])
            close(ov, d["logits_val"], "validation logits")
            if kind == "kw":
                # the same window handed over as a pre-built BatchedCSR (shorter than T): ranks whose
                # slices all lie behind the window must get an empty shard, not fail alone while the
                # others wait in gather_rows' all-gather (layers._Sharding._own)
                from tmgcn_amd.csr import BatchedCSR
                A_csr = BatchedCSR.from_coo_list(v["A"], N=v["N"], device="cpu")
                assert A_csr.T < i["T"]
                with torch.no_grad():
                    ov2 = m(A_csr, v["X"], v["edges"])
                assert torch.equal(ov2, ov), "BatchedCSR window differs from the list window"
        dist.barrier()
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _spawn(world, name, base=None):
    from _util import free_port
    port = free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, name, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}:\n{ret.get(r)}"


@pytest.mark.parametrize("name", sorted(CASES))
def test_sharded_models_reproduce_the_reference_fixtures_world2(name):
    _spawn(2, name, 30100)


@pytest.mark.parametrize("name", ["g3_gcn2_three_relu_condensed1", "g4_kwgcn_2layer_selu"])
def test_sharded_models_uneven_shards_world3(name):
    """T = 10 (G3) / 9 (G4) slices over 3 ranks: shards of 4+3+3 slices, padded collectives; the G4
    validation window (4 slices) leaves the last rank without any slice of it."""
    _spawn(3, name, 30400)


def _too_few_slices_worker(rank, world, port, ret):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from tmgcn_amd import ops, synth
        from _oracle_kernels import OracleKernels
        import tmgcn_amd.layers as ehf
        ops.kernels = OracleKernels()
        g = synth.dynamic_graph(2, 12, edges_per_slice=20, seed=0, no_diag=1, F0=2)
        try:
            ehf.EmbeddingGCN(g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.edges), torch.from_numpy(g.M),
                             hidden_feat=[3, 2], condensed_W=True, use_Minv=False, device="cpu", group=dist.group.WORLD)
            ret[rank] = "no error"
        except RuntimeError as e:
            ret[rank] = "raised" if "cannot be sharded" in str(e) else f"other: {e}"
    except Exception as e:
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def test_fewer_slices_than_ranks_raises_on_every_rank():
    """T = 2 over 3 ranks: EVERY rank raises (the verdict is a function of T and G only), none is
    left waiting in a collective for a rank that has already failed."""
    from _util import free_port
    ret = mp.Manager().dict()
    mp.spawn(_too_few_slices_worker, args=(3, free_port(), ret), nprocs=3, join=True)
    assert [ret.get(r) for r in range(3)] == ["raised"] * 3, dict(ret)

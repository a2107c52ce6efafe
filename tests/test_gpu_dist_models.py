"""GPU, world size 2 and 3 on ONE device (gloo transport for the CUDA tensors — RCCL refuses two ranks
on one GPU): the slice-sharded drop-in models (group=…) with the REAL HIP kernels reproduce the
reference's fixtures G2 / G3 / G4 — logits on every rank, all parameter gradients after the
all-reduce, the validation-style call — and a whole SGD trajectory (G6) stays on the reference's
loss curve with the parameters replicated bit-identically across the ranks.
The CPU counterpart with the oracle kernels is tests/test_dist_models_gloo.py."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
pytestmark = pytest.mark.gpu

CASES = {
    "g3_gcn2_twice_selu_condensed1": ("gcn2", dict(condensed_W=True, use_Minv=False, apply_M_twice=True, nonlin2="selu")),
    "g3_gcn2_twice_selu_condensed0": ("gcn2", dict(condensed_W=False, use_Minv=False, apply_M_twice=True, nonlin2="selu")),
    "g3_gcn2_three_relu_condensed1": ("gcn2", dict(condensed_W=True, use_Minv=False, apply_M_twice=True,
                                                    apply_M_three_times=True, nonlin2="relu")),
    "g3_gcn2_default_leaky_condensed1": ("gcn2", dict(condensed_W=True, use_Minv=False, nonlin2="leaky")),
    "g2_gcn_minv_fp32": ("gcn", dict(condensed_W=True, use_Minv=True)),
    "g4_kwgcn_2layer_selu": ("kw", dict(nonlin2="selu")),
}


def _build(ehf, kind, kw, i, group):
    if kind == "gcn":
        return ehf.EmbeddingGCN(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 2], group=group, **kw)
    if kind == "gcn2":
        return ehf.EmbeddingGCN2(i["At"], i["X"], i["edges"], i["M"], hidden_feat=[6, 6, 2], group=group, **kw)
    return ehf.EmbeddingKWGCN(i["A"], i["X"], i["edges"], hidden_feat=[6, 5, 2], group=group, **kw)


def _worker(rank, world, port, name, ret, backend="gloo"):
    try:
        for p in (ROOT, HERE):
            if p not in sys.path:
                sys.path.insert(0, p)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
        if backend == "nccl":               # one GPU per rank, RCCL over xGMI (tests/test_gpu_multi.py)
            torch.cuda.set_device(rank)
            os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        else:                               # the ranks share cuda:0, gloo carries the device tensors
            torch.cuda.set_device(0)
            dist.init_process_group("gloo", rank=rank, world_size=world)
        from _util import coo_list, golden
        import tmgcn_amd.layers as ehf
        from tmgcn_amd import ops
        assert ops.kernels.name == "hip"

        def inputs(d, prefix=""):
            X = torch.from_numpy(d[prefix + "X"])
            T, N = X.shape[0], X.shape[1]
            return dict(T=T, N=N, X=X, M=torch.from_numpy(d[prefix + "M"]), edges=torch.from_numpy(d[prefix + "edges"]),
                        labels=torch.from_numpy(d[prefix + "labels"]), At=coo_list(d, "At", T, N, prefix=prefix),
                        A=coo_list(d, "A", T, N, prefix=prefix) if prefix + "A_k" in d else None)

        def close(a, b, what, tol=1e-5):
            a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
            err = float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))
            if rank == 0:
                from _util import record_tolerance
                record_tolerance(f"sharded model {name} world={world} {what}", err, tol)
            assert err <= tol, f"{what}: {err:.2e}"

        group = dist.group.WORLD
        crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1], device="cuda"))
        if name.startswith("g6_"):
            d = golden(name)
            kind = "gcn" if name.endswith("_gcn") else "gcn2"
            kw = dict(condensed_W=True, use_Minv=False) if kind == "gcn" else \
                dict(condensed_W=True, use_Minv=False, nonlin2="selu")          # as tests/golden/make_golden.py:g6
            i = inputs(d)
            torch.manual_seed(int(d["seed"]))
            m = _build(ehf, kind, kw, i, group)
            opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
            tgt = i["labels"].cuda()
            losses = []
            for _ in range(len(d["losses"])):
                opt.zero_grad()
                loss = crit(m(), tgt)
                loss.backward()
                opt.step()
                losses.append(float(loss.detach()))
            close(torch.tensor(losses), d["losses"], "SGD loss trajectory", 1e-5)
            for n, q in m.named_parameters():
                close(q.detach(), d[n + "_final"], "final " + n, 1e-5)
            for n, q in m.named_parameters():            # replicas stay bit-identical: same all-reduced gradients
                mine = q.detach().cpu().contiguous()
                others = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(others, mine)
                assert all(torch.equal(o, mine) for o in others), n
        else:
            d = golden(name)
            kind, kw = CASES[name]
            i = inputs(d)
            torch.manual_seed(int(d["seed"]))
            m = _build(ehf, kind, kw, i, group)
            sh = m._shard
            assert sh.G == world
            held = m.AtXt if kind != "kw" else m.AX
            assert held.is_cuda and held.shape[0] == sh.Tl
            out = m()
            close(out.detach(), d["logits"], "logits")
            loss = crit(out, i["labels"].cuda())
            close(loss.detach(), float(d["loss"]), "loss")
            m.zero_grad()
            loss.backward()
            for n, q in m.named_parameters():
                close(q.grad, d["d" + n], "d" + n)
            if "logits_val" in d.files:
                v = inputs(d, "val_")
                with torch.no_grad():
                    ov = m(v["A"] if kind == "kw" else v["At"], v["X"], v["edges"])
                close(ov, d["logits_val"], "validation logits")
        torch.cuda.synchronize()
        dist.barrier()
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


def _spawn(world, name, base):
    from _util import free_port
    port = free_port()
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, port, name, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}:\n{ret.get(r)}"


@pytest.mark.parametrize("name", sorted(CASES))
def test_sharded_models_two_ranks_one_gpu(name):
    _spawn(2, name, 30700)


@pytest.mark.parametrize("name", ["g3_gcn2_three_relu_condensed1", "g4_kwgcn_2layer_selu"])
def test_sharded_models_three_ranks_uneven_shards(name):
    _spawn(3, name, 31000)


@pytest.mark.parametrize("name", ["g6_sgd_gcn", "g6_sgd_gcn2"])
def test_sharded_sgd_trajectory_two_ranks(name):
    _spawn(2, name, 31300)

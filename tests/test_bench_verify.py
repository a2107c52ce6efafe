"""CPU (gloo, world 1 / 2 / 4): bench.py's verify leg (tools/bench_verify.py) on bench.py's own problem
builder at a toy size, with the oracle standing in for the device kernels.  What is under test is the
CHECKER: that it accepts a correct sharded step in both exchange modes — regenerating other ranks'
inputs from their seeds —, and that it rejects a wrong Y, dX or dW on EVERY rank (the verdict is
collective), so that a multi-GPU bench line carries a verified result or a non-zero exit."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def _worker(rank, world, port, exchange, ret):
    try:
        for p in (ROOT, HERE, os.path.join(ROOT, "tools")):
            if p not in sys.path:
                sys.path.insert(0, p)
        if world > 1:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
            dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.set_num_threads(2)
        import bench
        from bench_verify import verify_layer
        from tmgcn_amd import ops
        from _oracle_kernels import OracleKernels
        ops.kernels = OracleKernels()
        args = bench.parse(["--gpus", str(world), "--nodes", "24", "--slices-per-gpu", "3", "--deg", "4", "--feat", "8",
                            "--band", "4", "--gather-chunk-nodes", "10"])
        dev = torch.device("cpu")
        pb = bench.build_problem(args, dev, rank, world, exchange, args.nodes)
        layer, X, W, dY = pb["layer"], pb["X"], pb["W"], pb["dY"]
        assert layer.collective == (world > 1)
        Y = layer(X, W)
        Y.backward(dY)

        def run(Y_=None, dX_=None, dW_=None):
            return verify_layer(dist=dist, rank=rank, world=world, dev=dev, node_sharded_input=pb["node_sharded"],
                                A=pb["A"], M64=pb["M"], T=pb["T"], k0=pb["k0"], N=args.nodes, W=W, X=X, dY=dY,
                                Y=Y.detach() if Y_ is None else Y_, dX=X.grad if dX_ is None else dX_,
                                dW=W.grad if dW_ is None else dW_, x_slice=pb["x_slice"], dy_slice=pb["dy_slice"],
                                a_slice=pb["a_slice"], rows=16)

        v = run()
        assert v["ok"], v
        assert v["max_rel_err_Y"] <= 1e-5 and v["max_rel_err_dX"] <= 1e-5 and v["max_rel_err_dW"] <= 1e-5, v
        assert v["identity_YdY_vs_WdW"] <= 1e-5 and v["identity_YdY_vs_XdX"] <= 1e-5, v
        assert v["slices_checked"] == pb["T"] and v["rows_Y_per_slice"] == 16
        # a defect on ONE rank (the last) must fail the verdict on EVERY rank
        last = rank == world - 1
        Yb = Y.detach().clone()
        if last:
            Yb[1] *= 1.001                                   # one slice off by 1e-3
        vb = run(Y_=Yb)
        assert not vb["ok"] and vb["max_rel_err_Y"] > 1e-5, vb
        dXb = X.grad.clone()
        if last:
            dXb[-1] += 1e-3 * dXb.abs().max()                # the last slice of this rank's dX rows
        vb = run(dX_=dXb)
        assert not vb["ok"] and vb["max_rel_err_dX"] > 1e-5, vb
        vb = run(dW_=W.grad * (1.0 + 1e-3))                   # replicated tensor: wrong everywhere
        assert not vb["ok"] and vb["max_rel_err_dW"] > 1e-5 and vb["identity_YdY_vs_WdW"] > 1e-5, vb
        if world > 1:
            dist.barrier()
        ret[rank] = "ok"
    except Exception as e:
        import traceback
        ret[rank] = "".join(traceback.format_exception(type(e), e, e.__traceback__))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()


@pytest.mark.parametrize("world,exchange", [(1, "a2a"), (2, "a2a"), (2, "allgather"), (4, "a2a"), (4, "allgather")])
def test_verify_leg_accepts_correct_and_rejects_wrong_results(world, exchange):
    from _util import free_port
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, free_port(), exchange, ret), nprocs=world, join=True)
    for r in range(world):
        assert ret.get(r) == "ok", f"rank {r}:\n{ret.get(r)}"

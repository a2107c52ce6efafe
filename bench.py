#!/usr/bin/env python3
"""bench.py — TM-GCN layer forward+backward throughput, (edges×T)/s  (BASELINE.json metric).

A "step" is one pass of the hot path over one batch of synthetic input:
    forward   Y  = ((Â ⋆ (M ×₁ X)) · W)            P1 -> [exchange] -> P2 -> P3
    backward  dX, dW from a given dY               P3ᵀ (dA, dW) -> P2ᵀ -> [exchange] -> P1ᵀ
Workload (SURVEY §8d "S4", the config the metric is quoted on): per GPU 16 frontal slices of an
N = 2,000,000-node graph, 32 random out-neighbours per row + self loop (66 M stored non-zeros
per slice), F = 128 -> 128 features, band M with 20 diagonals, fp32.  Weak scaling: T = 16·G.
Inputs are generated on the device (seeded by global slice index) and are resident in HBM
before the timed region.  One unit of work = one stored non-zero of one slice ("edge-slice").

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        --master-port 29500 bench.py --gpus 8 --steps 10 --warmup 3

Rank 0 prints ONE JSON line (driver contract) with `roofline` (dominant kernel = forward
SpMM, HIP events on the launch stream) and `cpu_baseline` (the oracle executed the
reference's way on the host cores, on a bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--warmup", type=int, default=3)
    p.add_argument("--nodes", type=int, default=2_000_000)
    p.add_argument("--slices-per-gpu", type=int, default=16)
    p.add_argument("--deg", type=int, default=32)
    p.add_argument("--feat", type=int, default=128)
    p.add_argument("--band", type=int, default=20)
    p.add_argument("--exchange", choices=["a2a", "allgather"], default="a2a")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-fuse", action="store_true", help="run P2 and P3 as separate kernels")
    p.add_argument("--no-pipeline", action="store_true", help="exchange all slices before computing")
    p.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for "
                   "the single-device diagnostic below)")
    p.add_argument("--single-device", action="store_true",
                   help="diagnostic: all ranks share cuda:0 (needs --backend gloo; RCCL refuses it)")
    p.add_argument("--grid-reserve", type=int, default=None,
                   help="block slots the fused kernel leaves free for RCCL's kernels (default: the layer's choice)")
    p.add_argument("--force-collectives", action="store_true",
                   help="run the exchange (RCCL) code path even at --gpus 1 (diagnostic)")
    p.add_argument("--cpu-nodes", type=int, default=250_000, help="N of the CPU-baseline sample")
    p.add_argument("--watchdog", type=int, default=600,
                   help="diagnostic: dump every thread's Python traceback to stderr once, after this many seconds "
                        "(a default run takes 2-3 minutes; 0 disables)")
    return p.parse_args()


def cpu_baseline(args):
    """The oracle's layer fwd+bwd (list of COO fp64, one sparse.mm per slice, autograd) on the
    host cores, on a bounded sample: 2 slices of the same degree/F at N = cpu-nodes.  Timed at
    two thread counts (all cores, and 32: torch's sparse kernels do not scale to hundreds of
    threads) and the faster one is reported with the threads it used."""
    from oracle import tmgcn_oracle as orc
    from tmgcn_amd import synth

    ncpu = os.cpu_count() or 1
    Tc, Nc, F = 2, min(args.nodes, args.cpu_nodes), args.feat
    A = synth.device_er_csr(Tc, Nc, args.deg, "cpu")
    At = A.to_coo_list(torch.float64)
    X = synth.device_features(Tc, Nc, F, "cpu").double()
    M = torch.from_numpy(synth.band_M(Tc, args.band, "matlab"))
    g = torch.Generator().manual_seed(1)
    W = torch.randn(F, F, generator=g) * 0.1
    dY = torch.randn(Tc, Nc, F, generator=g)
    best = None
    for threads in sorted({ncpu, min(32, ncpu)}):
        torch.set_num_threads(threads)
        orc.layer_fwd_bwd(M, At, X, W, dY)  # warm-up (allocator, thread pool)
        reps, t0 = 0, time.perf_counter()
        while True:
            orc.layer_fwd_bwd(M, At, X, W, dY)
            reps += 1
            el = time.perf_counter() - t0
            if el > 6.0 or reps >= 3:
                break
        rate = A.nnz * reps / el
        if best is None or rate > best[0]:
            best = (rate, threads, reps, el)
    rate, threads, reps, el = best
    try:
        model = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
    except Exception:
        model = "unknown"
    return {"value": rate, "unit": "edge-slices/s", "cores": threads, "kind": "port",
            "sample": f"{reps} x fwd+bwd of {Tc} slices, N={Nc}, deg={args.deg}+1, F={F}->{F} "
                      f"(oracle = reference's way on torch CPU, best of {{{ncpu}, {min(32, ncpu)}}} threads, {model}); "
                      f"{el / reps:.2f} s each"}


def main():
    args = parse()
    if args.watchdog > 0:
        import faulthandler
        faulthandler.dump_traceback_later(args.watchdog, exit=False)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    assert torch.cuda.is_available(), "bench.py needs MI355X GPUs"
    if args.single_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    import torch.distributed as dist
    if world > 1 or args.force_collectives:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    from tmgcn_amd import _lib, ops, synth
    from tmgcn_amd.dist import ShardedTMGCNLayer
    _lib.load()  # fail loudly if the HIP library is missing

    G, Tl, N, F = world, args.slices_per_gpu, args.nodes, args.feat
    T = Tl * G
    k0 = rank * Tl
    A = synth.device_er_csr(Tl, N, args.deg, dev, first_slice=k0)
    A.transpose()  # backward operand, built once (plan time, not timed)
    M = synth.band_M(T, args.band, "matlab")
    layer = ShardedTMGCNLayer(A, M, T, group=None, exchange=args.exchange, fuse=False if args.no_fuse else None,
                              pipeline=not args.no_pipeline, force_collectives=args.force_collectives,
                              grid_reserve=args.grid_reserve)
    shape = layer.input_shape(F)
    if layer.collective and args.exchange == "a2a":
        # node shard of the synthetic features: slice k seeded by k, columns of this rank's nodes
        X = synth.device_features(T, shape[1], F, dev, first_slice=1000 * rank)
    else:
        X = synth.device_features(shape[0], N, F, dev, first_slice=k0)
    X.requires_grad_(True)
    g = torch.Generator(device=dev).manual_seed(1234)
    W = (torch.randn(F, F, device=dev, generator=g) * 0.1).requires_grad_(True)  # same on every rank
    g.manual_seed(99 + rank)
    dY = torch.randn(Tl, N, F, device=dev, generator=g)

    def step():
        X.grad = None
        W.grad = None
        Y = layer(X, W)
        Y.backward(dY)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ops.kernels.timer = ops.KernelTimer()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    kt = ops.kernels.timer.summary()
    ops.kernels.timer = None
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        n = torch.tensor([A.nnz], device=dev, dtype=torch.int64)
        dist.all_reduce(n)
        total_nnz = int(n.item())
    else:
        total_nnz = A.nnz

    out = None
    if rank == 0:
        # roofline of the dominant kernel (forward SpMM): SURVEY §8d no-reuse gather model,
        # bytes per edge-slice = 8 (col+val) + F*4 (gathered row) + (4 + F*4)/d (rowptr + output row)
        d = A.nnz / A.n_rows
        bytes_per_unit = 8 + F * 4 + (4 + F * 4) / d
        # dominant kernel: the forward SpMM — fused with the GEMM epilogue when the widths allow
        # (then it also writes Y; only P2's own bytes are counted, conservatively)
        dom = "spmm_gemm" if "spmm_gemm" in kt else "spmm"
        sp = kt[dom]
        # one launch per step at N = 1; the pipelined multi-GPU path launches slice by slice
        units_per_launch = A.nnz * args.steps / sp["launches"]
        achieved = bytes_per_unit * units_per_launch / (sp["avg_ms"] * 1e-3) / 1e9
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                rec = json.load(open(pmc))
                if rec.get("nodes") == N and rec.get("feat") == F and rec.get("slices_per_gpu") == Tl:
                    traffic = rec["spmm_hbm_bytes_per_launch"]
            except Exception:
                traffic = None
        out = {
            "metric": "TM-GCN layer fwd+bwd throughput (edges x T)/s",
            "value": total_nnz * args.steps / elapsed,
            "unit": "edge-slices/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"S4 TM-GCN layer fwd+bwd: {Tl} slices/GPU (T={T}), N={N}, "
                                   f"deg={args.deg}+self, F={F}->{F}, band-M b={args.band}, fp32",
                       "exchange": args.exchange if layer.collective else "none", "grid_reserve": layer.grid_reserve,
                       "edge_slices_per_step": total_nnz},
            "roofline": {"kernel": "spmm_gemm_kernel (forward P2 + fused P3)" if dom == "spmm_gemm" else "spmm_vec4_kernel (forward P2)", "bound": "hbm", "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic if sp["launches"] == args.steps else None,
                         "bytes_per_edge_slice": bytes_per_unit, "edge_slices_per_launch": units_per_launch,
                         "avg_launch_ms": sp["avg_ms"]},
            "kernels_ms": {k: round(v["avg_ms"], 4) for k, v in kt.items()},
            "peak_hbm_gb_rank0": round(torch.cuda.max_memory_allocated(dev) / 1e9, 1),
        }
        if "gemm_dW" in kt:
            # the one GEMM that still runs as its own kernel (dW = AXᵀ·dY); the forward / dA GEMMs run
            # inside the fused SpMM kernels, hidden under the gather.  It multiplies on the bf16 matrix
            # cores after an exact 3-way split of the fp32 operands: 6 bf16 MFMA products per fp32 term.
            t_dw = kt["gemm_dW"]["avg_ms"] * 1e-3
            fp32_tf = 2.0 * A.n_rows * F * F / t_dw / 1e12
            out["mfma"] = {"kernel": "gemm_dw_bf16x3_kernel (dW)", "bound": "mfma", "achieved": 6.0 * fp32_tf,
                           "peak": 2500.0, "unit": "TFLOP/s", "frac": 6.0 * fp32_tf / 2500.0,
                           "dtype": "bf16 planes of an exact 3-way fp32 split (v_mfma_f32_32x32x16_bf16), fp32 accumulate",
                           "fp32_equivalent_tflops": fp32_tf, "f32_mfma_peak_tflops": 157.3,
                           "hbm_gbs": A.n_rows * 2 * F * 4 / t_dw / 1e9}
        if not args.no_cpu_baseline and world == 1:  # the CPU leg is reported at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args)
    # RCCL prints its version banner through C stdio, which is flushed only at exit when stdout is
    # a pipe/file: every rank flushes it BEFORE the last barrier so that rank 0's JSON line is the
    # last line of the job's stdout.
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass
    sys.stdout.flush()
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

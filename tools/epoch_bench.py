#!/usr/bin/env python3
"""Training-epoch time of the reference-shaped configs S1-S3 on the MI355X module vs
the CPU oracle executed the reference's way, same box, same run.

    python tools/epoch_bench.py [S1 S2 S3] [--epochs 50] [--cpu-epochs 3]

An "epoch" is what the reference scripts do per iteration (experiment_bitcoin_our.py:118-123,
experiment_reddit_our_link_prediction.py:75-81): zero_grad, gcn(), weighted CE, backward, SGD step.
Also checks that both sides agree on the first-epoch loss (parity at the BASELINE sizes)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tmgcn_amd.layers as ehf  # noqa: E402
from tmgcn_amd import synth  # noqa: E402
from oracle import tmgcn_oracle as orc  # noqa: E402  (cpu baseline leg only)

MODELS = {  # the scripts' model per config
    "S1": dict(kind="gcn2", hidden=[6, 6, 2], nonlin="selu"),   # experiment_bitcoin_our.py:107 (2-layer)
    "S2": dict(kind="gcn", hidden=[6, 2]),                      # experiment_reddit_our_link_prediction.py:65
    "S3": dict(kind="gcn2", hidden=[6, 6, 2], nonlin="selu", param_dtype=torch.bfloat16),  # AMLSim, bf16 weights
    "P128": dict(kind="gcn2", hidden=[128, 128, 2], nonlin="relu", scale=0.05),  # BASELINE.md §2 probe, wide features
}


def gpu_epochs(g, spec, epochs, graph, fused_loss=False, script_mode=False):
    global ehf
    At, X, M = g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.M)
    edges, labels = torch.from_numpy(g.edges), torch.from_numpy(g.labels).cuda()
    if script_mode:  # an untouched reference script: tmgcn_amd.ehf, targets / class weights / criterion as the script has them
        import tmgcn_amd.ehf as ehf
        labels, graph = labels.cpu(), False
    torch.manual_seed(0)
    kw = dict(condensed_W=True, use_Minv=False, param_dtype=spec.get("param_dtype", torch.float32))
    if spec["kind"] == "gcn":
        m = ehf.EmbeddingGCN(At, X, edges, M, hidden_feat=spec["hidden"], **kw)
    else:
        m = ehf.EmbeddingGCN2(At, X, edges, M, hidden_feat=spec["hidden"], nonlin2=spec["nonlin"], **kw)
    if "scale" in spec:  # N(0,1) weights at width 128 overflow the activations; both sides scale the same way
        with torch.no_grad():
            for q in m.parameters():
                q.mul_(spec["scale"])
    opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
    if fused_loss:
        from tmgcn_amd.losses import WeightedCrossEntropy
        crit = WeightedCrossEntropy(torch.tensor([0.9, 0.1])).cuda()
    elif script_mode:
        crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))
    else:
        crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1], device="cuda"))

    def epoch():
        opt.zero_grad(set_to_none=True)
        loss = crit(m(), labels)
        loss.backward()
        opt.step()
        return loss

    first = float(epoch().detach())
    for _ in range(3):
        epoch()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(epochs):
        epoch()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / epochs
    graphed = None
    if graph:
        # capture one whole epoch (forward, loss, backward, SGD) into a hipGraph: the real configs
        # are launch-bound (tens of µs of kernels), so replay removes the Python/launch overhead
        opt.zero_grad(set_to_none=False)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(3):
                opt.zero_grad(set_to_none=False)
                crit(m(), labels).backward()
                opt.step()
        torch.cuda.current_stream().wait_stream(s)
        gr = torch.cuda.CUDAGraph()
        opt.zero_grad(set_to_none=False)
        with torch.cuda.graph(gr):
            loss = crit(m(), labels)
            loss.backward()
            opt.step()
        for _ in range(3):
            gr.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(epochs):
            gr.replay()
        torch.cuda.synchronize()
        graphed = (time.perf_counter() - t0) / epochs
    return first, eager, graphed


def cpu_epochs(g, spec, epochs, threads):
    torch.set_num_threads(threads)
    At, X, M = g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.M)
    edges, labels = torch.from_numpy(g.edges), torch.from_numpy(g.labels)
    torch.manual_seed(0)
    F = [X.shape[-1]] + spec["hidden"]
    p = {k: torch.nn.Parameter(v * spec.get("scale", 1.0)) for k, v in orc.draw_params(spec["kind"], g.T, F).items()}
    AtXt = orc.compute_AtXt(M, At, X)  # cached at construction, as the reference does (ehf:195)
    src, dst = orc.flat_edge_index(edges, g.N)
    opt = torch.optim.SGD(list(p.values()), lr=0.01, momentum=0.9)
    crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))

    def epoch():
        opt.zero_grad()
        if spec["kind"] == "gcn":
            out = orc.gcn_forward(AtXt, p["W"], p["U"], src, dst)
        else:
            out = orc.gcn2_forward(AtXt, At, M, p["W1"], p["W2"], p["U"], src, dst, nonlin=spec["nonlin"])
        loss = crit(out, labels)
        loss.backward()
        opt.step()
        return float(loss)

    first = epoch()
    t0 = time.perf_counter()
    for _ in range(epochs):
        epoch()
    return first, (time.perf_counter() - t0) / epochs


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["S1", "S2", "S3"])
    ap.add_argument("--epochs", type=int, default=50)
    ap.add_argument("--cpu-epochs", type=int, default=3)
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--fused-loss", action="store_true", help="tmgcn_amd.WeightedCrossEntropy instead of nn.CrossEntropyLoss")
    ap.add_argument("--script-mode", action="store_true",
                    help="what an untouched reference script does: import tmgcn_amd.ehf, host-side targets and criterion")
    args = ap.parse_args()
    for name in args.configs:
        g = synth.dynamic_graph(**synth.CONFIGS[name], seed=0)
        spec = MODELS[name]
        l_gpu, t_eager, t_graph = gpu_epochs(g, spec, args.epochs, not args.no_graph, args.fused_loss, args.script_mode)
        cpu = {}
        for th in ((8, 32) if args.cpu_epochs > 0 else ()):  # all 256 threads is pathological on these small ops (measured: 1000x slower)
            l_cpu, cpu[th] = cpu_epochs(g, spec, args.cpu_epochs, th)
        th_best = min(cpu, key=cpu.get) if cpu else 0
        t_cpu = cpu[th_best] if cpu else float("nan")
        l_cpu = l_cpu if cpu else float("nan")
        best = min(t for t in (t_eager, t_graph) if t)
        print(json.dumps({"config": name, "model": spec["kind"], "T": g.T, "N": g.N, "E": int(g.edges.shape[1]),
                          "nnz_At": int(sum(c.nnz for c in g.Ct)),
                          "gpu_epoch_ms_eager": round(t_eager * 1e3, 3),
                          "gpu_epoch_ms_hipgraph": round(t_graph * 1e3, 3) if t_graph else None,
                          "cpu_epoch_ms": round(t_cpu * 1e3, 1), "cpu_threads": th_best,
                          "cpu_epoch_ms_by_threads": {str(k): round(v * 1e3, 1) for k, v in cpu.items()},
                          "speedup": round(t_cpu / best, 1),
                          "fused_loss": args.fused_loss, "script_mode": args.script_mode, "first_loss_gpu": l_gpu, "first_loss_cpu": l_cpu}), flush=True)

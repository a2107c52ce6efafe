#!/usr/bin/env python3
"""Which statements of an untouched script's epoch the host time of "script mode" belongs to.

The epoch of experiment_reddit_our_link_prediction.py:75-81 — optimizer.zero_grad(); output = gcn(); loss = criterion(output,
target); loss.backward(); optimizer.step() — is host-bound outside a hipGraph (kernels: 0.03-0.08 ms, epoch: 0.15-0.3 ms).
This probe times each statement on the host clock (no device sync inside the epoch; the loop is host-bound, so a
statement's host time is its share of the epoch), for the library's modules under `import tmgcn_amd.ehf as ehf` with the
script's own torch.optim.SGD and nn.CrossEntropyLoss on host-side targets — and, beside it, the same loop with the model
replaced by ONE trivial differentiable torch op on the same parameters (what zero_grad / backward / step cost in torch alone).
    python3 tools/script_floor_probe.py [S1 S2 S3] [--epochs 2000]"""
import argparse
import json
import os
import sys
import time

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)


def timed_loop(stmts, epochs):
    """stmts: list of (name, callable(state) -> None); returns ms per epoch per statement and in total (median of 5 passes)."""
    names = [n for n, _ in stmts]
    passes = []
    for _ in range(5):
        acc = dict.fromkeys(names, 0.0)
        st = {}
        torch.cuda.synchronize()
        t_all = time.perf_counter()
        for _ in range(epochs):
            for n, f in stmts:
                t = time.perf_counter()
                f(st)
                acc[n] += time.perf_counter() - t
        torch.cuda.synchronize()
        acc["epoch"] = time.perf_counter() - t_all
        passes.append(acc)
    passes.sort(key=lambda a: a["epoch"])
    return {k: round(v / epochs * 1e3, 4) for k, v in passes[2].items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["S2"])
    ap.add_argument("--epochs", type=int, default=2000)
    a = ap.parse_args()
    import bench
    import tmgcn_amd.ehf as ehf
    from tmgcn_amd import synth
    out = {}
    for name in a.configs:
        g = synth.dynamic_graph(**synth.CONFIGS[name], seed=0)
        spec = bench.EPOCH_MODELS[name]
        At, X, M = g.At_list(), torch.from_numpy(g.X), torch.from_numpy(g.M)
        edges, labels = torch.from_numpy(g.edges), torch.from_numpy(g.labels)        # host-side, as in the scripts
        torch.manual_seed(0)
        if spec["kind"] == "gcn":
            m = ehf.EmbeddingGCN(At, X, edges, M, hidden_feat=spec["hidden"], condensed_W=True, use_Minv=False)
        else:
            m = ehf.EmbeddingGCN2(At, X, edges, M, hidden_feat=spec["hidden"], nonlin2=spec["nonlin"], condensed_W=True, use_Minv=False)
        opt = torch.optim.SGD(m.parameters(), lr=0.01, momentum=0.9)
        crit = torch.nn.CrossEntropyLoss(weight=torch.tensor([0.9, 0.1]))

        def s_zero(st): opt.zero_grad()
        def s_fwd(st): st["out"] = m()
        def s_crit(st): st["loss"] = crit(st["out"], labels)
        def s_bwd(st): st["loss"].backward()
        def s_step(st): opt.step()
        script = [("zero_grad", s_zero), ("gcn()", s_fwd), ("criterion", s_crit), ("backward", s_bwd), ("step", s_step)]
        for _ in range(20):
            for _, f in script:
                f(globals().setdefault("_st", {}))
        rec = {"script": timed_loop(script, a.epochs)}
        # torch alone: the same parameters, the same optimizer, a one-op "model"
        params = list(m.parameters())

        def t_fwd(st): st["loss"] = sum((p.sum() for p in params[1:]), params[0].sum())
        torch_only = [("zero_grad", s_zero), ("model+loss (one sum per parameter)", t_fwd), ("backward", s_bwd), ("step", s_step)]
        for _ in range(20):
            for _, f in torch_only:
                f(globals().setdefault("_st2", {}))
        rec["torch_alone"] = timed_loop(torch_only, a.epochs)
        rec["parameters"] = len(params)
        out[name] = rec
        print(name, json.dumps(rec))
    os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
    with open(os.path.join(root, "gpurun_out", "script_floor_probe.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()

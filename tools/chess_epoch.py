#!/usr/bin/env python3
"""One training epoch on the only real data set the reference ships — chess, 7 301 players, 80 training slices, the 2-layer
model of experiment_chess_our.py (2 -> 6 -> 6 -> 3 classes) — from the raw edges of fixture G10 (tests/golden):
adjacency pipeline on the device, then eager / captured / fused-captured epochs and, with --cpu, the CPU oracle's epoch.
    python tools/chess_epoch.py [--epochs 200] [--cpu 3]
Under rocprofv3 --kernel-trace --stats it gives the kernel sequence of a real-data step (tools/epoch_sequence.py)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
from _g10 import G10  # noqa: E402
import tmgcn_amd.layers as ehf  # noqa: E402
from tmgcn_amd import adjacency  # noqa: E402
from tmgcn_amd.graphs import GraphedTrainStep  # noqa: E402
from tmgcn_amd.optim import FusedSGD  # noqa: E402


def run(epochs=200, cpu=0, cpu_threads=(8, 32), only=None):
    """The record (dict) of one session: see the module docstring.  Also what bench.py's `epochs.chess` holds."""
    a = argparse.Namespace(epochs=epochs, cpu=cpu, cpu_threads=list(cpu_threads), only=only)
    g = G10()
    k, i, j = g.raw
    t0 = time.perf_counter()
    Chat, _ = adjacency.build_adjacency(k, i, j, np.ones(len(k), np.float32), g.TT, g.N, M=None, window=10)
    A = adjacency.m_product_csr(Chat.slices(0, g.T), g.M)
    torch.cuda.synchronize()
    rec = {"T": g.T, "N": g.N, "edges": int(g.edges_train.shape[1]), "nnz_At": int(A.nnz),
           "nnz_per_row": round(A.nnz / (g.T * g.N), 2), "adjacency_pipeline_s": round(time.perf_counter() - t0, 3)}
    tgt = torch.from_numpy(g.target_train).cuda()
    crit = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g.class_weights).cuda())

    def make(opt_cls):
        torch.manual_seed(0)
        m = ehf.EmbeddingGCN2(A, torch.from_numpy(g.X_train), torch.from_numpy(g.edges_train), torch.from_numpy(g.M),
                              hidden_feat=[6, 6, 3], condensed_W=True, use_Minv=False, nonlin2="selu")
        return m, opt_cls(m.parameters(), lr=0.01, momentum=0.9)

    def timed(run, n, per=1):
        for _ in range(5):
            run()
        best = []
        for _ in range(3):
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(n // per):
                run()
            torch.cuda.synchronize()
            best.append((time.perf_counter() - t) / (n // per * per))
        return round(sorted(best)[1] * 1e3, 4)

    m, o = make(torch.optim.SGD)

    def eager():
        o.zero_grad(set_to_none=True)
        crit(m(), tgt).backward()
        o.step()
    rec["gpu_ms_eager"] = timed(eager, a.epochs)
    m, o = make(FusedSGD)
    rec["gpu_ms_graph_fused"] = timed(GraphedTrainStep(m, crit, o, tgt), a.epochs)
    m, o = make(FusedSGD)
    rec["gpu_ms_graph_fused8"] = timed(GraphedTrainStep(m, crit, o, tgt, steps_per_replay=8), a.epochs, 8)
    if a.only == "graph_fused":                      # leave single captured steps at the end of the trace
        m, o = make(FusedSGD)
        timed(GraphedTrainStep(m, crit, o, tgt), 20)
    if a.cpu:
        from oracle import tmgcn_oracle as orc
        At = [c.coalesce() for c in A.to_coo_list()]
        X, M = torch.from_numpy(g.X_train), torch.from_numpy(g.M)
        torch.manual_seed(0)
        p = {kk: torch.nn.Parameter(v) for kk, v in orc.draw_params("gcn2", g.T, [2, 6, 6, 3]).items()}
        AtXt = orc.compute_AtXt(M, At, X)
        src, dst = orc.flat_edge_index(torch.from_numpy(g.edges_train), g.N)
        opt = torch.optim.SGD(list(p.values()), lr=0.01, momentum=0.9)
        critc = torch.nn.CrossEntropyLoss(weight=torch.from_numpy(g.class_weights))
        tc = torch.from_numpy(g.target_train)
        by = {}
        for th in a.cpu_threads:
            torch.set_num_threads(th)
            ts = []
            for _ in range(a.cpu + 1):
                t = time.perf_counter()
                opt.zero_grad()
                loss = critc(orc.gcn2_forward(AtXt, At, M, p["W1"], p["W2"], p["U"], src, dst, nonlin="selu"), tc)
                loss.backward()
                opt.step()
                ts.append(time.perf_counter() - t)
            by[th] = round(sorted(ts[1:])[len(ts[1:]) // 2] * 1e3, 1)
        rec["cpu_ms_by_threads"] = by
        rec["cpu_ms"] = min(by.values())
        rec["speedup_best_mode"] = round(rec["cpu_ms"] / min(rec["gpu_ms_graph_fused"], rec["gpu_ms_graph_fused8"]), 1)
    return rec


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--epochs", type=int, default=200)
    ap.add_argument("--cpu", type=int, default=0, help="CPU oracle epochs to time (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, nargs="*", default=[8, 32], help="torch thread counts tried for the CPU epoch")
    ap.add_argument("--only", default=None, help="run only this GPU mode last (for a kernel trace): graph_fused")
    a = ap.parse_args()
    print(json.dumps(run(a.epochs, a.cpu, a.cpu_threads, a.only)))


if __name__ == "__main__":
    main()

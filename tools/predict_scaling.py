#!/usr/bin/env python3
"""The predicted 1/2/4/8-GPU curve of DESIGN.md §6, from its stated inputs — so that a SCALE record
can be diffed against it and the assumptions (link rate, HBM interference of the exchange) can be
changed on the command line.   python tools/predict_scaling.py [--link 45 60] [--step 183 …]

Model (weak scaling, 16 slices of S4 per GPU, fp32):
  a2a        forward  = P1 + e + 15·max(e, c_f) + c_f          e = per-slice exchange time on one link
             backward = max(16·c_b + dW, c_b + 16·e) + P1ᵀ     c_f, c_b = per-slice fused-kernel time
             + pipelined-form overhead (one-slice launches, send layout)
             + the exchange's HBM traffic beside an HBM-bound kernel: 2 passes x 2 x (G-1)/G x shard bytes,
               at `--hbm-ms-per-gb` (low = queueing only, high = as measured with local RCCL copies)
  all-gather step + 2 x (shard bytes / link rate): every peer's whole shard crosses its direct link per
             pass, and only P1 hides under it (a slice's SpMM needs all nodes)"""
import argparse
import json


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--step", type=float, default=183.0, help="measured single-GPU ms/step")
    ap.add_argument("--fwd", type=float, default=86.0)
    ap.add_argument("--bwd", type=float, default=77.5)
    ap.add_argument("--p1", type=float, default=5.6)
    ap.add_argument("--dw", type=float, default=7.5)
    ap.add_argument("--shard-gb", type=float, default=16.384, help="one [16, N, F] fp32 activation shard")
    ap.add_argument("--link", type=float, nargs=2, default=[45.0, 60.0], help="sustained GB/s per xGMI link and direction (low high)")
    ap.add_argument("--pipeline-ms", type=float, default=2.3, help="one-slice launches + send layout, measured at world size 1")
    ap.add_argument("--hbm-ms-per-gb", type=float, nargs=2, default=[0.149, 0.47],
                    help="cost of the exchange's own HBM traffic (183 ms / 1230 GB ... measured with local RCCL copies)")
    ap.add_argument("--units", type=float, default=1.056e9, help="edge-slices per GPU and step")
    a = ap.parse_args()
    slices = 16
    cf, cb = a.fwd / slices, a.bwd / slices
    rows = [{"gpus": 1, "a2a_ms": [a.step, a.step], "a2a_eff": [1.0, 1.0], "allgather_ms": [a.step, a.step], "allgather_eff": [1.0, 1.0]}]
    for G in (2, 4, 8):
        a2a, ag = [], []
        for link, hbm in ((a.link[1], a.hbm_ms_per_gb[0]), (a.link[0], a.hbm_ms_per_gb[1])):   # best case, worst case
            e = a.shard_gb / (slices * G) / link * 1e3
            fwd = a.p1 + e + (slices - 1) * max(e, cf) + cf
            bwd = max(slices * cb + a.dw, cb + slices * e) + a.p1
            traffic_gb = 2 * 2 * (G - 1) / G * a.shard_gb
            a2a.append(fwd + bwd + a.pipeline_ms + hbm * traffic_gb)
            ag.append(a.step + 2 * a.shard_gb / link * 1e3)
        rows.append({"gpus": G, "a2a_ms": [round(x, 1) for x in a2a], "a2a_eff": [round(a.step / x, 3) for x in a2a],
                     "a2a_edge_slices_per_s": [round(G * a.units / (x * 1e-3)) for x in a2a],
                     "allgather_ms": [round(x, 1) for x in ag], "allgather_eff": [round(a.step / x, 3) for x in ag]})
    print(json.dumps({"inputs": vars(a), "rows": rows}, indent=1))


if __name__ == "__main__":
    main()

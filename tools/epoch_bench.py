#!/usr/bin/env python3
"""Training-epoch time of the reference-shaped configs (plus the wide-feature
probe P128) on the MI355X module vs the CPU oracle executed the reference's way — the same code
as bench.py's `epochs` block (bench.epochs_block), callable for a chosen subset:

    python tools/epoch_bench.py [S1 S2 S3 P128] [--epoch-reps 50] [--cpu-epoch-reps 5] [--modes eager graph fused graph_fused graph_fused8 script]

An "epoch" is what the reference scripts do per iteration (experiment_bitcoin_our.py:118-123,
experiment_reddit_our_link_prediction.py:75-81): zero_grad, gcn(), weighted CE, backward, SGD step."""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("configs", nargs="*", default=["S1", "S2", "S3"])
    ap.add_argument("--epoch-reps", type=int, default=50)
    ap.add_argument("--cpu-epoch-reps", type=int, default=5)
    ap.add_argument("--modes", nargs="*", default=["eager", "graph", "fused", "graph_fused", "graph_fused8", "script"])
    a = ap.parse_args()
    args = bench.parse(["--epoch-reps", str(a.epoch_reps), "--cpu-epoch-reps", str(a.cpu_epoch_reps)])
    print(json.dumps(bench.epochs_block(args, tuple(a.configs), tuple(a.modes)), indent=1), flush=True)

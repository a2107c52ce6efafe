#!/usr/bin/env python3
"""Where the HOST time of an eager / script-mode epoch goes (the tiny-F epochs are host-bound outside a hipGraph:
kernels ≈ 0.1 ms, epoch ≈ 0.2 ms).  cProfile of `--epochs` epochs of bench.py's epoch loop; prints the top
functions by own time.   python tools/host_profile_epoch.py S1 script [--epochs 300]"""
import argparse
import cProfile
import io
import os
import pstats
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("mode")
    ap.add_argument("--epochs", type=int, default=300)
    ap.add_argument("--top", type=int, default=45)
    ap.add_argument("--callers", default=None, help="also print the callers of functions matching this regex")
    a = ap.parse_args()
    import torch
    import bench
    from tmgcn_amd import synth
    g = synth.dynamic_graph(**synth.CONFIGS[a.config], seed=0)
    spec = bench.EPOCH_MODELS[a.config]
    pr = cProfile.Profile()
    orig = bench.time.perf_counter
    bench.gpu_epochs(g, spec, 20, a.mode)                       # warm: plans, allocator, lazy init
    pr.enable()
    first, med, best = bench.gpu_epochs(g, spec, a.epochs, a.mode)
    pr.disable()
    print(f"# {a.config} {a.mode}: median pass {med * 1e3:.4f} ms/epoch, fastest {best * 1e3:.4f} (under cProfile)")
    s = io.StringIO()
    st = pstats.Stats(pr, stream=s).sort_stats("tottime")
    st.print_stats(a.top)
    if a.callers:
        st.print_callers(a.callers)
    print(s.getvalue())


if __name__ == "__main__":
    main()

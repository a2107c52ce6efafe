#!/usr/bin/env python3
"""Where the time of a fused SpMM+GEMM launch goes, phase by phase.

A development build of the library (tools/ab_variants.sh trace "-DTMGCN_FUSED_TRACE") sums, in thread 0 of every block, the
100 MHz wall-clock time of each phase of the block's tiles — tile draw (atomic + barrier), row pointers, gather (wave 0's
share), barrier behind the gather, products + store issue, barrier behind the products — separately for short tiles
(entry-major walk, csrc/spmm_row.h) and the others.  This script launches the kernel on a chosen graph through that
library (loaded beside the in-tree one, as tools/ab_fused.py does) and prints the mean time per tile and phase.
    python tools/fused_trace.py [chess_tiled|er|powerlaw] [--deg 32] [--slices 16] [--nodes 2000000] [--ax]"""
import argparse
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from tmgcn_amd import synth  # noqa: E402

p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
PHASES = ("draw", "row_pointers", "gather_wave0", "barrier_after_gather", "products_and_stores", "barrier_after_products")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("graph", nargs="?", default="chess_tiled")
    ap.add_argument("--deg", type=int, default=32)
    ap.add_argument("--slices", type=int, default=16)
    ap.add_argument("--nodes", type=int, default=2_000_000)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--ax", action="store_true")
    ap.add_argument("--variant", default="trace")
    ap.add_argument("--json", default=None)
    a = ap.parse_args()
    lib = C.CDLL(f"{root}/build/variants/{a.variant}/libtmgcn_hip.so")
    lib.tmgcn_spmm_gemm_f32_hint.argtypes = [p, p, p, p, i64, i32, i32, p, i32, i32, i64, i64, i32, p, p, p, i32, C.c_float, p]
    lib.tmgcn_debug_fused_trace.argtypes = [p, C.c_long, C.c_int]
    A = synth.device_csr(a.graph, a.slices, a.nodes, a.deg, "cuda")
    N, F = A.N, a.feat
    X = torch.rand(A.T, N, F, device="cuda")
    W = torch.randn(F, F, device="cuda") * 0.1
    Y = torch.empty_like(X)
    AX = torch.empty_like(X) if a.ax else None
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None

    def run():
        rc = lib.tmgcn_spmm_gemm_f32_hint(ptr(A.rowptr), ptr(A.col), ptr(A.val), ptr(X), A.n_rows, N, F, ptr(W), F, 0, 0, 0, 0,
                                          ptr(Y), ptr(AX), None, 0, float(A.avg_nnz_per_row), st)
        assert rc == 0
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 16, np.uint64)
    lib.tmgcn_debug_fused_trace(buf.ctypes.data_as(p), buf.size, 1)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    run()
    e.record()
    torch.cuda.synchronize()
    lib.tmgcn_debug_fused_trace(buf.ctypes.data_as(p), buf.size, 0)
    w = buf.reshape(4096, 16).astype(np.float64)
    blocks = int((w[:, 6] + w[:, 14] > 0).sum())
    rec = {"graph": a.graph, "T": A.T, "N": N, "F": F, "entries_per_row": round(A.avg_nnz_per_row, 3), "with_AX": a.ax,
           "launch_ms": round(s.elapsed_time(e), 3), "blocks": blocks}
    for name, off in (("other_tiles", 0), ("short_tiles", 8)):
        n = w[:, off + 6].sum()
        if n == 0:
            continue
        per = {ph: round(float(w[:, off + i].sum() / n) * 0.01, 3) for i, ph in enumerate(PHASES)}      # 100 MHz ticks -> us
        per["sum_us"] = round(sum(per.values()), 3)
        rec[name] = {"tiles": int(n), "us_per_tile": per}
    tot = (w[:, :6].sum() + w[:, 8:14].sum()) * 0.01
    rec["mean_block_busy_ms"] = round(tot / max(1, blocks) / 1e3, 3)
    print(json.dumps(rec, indent=1))
    if a.json:
        json.dump(rec, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

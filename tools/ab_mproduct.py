#!/usr/bin/env python3
"""The adjacency M-product as a segmented merge vs expand + sort + reduce: time and peak device memory at
growing sizes (one process, interleaved).   python tools/ab_mproduct.py"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmgcn_amd import adjacency, synth  # noqa: E402


def run(A, M, algo, reps=3):
    torch.cuda.synchronize()
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    ts = []
    out = None
    for _ in range(reps):
        del out
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = adjacency.m_product_csr(A, M, algo=algo)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return out, min(ts) * 1e3, (torch.cuda.max_memory_allocated() - base) / 1e9


for T, N, deg, b in ((16, 200_000, 8, 20), (32, 1_000_000, 8, 20), (16, 2_000_000, 16, 20), (95, 6000, 4, 20)):
    A = synth.device_er_csr(T, N, deg, "cuda")
    M = synth.band_M(T, b, "matlab")
    rec = {"T": T, "N": N, "nnz_in": A.nnz, "band": b}
    outs = {}
    for algo in ("merge", "expand"):
        try:
            outs[algo], ms, gb = run(A, M, algo)
            rec[algo] = {"ms": round(ms, 1), "peak_extra_gb": round(gb, 2), "nnz_out": outs[algo].nnz}
        except (RuntimeError, torch.OutOfMemoryError) as e:   # noqa: PERF203
            rec[algo] = {"error": str(e)[:120]}
            torch.cuda.empty_cache()
    if len(outs) == 2:
        rec["same_pattern"] = bool(torch.equal(outs["merge"].rowptr, outs["expand"].rowptr) and torch.equal(outs["merge"].col, outs["expand"].col))
        rec["max_rel_diff"] = float((outs["merge"].val - outs["expand"].val).abs().max() / outs["expand"].val.abs().max())
    print(json.dumps(rec), flush=True)
    del A, outs
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""The wide (F = 128) SpMM kernels on the reference's REAL operand structure: the chess Ât (read_data.py:116-127, 204-223;
mean 4 entries per row, half the rows the self loop only, community-local columns) replicated on the block diagonal to
bench size (synth.device_chess_tiled_csr).  Times, per launch and interleaved in one process:
    plain SpMM, fused SpMM+GEMM (+AX) on Â and on Âᵀ, the standalone GEMM of the same rows (the MFMA floor of the fused kernel)
and prints model bytes (SURVEY §8d no-reuse formula), the compulsory bytes, and the rates.
    python tools/real_structure_probe.py [--nodes 2000000] [--slices 16] [--feat 128] [--graph chess_tiled|er|powerlaw] [--reps 5]
"""
import argparse
import json
import os
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from tmgcn_amd import ops, synth  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ms.append(s.elapsed_time(e))
    ms.sort()
    return ms[len(ms) // 2], ms[0]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--nodes", type=int, default=2_000_000)
    p.add_argument("--slices", type=int, default=16)
    p.add_argument("--feat", type=int, default=128)
    p.add_argument("--deg", type=int, default=32)
    p.add_argument("--graph", default="chess_tiled")
    p.add_argument("--reps", type=int, default=5)
    p.add_argument("--json", default=None)
    a = p.parse_args()
    dev = torch.device("cuda", 0)
    K = ops.kernels
    A = synth.device_csr(a.graph, a.slices, a.nodes, a.deg, dev)
    At = A.transpose()
    T, N, F = A.T, A.N, a.feat
    cnt = A.rowptr[1:] - A.rowptr[:-1]
    cntT = At.rowptr[1:] - At.rowptr[:-1]
    d = A.nnz / A.n_rows
    rec = {"graph": a.graph, "T": T, "N": N, "F": F, "rows": A.n_rows, "entries": A.nnz, "entries_per_row": round(d, 3),
           "row_lengths": {"max": int(cnt.max()), "median": int(cnt.median()), "min": int(cnt.min()),
                           "share_rows_1_entry": round(float((cnt == 1).sum()) / A.n_rows, 4),
                           "share_rows_le_4": round(float((cnt <= 4).sum()) / A.n_rows, 4),
                           "share_rows_le_8": round(float((cnt <= 8).sum()) / A.n_rows, 4),
                           "transpose_max": int(cntT.max())}}
    model = 8 + F * 4 + (4 + F * 4) / d
    # what a launch must move at least: (col, val) + rowptr once, every X row that is referenced once, every output row once
    compulsory_plain = A.nnz * 8 + A.n_rows * 8 + 2 * A.n_rows * F * 4
    rec["model_bytes_per_entry"] = round(model, 1)
    rec["model_gb_per_launch"] = round(model * A.nnz / 1e9, 2)
    rec["compulsory_gb_plain"] = round(compulsory_plain / 1e9, 2)
    rec["compulsory_gb_fused_ax"] = round((compulsory_plain + A.n_rows * F * 4) / 1e9, 2)
    X = synth.device_features(T, N, F, dev)
    g = torch.Generator(device=dev).manual_seed(1)
    W = torch.randn(F, F, device=dev, generator=g) * 0.1
    out = (torch.empty(T, N, F, device=dev), torch.empty(T, N, F, device=dev), None)
    outT = (torch.empty(T, N, F, device=dev), None, None)
    legs = {
        "spmm": lambda: K.spmm(A, X),
        "spmm_T": lambda: K.spmm(At, X),
        "spmm_gemm_ax": lambda: K.spmm_gemm(A, X, W, out=out),
        "spmm_gemm_T": lambda: K.spmm_gemm(At, X, W, trans_w=True, out=outT),
        "gemm": lambda: K.gemm(X, W),
    }
    rec["ms"] = {}
    for name, fn in legs.items():
        med, best = timed(fn, a.reps)
        rec["ms"][name] = {"median": round(med, 3), "best": round(best, 3)}
    f = rec["ms"]["spmm_gemm_ax"]["median"] * 1e-3
    rec["fused_forward"] = {"model_tbs": round(model * A.nnz / f / 1e12, 3), "frac_model": round(model * A.nnz / f / 8e12, 3),
                            "compulsory_tbs": round((compulsory_plain + A.n_rows * F * 4) / f / 1e12, 3),
                            "frac_compulsory": round((compulsory_plain + A.n_rows * F * 4) / f / 8e12, 3),
                            "gemm_tflops_f32": round(2.0 * A.n_rows * F * F / f / 1e12, 1)}
    s = rec["ms"]["spmm"]["median"] * 1e-3
    rec["plain_forward"] = {"model_tbs": round(model * A.nnz / s / 1e12, 3), "compulsory_tbs": round(compulsory_plain / s / 1e12, 3),
                            "frac_compulsory": round(compulsory_plain / s / 8e12, 3)}
    print(json.dumps(rec, indent=1))
    if a.json:
        json.dump(rec, open(a.json, "w"), indent=1)


if __name__ == "__main__":
    main()

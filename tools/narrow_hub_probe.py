#!/usr/bin/env python3
"""The narrow SpMM kernels (F <= 8: G lanes per row) on a graph with one hub row, this tree against a library variant built
from an earlier commit (tools/ab_r4_baseline.sh REV prev): T = 32 slices of N = 20 000 nodes, 17 entries per row, F = 6,
one row replaced by a hub of H entries.   python tools/narrow_hub_probe.py [variant]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tmgcn_amd import synth  # noqa: E402
from tmgcn_amd.csr import BatchedCSR  # noqa: E402

p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64


def load(path):
    lib = C.CDLL(path)
    lib.tmgcn_spmm_csr_batched_f32_hint.argtypes = [p, p, p, p, p, i64, i32, i32, C.c_float, p]
    lib.tmgcn_spmm_gemm_f32_hint.argtypes = [p, p, p, p, i64, i32, i32, p, i32, i32, i64, i64, i32, p, p, p, i32, C.c_float, p]
    return lib


libs = {"this tree": load(ROOT + "/tm-gcn_amd/libtmgcn_hip.so")}
for v in sys.argv[1:]:
    libs[v] = load(f"{ROOT}/build/variants/{v}/libtmgcn_hip.so")
dev = "cuda"
T, N, F = 32, 20_000, 6
base = synth.device_er_csr(T, N, 16, dev)
X = torch.rand(T, N, F, device=dev)
W = torch.randn(F, F, device=dev)
Y = torch.empty_like(X)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())
g = torch.Generator(device=dev).manual_seed(1)
for H in (0, 1_000, 10_000, 100_000):
    A = base
    if H:
        r = 5 * N + 777
        a, b = int(base.rowptr[r]), int(base.rowptr[r + 1])
        hub = torch.sort(torch.randint(0, N, (H,), device=dev, dtype=torch.int32, generator=g)).values
        rowptr = base.rowptr.clone()
        rowptr[r + 1:] += H - (b - a)
        A = BatchedCSR(rowptr, torch.cat([base.col[:a], hub, base.col[b:]]),
                       torch.cat([base.val[:a], torch.full((H,), 1.0 / H, device=dev), base.val[b:]]), T, N)
    avg = C.c_float(A.nnz / A.n_rows)
    line = f"hub of {H:>7,d} entries:"
    outs = {}
    for name, lib in libs.items():
        for kind in ("spmm", "spmm+gemm"):
            def run():
                if kind == "spmm":
                    return lib.tmgcn_spmm_csr_batched_f32_hint(ptr(A.rowptr), ptr(A.col), ptr(A.val), ptr(X), ptr(Y), A.n_rows, N, F, avg, st)
                return lib.tmgcn_spmm_gemm_f32_hint(ptr(A.rowptr), ptr(A.col), ptr(A.val), ptr(X), A.n_rows, N, F, ptr(W), F, 0, 0, 0, 3,
                                                    ptr(Y), None, None, 0, avg, st)
            for _ in range(3):
                assert run() == 0
            torch.cuda.synchronize()
            ts = []
            for _ in range(7):
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                run()
                e.record()
                torch.cuda.synchronize()
                ts.append(s.elapsed_time(e))
            ts.sort()
            line += f"  {name} {kind} {ts[3] * 1e3:8.1f} us"
            outs[(name, kind)] = Y.clone()
    names = list(libs)
    if len(names) > 1:
        d = float((outs[(names[0], "spmm+gemm")] - outs[(names[1], "spmm+gemm")]).abs().max())
        line += f"  max|diff| {d:.1e}"
    print(line, flush=True)

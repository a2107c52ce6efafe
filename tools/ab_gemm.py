#!/usr/bin/env python3
"""Interleaved A/B timing of the standalone GEMM / dW kernels across library variants
(tools/ab_variants.sh).   python tools/ab_gemm.py [variant ...]"""
import ctypes as C
import glob
import os
import statistics
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64


def load(path):
    lib = C.CDLL(path)
    lib.tmgcn_gemm_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, i64, i64, i32, i32, p]
    if hasattr(lib, "tmgcn_gemm_bf16w_f32"):
        lib.tmgcn_gemm_bf16w_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, i64, i64, i32, i32, p]
    lib.tmgcn_gemm_dw_f32.argtypes = [p, p, p, i64, i32, i32, i64, i32, p, i64, p]
    lib.tmgcn_gemm_dw_workspace_bytes.restype = i64
    lib.tmgcn_gemm_dw_workspace_bytes.argtypes = [i64, i32, i32, i64]
    return lib


ALGO = int(os.environ.get("TMGCN_AB_GEMM_ALGO", "0"))   # 0 = auto (bf16x3), 1 = exact-f32 MFMA
names = sys.argv[1:] or sorted(os.path.basename(os.path.dirname(f)) for f in glob.glob(root + "/build/variants/*/libtmgcn_hip.so"))
libs = {"default": load(root + "/tm-gcn_amd/libtmgcn_hip.so")}
for n in names:
    libs[n] = load(f"{root}/build/variants/{n}/libtmgcn_hip.so")
R, K, Nf = int(os.environ.get("AB_R", 8_000_000)), int(os.environ.get("AB_K", 128)), int(os.environ.get("AB_NF", 128))
A = torch.rand(R, K, device="cuda")
W = torch.randn(K, Nf, device="cuda")
Wh = W.to(torch.bfloat16)
dY = torch.rand(R, Nf, device="cuda")
Y = torch.empty(R, max(K, Nf), device="cuda")     # gemm writes [R,Nf], gemm_dA [R,K]
dW = torch.empty(K, Nf, device="cuda")
ws = torch.empty(int(libs["default"].tmgcn_gemm_dw_workspace_bytes(R, K, Nf, 0)), dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())


def run(lib, which):
    if which == "gemm":
        return lib.tmgcn_gemm_f32(ptr(A), ptr(W), ptr(Y), None, R, K, Nf, 0, 0, 0, 0, ALGO, st)
    if which == "gemm_dA":
        return lib.tmgcn_gemm_f32(ptr(dY), ptr(W), ptr(Y), None, R, Nf, K, 1, 0, 0, 0, ALGO, st)
    if which == "gemm_bf16w":      # weight stored in bf16: three plane products per term
        if not hasattr(lib, "tmgcn_gemm_bf16w_f32"):
            return 0
        return lib.tmgcn_gemm_bf16w_f32(ptr(A), ptr(Wh), ptr(Y), None, R, K, Nf, 0, 0, 0, 0, ALGO, st)
    return lib.tmgcn_gemm_dw_f32(ptr(A), ptr(dY), ptr(dW), R, K, Nf, 0, 0, ptr(ws), ws.numel(), st)


res = {}
for which in ("gemm", "gemm_bf16w", "gemm_dA", "gemm_dW"):
    for lib in libs.values():
        assert run(lib, which) == 0
    torch.cuda.synchronize()
    for rnd in range(7):
        for name, lib in libs.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            run(lib, which)
            e.record()
            torch.cuda.synchronize()
            res.setdefault((which, name), []).append(s.elapsed_time(e))
# results of the variants agree with the default library (dW: bit for bit is not required — a variant may
# change the summation order — but it must stay fp32-accurate)
ref_dw = None
for name, lib in libs.items():
    dW.zero_()
    assert run(lib, "gemm_dW") == 0
    torch.cuda.synchronize()
    if ref_dw is None:
        ref_dw = dW.clone()
        truth = A[:1_000_000].double().t() @ dY[:1_000_000].double() if R <= 1_000_000 else None
    else:
        d = float((dW.double() - ref_dw.double()).abs().max() / ref_dw.double().abs().max())
        print(f"gemm_dW    {name:14s} max|dW - dW_default|/max|dW| = {d:.2e}  bit-equal: {bool(torch.equal(dW, ref_dw))}")
        assert d <= 2e-6, "variant result differs from the default library"

if os.environ.get("AB_POWER"):   # board power / shader clock while each library's dW runs back to back
    import json, re, subprocess, threading, time

    def smi():
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
            card = next(iter(json.loads(out).values()))
            power = next((float(v) for k, v in card.items() if "power" in k.lower() and re.match(r"^[0-9.]+$", str(v))), None)
            return power, next((v for k, v in card.items() if k.lower().startswith("sclk")), None)
        except Exception as e:  # noqa: BLE001
            return None, str(e)

    for name, lib in libs.items():
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                samples.append(smi())
                time.sleep(0.4)
        th = threading.Thread(target=sampler)
        th.start()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < 6.0:
            for _ in range(10):
                run(lib, "gemm_dW")
            torch.cuda.synchronize()
            n += 10
        el = time.perf_counter() - t0
        stop.set()
        th.join()
        pw = [q for q, _ in samples[2:] if q is not None]
        print(json.dumps({"kernel": "gemm_dW", "lib": name, "ms": round(el / n * 1e3, 3),
                          "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None, "power_w_max": max(pw) if pw else None,
                          "sclk_samples": [c for _, c in samples[2:8]]}), flush=True)

fl = 2.0 * R * K * Nf
for (which, name), ms in res.items():
    med = statistics.median(ms)
    print(f"{which:10s} {name:14s} median {med:6.2f} ms  {fl / med / 1e9:6.1f} TFLOP/s  {(R * (K + Nf) * 4) / med / 1e6:6.0f} GB/s")

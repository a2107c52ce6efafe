#!/usr/bin/env python3
"""Board power and shader clock while one kernel runs back to back (rocm-smi sampled from a thread).
Answers "is this kernel power-limited?": a kernel whose memory phase and matrix phase ADD UP instead of
overlapping, at a clock well below 2.4 GHz and at the board's power cap, is.
    python tools/power_probe.py [gemm] [gemm_f32] [dw] [dense] [spmm] [band]"""
import json
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmgcn_amd import ops, synth  # noqa: E402

K = ops.kernels
dev = "cuda"


def smi():
    try:
        out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=10).stdout
        d = json.loads(out)
        card = next(iter(d.values()))
        power = next((float(v) for k, v in card.items() if "power" in k.lower() and re.match(r"^[0-9.]+$", str(v))), None)
        sclk = next((v for k, v in card.items() if k.lower().startswith("sclk")), None)
        return power, sclk
    except Exception as e:  # noqa: BLE001
        return None, str(e)


def probe(name, fn, seconds=6.0):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.4)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        n += 10
    el = time.perf_counter() - t0
    stop.set()
    th.join()
    pw = [p for p, _ in samples[2:] if p is not None]
    print(json.dumps({"kernel": name, "ms": round(el / n * 1e3, 3), "power_w_mean": round(sum(pw) / len(pw), 1) if pw else None,
                      "power_w_max": max(pw) if pw else None, "sclk_samples": [s for _, s in samples[2:8]]}), flush=True)


which = sys.argv[1:] or ["idle", "gemm", "gemm_f32", "dw", "dense", "band", "spmm"]
R, F = 8_000_000, 128
A = torch.rand(1, R, F, device=dev)
W = torch.randn(F, F, device=dev)
dY = torch.rand(1, R, F, device=dev)
for w in which:
    if w == "idle":
        time.sleep(1.0)
        print(json.dumps({"kernel": "idle", "smi": smi()}), flush=True)
    elif w == "gemm":
        probe("gemm_bf16x3", lambda: K.gemm(A, W))
    elif w == "gemm_f32":
        probe("gemm_mfma (exact f32)", lambda: K.gemm(A, W, algo="f32mfma"))
    elif w == "dw":
        probe("gemm_dw_bf16x3", lambda: K.gemm_dw(A, dY, False))
    elif w == "dense":
        X = A.view(128, -1, F)
        op = ops.MOperator(synth.band_M(128, 20, "matlab"), dev).inverse()
        probe("mtransform_bf16x3 (Minv, T=128)", lambda: K.mtransform(op, X))
    elif w == "band":
        X = A.view(16, -1, F)
        op = ops.MOperator(synth.band_M(16, 20, "matlab"), dev)
        probe("mtransform_band", lambda: K.mtransform(op, X))
    elif w == "spmm":
        Ac = synth.device_er_csr(2, 2_000_000, 32, dev)
        Xs = torch.rand(2, 2_000_000, F, device=dev)
        probe("spmm_gemm (fused, 2 slices)", lambda: K.spmm_gemm(Ac, Xs, W))

#!/usr/bin/env python3
"""Fabric-side traffic of the TM-GCN kernels from rocprofv3 PMC passes -> profiles/pmc_traffic.json.

The two counters cannot share a pass on gfx950 (TCC slots) and `gpurun` wants PMC passes without
tracing domains, so take them separately, kernel trace only:

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-epochs
    rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-epochs
    python3 tools/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write --profile-id r02x > profiles/pmc_traffic.json

What the numbers are (MI355X_MICROARCH.md, HBM section): FETCH_SIZE / WRITE_SIZE derive from the L2's
memory-side (fabric) request counters, unit KiB, and on gfx950 FETCH_SIZE reports HALF the bytes of
16-B-per-lane reads (x2 below; calibrated in the same pass on the band M-transform, whose bytes are
known exactly: it reads X once).  Infinity-Cache hits are COUNTED: this is traffic between L2 and the
fabric, an upper bound on what HBM itself moved — never call it "HBM bytes".
"""
import argparse
import csv
import glob
import json
import re
import sys
from collections import defaultdict


def read_pass(d, counter):
    """kernel name -> list of per-dispatch counter values (KiB), in dispatch order."""
    rows = []
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if r.get("Counter_Name") != counter:
                continue
            rows.append((int(r.get("Dispatch_Id", 0)), r["Kernel_Name"], float(r["Counter_Value"]), int(r.get("Grid_Size", 0))))
    rows.sort()
    out = defaultdict(list)
    for _, name, v, _g in rows:
        out[name].append(v)
    return out


def short(name):
    m = re.search(r"tmgcn::(\w+)", name)
    return m.group(1) if m else name[:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("fetch_dir")
    ap.add_argument("write_dir")
    ap.add_argument("--profile-id", default="")
    ap.add_argument("--nodes", type=int, default=2_000_000)
    ap.add_argument("--feat", type=int, default=128)
    ap.add_argument("--slices-per-gpu", type=int, default=16)
    ap.add_argument("--deg", type=int, default=32)
    a = ap.parse_args()
    fetch = read_pass(a.fetch_dir, "FETCH_SIZE")
    write = read_pass(a.write_dir, "WRITE_SIZE")
    T, N, F, d = a.slices_per_gpu, a.nodes, a.feat, a.deg + 1
    slab = T * N * F * 4                                   # one [T,N,F] fp32 tensor
    nnz = T * N * d
    alg_gather = nnz * (8 + F * 4 + (4 + F * 4) / d)       # SURVEY §8d no-reuse model, P2 only
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        if "tmgcn::" not in name:
            continue
        f, w = fetch.get(name, []), write.get(name, [])
        kernels[short(name)] = {
            "kernel": name[:100], "dispatches": max(len(f), len(w)),
            "fetch_kib_raw_per_dispatch": [round(x, 1) for x in f[:8]],
            "write_kib_per_dispatch": [round(x, 1) for x in w[:8]],
        }
    # calibration: the band M-transform reads X once (slab bytes) and writes Y once
    cal = None
    for k, v in kernels.items():
        if k.startswith("mtransform_band") and v["fetch_kib_raw_per_dispatch"]:
            fr = v["fetch_kib_raw_per_dispatch"][0] * 1024
            wr = v["write_kib_per_dispatch"][0] * 1024 if v["write_kib_per_dispatch"] else None
            cal = {"kernel": k, "bytes_read_exactly": slab, "fetch_size_raw_bytes": fr, "fetch_correction": round(slab / fr, 4),
                   "bytes_written_exactly": slab, "write_size_bytes": wr}
            break
    corr = 2.0
    out = {
        "derived_by": "tools/pmc_traffic.py", "profile": a.profile_id,
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, kernel trace only), bench.py --steps 2 --warmup 1",
        "meaning": "fabric-side bytes between L2 and the Infinity Fabric (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), Infinity-Cache hits included: "
                   "an upper bound on HBM traffic, not HBM traffic",
        "nodes": N, "feat": F, "slices_per_gpu": T,
        "units": "counter values are KiB; FETCH_SIZE doubled (gfx950 counts 16-B-per-lane reads at half their bytes; see `calibration`)",
        "calibration": cal, "algorithmic_bytes_per_launch": alg_gather, "kernels": kernels,
    }
    fused = [k for k in kernels if k.startswith("spmm_gemm_kernel")]
    if fused:
        v = kernels[fused[0]]
        # dispatches alternate forward (stores AX and Y: larger WRITE_SIZE) and backward
        pairs = list(zip(v["fetch_kib_raw_per_dispatch"], v["write_kib_per_dispatch"]))
        if pairs:
            fwd = max(pairs, key=lambda p: p[1])
            bwd = min(pairs, key=lambda p: p[1])
            out["kernel"] = fused[0] + " forward (P2 + fused P3, stores AX and Y)"
            out["fetch_size_kib_raw"] = fwd[0]
            out["write_size_kib"] = fwd[1]
            out["spmm_hbm_bytes_per_launch"] = int(fwd[0] * 1024 * corr + fwd[1] * 1024)   # key kept for bench.py; see `meaning`
            out["fabric_bytes_per_launch_forward"] = out["spmm_hbm_bytes_per_launch"]
            out["fabric_bytes_per_launch_backward"] = int(bwd[0] * 1024 * corr + bwd[1] * 1024)
            out["forward_expected"] = {"gather_model": alg_gather, "AX_and_Y_stores": 2 * slab,
                                       "ratio_fabric_to_model": round(out["spmm_hbm_bytes_per_launch"] / (alg_gather + 2 * slab), 4)}
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

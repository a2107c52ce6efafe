#!/usr/bin/env python3
"""Summarise a matrix-core utilisation PMC pass:
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d DIR -- python tools/ab_gemm.py
    python tools/mfma_util.py DIR  ->  JSON: per kernel, MFMA-busy cycles / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)
and the shader clock implied by the kernel-trace durations."""
import csv
import glob
import json
import sys
from collections import defaultdict

d = sys.argv[1]
cnt = defaultdict(lambda: defaultdict(float))
n = defaultdict(int)
for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "tmgcn::" not in name:
            continue
        cnt[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[name] += 1
dur = defaultdict(float)
for path in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        if "tmgcn::" in r["Kernel_Name"]:
            dur[r["Kernel_Name"]] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
out = []
for name, c in cnt.items():
    gui = c.get("GRBM_GUI_ACTIVE", 0.0)
    if not gui:
        continue
    cycles = gui / 8.0                                   # the counter sums the 8 XCDs
    out.append({"kernel": name[:60], "dispatches": n[name],
                "mfma_busy_fraction": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (cycles * 1024), 4),
                "clock_ghz": round(cycles / dur[name], 3) if dur.get(name) else None,
                "avg_ms_profiled": round(dur[name] / n[name] / 1e6, 3) if dur.get(name) else None})
print(json.dumps({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace; utilisation = MFMA busy "
                            "cycles / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs)", "kernels": sorted(out, key=lambda k: k["kernel"])}, indent=1))

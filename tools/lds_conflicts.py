#!/usr/bin/env python3
"""Summarise an LDS-conflict PMC pass: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE
--kernel-trace --output-format csv -d DIR -- python tools/perf_kernels.py spmm gemm
    python tools/lds_conflicts.py DIR  ->  JSON: per kernel, conflict cycles / active LDS cycles."""
import csv
import glob
import json
import sys
from collections import defaultdict

rows = defaultdict(lambda: defaultdict(float))
calls = defaultdict(int)
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        if "tmgcn::" not in name:
            continue
        rows[name][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_LDS_IDX_ACTIVE":
            calls[name] += 1
out = []
for name, c in rows.items():
    act = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
    out.append({"kernel": name[:60], "dispatches": calls[name], "lds_active_cycles": act,
                "lds_bank_conflict_cycles": c.get("SQ_LDS_BANK_CONFLICT", 0.0),
                "conflict_fraction": round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / act, 4) if act else None})
print(json.dumps({"source": "rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace; "
                            "fraction = extra cycles lost to bank conflicts / cycles the LDS array was active",
                  "kernels": sorted(out, key=lambda k: -k["lds_active_cycles"])}, indent=1))

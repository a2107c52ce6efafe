#!/usr/bin/env python3
"""Per-kernel means of every counter of a rocprofv3 --pmc pass (tmgcn kernels only).
    python tools/pmc_kernel_summary.py DIR [DIR ...]   -> JSON {kernel: {counter: mean per dispatch, n}}"""
import csv
import glob
import json
import re
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(path)):
            if "tmgcn::" not in r["Kernel_Name"]:
                continue
            m = re.search(r"tmgcn::(\w+)", r["Kernel_Name"])
            acc[m.group(1) + "|grid=" + r.get("Grid_Size", "?") + "|wg=" + r.get("Workgroup_Size", "?")][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {c: round(sum(v) / len(v), 1) for c, v in cs.items()} | {"dispatches": max(len(v) for v in cs.values())} for k, cs in acc.items()}
print(json.dumps(out, indent=1))

#!/usr/bin/env python3
"""Condense gpurun_out/tolerance_record.jsonl (written by tests/_util.record_tolerance: the measured
error of every assertion whose bound is looser than the stated 1e-5) into one record per
(test function, quantity): worst measured error, the bound, and the fraction of the bound used.
    python tools/tolerance_summary.py [in.jsonl] [out.json]"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "tolerance_record.jsonl")
    dst = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "tolerance_summary.json")
    groups = {}
    for line in open(src):
        r = json.loads(line)
        fn = re.sub(r"\[.*$", "", r["test"])                 # test function without its parameters
        key = (fn, r["bound"], r["kind"])
        g = groups.setdefault(key, {"test": fn, "bound": r["bound"], "kind": r["kind"], "assertions": 0, "worst": 0.0,
                                    "worst_what": None})
        g["assertions"] += 1
        if r["measured"] >= g["worst"]:
            g["worst"], g["worst_what"] = r["measured"], f"{r['test']} :: {r['what']}"
    out = sorted(groups.values(), key=lambda g: -(g["worst"] / g["bound"] if g["bound"] else 0))
    for g in out:
        g["fraction_of_bound_used"] = round(g["worst"] / g["bound"], 4) if g["bound"] else None
    json.dump(out, open(dst, "w"), indent=1)
    for g in out:
        print(f"{g['fraction_of_bound_used']:7.3f} of {g['bound']:.0e} ({g['kind']}), {g['assertions']:4d} assertions  {g['test']}")
    print(f"-> {dst}")


if __name__ == "__main__":
    main()

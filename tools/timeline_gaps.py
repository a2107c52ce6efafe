#!/usr/bin/env python3
"""Where does a step's wall time go that the kernels' own durations do not explain?
Reads a `rocprofv3 --kernel-trace` CSV (…_kernel_trace.csv), takes the window of the last `--steps`
steps (a step = the dispatches from one `mtransform_band` forward launch to the next), and reports per
step: wall time, the UNION of the kernels' busy intervals (kernels of different streams overlap), the
idle time inside the window, the summed kernel time by name, and the longest idle gaps with the
kernels on either side.   python tools/timeline_gaps.py TRACE.csv [--steps 3] [--anchor NAME]"""
import argparse
import csv
import glob
import json
import os
import re


def short(name):
    m = re.search(r"(?:tmgcn::)?(\w+)(?:<[^(]*)?\(", name)
    return (m.group(1) if m else name)[:48]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--anchor", default="mtransform_band_kernel", help="a step starts at every 2nd launch of this kernel (fwd, bwd)")
    a = ap.parse_args()
    path = a.trace
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "")) for r in csv.DictReader(open(path))]
    rows.sort()
    anchors = [i for i, r in enumerate(rows) if a.anchor in r[2]]
    starts = anchors[0::2]                                  # forward P1 of every step
    if len(starts) < a.steps + 1:
        raise SystemExit(f"only {len(starts)} steps in the trace")
    out = []
    for s in range(len(starts) - a.steps - 1, len(starts) - 1):
        win = rows[starts[s]:starts[s + 1]]
        t0, t1 = win[0][0], rows[starts[s + 1]][0]
        busy, cur_s, cur_e, gaps = 0, None, None, []
        prev_name = None
        for st, en, name, _ in win:
            if cur_e is None:
                cur_s, cur_e, prev_name = st, en, name
            elif st <= cur_e:
                if en > cur_e:
                    cur_e, prev_name = en, name
            else:
                busy += cur_e - cur_s
                gaps.append((st - cur_e, short(prev_name), short(name)))
                cur_s, cur_e, prev_name = st, en, name
        busy += cur_e - cur_s
        gaps.append((t1 - cur_e, short(prev_name), "next step"))
        by = {}
        for st, en, name, _ in win:
            k = short(name)
            by[k] = by.get(k, [0, 0.0])
            by[k][0] += 1
            by[k][1] += (en - st) / 1e6
        gaps.sort(reverse=True)
        out.append({"wall_ms": round((t1 - t0) / 1e6, 3), "busy_union_ms": round(busy / 1e6, 3), "idle_ms": round((t1 - t0 - busy) / 1e6, 3),
                    "kernel_ms_sum": round(sum(v[1] for v in by.values()), 3),
                    "by_kernel": {k: {"launches": v[0], "ms": round(v[1], 3)} for k, v in sorted(by.items(), key=lambda kv: -kv[1][1])[:10]},
                    "largest_gaps_ms": [{"ms": round(g / 1e6, 3), "after": p, "before": n} for g, p, n in gaps[:6]]})
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

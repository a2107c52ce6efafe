#!/usr/bin/env python3
"""The kernel sequence of ONE training epoch out of a `rocprofv3 --kernel-trace` CSV: the dispatches between
the last two launches of an anchor kernel (default: the last kernel name that occurs in every epoch), with
each kernel's duration and the idle gap in front of it.  The launch count of an epoch is the number of lines.
    python tools/epoch_sequence.py TRACE.csv [--anchor NAME] [--epochs-back 2]"""
import argparse
import csv
import glob
import os
import re


def short(name):
    m = re.search(r"(?:tmgcn::)?(\w+)(?:<[^(]*)?\(", name)
    return (m.group(1) if m else name)[:60]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--anchor", default=None)
    ap.add_argument("--epochs-back", type=int, default=2, help="which epoch from the end (1 = the very last)")
    a = ap.parse_args()
    path = a.trace
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True))[0]
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path))]
    rows.sort()
    anchor = a.anchor or short(rows[-1][2])
    idx = [i for i, r in enumerate(rows) if anchor in r[2]]
    if len(idx) < a.epochs_back + 1:
        raise SystemExit(f"anchor {anchor!r} occurs only {len(idx)} times")
    lo, hi = idx[-a.epochs_back - 1] + 1, idx[-a.epochs_back] + 1
    win = rows[lo:hi]
    prev_end = rows[lo - 1][1]
    t0 = prev_end
    busy = 0
    print(f"# epoch = dispatches after one `{anchor}` up to and including the next; {len(win)} launches")
    for st, en, name in win:
        print(f"{(st - t0) / 1e3:9.2f} us  gap {(st - prev_end) / 1e3:7.2f}  dur {(en - st) / 1e3:8.2f}  {short(name)}")
        busy += en - st
        prev_end = en
    print(f"# wall {(win[-1][1] - t0) / 1e3:.2f} us, kernels {busy / 1e3:.2f} us, idle {(win[-1][1] - t0 - busy) / 1e3:.2f} us")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What a one-query step consists of, from a rocprofv3 kernel trace of `bench.py` (config C2):

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o bench -- python3 bench.py --steps 300 --warmup 30 ...
    python3 scripts/step_overhead.py <dir>

Per step: duration of the scan kernel, of the select kernel, the idle gap between the scan's end and the select's
start, and between the select's end and the next scan's start (timestamps of the trace, ns).  Used for DESIGN.md §5
(is fusing the select into the scan's tail worth a kernel boundary?).
"""
import csv
import glob
import statistics as st
import sys

root = sys.argv[1]
paths = glob.glob(f"{root}/**/*kernel_trace.csv", recursive=True)
rows = []
for p in paths:
    with open(p) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
scan_d, sel_d, gap_a, gap_b = [], [], [], []
for i in range(1, len(rows) - 1):
    s0, e0, n0 = rows[i - 1]
    s1, e1, n1 = rows[i]
    s2, e2, n2 = rows[i + 1]
    if "scan_rows_f32" in n0 and "select_rerank" in n1 and "scan_rows_f32" in n2:
        scan_d.append(e0 - s0)
        sel_d.append(e1 - s1)
        gap_a.append(s1 - e0)
        gap_b.append(s2 - e1)
if not scan_d:
    sys.exit("no scan -> select -> scan triples in the trace")
skip = len(scan_d) // 5          # the first fifth: warm-up / conditioning


def med(v):
    return st.median(v[skip:]) / 1e3


print(f"{len(scan_d) - skip} steps (after skipping {skip}): scan {med(scan_d):.2f} us, select {med(sel_d):.2f} us, "
      f"gap scan->select {med(gap_a):.2f} us, gap select->scan {med(gap_b):.2f} us, "
      f"step {med([a + b + c + d for a, b, c, d in zip(scan_d, sel_d, gap_a, gap_b)]):.2f} us")

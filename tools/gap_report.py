#!/usr/bin/env python3
"""Idle time between consecutive kernels of one stream from a rocprofv3 kernel trace: tools/gap_report.py <kernel_trace.csv>.
Prints, per (previous kernel -> next kernel) pair, the count and mean / median gap (next start - previous end), and the total per PPO update."""
import csv, sys, collections, statistics

def short(n):
    return n.split("(")[0].replace("void ", "").split("<")[0]

rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
gaps = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    g = s1 - e0
    if g < 50_000:   # host-side pauses between phases are not launch gaps
        gaps[(n0, n1)].append(g)
tot = 0.0
for (a, b), v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) < 8:
        continue
    print("%-26s -> %-26s n=%5d  mean %7.0f ns  median %7.0f ns  p90 %7.0f ns" % (a, b, len(v), statistics.mean(v), statistics.median(v), sorted(v)[int(0.9 * len(v))]))

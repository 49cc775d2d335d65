#!/usr/bin/env python3
"""Idle time between consecutive kernels of a rocprofv3 --kernel-trace csv: for every pair (previous kernel -> next kernel) the number of
boundaries, the mean / median gap (next start - previous end) and the mean duration of the next kernel.

usage: python tools/prof/kernel_gaps.py <..._kernel_trace.csv> [min_count]"""
import csv
import statistics
import sys
from collections import defaultdict


def short(name):
    return name.split("<")[0].split("(")[0].replace("void ", "")[:40]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])))
    rows.sort()
    gaps, durs = defaultdict(list), defaultdict(list)
    for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
        gaps[(n0, n1)].append(s1 - e0)
        durs[(n0, n1)].append(e1 - s1)
    mincount = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    print(f"{'previous -> next':72s} {'count':>6s} {'gap mean':>9s} {'median':>8s} {'next dur':>9s}   (us)")
    for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
        if len(v) < mincount:
            continue
        print(f"{k[0] + ' -> ' + k[1]:72s} {len(v):6d} {statistics.mean(v) / 1e3:9.2f} {statistics.median(v) / 1e3:8.2f} {statistics.mean(durs[k]) / 1e3:9.2f}")


if __name__ == "__main__":
    main()

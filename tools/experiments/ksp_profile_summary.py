"""Turns gpurun_out/chebprof (tools/experiments/cheb_profile.sh) into profiles/r01_ksp_kernels_summary.json.

Per kernel: launches and average duration from the rocprofv3 kernel trace, HBM bytes per launch from the FETCH_SIZE and
WRITE_SIZE passes (KiB units; FETCH_SIZE doubled on gfx950 as MI355X_MICROARCH.md prescribes, calibrated here on streaming
copies), and for the kernels with a stated algorithmic byte count the roofline fraction of 8 TB/s.
"""
import csv
import json
import os
import sys

SRC = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/chebprof"
N = 512 ** 3
ALGO = {"k_cg_A<": 64, "k_cg_B<": 24, "k_cheb<": 40}  # B/cell, DESIGN.md 5

dur, cnt = {}, {}
for r in csv.DictReader(open(os.path.join(SRC, "trace", "k_kernel_trace.csv"))):
    dur.setdefault(r["Kernel_Name"], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
for f in ("fetch", "write"):
    for r in csv.DictReader(open(os.path.join(SRC, f, "k_counter_collection.csv"))):
        cnt.setdefault((f, r["Kernel_Name"]), []).append(float(r["Counter_Value"]))


def full(v):  # the 512^3 launches only (the same kernels also run on small set-up problems)
    m = max(v)
    return [x for x in v if x > 0.5 * m]


out = {}
for k, v in dur.items():
    if not k.startswith("fl::") or "(" in k:
        pass
    name = k.split("(")[0].replace("void ", "")
    if not name.startswith("fl::"):
        continue
    v = full(v)
    if len(v) < 40 or sum(v) / len(v) < 0.2:
        continue
    e = {"launches": len(v), "avg_ms": round(sum(v) / len(v), 4)}
    fe, wr = cnt.get(("fetch", k)), cnt.get(("write", k))
    if fe and wr:
        fe, wr = full(fe), full(wr)
        fb, wb = 2 * 1024 * sum(fe) / len(fe), 1024 * sum(wr) / len(wr)
        e.update(fetch_B_per_cell=round(fb / N, 2), write_B_per_cell=round(wb / N, 2), hbm_GB_per_launch=round((fb + wb) / 1e9, 3),
                 moved_TBps=round((fb + wb) / (e["avg_ms"] * 1e-3) / 1e12, 3))
    for pat, b in ALGO.items():
        if pat in name:
            e.update(algorithmic_B_per_cell=b, algorithmic_TBps=round(b * N / (e["avg_ms"] * 1e-3) / 1e12, 3))
            e["roofline_frac_of_8TBps"] = round(e["algorithmic_TBps"] / 8.0, 3)
    out[name] = e
rates = [l.strip() for l in open(os.path.join(SRC, "trace.log")) if l.startswith("n= 512")]
json.dump({"command": "rocprofv3 {--kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE} -- python3 tools/ksp_bench.py 512",
           "rates_under_trace": rates, "kernels": out}, sys.stdout, indent=1)
print()

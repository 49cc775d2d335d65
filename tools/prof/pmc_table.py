"""Per-kernel table from the passes of tools/prof/pmc_kernel.sh: mean duration (kernel trace) and mean counter values per launch.
FETCH_SIZE is doubled (MI355X_MICROARCH.md, HBM section: gfx950 reports half the bytes of wide streaming reads); FETCH / WRITE are KB.
usage: python tools/prof/pmc_table.py <dir> <kernel-name substring> [min_ms]"""
import collections
import csv
import glob
import json
import os
import statistics
import sys

R, pat = sys.argv[1], sys.argv[2]
min_ms = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0


def name(row):
    return row["Kernel_Name"].split("(")[0].replace("void ", "")


dur = collections.defaultdict(list)
for f in glob.glob(os.path.join(R, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        dur[name(row)].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(R, "pmc*", "**", "*counter_collection.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        cnt[name(row)][row["Counter_Name"]].append(float(row["Counter_Value"]))
out = {}
for k in sorted(dur):
    if pat not in k:
        continue
    big = [d for d in dur[k] if d >= max(min_ms, 0.5 * max(dur[k]))]
    if not big:
        continue
    e = {"launches": len(big), "mean_ms": round(statistics.mean(big), 4), "median_ms": round(statistics.median(big), 4)}
    for c, v in cnt.get(k, {}).items():
        top = [x for x in v if x >= 0.5 * max(v)] if max(v) > 0 else v
        e[c] = statistics.mean(top)
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        e["fetch_GB_corrected"] = round(e["FETCH_SIZE"] * 2 * 1024 / 1e9, 3)
        e["write_GB"] = round(e["WRITE_SIZE"] * 1024 / 1e9, 3)
        e["hbm_GB"] = round(e["fetch_GB_corrected"] + e["write_GB"], 3)
        e["hbm_TBps"] = round(e["hbm_GB"] / e["mean_ms"], 3)
    out[k] = e
print(json.dumps(out, indent=1))

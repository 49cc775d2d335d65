"""Turn the outputs of tools/experiments/r02_profile.sh (gpurun_out/r02_prof) into profiles/: per-kernel HBM traffic of the 512^3
launches from the FETCH_SIZE / WRITE_SIZE passes (KB; FETCH doubled per MI355X_MICROARCH.md, calibrated on k_pad_copy: a 1 GiB read
reports 524 300 KB) and rocprofv3's kernel durations of the trace pass.  usage: python tools/experiments/pmc_summary.py <dir> <tag>"""
import collections
import csv
import json
import os
import statistics
import sys

R, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from fluca_amd import provenance  # noqa: E402  (the pass records which sources it counted: bench.py's traffic_stale)
N = 512 ** 3


def pmc(path):
    acc = collections.defaultdict(list)
    for row in csv.DictReader(open(path)):
        acc[row["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(row["Counter_Value"]))
    return acc


def top_cluster(vals):
    """launches of the largest problem (the bench's 512^3 solve): values within 2 % of the maximum"""
    m = max(vals)
    return [v for v in vals if v >= 0.98 * m]


f, w = pmc(os.path.join(R, "fetch", "k_counter_collection.csv")), pmc(os.path.join(R, "write", "k_counter_collection.csv"))
dur = collections.defaultdict(list)
for row in csv.DictReader(open(os.path.join(R, "trace", "k_kernel_trace.csv"))):
    dur[row["Kernel_Name"].split("(")[0].replace("void ", "")].append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6)
calib = statistics.median(top_cluster(f["fl::k_pad_copy"])) * 2 * 1024 / (N * 8.0)
out = {"calibration_k_pad_copy_fetch_over_1GiB": calib}
for k in sorted(f):
    if not any(t in k for t in ("k_cg_A<2, 8", "k_cg_Bq<2, 8", "k_cheb2<2, 8", "k_cg_B<4")):
        continue
    fv, wv = top_cluster(f[k]), top_cluster(w.get(k, [0.0]))
    fb, wb = statistics.median(fv) * 2 * 1024, statistics.median(wv) * 1024
    d = top_cluster(dur.get(k, [0.0]))
    # the 512^3 launches: everything at least half as long as the longest, minus the few first-touch / cold outliers (more than 15 % off
    # the median of that set) -- a plain ">= 0.8 max" kept only the outliers once a run had a dozen slow launches
    big = [x for x in dur.get(k, []) if x >= 0.5 * max(dur[k])]
    if big:
        med = statistics.median(big)
        big = [x for x in big if abs(x - med) <= 0.15 * med]
    out[k] = {"launches_counted": len(fv), "fetch_GB": round(fb / 1e9, 3), "write_GB": round(wb / 1e9, 3), "hbm_bytes_per_launch": fb + wb,
              "B_per_cell": round((fb + wb) / N, 2), "rocprof_avg_ms_512cubed_launches": round(statistics.mean(big), 4) if big else None,
              "rocprof_median_ms": round(statistics.median(big), 4) if big else None, "rocprof_launches": len(big)}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
# k_cg_Bq alternates between two instantiations (even iterations: r-update only; odd ones: + both x-updates): the per-launch figure
# quoted by bench.py is their mean
bq = [out[k] for k in ("fl::k_cg_Bq<2, 8, true, 2, 0>", "fl::k_cg_Bq<2, 8, true, 2, 2>") if k in out]
if len(bq) == 2:
    out["fl::k_cg_Bq (mean of the even- and odd-iteration launches)"] = {
        "fetch_GB": round((bq[0]["fetch_GB"] + bq[1]["fetch_GB"]) / 2, 3), "write_GB": round((bq[0]["write_GB"] + bq[1]["write_GB"]) / 2, 3),
        "hbm_bytes_per_launch": (bq[0]["hbm_bytes_per_launch"] + bq[1]["hbm_bytes_per_launch"]) / 2,
        "B_per_cell": round((bq[0]["B_per_cell"] + bq[1]["B_per_cell"]) / 2, 2),
        "rocprof_avg_ms_512cubed_launches": round((bq[0]["rocprof_avg_ms_512cubed_launches"] + bq[1]["rocprof_avg_ms_512cubed_launches"]) / 2, 4)}
    json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
for k, name in (("fl::k_cg_A<2, 8, true, 1, 2, false>", "pmc_k_cg_A.json"), ("fl::k_cheb2<2, 8, true, 2, false, 0>", "pmc_k_cheb2.json"),   # the two-step sweep of config 3 (Z = 0; round 5 added the template argument)
                ("fl::k_cg_Bq (mean of the even- and odd-iteration launches)", "pmc_k_cg_Bq.json")):
    if k in out:
        o = dict(out[k])
        key = name[len("pmc_"):-len(".json")]
        o.update({"kernel_key": key, "sources_at_profiling": provenance.source_hashes(key)})
        o.update({"kernel": k, "fetch_bytes_corrected": o["fetch_GB"] * 1e9, "write_bytes": o["write_GB"] * 1e9,
                  "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over `python3 bench.py --steps 8 --warmup 2 --skip-cpu --skip-extras` "
                            f"(tools/experiments/r0N_profile.sh, summarised by tools/experiments/pmc_summary.py, {tag}); FETCH_SIZE in KB doubled per MI355X_MICROARCH.md"})
        json.dump(o, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
# the other sizes the same passes saw (bench.py's configs run behind the headline): 256^3 (config 2) and the 512 x 512 x 256 block of the config-5
# rehearsal -- counter values near 1/8 and 1/2 of the 512^3 ones.  One record per workload: HBM bytes per launch of k_cg_A and of k_cg_Bq (mean of
# the even- and odd-iteration instantiations), stamped like the others.
for wl, lo, hi, cells in (("c2_256", 0.09, 0.16, 256 ** 3), ("c5_block", 0.45, 0.55, 512 * 512 * 256)):
    rec = {}
    for k in sorted(f):
        if not ("k_cg_A<2, 8" in k or "k_cg_Bq<2, 8, true, 2, 0>" in k or "k_cg_Bq<2, 8, true, 2, 2>" in k):
            continue
        mf, mw = max(f[k]), max(w.get(k, [0.0]))
        fv, wv = [v for v in f[k] if lo * mf <= v <= hi * mf], [v for v in w.get(k, []) if lo * mw <= v <= hi * mw]
        if fv and wv:
            rec[k] = {"launches_counted": len(fv), "fetch_GB": round(statistics.median(fv) * 2 * 1024 / 1e9, 4), "write_GB": round(statistics.median(wv) * 1024 / 1e9, 4),
                      "hbm_bytes_per_launch": statistics.median(fv) * 2 * 1024 + statistics.median(wv) * 1024}
            rec[k]["B_per_cell"] = round(rec[k]["hbm_bytes_per_launch"] / cells, 2)
    a = [v for k, v in rec.items() if "k_cg_A" in k]
    b = [v for k, v in rec.items() if "k_cg_Bq" in k]
    if len(a) == 1 and len(b) == 2:
        o = {"workload": wl, "cells": cells, "k_cg_A": a[0]["hbm_bytes_per_launch"], "k_cg_Bq": (b[0]["hbm_bytes_per_launch"] + b[1]["hbm_bytes_per_launch"]) / 2,
             "B_per_cell_per_iteration": round((a[0]["hbm_bytes_per_launch"] + (b[0]["hbm_bytes_per_launch"] + b[1]["hbm_bytes_per_launch"]) / 2) / cells, 2),
             "kernels": rec, "sources_at_profiling": {"k_cg_A": provenance.source_hashes("k_cg_A"), "k_cg_Bq": provenance.source_hashes("k_cg_Bq")},
             "source": f"the launches of this size inside the FETCH_SIZE / WRITE_SIZE passes of the bench command (tools/experiments/pmc_summary.py, {tag})"}
        json.dump(o, open(os.path.join(ROOT, "profiles", f"pmc_workload_{wl}.json"), "w"), indent=1)
        out["workload " + wl] = {kk: o[kk] for kk in ("k_cg_A", "k_cg_Bq", "B_per_cell_per_iteration")}
json.dump(out, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

"""profiles/pmc_k_mom3.json (what bench.py quotes as roofline.traffic of the momentum operator) from the table of tools/prof/pmc_table.py over
the passes of tools/prof/pmc_kernel.sh, stamped with the git blob hashes of the kernel's sources (fluca_amd/provenance.py).
usage: python tools/prof/pmc_mom3_json.py gpurun_out/<dir>/table.json <tag> [k_mom2]     (k_mom2: the stored path, mom_bench --fly 0 -> profiles/pmc_k_mom2.json)"""
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
sys.path.insert(0, ROOT)
from fluca_amd import provenance  # noqa: E402

tab, tag = json.load(open(sys.argv[1])), sys.argv[2]
kk = sys.argv[3] if len(sys.argv) > 3 else "k_mom3"
key = next(k for k in tab if kk + "<8, 0, false, 0, 1>" in k)   # the plain product (DOT 0, no Jacobi, padded output)
e = tab[key]
N = 512 ** 3
out = {"kernel": key, "kernel_key": kk, "hbm_bytes_per_launch": e["hbm_GB"] * 1e9, "fetch_bytes_corrected": e["fetch_GB_corrected"] * 1e9,
       "write_bytes": e["write_GB"] * 1e9, "B_per_cell": round(e["hbm_GB"] * 1e9 / N, 2), "rocprof_avg_ms_512cubed_launches": e["mean_ms"], "launches": e["launches"],
       "sources_at_profiling": provenance.source_hashes(kk),
       "source": f"tools/experiments/{tag}_profile.sh: tools/prof/pmc_kernel.sh over `python3 tools/mom_bench.py --cells 512 --fly {1 if kk == 'k_mom3' else 0} --nosolve --reps 5` (separate FETCH_SIZE / "
                 f"WRITE_SIZE passes, FETCH doubled), summarised by tools/prof/pmc_table.py; all variants and the other counters in profiles/{tag}_{kk[2:]}_pmc.json"}
json.dump(out, open(os.path.join(ROOT, "profiles", f"pmc_{kk}.json"), "w"), indent=1)
json.dump(tab, open(os.path.join(ROOT, "profiles", f"{tag}_{kk[2:]}_pmc.json"), "w"), indent=1)
print(json.dumps(out, indent=1))

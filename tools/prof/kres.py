"""Per-kernel resource usage of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage), ISA left in /tmp/kres/.
usage: python tools/prof/kres.py fluca_amd/csrc/fl_momentum.hip k_mom3"""
import re
import subprocess
import sys

src, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
subprocess.run(["mkdir", "-p", "/tmp/kres"])
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Iinclude", "-Ifluca_amd/csrc", "-x", "hip", "-c", src, "-o", "/tmp/kres/out.o",
                      "-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"] + ["-D" + d for d in __import__("os").environ.get("FL_DEFINES", "").split()], capture_output=True, text=True).stderr
cur, rows = None, {}
for l in out.splitlines():
    m = re.search(r"remark:\s+\S+:\d+:\d+:\s+(.*?) \[-Rpass", l)
    if not m:
        if "error" in l:
            print(l)
        continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        cur = t.split(": ")[1]
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.split(":", 1)
        rows[cur][k.strip()] = v.strip()
for f, r in rows.items():
    if pat not in f:
        continue
    d = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip().split("(")[0]
    print("%-58s VGPR %s scratch %s sgpr %s sspill %s vspill %s lds %s" % (d, r.get("VGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("TotalSGPRs"), r.get("SGPRs Spill"), r.get("VGPRs Spill"),
                                                                         r.get("LDS Size [bytes/block]")))

"""one line of the driver's bench command (K = 20) for box-to-box comparisons: python tools/experiments/bench_spread.py"""
import json
import subprocess
import sys

out = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--skip-cpu", "--skip-configs"], capture_output=True, text=True).stdout
d = json.loads([l for l in out.splitlines() if l.startswith("{")][-1])
k = d["roofline"]["kernels"]
print("value %.1f it/s  ms/step %.4f  unplaced %s  probe %.3f -> %.3f  k_cg_A %.4f k_cg_Bq %.4f  mg %.4f s" % (
    d["value"], d["ms_per_step"], d.get("value_unplaced"), d["placement"]["probe_ms_all_vectors_in_one_block"], d["placement"]["probe_ms_chosen"],
    k[0]["avg_launch_ms"], k[1]["avg_launch_ms"], d["time_to_solution"]["multigrid_pcg"]["seconds"]))

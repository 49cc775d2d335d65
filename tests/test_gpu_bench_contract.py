"""-m gpu: the driver's bench.py contract, at toy size -- one JSON line with the required keys at N = 1, and the N = 2 launch
line of the driver (torch.distributed.run, one rank per process) over the host-staged transport, because a one-GPU box cannot
give each rank its own device (RCCL refuses two ranks on one GPU; the production transport is exercised by
tests/test_gpu_rccl_loopback.py)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
        "roofline", "cpu_baseline"}


def _line(out):
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout + out.stderr          # ONE JSON line
    return json.loads(lines[0])


def test_single_gpu_line_has_the_contract_keys():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cells", "64", "--steps", "20", "--warmup", "3", "--cpu-cells", "32", "--cpu-iters", "20", "--skip-configs"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    d = _line(out)
    assert KEYS <= set(d)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 3 and d["scaling"] == "weak" and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["value"] > 0 and abs(d["ms_per_step"] * d["value"] * (512 / 64) ** 3 / 1e3 - 1.0) < 1e-6      # value = 512^3-equivalent iterations/s
    r, c = d["roofline"], d["cpu_baseline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    # frac IS a fraction of the roofline: bytes that crossed the HBM interface (counter pass, or the kernel's own floor where the workload has
    # none, as at this toy size) / time / peak -- never above 1, in the headline entry, in the whole iteration and in every kernel; the
    # textbook figure of SURVEY 8(d) travels beside it under its own name
    for e in [r, r["iteration"]] + r["kernels"]:
        assert 0 < e["frac"] <= 1.0 and abs(e["frac"] - e["achieved"] / r["peak"]) < 1e-12, e
        assert e["frac"] == e["traffic_frac"] and e["frac_textbook"] > 0 and e["bytes_source"], e
        assert e["frac_textbook"] >= e["frac"] * 0.999      # the textbook steps move at least the bytes the fused kernels move
    assert r["traffic_stale"] in (None, False)
    assert r["measured_copy_GBps"] > 3000.0 and r["measured_3r3w_GBps"] > 3000.0
    assert d["placement"]["mode"] == "auto" and d["hbm_bytes_held"] > 0
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and "sample" in c and "petsc_cpu" in c
    assert c["parity_on_sample"]["iters_gpu"] == c["parity_on_sample"]["iters_cpu"] and c["parity_on_sample"]["rel_max_diff_x"] < 1e-9
    assert "workload" in d["config"] and "model" not in d["config"]


def test_two_ranks_through_the_driver_launch_line():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "10", "--warmup", "2", "--cells", "32", "--skip-cpu", "--transport", "host"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    d = _line(out)
    assert d["n_gpus"] == 2 and d["config"]["rank_grid"] == [1, 1, 2] and d["config"]["cells_per_gpu"] == 32 ** 3
    assert "NOT the production transport" in d["config"]["halo"] and d["cpu_baseline"] is None and d["value"] > 0
    assert [r["rank"] for r in d["ranks"]] == [0, 1] and d["transport_ranks"] == 2
    # the line says which wire carried the halos, how much, and how evenly the ranks ran (the first SCALE record can be judged without a second run)
    w = d["wire"]
    assert w["transport"] == "host" and w["nranks_reported_by_the_communicator"] == 2 and w["ranks_reported"] == [0, 1]
    assert w["neighbours_per_rank"] == [1, 1] and w["halo_bytes_per_iter_per_rank"] == [32 * 32 * 8] * 2 and w["halo_bytes_per_iter"] == 2 * 32 * 32 * 8
    assert w["k_cg_A_ms_spread"]["max_over_min"] >= 1.0


def test_strong_scaling_option_splits_one_grid():
    """--scaling strong: ONE --cells^3 grid over the ranks (the other reading of BASELINE.json's '512^3 grid at 1/2/4/8 GPUs')."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--cells", "32", "--skip-cpu", "--transport", "host", "--scaling", "strong"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    d = _line(out)
    assert d["scaling"] == "strong" and d["config"]["cells_per_gpu"] == 32 * 32 * 16 and d["config"]["workload"].startswith("32x32x32")
    assert d["value"] > 0


def test_two_ranks_without_rccl_fail_instead_of_falling_back():
    """One GPU, two ranks, production transport: RCCL cannot give both ranks the device, so the run must exit non-zero and print
    no JSON line (a silently host-staged scaling curve would be worthless)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--cells", "32", "--skip-cpu"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]


def test_quoted_counter_passes_are_of_the_kernels_in_the_tree():
    """roofline.traffic is replayed from profiles/pmc_*.json (counters cannot be read inside the run): every pass bench.py quotes must
    carry the git blob hashes of the kernel's sources as they are in this tree (fluca_amd/provenance.py), or the line says traffic_stale."""
    sys.path.insert(0, ROOT)
    import bench
    for kernel in ("k_cg_A", "k_cg_Bq", "k_cheb2", "k_mom3"):
        traffic, stale = bench.pmc_traffic(kernel)
        assert traffic and traffic > 0, kernel
        assert stale is False, f"profiles/pmc_{kernel}.json was taken from other sources than the tree's: re-run the counter pass"
    for workload in ("c2_256", "c5_block"):     # the same pair on the grids of config 2 and of the config-5 rehearsal
        rec = bench.workload_traffic(workload)
        assert rec and set(rec) == {"k_cg_A", "k_cg_Bq"}, workload
        for kernel, (traffic, stale) in rec.items():
            assert traffic > 0 and stale is False, (workload, kernel)

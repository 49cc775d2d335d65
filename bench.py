#!/usr/bin/env python3
"""bench.py -- pressure-Poisson Krylov iterations/s on MI355X (BASELINE.json metric).

A "step" is ONE Jacobi-PCG iteration of KSPSolve(kspS) (fluca/src/ns/utils/abfpc/abfpc.c:77) on the cavity-flow
Schur complement (BCs of fluca/tests/cavity_flow/cavity_flow_3d.c:72-77, box [0,1]x[0,1]x[0,0.5], kappa = dt/rho = 1e-3)
with a 512^3 block of cells per GPU (weak scaling: N GPUs hold a (512 rx) x (512 ry) x (512 rz) grid, split like the
reference's -cart_ranks_{x,y,z}).  Right-hand side: SURVEY 8d micro-benchmark b = S p*, p* seeded uniform(-1,1); fixed
iteration count (rtol = atol = 0), inputs resident in HBM before the timed region.

One JSON line on rank 0.  value = (cells of the whole job / 512^3) * K / seconds  ==  iterations/s of a 512^3 grid at N=1.
At N = 1 the line also carries "configs": driver-timed lines for the other single-GPU configurations of BASELINE.json (C2 256^3
cavity CG, C3 512^3 channel Chebyshev-Jacobi sweeps, C4 512^3 with the IBM kernels on an immersed sphere), each with its own
roofline object, "C1" (config 1: the 64^3 cavity's CPU KSP CG + Jacobi, iterations / seconds / cores), "C5_rank_rehearsal" (one rank's share of config 5
without the halo exchange, with its parity against the oracle's assembled block), "flow_step" (a whole 512^3 time step, and the velocity-field parity of two
steps at 128^3), and "value_unplaced": the same K steps on plainly allocated vectors (see "placement" in include/fluca_hip.h).
For N > 1 the halo transport is RCCL; if RCCL cannot be initialised the run FAILS (a host-staged curve would be worthless) --
the host-staged rehearsal has to be asked for with --transport host.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch
import torch.distributed as dist

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); measured streaming peak ~6290 GB/s
B_ITER_ALGO = 88               # algorithmic bytes / cell / PCG iteration
# The textbook steps each kernel of an iteration replaces (direction 24 + SpMV/dot 16 + update 48 = 88 B/cell) and what it moves:
#   variant 0 (default)  k_cg_A: direction + SpMV/dot = 40 algorithmic; reads r,p writes p' = 24 moved (q is formed, never stored)
#                        k_cg_Bq: update = 48 algorithmic; reads p',r writes r = 24 on even iterations, + p_old, x read and x written = 48 on
#                        odd ones (both x-updates of the pair) = 36 moved on average
#   variant 2            k_cg_A: direction + SpMV/dot + x-half of the update = 64; reads r,p,x writes p',q,x = 48.  k_cg_B: r-half = 24; 24
CG_KERNELS = {0: (("k_cg_A (p-update + S*p + dot; q never stored)", 40, 24), ("k_cg_Bq (q = S*p again, r-update + sums; both x-updates on odd iterations)", 48, 36)),
              2: (("k_cg_A (p-update + S*p + dot + deferred x-update, q stored)", 64, 48), ("k_cg_B (r-update + sums)", 24, 24)),
              1: (("k_cg_apply_dot (unfused)", 16, 16), ("k_cg_B", 24, 24))}
B_ITER_REAL = {0: 60, 2: 72, 1: 136}


def phys(bytes_counted, bytes_floor, bytes_textbook, ms):
    """The roofline entries of one kernel launch.  `frac` is a FRACTION of the HBM roofline: bytes that crossed the HBM interface per launch
    (rocprofv3 FETCH_SIZE x 2 + WRITE_SIZE from the last committed counter pass, profiles/pmc_*.json; where no pass exists for the workload, the
    bytes the kernel must move by construction -- its own floor) / the launch's duration (HIP events inside the timed region) / 8 TB/s.
    Never above 1.  SURVEY 8(d)'s textbook figure (the bytes of the BLAS-1 / SpMV steps the kernel REPLACES, which it no longer moves: q is
    never stored, x is touched every second iteration, two Chebyshev steps share one sweep) is kept beside it as achieved_textbook /
    frac_textbook: a speed-up over the textbook sequence, not a bandwidth."""
    used = bytes_counted if bytes_counted else bytes_floor
    g = used / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
    gt = bytes_textbook / (ms * 1e-3) / 1e9 if ms and ms > 0 else None
    return {"achieved": g, "frac": g / HBM_PEAK_GBS if g else None, "bytes_per_launch": used,
            "bytes_source": "rocprofv3 PMC pass (profiles/)" if bytes_counted else "the kernel's own byte floor x cells (no counter pass for this workload)",
            "traffic": bytes_counted, "traffic_GBps": g, "traffic_frac": g / HBM_PEAK_GBS if g else None,
            "floor_bytes_per_launch": bytes_floor, "floor_GBps": bytes_floor / (ms * 1e-3) / 1e9 if ms and ms > 0 else None,
            "achieved_textbook": gt, "frac_textbook": gt / HBM_PEAK_GBS if gt else None, "avg_launch_ms": ms}


FRAC_NOTE = ("frac = achieved / peak with achieved = bytes that crossed the HBM interface per launch (counter pass; the kernel's byte floor where none exists) / "
             "HIP-event duration: the physical fraction of the 8 TB/s roofline, <= 1.  *_textbook = SURVEY 8(d)'s algorithmic bytes of the steps the kernel "
             "replaces / the same duration (exceeds the peak once a kernel stops moving bytes the textbook sequence moves)")


def cg_roofline(info, ncell, variant, ms_per_iter, traffic=None):
    """roofline object of a CG run: the kernel with the larger share of the iteration is the headline entry ("dominant kernel"), the other
    one and the whole iteration (every kernel, driver-timed) are listed beside it.  Durations: HIP events around each launch inside the
    timed region (fl_ksp_opts.profile).  Fractions as phys() defines them."""
    ks = []
    for (name, algo, moved), ms, n in zip(CG_KERNELS[variant], (info["kernel_ms"], info["kernel2_ms"]), (info["kernel_launches"], info["kernel2_launches"])):
        tr, tstale = traffic.get(name.split(" ")[0], (None, None)) if isinstance(traffic, dict) else (None, None)
        k = {"kernel": name, "algorithmic_bytes_per_cell": algo, "moved_bytes_per_cell": moved, "launches_timed": n, "traffic_stale": tstale}
        k.update(phys(tr, moved * ncell, algo * ncell, ms))
        ks.append(k)
    dom = max(ks, key=lambda k: k["avg_launch_ms"] or 0.0)
    it_tb = sum(k["bytes_per_launch"] for k in ks)
    it = phys(it_tb if any(k["traffic"] for k in ks) else None, sum(k["floor_bytes_per_launch"] for k in ks), B_ITER_ALGO * ncell, ms_per_iter)
    it.update({"algorithmic_bytes_per_cell": B_ITER_ALGO, "moved_bytes_per_cell": B_ITER_REAL.get(variant), "ms": ms_per_iter,
               "note": "all kernels of one iteration, driver-timed (ms_per_step)"})
    out = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac_note": FRAC_NOTE,
           "traffic_stale": (any(k["traffic_stale"] for k in ks) if any(k["traffic"] for k in ks) else None)}
    out.update({k: v for k, v in dom.items() if k != "traffic_stale"})
    out.update({"kernels": ks, "iteration": it})
    return out
B_CHEB_ALGO = 40               # algorithmic bytes / cell / Chebyshev-Jacobi step: read x, b, d; write x', d'
IBM_B_PER_MARKER = 1584        # L * (4^3 * 3 * 8 + 6 * 8) bytes per interp or spread of three components
RANK_GRIDS = {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}


def _host_mem_available_gb():
    try:
        with open("/proc/meminfo") as fh:
            for line in fh:
                if line.startswith("MemAvailable:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return None


def _cpu_leg(n1, iters):
    """The oracle's assembled-CSR Jacobi-PCG (PETSc-style) on an n1^3 cavity grid, `iters` iterations, timed; then the same right-hand
    side through the HIP path, compared with what was just timed."""
    from oracle import fluca_oracle as fo
    bc = [fo.BC_VELOCITY] * 4 + [fo.BC_SYMMETRY, fo.BC_VELOCITY]
    n = (n1,) * 3
    g = fo.Grid.uniform(n, [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
    t0 = time.perf_counter()
    S = g.assemble_S()
    t_asm = time.perf_counter() - t0
    rng = np.random.default_rng(20260313)
    p = rng.uniform(-1, 1, g.ncell)
    p -= p.mean()
    b = S.mult(p)
    del p
    xc, info = S.solve(b, rtol=0.0, atol=0.0, maxit=iters, history=False)
    its_per_s = info["iters"] / info["seconds"]
    parity = None
    try:
        from fluca_amd import poisson as flp
        Pc = flp.Poisson.uniform(n, [(0, 1), (0, 1), (0, 0.5)], bc, 1e-3)
        xg, ig = Pc.solve(torch.as_tensor(b, device="cuda"), rtol=0.0, atol=0.0, maxit=iters)
        xg = xg.cpu().numpy()
        xg -= xg.mean()
        xc = xc - xc.mean()
        parity = {"iters_gpu": ig["iters"], "iters_cpu": info["iters"], "rel_max_diff_x": float(np.abs(xg - xc).max() / np.abs(xc).max()),
                  "rnorm_gpu": ig["rnorm"], "rnorm_cpu": info["rnorm"]}
        Pc.close()
    except Exception as e:  # noqa: BLE001  (never lose the bench line over the cross-check)
        parity = {"error": repr(e)}
    return {"cells_per_axis": n1, "iters": info["iters"], "seconds": info["seconds"], "assemble_seconds": t_asm, "its_per_s": its_per_s, "parity": parity,
            "host_GBps_at_104B_per_row": 104.0 * n1 ** 3 * its_per_s / 1e9}


def cpu_baseline(sample_n, sample_iters, full_n, full_iters):
    """The oracle (CPU restatement: assembled CSR + PETSc-style PCG, OpenMP) timed on this box's host cores.  Where the
    host has the memory (>= 32 GB available for the 12 GB CSR + vectors) the headline's OWN grid is run for the headline's own number
    of iterations -- `value` is then a measurement at the metric's size -- and the bounded 256^3 sample is kept as a second entry;
    otherwise the sample, scaled by the cell count, is the value and says so."""
    fo = _oracle_threads()
    mem = _host_mem_available_gb()
    full_ok = bool(mem and mem >= 32.0) and full_n is not None
    sample = _cpu_leg(sample_n, sample_iters)
    sample_value = sample["its_per_s"] * (sample_n ** 3) / 512.0 ** 3
    out = {"unit": "512^3-equivalent PCG iterations/s", "cores": fo.num_threads(), "kind": "port", "host_mem_available_GB": mem, "full_grid_possible": full_ok,
           "petsc_cpu": petsc_cpu(sample_n, sample_iters, fo.num_threads()),
           "sample_256": {"value": sample_value, "parity_on_sample": sample["parity"], "seconds": sample["seconds"],
                          "sample": f"{sample_n}^3 cavity grid (1/{(512 // sample_n) ** 3} of the cells), {sample['iters']} Jacobi-PCG iterations, raw {sample['its_per_s']:.3f} it/s"},
           "parity_on_sample": sample["parity"]}
    full = None
    if full_ok:
        try:
            full = _cpu_leg(full_n, full_iters)
        except MemoryError as e:
            out["full_grid_error"] = repr(e)
    if full:
        out.update({"value": full["its_per_s"] * (full_n ** 3) / 512.0 ** 3, "parity_on_full_grid": full["parity"],
                    "sample": f"the headline's own grid: {full_n}^3 cavity, {full['iters']} Jacobi-PCG iterations on the assembled CSR (AIJ cost model), "
                              f"{full['seconds']:.2f} s (+ {full['assemble_seconds']:.1f} s assembling S, not timed)",
                    "host_GBps_at_104B_per_row": full["host_GBps_at_104B_per_row"]})
    else:
        out.update({"value": sample_value, "parity_on_full_grid": None,
                    "sample": out["sample_256"]["sample"] + ", scaled by the cell count (host memory too small for the 512^3 CSR)",
                    "host_GBps_at_104B_per_row": sample["host_GBps_at_104B_per_row"]})
    return out


def petsc_cpu(sample_n, iters, cores):
    """SURVEY 8(d), opportunistic true-reference timing: where a PETSc installation exists, compile oracle/petsc_cg_driver.c
    (the same S as AIJ, KSPCG + PCJACOBI, fixed iteration count) and run it on the host cores.  Returns a dict with the rate,
    or a string saying why there is no number (this image ships no PETSc)."""
    import re
    import shutil
    import subprocess
    import tempfile
    src = os.path.join(ROOT, "oracle", "petsc_cg_driver.c")
    flags = None
    pd, arch = os.environ.get("PETSC_DIR"), os.environ.get("PETSC_ARCH", "")
    for root in ([os.path.join(pd, arch), pd] if pd else []):
        if os.path.exists(os.path.join(root, "lib", "libpetsc.so")):
            incs = {os.path.join(pd, "include"), os.path.join(root, "include")}
            flags = [f"-I{i}" for i in sorted(incs)] + [f"-L{root}/lib", f"-Wl,-rpath,{root}/lib", "-lpetsc", "-lm"]
            break
    if flags is None and shutil.which("pkg-config"):
        for name in ("PETSc", "petsc"):
            if subprocess.run(["pkg-config", "--exists", name]).returncode == 0:
                flags = subprocess.run(["pkg-config", "--cflags", "--libs", name], capture_output=True, text=True).stdout.split() + ["-lm"]
                break
    if flags is None:
        return "unavailable (no PETSC_DIR with lib/libpetsc.so, no pkg-config PETSc)"
    cc = shutil.which("mpicc") or shutil.which("cc") or "gcc"
    try:
        with tempfile.TemporaryDirectory() as tmp:
            exe = os.path.join(tmp, "petsc_cg_driver")
            r = subprocess.run([cc, "-O2", src, "-o", exe] + flags, capture_output=True, text=True, timeout=300)
            if r.returncode != 0:
                return "unavailable (driver did not compile: " + r.stderr.strip().splitlines()[-1][:200] + ")"
            mpi = shutil.which("mpiexec") or shutil.which("mpirun")
            cmd = ([mpi, "-n", str(cores)] if mpi else []) + [exe, "-n", str(sample_n), "-its", str(iters), "-kappa", "1e-3"]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            m = re.search(r"FLUCA_PETSC n=(\d+) ranks=(\d+) its=(\d+) seconds=(\S+) rnorm=(\S+)", r.stdout)
            if not m:
                return "unavailable (driver failed: " + (r.stderr.strip().splitlines() or ["no output"])[-1][:200] + ")"
            its, sec = int(m.group(3)), float(m.group(4))
            return {"value": its / sec * (sample_n ** 3) / 512.0 ** 3, "unit": "512^3-equivalent PCG iterations/s", "kind": "reference library (PETSc KSPCG + PCJACOBI, AIJ)",
                    "ranks": int(m.group(2)), "sample": f"{sample_n}^3, {its} iterations, {sec:.2f} s"}
    except Exception as e:  # noqa: BLE001
        return f"unavailable ({e!r})"


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the last committed counter pass (profiles/pmc_<kernel>.json; counters cannot be read from inside
    the run) and whether that pass is STALE: it records the git blob hashes of the kernel's source files and the hipcc flags of the day it was
    taken (fluca_amd/provenance.py); a tree whose hashes differ is running another kernel than the one that was counted."""
    from fluca_amd import provenance
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{kernel}.json")))
    except Exception:  # noqa: BLE001
        return None, None
    return d.get("hbm_bytes_per_launch"), provenance.stale(kernel, d.get("sources_at_profiling"))


def workload_traffic(workload):
    """{kernel: (HBM bytes per launch, stale)} of the CG pair on another grid size than the headline's, from the launches of that size inside the last
    committed counter passes of the bench command (profiles/pmc_workload_<workload>.json); None where no such record exists."""
    from fluca_amd import provenance
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", f"pmc_workload_{workload}.json")))
    except Exception:  # noqa: BLE001
        return None
    return {k: (d[k], provenance.stale(k, d.get("sources_at_profiling", {}).get(k))) for k in ("k_cg_A", "k_cg_Bq") if k in d}


def _oracle_threads():
    from oracle import fluca_oracle as fo
    # a one-GPU box's CPU share is 16 cores; never spawn more OpenMP threads than that (or than the affinity mask)
    fo.set_num_threads(min(16, len(os.sched_getaffinity(0)), fo.num_threads()))
    return fo


def c3_parity(P, b, box, bc, steps=20):
    """BASELINE config 3 at its own size against the oracle (not against another HIP kernel): the oracle's assembled 512^3 channel S and its
    KSPCHEBYSHEV + PCJACOBI restatement, `steps` steps without a norm, on the right-hand side the timed sweep used; the HIP path runs the
    kernel the sweep ran (the fused two-step kernel) with its own default interval (separable Gershgorin bound x (0.1, 1.1))."""
    mem = _host_mem_available_gb()
    if not (mem and mem >= 32.0):
        return {"skipped": f"host memory {mem} GB < 32 GB for the 512^3 CSR"}
    fo = _oracle_threads()
    t0 = time.perf_counter()
    g = fo.Grid.uniform((512,) * 3, box, bc, 1e-3)
    S = g.assemble_S()
    lam = S.gershgorin(fo.PC_JACOBI)
    P.synchronize()
    bh = b.cpu().numpy()
    xo, io = S.solve(bh, ksp=fo.KSP_CHEBYSHEV, pc=fo.PC_JACOBI, norm=fo.NORM_NONE, nullspace=False, maxit=steps, emin=0.1 * lam, emax=1.1 * lam, history=False)
    xg, ig = P.solve(b, type=2, norm_type=3, remove_nullspace=0, maxit=steps, check_every=100, profile=1)
    P.synchronize()
    xg = xg.cpu().numpy()
    scale = float(np.abs(xo).max())
    return {"steps": steps, "steps_gpu": ig["iters"], "steps_cpu": io["iters"], "reason_gpu": ig["reason"], "reason_cpu": io["reason"],
            "fused_kernel": ig["kernel_launches"] * 2 == steps, "rel_max_diff_x": float(np.abs(xg - xo).max() / scale),
            "rel_l2_diff_x": float(np.linalg.norm(xg - xo) / np.linalg.norm(xo)), "oracle_gershgorin": lam, "oracle_seconds": io["seconds"],
            "seconds_total": time.perf_counter() - t0, "oracle": "assembled 512^3 CSR (oracle/fluca_oracle.c), KSPCHEBYSHEV + PCJACOBI restatement"}


def momentum_parity(n1=256, its=5):
    """The momentum block against the oracle's ASSEMBLED A (cnlinearcart3d.c:425-646, 873-1294 restated row by row) at 256^3 (50 M rows,
    650 M non-zeros): MatMult(A) through k_mom3 (state handed over with v0, the way NSStep does it), diag(A), and `its` Jacobi-BiCGStab
    iterations -- cavity boundary types, random V0 / v0, the bench's dt and viscosity scaled to this grid."""
    mem = _host_mem_available_gb()
    if not (mem and mem >= 40.0):
        return {"skipped": f"host memory {mem} GB < 40 GB for the assembled 256^3 momentum matrix"}
    fo = _oracle_threads()
    from fluca_amd import poisson as flp
    t0 = time.perf_counter()
    box = [(0.0, 1.0), (0.0, 1.0), (0.0, 0.5)]
    bc = [1, 1, 1, 1, 4, 1]
    n = (n1,) * 3
    g = fo.Grid.uniform(n, box, bc, 1e-3)
    rng = np.random.default_rng(11)
    V0 = [rng.uniform(-1, 1, g.nface[d]) for d in range(3)]
    v0 = rng.uniform(-1, 1, 3 * g.ncell)
    x = rng.uniform(-1, 1, 3 * g.ncell)
    hh = 1.0 / n1
    dt, rho, mu = 0.5 * hh, 1.0, 0.5 * hh
    W = g.apply_B(v0)
    A = g.assemble_momentum(1.0, dt, -0.5 * mu * dt / rho, V0, W)
    del W
    yo, do = A.mult(x), A.diag()
    xo, io = A.solve(x, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=0.0, atol=0.0, maxit=its, history=True)
    t_cpu = time.perf_counter() - t0
    P = flp.Poisson.uniform(n, box, bc, 1e-3)      # on its own stream, ordered against torch's current one by the wrapper
    M = flp.Momentum(P)
    dev = lambda a: torch.as_tensor(a, device="cuda")  # noqa: E731
    v0d = dev(v0)
    Wd = M.interp_faces(v0d, ends_only=True)            # only the block-end faces are read from the stored fields (fl_momentum_interp_faces_ends)
    M.set_state(dt, rho, mu, [dev(a) for a in V0], Wd, v0=v0d)
    xd = dev(x)
    yg, dg = M.apply(xd).cpu().numpy(), M.diagonal().cpu().numpy()
    xg, ig = M.solve(xd, rtol=0.0, atol=0.0, maxit=its, history=True)
    xg = xg.cpu().numpy()
    hist_g, hist_o = np.asarray(ig["history"][:its + 1]), np.asarray(io["history"][:its + 1])
    out = {"cells_per_axis": n1, "rows": int(A.nrow), "nnz": int(A.nnz), "kernel": "k_mom3 (fl_momentum_set_state_v0)",
           "rel_max_diff_apply": float(np.abs(yg - yo).max() / np.abs(yo).max()), "rel_max_diff_diag": float(np.abs(dg - do).max() / np.abs(do).max()),
           "bcgs_iters_gpu": ig["iters"], "bcgs_iters_cpu": io["iters"], "rel_max_diff_x": float(np.abs(xg - xo).max() / np.abs(xo).max()),
           "rel_max_diff_history": float(np.abs(hist_g - hist_o).max() / np.abs(hist_o).max()), "rnorm_gpu": ig["rnorm"], "rnorm_cpu": io["rnorm"],
           "oracle_seconds": t_cpu, "seconds_total": time.perf_counter() - t0,
           "oracle": "assembled A = I + dt C - (mu dt / 2 rho) L (oracle/fluca_oracle.c fo_assemble_momentum), KSPBCGS + PCJACOBI restatement"}
    M.close()
    P.close()
    return out


def c1_cpu_ksp():
    """BASELINE config 1 (fluca/tests/cavity_flow/cavity_flow_3d.c:39-42,72-77 at -cart_grid 64): the 64^3 lid-driven cavity's
    KSPSolve(kspS) as the reference runs it on the CPU -- KSPCG + PCJACOBI, PETSc defaults (rtol 1e-5, preconditioned norm, constant null space
    removed) -- here the oracle's restatement on the host cores (PETSc is absent), on the SURVEY 8(d) right-hand side; and the same solve on
    the GPU path beside it."""
    fo = _oracle_threads()
    from fluca_amd import poisson as flp
    n, box, bc = (64,) * 3, [(0.0, 1.0), (0.0, 1.0), (0.0, 0.5)], [1, 1, 1, 1, 4, 1]
    g = fo.Grid.uniform(n, box, bc, 1e-3)
    S = g.assemble_S()
    rng = np.random.default_rng(20260313)
    p = rng.uniform(-1, 1, g.ncell)
    p -= p.mean()
    b = S.mult(p)
    S.solve(b, history=False)                      # warm caches / thread team
    xo, io = S.solve(b, history=False)
    P = flp.Poisson.uniform(n, box, bc, 1e-3)
    bd = torch.as_tensor(b, device="cuda")
    P.solve(bd)
    xg, ig = P.solve(bd)
    xg = xg.cpu().numpy()
    P.close()
    return {"workload": "64^3 lid-driven cavity (cavity_flow_3d.c:39-42,72-77), KSPCG + PCJACOBI to PETSc's default rtol 1e-5 (preconditioned norm, constant null space), "
                        "b = S p* seeded; CPU = the oracle's restatement (PETSc absent from the image), not the reference binary",
            "metric": "KSPSolve seconds (CPU reference plumbing)", "value": io["seconds"], "higher_is_better": False, "cores": fo.num_threads(),
            "iterations": io["iters"], "reason": io["reason"], "its_per_s": io["iters"] / io["seconds"], "rnorm_over_rnorm0": io["rnorm"] / io["rnorm0"],
            "gpu": {"iterations": ig["iters"], "reason": ig["reason"], "seconds": ig["seconds"], "its_per_s": ig["iters"] / ig["seconds"],
                    "rel_l2_diff_x": float(np.linalg.norm((xg - xg.mean()) - (xo - xo.mean())) / np.linalg.norm(xo - xo.mean()))}}


def c5_block_parity(P, b, nb, box, bc, its=20):
    """One rank's 512 x 512 x 256 block of config 5 against the oracle's ASSEMBLED S of that block (as c3_parity does for config 3): `its`
    Jacobi-PCG iterations on the right-hand side the timed run used, no null space (the outlet)."""
    mem = _host_mem_available_gb()
    if not (mem and mem >= 24.0):
        return {"skipped": f"host memory {mem} GB < 24 GB for the 67 M-row CSR"}
    fo = _oracle_threads()
    t0 = time.perf_counter()
    g = fo.Grid.uniform(nb, box, bc, 1e-3)
    S = g.assemble_S()
    P.synchronize()
    bh = b.cpu().numpy()
    xo, io = S.solve(bh, nullspace=False, rtol=0.0, atol=0.0, maxit=its, history=True)
    xg, ig = P.solve(b, rtol=0.0, atol=0.0, maxit=its, remove_nullspace=0, history=True)
    P.synchronize()
    xg = xg.cpu().numpy()
    hg, ho = np.asarray(ig["history"][:its + 1]), np.asarray(io["history"][:its + 1])
    return {"iters_gpu": ig["iters"], "iters_cpu": io["iters"], "rel_max_diff_x": float(np.abs(xg - xo).max() / np.abs(xo).max()),
            "rel_max_diff_history": float(np.abs(hg - ho).max() / np.abs(ho).max()), "rnorm_gpu": ig["rnorm"], "rnorm_cpu": io["rnorm"],
            "oracle_seconds": io["seconds"], "seconds_total": time.perf_counter() - t0,
            "oracle": "assembled CSR of the 512 x 512 x 256 block (oracle/fluca_oracle.c), KSPCG + PCJACOBI restatement"}


def measured_stream_rates():
    import ctypes as C

    from fluca_amd import capi
    from fluca_amd import poisson as flp
    f = capi.lib.fldbg_bench
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]
    P = flp.Poisson.uniform((512,) * 3, [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
    src = torch.rand(P.ncell, dtype=torch.float64, device="cuda") - 0.5
    torch.cuda.synchronize()
    padded = 544 * 514 * 514
    res = {}
    first = True
    for key, mix in (("measured_copy_GBps", 11), ("measured_3r3w_GBps", 33)):
        best = None
        for blocks in (8192, 65536):
            ms, nbk = C.c_double(), C.c_int()
            capi.check(f(P.h, 3, mix, 81, blocks, 10, C.c_void_p(src.data_ptr()) if first else None, C.byref(ms), C.byref(nbk)), "fldbg_bench")
            first = False
            rate = 8.0 * (mix // 10 + mix % 10) * padded / ms.value / 1e6
            best = rate if best is None else max(best, rate)
        res[key] = best
    res["measured_copy_note"] = "library streaming kernels (16 B per lane, 8-fold unrolled, non-temporal), best of 8192 / 65536 blocks, 10 launches each; 544 x 514 x 514 padded doubles per stream"
    P.close()
    del src
    return res


def other_configs(stream, parity=True):
    """Driver-timed lines for the other single-GPU configurations of BASELINE.json, each with its own roofline object (same
    conventions as the headline: inputs resident, host clock around a synchronised region, HIP events for the dominant kernel)."""
    import ctypes as C

    from fluca_amd import capi
    from fluca_amd import poisson as flp
    box = [(0.0, 1.0), (0.0, 1.0), (0.0, 0.5)]
    cfg = {}

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t0

    def rhs(P, seed):
        gen = torch.Generator(device="cuda").manual_seed(seed)
        p = torch.rand(P.ncell, generator=gen, dtype=torch.float64, device="cuda") * 2 - 1
        stream.wait_stream(torch.cuda.current_stream())
        return P.apply(p)

    # C2: 256^3 lid-driven cavity, matrix-free Jacobi-PCG
    P = flp.Poisson.uniform((256,) * 3, box, [1, 1, 1, 1, 4, 1], 1e-3)
    P.set_stream(stream)
    b, x = rhs(P, 2), P.empty()
    kw = dict(rtol=0.0, atol=0.0, check_every=64)
    P.solve(b, x=x, maxit=50, **kw)
    K = 400
    (_, info), dt = timed(lambda: P.solve(b, x=x, maxit=K, profile=4, **kw))
    cfg["C2"] = {"workload": "256^3 lid-driven cavity, matrix-free Jacobi-PCG, fixed 400 iterations", "metric": "PCG iterations/s", "value": K / dt, "steps": K,
                 "ms_per_step": dt / K * 1e3, "iteration_algorithmic_GBps": B_ITER_ALGO * P.ncell * K / dt / 1e9,
                 "roofline": cg_roofline(info, P.ncell, 0, dt / K * 1e3, workload_traffic("c2_256"))}
    P.close()
    del b, x

    # C3: 512^3 channel (inlet VELOCITY, PRESSURE_OUTLET, walls in y, periodic span), Chebyshev-Jacobi, fixed 100-step sweeps
    P = flp.Poisson.uniform((512,) * 3, box, [1, 2, 1, 1, 3, 3], 1e-3)
    P.set_stream(stream)
    b, x = rhs(P, 3), P.empty()
    kw = dict(type=2, norm_type=3, remove_nullspace=0, check_every=100)
    P.solve(b, x=x, maxit=20, **kw)
    K = 100
    (_, info), dt = timed(lambda: P.solve(b, x=x, maxit=K, profile=1, **kw))
    fused = info["kernel_launches"] * 2 == K            # the fused kernel applies two steps per launch
    spl = 2 if fused else 1
    traffic, tstale = pmc_traffic("k_cheb2") if fused else (None, None)
    # floor of a launch: x, b, d read, x', d' written once = 40 B/cell whether it applies one step or two; textbook: 40 B/cell per STEP
    r3 = {"bound": "hbm", "kernel": "k_cheb2 (two fused Chebyshev-Jacobi steps per launch)" if fused else "k_cheb_st", "peak": HBM_PEAK_GBS, "unit": "GB/s",
          "traffic_stale": tstale, "frac_note": FRAC_NOTE, "algorithmic_bytes_per_cell_per_step": B_CHEB_ALGO, "floor_bytes_per_cell_per_launch": B_CHEB_ALGO,
          "steps_per_launch": spl, "launches_timed": info["kernel_launches"]}
    r3.update(phys(traffic, B_CHEB_ALGO * P.ncell, spl * B_CHEB_ALGO * P.ncell, info["kernel_ms"]))
    cfg["C3"] = {"workload": "512^3 channel [VELOCITY, PRESSURE_OUTLET, wall, wall, PERIODIC, PERIODIC], Chebyshev-Jacobi, KSP_NORM_NONE, fixed 100 steps",
                 "metric": "Chebyshev-Jacobi steps/s", "value": K / dt, "steps": K, "ms_per_step": dt / K * 1e3,
                 "step_algorithmic_GBps": B_CHEB_ALGO * P.ncell * K / dt / 1e9, "roofline": r3}
    if parity:
        try:
            cfg["C3"]["parity_on_full_grid"] = c3_parity(P, b, box, [1, 2, 1, 1, 3, 3])
        except Exception as e:  # noqa: BLE001  (never lose the bench line over the cross-check)
            cfg["C3"]["parity_on_full_grid"] = {"error": repr(e)}
    P.close()
    del b, x

    # C4: 512^3 with an immersed sphere D = 64 h (Fibonacci lattice, spacing ~ h): interpolation + spreading of three components
    n = 512
    P = flp.Poisson.uniform((n,) * 3, [(0, 1), (0, 1), (0, 1)], [1] * 6, 1e-3)
    P.set_stream(stream)
    h = 1.0 / n
    R = 32 * h
    L = int(round(4 * np.pi * R * R / (h * h)))
    i = np.arange(L) + 0.5
    phi, th = np.arccos(1 - 2 * i / L), np.pi * (1 + 5 ** 0.5) * i
    X = [torch.as_tensor(a, device="cuda") for a in (0.5 + R * np.cos(th) * np.sin(phi), 0.5 + R * np.sin(th) * np.sin(phi), 0.5 + R * np.cos(phi))]
    u = torch.rand(3 * P.ncell, dtype=torch.float64, device="cuda")
    F = torch.rand(3 * L, dtype=torch.float64, device="cuda")
    dV = torch.full((L,), h ** 3, dtype=torch.float64, device="cuda")
    U = torch.empty(3 * L, dtype=torch.float64, device="cuda")
    f = torch.zeros(3 * P.ncell, dtype=torch.float64, device="cuda")
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    m = C.c_void_p()
    stream.wait_stream(torch.cuda.current_stream())
    capi.check(capi.lib.fl_ibm_create(P.h, capi.DELTA_PESKIN4, L, ptr(X[0]), ptr(X[1]), ptr(X[2]), C.byref(m)), "fl_ibm_create")

    def ibm_step():  # stream-ordered like inside a time step: no host wait between steps, one at the end of the timed region
        capi.check(capi.lib.fl_ibm_interp(m, 3, ptr(u), ptr(U)))
        capi.check(capi.lib.fl_ibm_spread(m, 3, ptr(F), ptr(dV), ptr(f)))
    ibm_step()
    P.synchronize()
    K = 50
    _, dt = timed(lambda: ([ibm_step() for _ in range(K)], P.synchronize()))
    ach = 2 * IBM_B_PER_MARKER * L * K / dt / 1e9
    cfg["C4"] = {"workload": f"512^3 grid, immersed sphere D = 64 h, {L} markers (Peskin 4-point): interpolation + spreading of 3 components per step",
                 "metric": "IBM interpolate+spread steps/s", "value": K / dt, "steps": K, "ms_per_step": dt / K * 1e3, "markers": L,
                 "roofline": {"bound": "hbm", "kernel": "k_ibm_interp + k_ibm_spread", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
                              "traffic": None, "algorithmic_bytes_per_marker": IBM_B_PER_MARKER, "bytes_source": "SURVEY 8(d): 1584 B per marker and pass (no counter pass)",
                              "note": "latency-bound by construction (20 MB of traffic per step): reported, not a roofline target (SURVEY 8d)"}}
    capi.lib.fl_ibm_destroy(m)
    P.close()
    del u, f, F, U, dV, X

    # The momentum block (KSPSolve(kspA), abfpc.c:72): matrix-free A = I + dt C - (mu dt / 2 rho) L on the same 512^3 cavity
    # grid, random V0 and v0, v0interp = B v0 (fl_momentum_interp_faces) handed over with v0 (fl_momentum_set_state_v0: the operator forms the inner
    # face values itself), the operator kernel alone and the Jacobi-BiCGStab iteration around it
    try:
        P = flp.Poisson.uniform((512,) * 3, box, [1, 1, 1, 1, 4, 1], 1e-3)
        P.set_stream(stream)
        M = flp.Momentum(P)
        gen = torch.Generator(device="cuda").manual_seed(11)
        rnd = lambda m_: torch.rand(m_, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1  # noqa: E731
        V0 = [rnd(P.nface[d]) for d in range(3)]
        v0 = rnd(3 * P.ncell)
        hh = 1.0 / 512
        stream.wait_stream(torch.cuda.current_stream())
        W = M.interp_faces(v0)
        M.set_state(0.5 * hh, 1.0, 0.5 * hh, V0, W, v0=v0)
        del V0, W, v0
        v = rnd(3 * P.ncell)
        stream.wait_stream(torch.cuda.current_stream())
        fa = capi.lib.fldbg_mom_apply
        fa.restype = C.c_int
        fa.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]
        kms = {}
        for mode, name in ((0, "plain"), (2, "jacobi_dot_first"), (3, "jacobi_dots_second")):
            ms = C.c_double()
            capi.check(fa(M.h, mode, 10, C.byref(ms)), "fldbg_mom_apply")
            kms[name] = ms.value
        K = 10
        M.solve(v, rtol=0.0, atol=0.0, maxit=2)
        (_, info), dt = timed(lambda: M.solve(v, rtol=0.0, atol=0.0, maxit=K))
        # per cell: x 24 + y 24 + v0 24 + V0 24 (k_mom3; the nine stored v0interp fields, 72 B more, are read by k_mom2 only); a BiCGStab iteration = 2 products
        # (one also reads the shadow residual) + 3 vector updates
        B_APPLY, B_BCGS = 96, 552
        tr, trstale = pmc_traffic("k_mom3")
        rm = {"bound": "hbm", "kernel": "k_mom3 (MatMult(A): two cells per lane on 128 x 8 tiles, v0interp formed from v0 in the kernel)", "peak": HBM_PEAK_GBS, "unit": "GB/s",
              "traffic_stale": trstale, "frac_note": FRAC_NOTE, "algorithmic_bytes_per_cell": B_APPLY, "launches_timed": 10}
        rm.update(phys(tr, B_APPLY * P.ncell, B_APPLY * P.ncell, kms["plain"]))
        itg = B_BCGS * P.ncell * info["iters"] / dt / 1e9
        rm["iteration"] = {"algorithmic_bytes_per_cell": B_BCGS, "ms": dt / max(info["iters"], 1) * 1e3, "achieved": itg, "frac": itg / HBM_PEAK_GBS,
                           "bytes_source": "552 B/cell: what the iteration's kernels move by construction (2 products + 3 vector updates; no counter pass of the whole iteration)"}
        cfg["momentum"] = {"workload": "512^3 cavity grid, momentum block A = I + dt C - (mu dt / 2 rho) L matrix-free (3 velocity components; V0 on faces, v0interp = B v0 formed in the kernel), "
                                       f"Jacobi-BiCGStab (KSPBCGS + PCJACOBI) fixed {K} iterations",
                           "metric": "momentum BiCGStab iterations/s", "value": info["iters"] / dt, "steps": info["iters"], "ms_per_step": dt / max(info["iters"], 1) * 1e3,
                           "iteration_algorithmic_GBps": itg, "kernel_ms": kms, "roofline": rm}
        # the two Krylov methods to rtol 1e-5 on the viscous-dominated operator of the 512^3 flow configurations (nu dt / h^2 = 2.56; the state
        # above has 0.25): Jacobi-BiCGStab against KSPCHEBYSHEV fused into the product (k_mom3, OUT 4: 144 B/cell per step against 552 per iteration)
        try:
            M.set_coefficients(1.0, 0.5 * hh, -0.5 * 2.56 * hh * hh)
            # (the bench box is 1 x 1 x 0.5: h_z = h / 2, so the z direction carries four times that number -- a stiffer operator than the unit-cube
            # flow configurations, where tools/mom_bench.py counts 16 BiCGStab iterations against 27 Chebyshev steps)
            tts = {"nu_dt_over_h2": [2.56, 2.56, 10.24], "rtol": 1e-5, "gershgorin_radius": M.gershgorin(), "chebyshev_interval": list(M.chebyshev_interval())}
            for name, kw in (("bcgs_jacobi", dict(type=1)), ("chebyshev_jacobi", dict(type=2))):
                M.solve(v, rtol=1e-5, maxit=400, **kw)
                (_, si), dts = timed(lambda: M.solve(v, rtol=1e-5, maxit=400, **kw))
                tts[name] = {"iters": si["iters"], "reason": si["reason"], "seconds": dts, "ms_per_iter": dts / max(si["iters"], 1) * 1e3}
            cfg["momentum"]["time_to_solution"] = tts
        except Exception as e:  # noqa: BLE001
            cfg["momentum"]["time_to_solution"] = {"error": repr(e)}
        M.close()
        P.close()
        del v
        torch.cuda.empty_cache()
        if parity:
            try:
                cfg["momentum"]["parity"] = momentum_parity()
            except Exception as e:  # noqa: BLE001
                cfg["momentum"]["parity"] = {"error": repr(e)}
            torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        cfg["momentum"] = {"error": repr(e)}

    # Multigrid-preconditioned CG on the 512^3 cavity operator (FL_PC_MG: V(3,3) cycles, Chebyshev-Jacobi smoothing, tri-linear prolongation) to
    # rtol 1e-8 on a random mean-free solution: time to the answer, not a fixed iteration count
    try:
        P = flp.Poisson.uniform((512,) * 3, [(0, 1), (0, 1), (0, 0.5)], [1, 1, 1, 1, 4, 1], 1e-3)
        P.set_stream(stream)
        gen = torch.Generator(device="cuda").manual_seed(1)
        pm = torch.rand(P.ncell, dtype=torch.float64, device="cuda", generator=gen) * 2 - 1
        pm -= pm.mean()
        stream.wait_stream(torch.cuda.current_stream())
        bm = P.apply(pm)
        P.solve(bm, type=0, pc=2, rtol=1e-8, maxit=200)
        xm, im = P.solve(bm, type=0, pc=2, rtol=1e-8, maxit=200)
        stream.synchronize()
        err = float(torch.linalg.norm((xm - xm.mean()) - pm) / torch.linalg.norm(pm))
        cfg["multigrid"] = {"workload": "512^3 cavity operator, multigrid-preconditioned CG (FL_PC_MG, V(3,3), Chebyshev-Jacobi smoother, tri-linear prolongation) to rtol 1e-8, "
                                        "random mean-free solution", "metric": "seconds per solve", "value": im["seconds"], "higher_is_better": False, "iterations": im["iters"],
                            "reason": im["reason"], "rel_error_of_answer": err, "ms_per_iteration": im["seconds"] / max(im["iters"], 1) * 1e3}
        P.close()
        del pm, bm, xm
        torch.cuda.empty_cache()
    except Exception as e:  # noqa: BLE001
        cfg["multigrid"] = {"error": repr(e)}

    # C5 needs 8 GPUs (1024 x 1024 x 512 over 2 x 2 x 2).  What ONE rank of it does, rehearsed on this GPU without the halo exchange:
    # a 512 x 512 x 256 block with config 5's boundary types, the Jacobi-PCG iteration on it and the IBM kernels on the cylinder
    # (diameter 64 h along the periodic span; markers are replicated on every rank, so the full set of the 256-plane block is used)
    nb = (512, 512, 256)
    hc = 1.0 / 1024
    P = flp.Poisson.uniform(nb, [(0, 0.5), (0, 0.5), (0, 0.25)], [1, 2, 1, 1, 3, 3], 1e-3)
    P.set_stream(stream)
    b, x = rhs(P, 5), P.empty()
    kw = dict(rtol=0.0, atol=0.0, check_every=64, remove_nullspace=0)
    P.solve(b, x=x, maxit=10, **kw)
    K = 40
    (_, info), dt = timed(lambda: P.solve(b, x=x, maxit=K, profile=1, **kw))
    Rc = 32 * hc
    nth = int(round(2 * np.pi * Rc / hc))
    th = (np.arange(nth) + 0.5) * 2 * np.pi / nth
    zc = (np.arange(nb[2]) + 0.5) * hc
    Xc = [torch.as_tensor(a, device="cuda") for a in (np.tile(0.25 + Rc * np.cos(th), nb[2]), np.tile(0.25 + Rc * np.sin(th), nb[2]), np.repeat(zc, nth))]
    Lc = nb[2] * nth
    uc = torch.rand(3 * P.ncell, dtype=torch.float64, device="cuda")
    Fc = torch.rand(3 * Lc, dtype=torch.float64, device="cuda")
    dVc = torch.full((Lc,), hc ** 3, dtype=torch.float64, device="cuda")
    Uc = torch.empty(3 * Lc, dtype=torch.float64, device="cuda")
    fc = torch.zeros(3 * P.ncell, dtype=torch.float64, device="cuda")
    mc = C.c_void_p()
    stream.wait_stream(torch.cuda.current_stream())
    capi.check(capi.lib.fl_ibm_create(P.h, capi.DELTA_PESKIN4, Lc, ptr(Xc[0]), ptr(Xc[1]), ptr(Xc[2]), C.byref(mc)), "fl_ibm_create")

    def cyl_step():
        capi.check(capi.lib.fl_ibm_interp(mc, 3, ptr(uc), ptr(Uc)))
        capi.check(capi.lib.fl_ibm_spread(mc, 3, ptr(Fc), ptr(dVc), ptr(fc)))
    cyl_step()
    P.synchronize()
    _, dti = timed(lambda: ([cyl_step() for _ in range(20)], P.synchronize()))
    cfg["C5_rank_rehearsal"] = {
        "workload": "ONE rank's share of config 5 on one GPU, no halo exchange: 512x512x256 block [VELOCITY, PRESSURE_OUTLET, wall, wall, PERIODIC, PERIODIC], "
                    f"Jacobi-PCG fixed {K} iterations; immersed cylinder D = 64 h along the span, {Lc} markers",
        "metric": "PCG iterations/s on the block", "value": K / dt, "steps": K, "ms_per_step": dt / K * 1e3,
        "ibm_interp_plus_spread_ms": dti / 20 * 1e3, "markers": Lc,
        "roofline": cg_roofline(info, P.ncell, 0, dt / K * 1e3, workload_traffic("c5_block"))}
    if parity:
        try:
            cfg["C5_rank_rehearsal"]["parity"] = c5_block_parity(P, b, nb, [(0, 0.5), (0, 0.5), (0, 0.25)], [1, 2, 1, 1, 3, 3])
        except Exception as e:  # noqa: BLE001
            cfg["C5_rank_rehearsal"]["parity"] = {"error": repr(e)}
    cfg["C5_rank_rehearsal"]["rank_grid_parity"] = ("the 2 x 2 x 2 rank grid itself (halo exchange over three split axes, fused smoother, multigrid, IBM cylinder, "
                                                    "whole time steps) runs against the single-domain oracle in tests/test_gpu_config5.py: eight handles on eight "
                                                    "host threads of one process, in-memory transport")
    capi.lib.fl_ibm_destroy(mc)
    P.close()
    del b, x
    torch.cuda.empty_cache()
    if parity:
        try:
            cfg["C1"] = c1_cpu_ksp()
        except Exception as e:  # noqa: BLE001
            cfg["C1"] = {"error": repr(e)}

    # A whole time step of the reference's integrator on the C host mirror (examples/flow_configs.c, a child process: it opens the GPU itself):
    # 512^3 channel with the immersed sphere of config 4, fractional step (-ns_ksp_type preonly: one PCApply_ABF per step -- BiCGStab + Jacobi on the
    # momentum block, multigrid-CG on the Schur complement, IBM interpolation and spreading).  Wall time of the later steps (the first one allocates).
    try:
        import re
        import subprocess
        from fluca_amd import build as flbuild
        exe = flbuild.build_example(name="flow_configs")
        # `value` = the best of steps 2..4 (the impulsive start: what rounds 3 and 4 quoted); `last_step_seconds` = step FLOW_STEPS, where the flow
        # has settled and a starting guess from the previous step is worth something
        FLOW_STEPS = 24
        def flow(extra):
            out = subprocess.run([exe, "-config", "sphere", "-n", "512", "-ns_max_steps", str(FLOW_STEPS), "-ns_ksp_type", "preonly", "-ns_abf_schur_pc_type", "mg"] + extra, capture_output=True, text=True, timeout=300)
            steps = re.findall(r"step\s+(\d+)\s+wall\s+(\S+) s\s+outer its\s+(\d+)\s+kspA its\s+(\d+)\s+kspS its\s+(\d+)", out.stdout)
            if out.returncode != 0 or len(steps) < 3:
                raise RuntimeError((out.stdout + out.stderr)[-400:])
            later = [float(w) for _, w, _, _, _ in steps[1:]]
            return {"value": min(later[:3]), "last_step_seconds": later[-1], "steps_timed": len(later), "seconds_per_step": later, "first_step_seconds": float(steps[0][1]),
                    "kspA_its": [int(a) for _, _, _, a, _ in steps], "kspS_its": [int(s_) for _, _, _, _, s_ in steps]}
        base = flow([])
        cfg["flow_step"] = {"workload": "512^3 channel + immersed sphere (12 868 markers), one CNLinear time step as a fractional step (PCApply_ABF: BiCGStab + Jacobi on A, "
                                        "multigrid-CG on S, IBM direct forcing) on the C host mirror, child process", "metric": "seconds per time step", "higher_is_better": False,
                            "cells": 512 ** 3, **base}
        try:  # the same step with -ns_abf_momentum_ksp_type chebyshev (KSPCHEBYSHEV fused into the momentum product, interval from the Gershgorin disc)
            cfg["flow_step"]["with_momentum_chebyshev"] = {"options": "-ns_abf_momentum_ksp_type chebyshev (default interval)", **flow(["-ns_abf_momentum_ksp_type", "chebyshev"])}
        except Exception as e:  # noqa: BLE001
            cfg["flow_step"]["with_momentum_chebyshev"] = {"error": repr(e)[:300]}
        # ... and with the first PCApply_ABF of a step starting kspA from the previous velocity (-ns_abf_momentum_guess_previous: KSPSetInitialGuessNonzero
        # semantics, the same convergence test against || M momrhs ||; round 5), for both Krylov types
        for key, extra in (("with_guess_previous", ["-ns_abf_momentum_guess_previous"]),
                           ("with_momentum_chebyshev_and_guess_previous", ["-ns_abf_momentum_ksp_type", "chebyshev", "-ns_abf_momentum_guess_previous"])):
            try:
                cfg["flow_step"][key] = {"options": " ".join(extra), **flow(extra)}
            except Exception as e:  # noqa: BLE001
                cfg["flow_step"][key] = {"error": repr(e)[:300]}
    except Exception as e:  # noqa: BLE001
        cfg["flow_step"] = {"error": repr(e)[:500]}
    if parity and isinstance(cfg.get("flow_step"), dict):
        # velocity-field parity of whole steps (north_star: "residual and velocity field"): the same set-up at 128^3, two converged steps through the
        # C host mirror against the oracle's step (tests/flow_parity.py; the 512^3 step above is a fractional step, this one iterates to 1e-9)
        try:
            from tests import flow_parity
            cfg["flow_step"]["parity"] = flow_parity.channel_sphere(n=128, nsteps=2)
        except Exception as e:  # noqa: BLE001
            cfg["flow_step"]["parity"] = {"error": repr(e)[:500]}
    return cfg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--cells", type=int, default=512, help="cells per axis per GPU")
    ap.add_argument("--variant", type=int, default=0)
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, what the driver runs): --cells^3 per GPU; strong: ONE --cells^3 grid split over the GPUs (the other "
                         "reading of BASELINE.json's '512^3 grid at 1/2/4/8 GPUs': 256^3 per GPU at N = 8, where the two all-reduces of an "
                         "iteration weigh as much as its kernels)")
    ap.add_argument("--cpu-cells", type=int, default=256)
    ap.add_argument("--cpu-iters", type=int, default=500, help="iterations of the CPU sample (about 10 s on 16 cores at 256^3)")
    ap.add_argument("--skip-cpu", action="store_true")
    ap.add_argument("--skip-extras", action="store_true", help="skip the time-to-solution comparison after the timed region")
    ap.add_argument("--placement", choices=["auto", "off"], default="auto",
                    help="auto: the library carves the solver vectors out of one arena where a probe of k_cg_A runs fastest (its "
                         "default); off: one plain allocation per vector")
    ap.add_argument("--placement-tries", type=int, default=None, help="(accepted for compatibility, ignored: placement is no longer a lottery)")
    ap.add_argument("--skip-configs", action="store_true", help="skip the C2 / C3 / C4 lines after the timed region")
    ap.add_argument("--transport", choices=["rccl", "host"], default="rccl",
                    help="halo transport for N > 1: RCCL Send/Recv (production) or the host-staged gloo callbacks "
                         "(rehearsal on a box with fewer GPUs than ranks; never used for a reported number)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    assert args.gpus in RANK_GRIDS, "supported: 1, 2, 4, 8 GPUs"
    ndev = torch.cuda.device_count()
    if args.transport == "rccl":
        assert world <= ndev, f"{world} ranks need {world} GPUs for the RCCL transport (found {ndev})"
    local = local % max(ndev, 1)
    torch.cuda.set_device(local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)   # control plane only; data path = RCCL inside the library

    from fluca_amd import capi
    from fluca_amd import poisson as flp
    from fluca_amd.capi import BC_SYMMETRY, BC_VELOCITY
    capi.check(capi.lib.fl_tuning_set(b"placement", 1 if args.placement == "auto" else 0), "fl_tuning_set")

    ranks = RANK_GRIDS[args.gpus]
    n = tuple(args.cells * r for r in ranks) if args.scaling == "weak" else (args.cells,) * 3
    assert all(n[d] % ranks[d] == 0 for d in range(3)), "--scaling strong: --cells must be divisible by the rank grid"
    bc = [BC_VELOCITY] * 4 + [BC_SYMMETRY, BC_VELOCITY]
    box = [(0.0, 1.0), (0.0, 1.0), (0.0, 0.5)]
    dec = flp.default_decomp(n, ranks, rank) if world > 1 else None
    P = flp.Poisson.uniform(n, box, bc, 1e-3, decomp=dec, device=local)
    transport = args.transport
    if world > 1 and transport == "rccl":
        # rank 0 creates the ncclUniqueId, the control plane (gloo) broadcasts it, every rank joins the RCCL communicator.
        # No fallback: a run that cannot use the production wire fails on every rank.
        ok, why = 1, ""
        try:
            idb = [flp.rccl_unique_id() if rank == 0 else None]
        except Exception as e:  # noqa: BLE001
            idb, ok, why = [None], 0, f"RCCL unique id failed: {e}"
        dist.broadcast_object_list(idb, src=0)
        if idb[0] is None:
            ok = 0
        if ok:
            try:
                P.comm_init_rccl(idb[0], rank, world)
            except Exception as e:  # noqa: BLE001
                ok, why = 0, f"RCCL init failed: {e}"
        t_ok = torch.tensor([ok], dtype=torch.int32)
        dist.all_reduce(t_ok, op=dist.ReduceOp.MIN)
        if int(t_ok[0]) == 0:
            print(f"[bench] rank {rank}: RCCL transport unavailable ({why or 'another rank failed'}); refusing to fall back to the "
                  f"host-staged transport (ask for it with --transport host)", file=sys.stderr, flush=True)
            dist.destroy_process_group()
            sys.exit(3)
    if world > 1 and transport == "host":
        from fluca_amd import hostcomm
        P.comm_init_host(hostcomm.gloo_exchange, hostcomm.gloo_allreduce, rank, world)
    stream = torch.cuda.Stream()
    P.set_stream(stream)

    gen = torch.Generator(device="cuda").manual_seed(20260313 + rank)
    pstar = torch.rand(P.ncell, generator=gen, dtype=torch.float64, device="cuda") * 2 - 1
    stream.wait_stream(torch.cuda.current_stream())
    b = P.apply(pstar)           # b = S p*  (consistent by construction; exercises the halo exchange when N > 1)
    x = P.empty()
    P.synchronize()

    def run(iters, profile):
        # profile = 2: the kernels of every second PAIR of iterations are bracketed by HIP events (every pair costs 1.3 % of the rate)
        return P.solve(b, x=x, rtol=0.0, atol=0.0, maxit=iters, variant=args.variant, profile=2 if profile else 0, check_every=64)[1]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if args.warmup > 0:
        run(args.warmup, False)      # the first solve places the vectors (one-off, outside the timed region)
    # placement is opt-in (tuning knob "placement" / fl_poisson_tune_placement): the bench asks for it unless told not to
    probe = P.tune_placement() if args.placement == "auto" else (0.0, 0.0)
    barrier()
    t0 = time.perf_counter()
    info = run(args.steps, True)
    P.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t[0])
    assert info["iters"] == args.steps, info
    per_rank, wire = None, None
    if world > 1:
        per_rank = [None] * world
        dist.all_gather_object(per_rank, {"rank": rank, "device": local, "k_cg_A_ms": info["kernel_ms"], "k_cg_Bq_ms": info["kernel2_ms"], "seconds_device": info["seconds"],
                                          "placement_probe_ms": list(probe), "comm": P.comm_info()})
        # the line checks itself: every rank's communicator must BE the transport that was asked for, span the whole job (as RCCL itself
        # reports it: ncclCommCount / ncclCommUserRank), and the ranks must be all different -- a curve over anything else is worthless
        comms = [r["comm"] for r in per_rank]
        want = 1 if transport == "rccl" else 2
        assert all(c["transport"] == want for c in comms), f"transport mismatch: {comms}"
        assert all(c["nranks"] == world for c in comms) and sorted(c["rank"] for c in comms) == list(range(world)), f"communicator does not span the job: {comms}"
        assert all(c["loopback"] == 0 for c in comms), "FLUCA_COMM_LOOPBACK is set: a rank would be talking to itself"
        ka = [r["k_cg_A_ms"] for r in per_rank]
        kb = [r["k_cg_Bq_ms"] for r in per_rank]
        wire = {"transport": "rccl" if want == 1 else "host", "nranks_reported_by_the_communicator": comms[0]["nranks"], "ranks_reported": sorted(c["rank"] for c in comms),
                "neighbours_per_rank": [c["neighbours"] for c in comms], "messages_per_exchange_per_rank": [c["messages"] for c in comms],
                # one ghost exchange of r per iteration (ghost p is recomputed from ghost r), two all-reduces of 8 doubles
                "halo_bytes_per_iter": int(sum(c["halo_bytes"] for c in comms)), "halo_bytes_per_iter_per_rank": [int(c["halo_bytes"]) for c in comms],
                "allreduce_bytes_per_iter_per_rank": 2 * 8 * 8,
                "k_cg_A_ms_spread": {"min": min(ka), "max": max(ka), "max_over_min": max(ka) / min(ka) if min(ka) > 0 else None},
                "k_cg_Bq_ms_spread": {"min": min(kb), "max": max(kb), "max_over_min": max(kb) / min(kb) if min(kb) > 0 else None}}

    cells_job = float(P.ncell) * world
    value = cells_job / 512.0 ** 3 * args.steps / dt
    # HBM traffic per 512^3 launch from rocprofv3 --pmc passes over this command (profiles/README.md): counters cannot be read from
    # inside the run, so the last committed pass is quoted (null when the workload is not the one that was profiled)
    traffic = None
    if args.cells == 512 and args.variant == 0:
        traffic = {kname: pmc_traffic(kname) for kname in ("k_cg_A", "k_cg_Bq")}
    out = {
        "metric": "pressure-Poisson Jacobi-PCG iterations/s, 512^3 cells per GPU",
        "value": value, "unit": "512^3-equivalent PCG iterations/s", "n_gpus": args.gpus, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{n[0]}x{n[1]}x{n[2]} lid-driven-cavity Schur complement S=-kappa*D*Gst (7-pt, Neumann), "
                               f"matrix-free Jacobi-PCG with constant-null-space removal, b=S*p* seeded, fixed {args.steps} iterations",
                   "cells_per_gpu": int(P.ncell), "rank_grid": list(ranks), "variant": {0: "fused, q = S p formed twice and never stored (k_cg_A + k_cg_Bq)", 1: "unfused (one kernel per step)",
                                                                          2: "fused, q stored (k_cg_A + k_cg_B)"}.get(args.variant, str(args.variant)),
                   "halo": ("RCCL Send/Recv" if transport == "rccl" else "host-staged gloo (NOT the production transport)") if world > 1 else "none"},
        "iteration_algorithmic_GBps_per_gpu": B_ITER_ALGO * P.ncell * args.steps / dt / 1e9,
        "solve_seconds_device": info["seconds"],
        "ranks": per_rank, "transport_ranks": (len(set(wire["ranks_reported"])) if wire else None), "wire": wire,
        "placement": {"mode": args.placement, "probe_ms_all_vectors_in_one_block": probe[0], "probe_ms_chosen": probe[1],
                      "note": "probe = one k_cg_A + one odd-iteration k_cg_Bq; the search arena is given back, the handle keeps five vectors (include/fluca_hip.h fl_poisson_tune_placement)"},
        "hbm_bytes_held": P.vector_bytes(),
        "roofline": cg_roofline(info, P.ncell, args.variant, dt / args.steps * 1e3, traffic),
    }
    if world == 1 and not args.skip_extras:
        # not part of the metric: what the iteration rate buys -- time to a converged pressure with the Jacobi preconditioner of
        # the metric and with the multigrid preconditioner (FL_PC_MG), same right-hand side, outside the timed region
        try:
            tts = {"rtol": 1e-8, "norm": "preconditioned (each solver's own)"}
            for name, kw in (("jacobi_pcg", dict(pc=1, maxit=20000)), ("multigrid_pcg", dict(pc=2, maxit=200))):
                xs, si = P.solve(b, rtol=1e-8, **kw)
                err = float(torch.linalg.norm((xs - xs.mean()) - (pstar - pstar.mean())) / torch.linalg.norm(pstar - pstar.mean()))
                tts[name] = {"iters": si["iters"], "seconds": si["seconds"], "reason": si["reason"], "rel_error_vs_p_star": err}
            out["time_to_solution"] = tts
        except Exception as e:  # noqa: BLE001
            out["time_to_solution"] = {"error": repr(e)}
    if world == 1 and not args.skip_extras and args.placement == "auto":
        # the same K steps on plainly allocated vectors (what the library did before the placement step existed)
        try:
            capi.check(capi.lib.fl_tuning_set(b"placement", 0))
            Pu = flp.Poisson.uniform(n, box, bc, 1e-3, device=local)
            Pu.set_stream(stream)
            xu = Pu.empty()
            kw = dict(rtol=0.0, atol=0.0, variant=args.variant, check_every=64)
            Pu.solve(b, x=xu, maxit=max(args.warmup, 1), **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            iu = Pu.solve(b, x=xu, maxit=args.steps, profile=1, **kw)[1]
            Pu.synchronize()
            torch.cuda.synchronize()
            du = time.perf_counter() - t0
            out["value_unplaced"] = cells_job / 512.0 ** 3 * args.steps / du
            out["unplaced_k_cg_A_ms"] = iu["kernel_ms"]
            out["unplaced_k_cg_Bq_ms"] = iu["kernel2_ms"]
            Pu.close()
            del xu
        except Exception as e:  # noqa: BLE001
            out["value_unplaced"] = {"error": repr(e)}
        finally:
            capi.check(capi.lib.fl_tuning_set(b"placement", 1))
    if world == 1 and not args.skip_configs:
        P.close()
        del b, x, pstar
        torch.cuda.empty_cache()
        try:
            out["configs"] = other_configs(stream, parity=not args.skip_cpu)
        except Exception as e:  # noqa: BLE001
            out["configs"] = {"error": repr(e)}
    if world == 1 and not args.skip_extras:
        # the attainable streaming rate on THIS box, reported beside the nominal peak: the library's own streaming kernels
        # (16 B per lane, non-temporal, grid-stride; tools/sbench.py) on 512^3-sized padded vectors -- a copy (1 read + 1 write stream)
        # and the 3 reads + 3 writes mix of the CG kernels
        try:
            out["roofline"].update(measured_stream_rates())
        except Exception as e:  # noqa: BLE001
            out["roofline"]["measured_copy_GBps"] = None
            out["roofline"]["measured_copy_error"] = repr(e)
    if rank == 0 and world == 1 and not args.skip_cpu:
        out["cpu_baseline"] = cpu_baseline(args.cpu_cells, args.cpu_iters, args.cells if args.cells >= 512 else None, args.steps)
    elif rank == 0:
        out["cpu_baseline"] = None
    if P.h:
        P.close()
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

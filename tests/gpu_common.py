"""Shared helpers of the -m gpu parity tests: build the same problem for the HIP path and for the CPU oracle."""
import numpy as np
import torch

from oracle import fluca_oracle as fo

V, O, PER, SYM = fo.BC_VELOCITY, fo.BC_PRESSURE_OUTLET, fo.BC_PERIODIC, fo.BC_SYMMETRY
CAVITY = [V, V, V, V, SYM, V]          # fluca/tests/cavity_flow/cavity_flow_3d.c:72-77
CAVITY_BOX = [(0.0, 1.0), (0.0, 1.0), (0.0, 0.5)]   # cavity_flow_3d.c:42


def stretched(n, lo, hi, beta=1.3):
    s = np.linspace(0.0, 1.0, n + 1)
    return lo + (hi - lo) * (np.tanh(beta * (2 * s - 1)) / np.tanh(beta) + 1) / 2


def make_pair(n, bc, kappa=1e-3, box=CAVITY_BOX, nonuniform=False):
    """-> (fluca_amd.Poisson, oracle Grid) on the same grid / BCs / kappa"""
    from fluca_amd.poisson import Poisson
    if nonuniform:
        xf = [stretched(n[d], box[d][0], box[d][1], 1.1 + 0.2 * d) for d in range(3)]
        return Poisson(n, xf, bc, kappa), fo.Grid(n, xf, bc, kappa)
    return Poisson.uniform(n, box, bc, kappa), fo.Grid.uniform(n, box, bc, kappa)


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")


def host(t):
    return t.detach().cpu().numpy()


def mean_free_rhs(S, ncell, seed=20260313):
    """SURVEY 8d micro-benchmark RHS: b = S p*, p* seeded uniform(-1,1) made mean-free."""
    rng = np.random.default_rng(seed)
    p = rng.uniform(-1.0, 1.0, ncell)
    p -= p.mean()
    return p, S.mult(p)


def kbench_build():
    """True when the loaded libflucahip.so is a -DFL_KBENCH_VARIANTS build (superseded solver variants compiled in, fluca_amd/csrc/fl_knobs.h)"""
    from fluca_amd import capi
    return b"+kbench" in capi.lib.fl_version()


def variants(*vs):
    """the solver variants a test may ask for: all of them in a kbench build, the shipped one (0) in the product"""
    return [v for v in vs if v == 0 or kbench_build()]

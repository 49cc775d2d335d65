"""The committed fixtures of the widened rows (tests/golden/momentum_fixtures.json, written by gen_momentum_fixtures.py from
the CPU oracle): CPU test = the oracle still reproduces them; -m gpu test = the HIP path hits the same numbers."""
import ctypes as C
import importlib.util
import json
import os

import numpy as np
import pytest

from oracle import fluca_oracle as fo

HERE = os.path.dirname(os.path.abspath(__file__))
FIX = json.load(open(os.path.join(HERE, "golden", "momentum_fixtures.json")))
_spec = importlib.util.spec_from_file_location("gen_momentum_fixtures", os.path.join(HERE, "golden", "gen_momentum_fixtures.py"))
gen = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(gen)


def _problem(fx):
    g = fo.Grid.uniform(tuple(fx["n"]), gen.BOX, fx["bc"], fx["dt"] / fx["rho"])
    return g, gen.fields(g)


@pytest.mark.parametrize("fx", FIX, ids=lambda f: "x".join(map(str, f["n"])))
def test_oracle_reproduces_momentum_fixture(fx):
    g, (V0, W, v, b) = _problem(fx)
    A = g.assemble_momentum(1.0, fx["dt"], -0.5 * fx["mu"] * fx["dt"] / fx["rho"], V0, W)
    y = A.mult(v)
    assert np.linalg.norm(y) == pytest.approx(fx["apply"]["norm2"], rel=1e-12)
    assert y.sum() == pytest.approx(fx["apply"]["sum"], rel=1e-9, abs=1e-9 * fx["apply"]["absmax"])
    idx = (0, 7, len(y) // 3, len(y) // 2 + 5, len(y) - 1)
    assert np.allclose([y[i] for i in idx], fx["apply"]["samples"], rtol=1e-12)
    assert A.diag().sum() == pytest.approx(fx["diag"]["sum"], rel=1e-12)
    x, info = A.solve(b, ksp=fo.KSP_BCGS, pc=fo.PC_JACOBI, nullspace=False, rtol=1e-10, maxit=200)
    assert info["iters"] == fx["bcgs"]["iters"] and info["reason"] == fx["bcgs"]["reason"]
    assert np.allclose(info["history"][:3], fx["bcgs"]["history"][:3], rtol=1e-9)
    mg = fo.MgOracle(g)
    assert mg.nlevels == fx["mg"]["levels"] and np.allclose(mg.bounds, fx["mg"]["bounds"], rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("fx", FIX, ids=lambda f: "x".join(map(str, f["n"])))
def test_hip_path_hits_momentum_fixture(fx):
    import torch
    from fluca_amd import capi
    from fluca_amd.poisson import Momentum, Poisson
    g, (V0, W, v, b) = _problem(fx)
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device="cuda")
    P = Poisson.uniform(tuple(fx["n"]), gen.BOX, fx["bc"], fx["dt"] / fx["rho"])
    M = Momentum(P)
    M.set_state(fx["dt"], fx["rho"], fx["mu"], [dev(a) for a in V0], [dev(a) for a in W])
    y = M.apply(dev(v)).cpu().numpy()
    assert np.linalg.norm(y) == pytest.approx(fx["apply"]["norm2"], rel=1e-12)
    idx = (0, 7, len(y) // 3, len(y) // 2 + 5, len(y) - 1)
    assert np.allclose([y[i] for i in idx], fx["apply"]["samples"], rtol=1e-11, atol=1e-13 * fx["apply"]["absmax"])
    d = M.diagonal().cpu().numpy()
    assert d.sum() == pytest.approx(fx["diag"]["sum"], rel=1e-12) and d.min() == pytest.approx(fx["diag"]["min"], rel=1e-12)
    x, info = M.solve(dev(b), history=True, rtol=1e-10, maxit=200)
    assert info["reason"] == fx["bcgs"]["reason"] and abs(info["iters"] - fx["bcgs"]["iters"]) <= 1
    m = min(len(info["history"]), len(fx["bcgs"]["history"]), 4)
    assert np.allclose(info["history"][:m], fx["bcgs"]["history"][:m], rtol=1e-6)
    assert float(x.norm()) == pytest.approx(fx["bcgs"]["x_norm2"], rel=1e-8)
    # multigrid-PCG: same hierarchy, same eigenvalue bounds per level (queried from the product), same history
    lam = C.c_double()
    capi.check(capi.lib.fl_poisson_gershgorin(P.h, capi.PC_JACOBI, C.byref(lam)))
    assert lam.value == pytest.approx(fx["mg"]["bounds"][0], rel=1e-12)      # uniform grid: the separable bound is the row-wise one
    xc = [0.5 * (a[1:] + a[:-1]) for a in g.xf]
    Z, Y, X = np.meshgrid(xc[2], xc[1], xc[0], indexing="ij")
    p = (np.cos(np.pi * X) * np.cos(np.pi * Y) * np.cos(2 * np.pi * Z)).ravel()
    p -= p.mean()
    bs = P.apply(dev(p))
    try:
        for knob, key in ((0, "mg"), (1, "mg_linear")):       # piecewise-constant, then the default tri-linear prolongation
            capi.check(capi.lib.fl_tuning_set(b"mg_prolong", knob))
            xm, im = P.solve(bs, history=True, type=0, pc=2, rtol=1e-8, maxit=50)
            assert im["reason"] == fx[key]["reason"] and abs(im["iters"] - fx[key]["iters"]) <= 1
            m = min(len(im["history"]), len(fx[key]["history"]))
            assert np.allclose(im["history"][:3], fx[key]["history"][:3], rtol=1e-6)
            assert np.allclose(im["history"][:m], fx[key]["history"][:m], rtol=5e-2)
            xm = xm.cpu().numpy()
            assert np.abs((xm - xm.mean()) - p).max() <= 10 * fx[key]["err_inf"] + 1e-9
    finally:
        capi.check(capi.lib.fl_tuning_set(b"mg_prolong", 1))
    M.close()
    P.close()

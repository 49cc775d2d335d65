"""-m gpu: KSPCG with -ksp_cg_single_reduction (fl_ksp_opts.cg_single_reduction; the reference reaches it as
-ns_abf_schur_ksp_cg_single_reduction through the prefix built at fluca/src/ns/utils/abfpc/abfpc.c:206) against the oracle's
restatement of the same rearrangement, and against the default two-reduction CG on the same handle."""
import numpy as np
import pytest

from oracle import fluca_oracle as fo
from tests.gpu_common import CAVITY, PER, SYM, V, dev, host, make_pair, mean_free_rhs

pytestmark = pytest.mark.gpu

GRIDS = [
    ((17, 9, 11), CAVITY, False),
    ((12, 10, 9), [PER] * 6, False),
    ((130, 37, 20), CAVITY, False),                       # two tiles in x, ragged
    ((136, 70, 12), [PER, PER, V, V, PER, PER], False),   # a periodic seam between different tiles
    ((64, 48, 40), [V, V, SYM, SYM, PER, PER], False),    # several z chunks
    ((2, 2, 2), [PER] * 6, False),
]


@pytest.mark.parametrize("n,bc,nonuni", GRIDS)
@pytest.mark.parametrize("pc", [fo.PC_JACOBI, fo.PC_NONE])
def test_single_reduction_cg_matches_oracle(n, bc, nonuni, pc):
    P, g = make_pair(n, bc, kappa=1e-3, nonuniform=nonuni)
    S = g.assemble_S()
    _, b = mean_free_rhs(S, g.ncell)
    xo, io = S.solve(b, pc=pc, rtol=1e-8, maxit=2000, single_reduction=True)
    xg, ig = P.solve(dev(b), pc=pc, rtol=1e-8, maxit=2000, history=True, cg_single_reduction=1, check_every=7)
    assert ig["reason"] == io["reason"] and abs(ig["iters"] - io["iters"]) <= 2, (ig, io["iters"], io["reason"])
    m = min(len(ig["history"]), len(io["history"]), 12)
    assert np.allclose(ig["history"][:m], io["history"][:m], rtol=1e-6)
    xg = host(xg)
    assert np.linalg.norm((xg - xg.mean()) - (xo - xo.mean())) <= 1e-6 * max(np.linalg.norm(xo), 1e-300)
    # the default pair on the same handle afterwards: same answer (the two forms are the same method)
    x2, i2 = P.solve(dev(b), pc=pc, rtol=1e-8, maxit=2000)
    assert abs(i2["iters"] - ig["iters"]) <= 2
    assert np.linalg.norm(host(x2) - xg) <= 1e-6 * max(np.linalg.norm(xg), 1e-300)
    P.close()


@pytest.mark.parametrize("norm", [fo.NORM_UNPRECONDITIONED, fo.NORM_NATURAL])
def test_single_reduction_cg_norm_types_and_iteration_cap(norm):
    P, g = make_pair((40, 36, 34), CAVITY, kappa=1e-3)
    S = g.assemble_S()
    _, b = mean_free_rhs(S, g.ncell)
    xo, io = S.solve(b, norm=norm, rtol=1e-7, maxit=3000, single_reduction=True)
    xg, ig = P.solve(dev(b), norm_type=norm, rtol=1e-7, maxit=3000, history=True, cg_single_reduction=1)
    assert ig["reason"] == io["reason"] and abs(ig["iters"] - io["iters"]) <= 2
    m = min(len(ig["history"]), len(io["history"]), 12)
    assert np.allclose(ig["history"][:m], io["history"][:m], rtol=1e-6)
    # iteration cap: exactly maxit iterations, DIVERGED_ITS, and x is the iterate of that iteration
    xo5, io5 = S.solve(b, norm=norm, rtol=0.0, atol=0.0, maxit=5, single_reduction=True)
    xg5, ig5 = P.solve(dev(b), norm_type=norm, rtol=0.0, atol=0.0, maxit=5, cg_single_reduction=1)
    assert ig5["iters"] == io5["iters"] == 5 and ig5["reason"] == io5["reason"] == -3
    assert np.linalg.norm(host(xg5) - xo5) <= 1e-10 * np.linalg.norm(xo5)
    P.close()

"""-m gpu: a whole CNLinear time step assembled from the C-ABI pieces, on a fully periodic box where every
boundary-condition vector of the reference vanishes:

    v0interp = B v0                                   cnlinearcart3d.c:2826-2829   fl_momentum_interp_faces
    A = I + dt C(V0, v0interp) - (mu dt/2 rho) L      cnlinearcart3d.c:2930-2941   fl_momentum_set_state
    momrhs = v0 + (mu dt/2 rho) L v0 - kappa G phalf  cnlinearcart3d.c:2976-2998   fl_momentum_rhs
    x = J^-1 (momrhs, 0, 0)  with PC_ABF              nssol.c:21-29                fl_abf_apply / fl_abf_jacobian_mult
    p = phalf + 1.5 dp ; phalf += dp                  cnlinearcart3d.c:2846-2854   fl_pressure_update

The driver loop itself (SNES / outer KSP) is the reference's control plane and stays on the host: here a few lines of
Python (Richardson iteration on the block system, preconditioned by PCApply_ABF).  Checked the way the reference checks
itself (fluca/tests/taylor_green_vortex/taylor_green_vortex.c): error against the analytical Taylor-Green solution,
second order in h and dt.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

PER = 3


def _tgv(n, x, y, t, nu):
    d = np.exp(-2.0 * nu * t)
    return np.sin(x) * np.cos(y) * d, -np.cos(x) * np.sin(y) * d


def _run(n, nsteps, t_final=0.4, rho=1.0, mu=0.1, outer=3):
    from fluca_amd import capi
    from fluca_amd.poisson import KspOptions, Momentum, Poisson
    L = 2 * np.pi
    dt = t_final / nsteps
    P = Poisson.uniform((n, n, 4), [(0, L), (0, L), (0, L * 4 / n)], [PER] * 6, dt / rho)
    M = Momentum(P)
    h = L / n
    xc = (np.arange(n) + 0.5) * h
    xf = np.arange(n) * h
    dev = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64).ravel(), device="cuda")
    Z = np.ones((4, 1, 1))
    grid = lambda xs, ys: (Z * xs[None, None, :], Z * ys[None, :, None])     # arrays (k, j, i)
    X, Y = grid(xc, xc)
    u, w = _tgv(n, X, Y, 0.0, mu / rho)
    v = dev(np.stack([u, w, np.zeros_like(u)]))
    Xf, Yc = grid(xf, xc)
    Xc, Yf = grid(xc, xf)
    Vf = [dev(_tgv(n, Xf, Yc, 0.0, mu / rho)[0]), dev(_tgv(n, Xc, Yf, 0.0, mu / rho)[1]), dev(np.zeros_like(u))]
    p = dev(rho / 4 * (np.cos(2 * X) + np.cos(2 * Y)))
    phalf = p.clone()
    mo = KspOptions(type=capi.KSP_BCGS, rtol=1e-10, maxit=200)
    so = KspOptions(rtol=1e-10, maxit=2000)
    N = P.ncell
    iters = []
    for step in range(nsteps):
        W = M.interp_faces(v)
        M.set_state(dt, rho, mu, Vf, W)
        f = M.rhs(dt, rho, mu, v, p if step == 0 else phalf)
        # Richardson on J x = (f, 0, 0), preconditioned by PCApply_ABF: x += P^-1 (f - J x)
        xv, xV, xp, info = M.abf_apply(f, momentum=mo, schur=so)
        assert info[0]["reason"] > 0 and info[1]["reason"] > 0
        iters.append(info[1]["iters"])
        for _ in range(outer):
            fv, fV, fp = M.jacobian_mult(xv, xV, xp)
            rv = f - fv
            dv, dV, dp_, info = M.abf_apply(rv, [-a for a in fV], -fp, momentum=mo, schur=so)
            xv, xp = xv + dv, xp + dp_
            xV = [a + b for a, b in zip(xV, dV)]
        res = torch.linalg.norm(f - M.jacobian_mult(xv, xV, xp)[0]) / torch.linalg.norm(f)
        v, Vf = xv, xV
        pn = torch.empty_like(p)
        P.pressure_update(step == 0, xp, p, phalf, pn)
        p = pn
    ue, we = _tgv(n, X, Y, t_final, mu / rho)
    vh = v.cpu().numpy().reshape(3, 4, n, n)
    err = np.sqrt(((vh[0] - ue) ** 2 + (vh[1] - we) ** 2).mean())
    div = np.abs(P.rhs(*Vf).cpu().numpy()).max()
    wmax = np.abs(vh[2]).max()
    M.close()
    P.close()
    return err, float(res), div, wmax


def test_taylor_green_vortex_second_order():
    e1, r1, d1, w1 = _run(16, 4)
    e2, r2, d2, w2 = _run(32, 8)
    # the flow stays two-dimensional and discretely divergence-free, the block system is solved
    assert w1 < 1e-12 and w2 < 1e-12
    assert d1 < 1e-7 and d2 < 1e-7
    assert r1 < 1e-4 and r2 < 1e-4
    # amplitude ~ 1: a few per cent error at 16^2 x 4 steps, and second-order convergence under joint refinement
    assert e1 < 0.05
    assert e2 < e1 / 3.0, (e1, e2)

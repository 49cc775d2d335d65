"""ctypes front end of oracle/libfluca_oracle.so -- TEST INFRASTRUCTURE ONLY.

The product package (fluca_amd/) must never import this module.  See the header
of fluca_oracle.c for what is restated and for the parity-pin status
(operator coefficients: pinned by the FlucaFD goldens; Krylov solve: PARITY
UNPINNED -- PETSc is absent).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

BC_NONE, BC_VELOCITY, BC_PRESSURE_OUTLET, BC_PERIODIC, BC_SYMMETRY = range(5)
KSP_CG, KSP_BCGS, KSP_CHEBYSHEV, KSP_GMRES = range(4)
PC_NONE, PC_JACOBI = range(2)
NORM_PRECONDITIONED, NORM_UNPRECONDITIONED, NORM_NATURAL, NORM_NONE = range(4)
DELTA_PESKIN4, DELTA_ROMA3 = range(2)

_dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")


class KspOpts(C.Structure):
    _fields_ = [("type", C.c_int), ("pc", C.c_int), ("norm_type", C.c_int), ("remove_nullspace", C.c_int),
                ("maxit", C.c_int), ("rtol", C.c_double), ("atol", C.c_double), ("dtol", C.c_double),
                ("emin", C.c_double), ("emax", C.c_double), ("cg_single_reduction", C.c_int)]


class KspStats(C.Structure):
    _fields_ = [("iters", C.c_int), ("reason", C.c_int), ("rnorm0", C.c_double), ("rnorm", C.c_double),
                ("seconds", C.c_double)]


def build(force=False):
    so = os.path.join(_HERE, "libfluca_oracle.so")
    src = os.path.join(_HERE, "fluca_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfluca_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        ip3 = C.POINTER(C.c_int)
        L.fo_grid_create.restype = C.c_void_p
        L.fo_grid_create.argtypes = [ip3] + [C.c_void_p] * 6 + [ip3, C.c_double]
        L.fo_grid_destroy.argtypes = [C.c_void_p]
        L.fo_grid_ncell.restype = C.c_int64
        L.fo_grid_ncell.argtypes = [C.c_void_p]
        L.fo_grid_nface.restype = C.c_int64
        L.fo_grid_nface.argtypes = [C.c_void_p, C.c_int]
        L.fo_gst_row_1d.restype = C.c_int
        L.fo_gst_row_1d.argtypes = [C.c_void_p, C.c_int, C.c_int, ip3, C.POINTER(C.c_double)]
        L.fo_gst_bc_coeff_1d.restype = C.c_double
        L.fo_gst_bc_coeff_1d.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.fo_div_row_1d.argtypes = [C.c_void_p, C.c_int, C.c_int, ip3, C.POINTER(C.c_double)]
        L.fo_assemble_S.restype = C.c_void_p
        L.fo_assemble_S.argtypes = [C.c_void_p]
        L.fo_csr_destroy.argtypes = [C.c_void_p]
        for f in ("fo_csr_nnz", "fo_csr_nrow"):
            getattr(L, f).restype = C.c_int64
            getattr(L, f).argtypes = [C.c_void_p]
        for f in ("fo_csr_rowptr", "fo_csr_col", "fo_csr_val"):
            getattr(L, f).restype = C.c_void_p
            getattr(L, f).argtypes = [C.c_void_p]
        L.fo_csr_mult.argtypes = [C.c_void_p, _dp, _dp]
        L.fo_csr_diag.argtypes = [C.c_void_p, _dp]
        L.fo_rhs.argtypes = [C.c_void_p, _dp, _dp, _dp, C.c_void_p, _dp]
        L.fo_apply_gst.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.fo_apply_G.restype = C.c_int
        L.fo_apply_G.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp]
        L.fo_remove_constant.argtypes = [C.c_int64, _dp]
        L.fo_gershgorin_dinvA.restype = C.c_double
        L.fo_gershgorin_dinvA.argtypes = [C.c_void_p, C.c_int]
        L.fo_ksp_solve.restype = C.c_int
        L.fo_ksp_solve.argtypes = [C.c_void_p, _dp, _dp, C.POINTER(KspOpts), C.POINTER(KspStats), C.c_void_p, C.c_int]
        L.fo_num_threads.restype = C.c_int
        L.fo_set_num_threads.argtypes = [C.c_int]
        L.fo_ibm_interp.argtypes = [C.c_void_p, C.c_int, C.c_int64, _dp, _dp, _dp, C.c_int, _dp, _dp]
        L.fo_ibm_spread.argtypes = [C.c_void_p, C.c_int, C.c_int64, _dp, _dp, _dp, _dp, C.c_int, _dp, _dp]
        L.fo_lap_row_1d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Row1d)]
        L.fo_conv_row_1d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.POINTER(Row1d)]
        L.fo_assemble_momentum.restype = C.c_void_p
        L.fo_assemble_momentum.argtypes = [C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
        L.fo_T_row_1d.restype = C.c_int
        L.fo_T_row_1d.argtypes = [C.c_void_p, C.c_int, C.c_int, ip3, C.POINTER(C.c_double)]
        L.fo_apply_T.restype = C.c_int
        L.fo_apply_T.argtypes = [C.c_void_p, _dp, C.c_void_p, C.c_void_p, C.c_void_p, _dp, _dp, _dp]
        L.fo_B_row_1d.restype = C.c_int
        L.fo_B_row_1d.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, ip3, C.POINTER(C.c_double)]
        L.fo_apply_B.restype = C.c_int
        L.fo_apply_B.argtypes = [C.c_void_p, _dp, C.c_void_p]
        L.fo_ibm_phi.restype = C.c_double
        L.fo_ibm_phi.argtypes = [C.c_int, C.c_double]
        # OpenMP sizes its team from the machine, not from this process's share of it: on a GPU box (hundreds of hardware threads, a CPU share
        # of 16) a 360-iteration solve then spends minutes in oversubscribed barriers.  Unless the caller chose (OMP_NUM_THREADS), never more
        # threads than the affinity mask, the cgroup quota or 16.
        if not os.environ.get("OMP_NUM_THREADS"):
            L.fo_set_num_threads(max(1, min(_cpu_share(), L.fo_num_threads())))
        _LIB = L
    return _LIB


def _cpu_share():
    n = 16
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:  # pragma: no cover
        n = min(n, os.cpu_count() or 1)
    try:   # cgroup v2 quota: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except (OSError, ValueError):
        pass
    return n


class Row1d(C.Structure):
    _fields_ = [("nc", C.c_int), ("off", C.c_int * 4), ("v", C.c_double * 4)]


def uniform_coords(n, lo, hi):
    """DMStagSetUniformCoordinatesProduct: face = lo + i*h, centre = lo + (i+0.5)*h (PETSc docs; unverified vs source)."""
    h = (hi - lo) / n
    xf = lo + np.arange(n + 1, dtype=np.float64) * h
    xc = lo + (np.arange(n, dtype=np.float64) + 0.5) * h
    return xf, xc


class Grid:
    """Cartesian grid + NS boundary conditions + kappa = dt/rho."""

    def __init__(self, n, xf, bc, kappa=1.0, xc=None):
        L = lib()
        self.n = tuple(int(v) for v in n)
        self.xf = [np.ascontiguousarray(a, dtype=np.float64) for a in xf]
        self.xc = [None if (xc is None or xc[d] is None) else np.ascontiguousarray(xc[d], dtype=np.float64) for d in range(3)]
        self.bc = tuple(int(b) for b in bc)
        self.kappa = float(kappa)
        for d in range(3):
            assert self.xf[d].shape == (self.n[d] + 1,)
        n3 = (C.c_int * 3)(*self.n)
        bc6 = (C.c_int * 6)(*self.bc)
        ptr = lambda a: None if a is None else a.ctypes.data_as(C.c_void_p)
        self.h = L.fo_grid_create(n3, ptr(self.xf[0]), ptr(self.xf[1]), ptr(self.xf[2]), ptr(self.xc[0]), ptr(self.xc[1]),
                                  ptr(self.xc[2]), bc6, self.kappa)
        if not self.h:
            raise ValueError("invalid grid / boundary conditions")
        self.periodic = tuple(self.bc[2 * d] == BC_PERIODIC for d in range(3))
        self.nf = tuple(self.n[d] + (0 if self.periodic[d] else 1) for d in range(3))
        self.ncell = L.fo_grid_ncell(self.h)
        self.nface = tuple(L.fo_grid_nface(self.h, d) for d in range(3))

    @classmethod
    def uniform(cls, n, box, bc, kappa=1.0):
        xf, xc = zip(*[uniform_coords(n[d], box[d][0], box[d][1]) for d in range(3)])
        return cls(n, xf, bc, kappa, xc)

    def __del__(self):
        try:
            lib().fo_grid_destroy(self.h)
        except Exception:
            pass

    # 1-D rows -----------------------------------------------------------
    def gst_row(self, d, f):
        col = (C.c_int * 2)()
        v = (C.c_double * 2)()
        nc = lib().fo_gst_row_1d(self.h, d, f, col, v)
        return [(col[i], v[i]) for i in range(nc)]

    def gst_bc_coeff(self, d, side):
        return lib().fo_gst_bc_coeff_1d(self.h, d, side)

    def div_row(self, d, i):
        col = (C.c_int * 2)()
        v = (C.c_double * 2)()
        lib().fo_div_row_1d(self.h, d, i, col, v)
        return [(col[0], v[0]), (col[1], v[1])]

    def T_row(self, d, f):
        """[(cell, weight)] of the face-normal velocity interpolation at face f of axis d (cnlinearcart3d.c:1934-2140)."""
        col = (C.c_int * 2)()
        v = (C.c_double * 2)()
        nc = lib().fo_T_row_1d(self.h, d, f, col, v)
        if nc < 0:
            raise ValueError("unsupported BC")
        return [(col[i], v[i]) for i in range(nc)]

    def apply_T(self, v, rhs=(None, None, None)):
        """V_d = rhs_d + T v_d: the face-normal velocity of stage 1 of PCApply_ABF (abfpc.c:73-74)."""
        V = [np.empty(self.nface[d]) for d in range(3)]
        keep = [None if r is None else np.ascontiguousarray(r, dtype=np.float64) for r in rhs]
        ptr = [None if r is None else r.ctypes.data_as(C.c_void_p) for r in keep]
        if lib().fo_apply_T(self.h, np.ascontiguousarray(v, dtype=np.float64), *ptr, *V):
            raise ValueError("unsupported BC in T")
        return V

    def B_row(self, d, f, c):
        """[(cell, weight)] of the face interpolation of component c at face f of axis d (cnlinearcart3d.c:1513-1747)."""
        col = (C.c_int * 2)()
        v = (C.c_double * 2)()
        nc = lib().fo_B_row_1d(self.h, d, f, c, col, v)
        if nc < 0:
            raise ValueError("unsupported BC")
        return [(col[i], v[i]) for i in range(nc)]

    def apply_B(self, v):
        """v0interp without its boundary-condition vector: 9 face arrays, [c*3+d] = component c on the d-faces."""
        out = [np.empty(self.nface[d]) for c in range(3) for d in range(3)]
        ptrs = (C.c_void_p * 9)(*[a.ctypes.data for a in out])
        if lib().fo_apply_B(self.h, np.ascontiguousarray(v, dtype=np.float64), ptrs):
            raise ValueError("unsupported BC in B")
        return out

    def lap_row(self, d, i, c):
        """[(offset, coeff)] of one axis' second-derivative row for component c (cnlinearcart3d.c:466-632)."""
        r = Row1d()
        lib().fo_lap_row_1d(self.h, d, i, c, C.byref(r))
        if r.nc < 0:
            raise ValueError("unsupported BC")
        return [(r.off[a], r.v[a]) for a in range(r.nc)]

    def conv_row(self, d, i, side, normal, vf):
        """[(offset, coeff)] one face's contribution to a convection row (cnlinearcart3d.c:931-1292)."""
        r = Row1d()
        lib().fo_conv_row_1d(self.h, d, i, side, int(normal), float(vf), C.byref(r))
        if r.nc < 0:
            raise ValueError("unsupported BC")
        return [(r.off[a], r.v[a]) for a in range(r.nc)]

    # operators ----------------------------------------------------------
    def assemble_momentum(self, cI, cC, cL, V0=None, W=None):
        """A = cI I + cC C + cL L on the component-major velocity vector; V0: 3 face arrays, W: 9 (c*3+d)."""
        def pack(arrs, n):
            if arrs is None:
                return None, None
            keep = [np.ascontiguousarray(a, dtype=np.float64) for a in arrs]
            assert len(keep) == n
            return keep, (C.c_void_p * n)(*[a.ctypes.data for a in keep])
        k1, p1 = pack(V0, 3)
        k2, p2 = pack(W, 9)
        if cC != 0.0:
            for d in range(3):
                assert k1[d].size == self.nface[d]
                for c in range(3):
                    assert k2[c * 3 + d].size == self.nface[d]
        return Csr(lib().fo_assemble_momentum(self.h, cI, cC, cL, p1, p2))

    def assemble_S(self):
        return Csr(lib().fo_assemble_S(self.h))

    def rhs(self, Vx, Vy, Vz, contrhs=None):
        b = np.empty(self.ncell)
        cp = None if contrhs is None else np.ascontiguousarray(contrhs, dtype=np.float64).ctypes.data_as(C.c_void_p)
        lib().fo_rhs(self.h, Vx, Vy, Vz, cp, b)
        return b

    def apply_gst(self, p):
        G = [np.empty(self.nface[d]) for d in range(3)]
        lib().fo_apply_gst(self.h, p, *G)
        return G

    def apply_G(self, p):
        W = [np.empty(self.ncell) for _ in range(3)]
        if lib().fo_apply_G(self.h, p, *W):
            raise ValueError("unsupported BC in G")
        return W

    # IBM ----------------------------------------------------------------
    def ibm_interp(self, kind, X, u):
        X = [np.ascontiguousarray(a, dtype=np.float64) for a in X]
        u = np.ascontiguousarray(u, dtype=np.float64).reshape(-1, self.ncell)
        U = np.empty((u.shape[0], X[0].size))
        lib().fo_ibm_interp(self.h, kind, X[0].size, X[0], X[1], X[2], u.shape[0], u, U)
        return U

    def ibm_spread(self, kind, X, dV, F, f=None):
        X = [np.ascontiguousarray(a, dtype=np.float64) for a in X]
        F = np.ascontiguousarray(F, dtype=np.float64).reshape(-1, X[0].size)
        dV = np.ascontiguousarray(np.broadcast_to(dV, X[0].shape), dtype=np.float64)
        if f is None:
            f = np.zeros((F.shape[0], self.ncell))
        lib().fo_ibm_spread(self.h, kind, X[0].size, X[0], X[1], X[2], dV, F.shape[0], F, f)
        return f


class Csr:
    def __init__(self, h):
        self.h = h
        L = lib()
        self.nrow = L.fo_csr_nrow(h)
        self.nnz = L.fo_csr_nnz(h)

    def __del__(self):
        try:
            lib().fo_csr_destroy(self.h)
        except Exception:
            pass

    def arrays(self):
        L = lib()
        rp = np.ctypeslib.as_array(C.cast(L.fo_csr_rowptr(self.h), C.POINTER(C.c_int64)), (self.nrow + 1,)).copy()
        col = np.ctypeslib.as_array(C.cast(L.fo_csr_col(self.h), C.POINTER(C.c_int32)), (self.nnz,)).copy()
        val = np.ctypeslib.as_array(C.cast(L.fo_csr_val(self.h), C.POINTER(C.c_double)), (self.nnz,)).copy()
        return rp, col, val

    def to_scipy(self):
        import scipy.sparse as sp
        rp, col, val = self.arrays()
        return sp.csr_matrix((val, col, rp), shape=(self.nrow, self.nrow))

    def mult(self, x):
        y = np.empty(self.nrow)
        lib().fo_csr_mult(self.h, np.ascontiguousarray(x, dtype=np.float64), y)
        return y

    def diag(self):
        d = np.empty(self.nrow)
        lib().fo_csr_diag(self.h, d)
        return d

    def gershgorin(self, pc=PC_JACOBI):
        return lib().fo_gershgorin_dinvA(self.h, pc)

    def solve(self, b, ksp=KSP_CG, pc=PC_JACOBI, norm=NORM_PRECONDITIONED, nullspace=True, rtol=1e-5, atol=1e-50,
              dtol=1e5, maxit=10000, emin=0.0, emax=0.0, history=True, single_reduction=False):
        """KSPSolve(kspS, b, x) with PETSc defaults (rtol 1e-5, atol 1e-50, dtol 1e5, maxit 1e4, zero guess).
        single_reduction: KSPCG with -ksp_cg_single_reduction."""
        o = KspOpts(ksp, pc, norm, int(nullspace), int(maxit), rtol, atol, dtol, emin, emax, int(bool(single_reduction)))
        st = KspStats()
        x = np.empty(self.nrow)
        hist = np.full(int(maxit) + 2, np.nan) if history else None
        rc = lib().fo_ksp_solve(self.h, np.ascontiguousarray(b, dtype=np.float64), x, C.byref(o), C.byref(st),
                                None if hist is None else hist.ctypes.data_as(C.c_void_p), 0 if hist is None else hist.size)
        if rc:
            raise ValueError("unknown KSP type")
        info = dict(iters=st.iters, reason=st.reason, rnorm0=st.rnorm0, rnorm=st.rnorm, seconds=st.seconds)
        if history:
            info["history"] = hist[:st.iters + 1].copy()
        return x, info


def gmres(A, b, pc=PC_JACOBI, rtol=1e-5, atol=1e-50, dtol=1e5, maxit=10000, restart=30):
    """KSPGMRES with PETSc's defaults, restated from PETSc's documented algorithm (the reference's default type for kspA,
    fluca/src/ns/utils/abfpc/abfpc.c:72; unverified vs PETSc source, PARITY UNPINNED like the other Krylov restatements):
    restart 30, classical Gram-Schmidt without refinement, left preconditioning (PCJACOBI = 1/diag(A) or PCNONE), zero initial
    guess, monitored norm = the preconditioned residual norm from the Givens recurrence, KSPConvergedDefault, the residual
    re-formed from x at every restart.  A: Csr.  -> (x, dict(iters, reason, rnorm0, rnorm, history))"""
    n = A.nrow
    dinv = 1.0 / A.diag() if pc == PC_JACOBI else np.ones(n)
    Mb = dinv * b
    x = np.zeros(n)
    r = Mb.copy()
    beta = float(np.sqrt(r @ r))
    rnorm0 = beta
    ttol = max(rtol * rnorm0, atol)

    def conv(v):
        if not np.isfinite(v):
            return -9
        if v <= ttol:
            return 3 if v < atol else 2
        if v >= dtol * rnorm0:
            return -4
        return 0
    hist = [beta]
    it, reason, res = 0, conv(beta), beta
    if not reason and maxit == 0:
        reason = -3
    while not reason:
        V = [r / beta]
        H = np.zeros((restart + 1, restart))
        cs, sn, g = np.zeros(restart), np.zeros(restart), np.zeros(restart + 1)
        g[0] = beta
        k = 0
        while k < restart and not reason:
            w = dinv * A.mult(V[k])
            h = np.array([w @ v for v in V])             # classical Gram-Schmidt: all dots against the unmodified w
            for hi, v in zip(h, V):
                w = w - hi * v
            hk1 = float(np.sqrt(w @ w))
            col = np.append(h, hk1)
            happy = not (hk1 > 1e-30 * rnorm0)
            if not happy:
                V.append(w / hk1)
            for i in range(k):
                t = cs[i] * col[i] + sn[i] * col[i + 1]
                col[i + 1] = -sn[i] * col[i] + cs[i] * col[i + 1]
                col[i] = t
            d = float(np.hypot(col[k], col[k + 1]))
            cs[k], sn[k] = (col[k] / d, col[k + 1] / d) if d > 0 else (1.0, 0.0)
            col[k], col[k + 1] = d, 0.0
            g[k + 1] = -sn[k] * g[k]
            g[k] = cs[k] * g[k]
            H[:k + 1, k] = col[:k + 1]
            res = abs(g[k + 1])
            it += 1
            hist.append(res)
            reason = conv(res)
            if not reason and happy:
                reason = 2
            if not reason and it >= maxit:
                reason = -3
            k += 1
        y = np.linalg.solve(np.triu(H[:k, :k]), g[:k]) if k else np.zeros(0)
        for yi, v in zip(y, V):
            x = x + yi * v
        if reason:
            break
        r = Mb - dinv * A.mult(x)
        beta = float(np.sqrt(r @ r))
        if not beta > 0:
            reason = 3
    return x, dict(iters=it, reason=reason, rnorm0=rnorm0, rnorm=res, history=np.array(hist))


def num_threads():
    return lib().fo_num_threads()


def set_num_threads(n):
    lib().fo_set_num_threads(int(n))


class MgOracle:
    """CPU restatement of the geometric multigrid preconditioner specified in fluca_amd/csrc/fl_mg.hip (no reference
    function behind it).  Levels: halve every axis with an even cell count >= 8, rediscretise S on
    the coarse grid; smoother: nu Chebyshev-Jacobi steps from a zero guess over (0.1, 1.1) x bound; restriction: volume-
    weighted average; prolongation: piecewise constant; coarsest level: Jacobi-PCG to rtol 1e-2; outer: PCG with the
    V(nu,nu) cycle as left preconditioner.  `bounds[l]` = the eigenvalue bound of D^-1 S the product uses on level l
    (fl_poisson_gershgorin); None = the row-wise Gershgorin bound of the assembled matrix."""

    def __init__(self, g, max_levels=0, nu=3, nullspace=True, bounds=None, prolong="constant", flexible=True):
        # flexible: the outer CG's beta in the Polak-Ribiere form, -alpha (q . z_new) / (r_old . z_old) (fl_mg.hip "mg_flexible", the default)
        self.nu, self.nullspace, self.prolong, self.flexible = int(nu), bool(nullspace), prolong, bool(flexible)
        self.grids, self.S, self.ratio = [g], [g.assemble_S()], []
        while max_levels <= 0 or len(self.grids) < max_levels:
            gf = self.grids[-1]
            r = [2 if (gf.n[d] % 2 == 0 and gf.n[d] >= 8) else 1 for d in range(3)]
            if max(r) == 1:
                break
            try:
                gc = Grid([gf.n[d] // r[d] for d in range(3)], [gf.xf[d][::r[d]] for d in range(3)], gf.bc, gf.kappa)
            except ValueError:
                break
            self.ratio.append(r)
            self.grids.append(gc)
            self.S.append(gc.assemble_S())
        self.bounds = list(bounds) if bounds is not None else [S.gershgorin(PC_JACOBI) for S in self.S]

    @property
    def nlevels(self):
        return len(self.grids)

    def _restrict(self, l, r):
        gf, gc, rr = self.grids[l], self.grids[l + 1], self.ratio[l]
        a = r.reshape(gc.n[2], rr[2], gc.n[1], rr[1], gc.n[0], rr[0])
        w = [np.diff(gf.xf[d]).reshape(gc.n[d], rr[d]) / np.diff(gc.xf[d])[:, None] for d in range(3)]
        a = a * w[2][:, :, None, None, None, None] * w[1][None, None, :, :, None, None] * w[0][None, None, None, None, :, :]
        return a.sum(axis=(1, 3, 5)).ravel()

    def _prolong(self, l, e):
        gf, gc, rr = self.grids[l], self.grids[l + 1], self.ratio[l]
        a = e.reshape(gc.n[2], gc.n[1], gc.n[0])
        if self.prolong == "constant":
            for ax, k in ((0, rr[2]), (1, rr[1]), (2, rr[0])):
                a = np.repeat(a, k, axis=ax)
            return a.ravel()
        # tri-linear (tuning knob "mg_prolong" = 1): along every coarsened axis a child takes (1 - w) parent + w (the parent's neighbour on
        # the child's side), w from the cell centres (1/4 on a uniform axis), periodic images included; no neighbour behind a wall: w = 0
        for d, ax in ((0, 2), (1, 1), (2, 0)):
            if rr[d] != 2:
                continue
            nc, per = gc.n[d], gf.bc[2 * d] == BC_PERIODIC
            Xc = 0.5 * (np.asarray(gc.xf[d])[:-1] + np.asarray(gc.xf[d])[1:])
            xc = 0.5 * (np.asarray(gf.xf[d])[:-1] + np.asarray(gf.xf[d])[1:])
            L = gc.xf[d][-1] - gc.xf[d][0]
            i = np.arange(gf.n[d])
            I = i // 2
            In = np.where(i % 2 == 1, I + 1, I - 1)
            ok = np.full(i.shape, True) if per else (In >= 0) & (In < nc)
            Xn = Xc[In % nc] + np.where(In < 0, -L, 0.0) + np.where(In >= nc, L, 0.0)
            w = np.where(ok, (xc - Xc[I]) / np.where(ok, Xn - Xc[I], 1.0), 0.0)
            shape = [1, 1, 1]
            shape[ax] = -1
            a = (1.0 - w).reshape(shape) * np.take(a, I, axis=ax) + w.reshape(shape) * np.take(a, In % nc, axis=ax)
        return a.ravel()

    def _smooth(self, l, b):
        lam = self.bounds[l]
        x, _ = self.S[l].solve(b, ksp=KSP_CHEBYSHEV, pc=PC_JACOBI, norm=NORM_NONE, nullspace=self.nullspace, maxit=self.nu,
                               emin=0.1 * lam, emax=1.1 * lam, history=False)
        return x

    def vcycle(self, b, l=0):
        S = self.S[l]
        if l + 1 == self.nlevels:
            x, _ = S.solve(b, ksp=KSP_CG, pc=PC_JACOBI, nullspace=self.nullspace, rtol=1e-2, maxit=200, history=False)
            return x
        x = self._smooth(l, b)
        r = b - S.mult(x)
        x = x + self._prolong(l, self.vcycle(self._restrict(l, r), l + 1))
        r = b - S.mult(x)
        return x + self._smooth(l, r)

    def pcg(self, b, rtol=1e-5, atol=1e-50, dtol=1e5, maxit=10000):
        """KSPCG, left preconditioning, preconditioned norm, zero initial guess (same recurrences as fl_solve_cg_mg)."""
        S = self.S[0]
        proj = (lambda v: v - v.mean()) if self.nullspace else (lambda v: v)
        x = np.zeros_like(b)
        r = b.copy()
        z = proj(self.vcycle(r))
        dp = np.linalg.norm(z)
        rz = r @ z
        p = z.copy()
        hist, it, reason = [dp], 0, 0
        rn0, ttol = dp, max(rtol * dp, atol)
        conv = lambda d: (-9 if not np.isfinite(d) else (3 if d < atol else 2) if d <= ttol else (-4 if d >= dtol * rn0 else 0))
        reason = conv(dp)
        while not reason:
            q = S.mult(p)
            pq = p @ q
            if not pq > 0:
                reason = -10
                break
            alpha = rz / pq
            x += alpha * p
            r -= alpha * q
            z = proj(self.vcycle(r))
            rz_old, dp, rz = rz, np.linalg.norm(z), r @ z
            it += 1
            hist.append(dp)
            reason = conv(dp)
            if not reason and it >= maxit:
                reason = -3
            if not reason and not rz > 0:
                reason = -8   # KSP_DIVERGED_INDEFINITE_PC: r.z <= 0 (KSPSolve_CG's test on beta; a non-symmetric S can get there)
            if reason:
                break
            beta = -alpha * (q @ z) / rz_old if self.flexible else rz / rz_old
            p = z + beta * p
        return proj(x), dict(iters=it, reason=reason, history=np.array(hist))


AINV_ID, AINV_DIAG, AINV_ROWSUM = 0, 1, 2   # PCABFAinvType, flucans.h:99-103


def abf_ainv(A, kind):
    """abfpc.c:84-90 / 155-160: MatGetDiagonal or MatGetRowSum of A, then VecReciprocal.  A: the Csr of assemble_momentum."""
    a = A.diag() if kind == AINV_DIAG else A.mult(np.ones(A.nrow))
    return 1.0 / a


def abf_schur_apply(g, ainv, p):
    """S p with S = D ((-T) diag(ainv) kappa G - (-R)) and -R = (-T)(kappa G) + kappa Gst, term by term as PCSetUp_ABF
    forms it (abfpc.c:161-170; cnlinearcart3d.c:2909-2911).  ainv = None is the ID type."""
    kGp = np.concatenate(g.apply_G(p))
    t1 = g.apply_T(kGp if ainv is None else ainv * kGp)      # T (a^-1 kappa G p)
    t2 = g.apply_T(kGp)
    kGst = g.apply_gst(p)
    tmp = [-t1[d] + t2[d] - kGst[d] for d in range(3)]
    return -g.rhs(*tmp)                                       # D tmp  (rhs = contrhs - D V)


def abf_schur_dense(g, ainv):
    """the assembled S of the DIAG / ROWSUM types, dense (small grids only): column j = S e_j"""
    n = g.ncell
    S = np.empty((n, n))
    e = np.zeros(n)
    for j in range(n):
        e[j] = 1.0
        S[:, j] = abf_schur_apply(g, ainv, e)
        e[j] = 0.0
    return S


class StepOracle:
    """CPU restatement of one CNLinear time step (NSStep_CNLinear_Cart3d_Internal, NSFormJacobian, NSFormFunction:
    cnlinearcart3d.c:2807-3060) for VELOCITY / PERIODIC / SYMMETRY boundaries, composed from the oracle's operators:

      v0interp = B v0 + vbcB(t)                                   :2826-2829, vbc :1749-1932
      A = I + dt C(V0, v0interp) - cv L,  cv = mu dt / (2 rho)    :2930-2941
      momrhs = v0 + cv (L v0 + vbcL(t)) - dt vbcC(t, t+dt) - (kappa G p + 0) + cv vbcL(t+dt)     :2976-2998
      interprhs = vbcT(t+dt)   (the Rhie-Chow boundary terms :3013-3044 vanish without outlets), contrhs = 0
      x = J^-1 f by Richardson preconditioned with PCApply_ABF (abfpc.c:48-111), unpreconditioned residual norm
      v, V <- x ; p = p0 + 2 dp, phalf = p0 + dp on the first step, p = phalf + 1.5 dp, phalf += dp afterwards  :2841-2854

    velocity(b, t, X) -> (3, npoints) array: the wall velocity callback of boundary b at the face centres X (npoints, 3).
    """

    def __init__(self, g, dt, rho, mu, velocity=None, krylov_rtol=1e-12, outer_rtol=1e-8, outer_maxit=50, pressure=None, ibm=None):
        """ibm = dict(kind, X=[x, y, z], dV, Ut=None): explicit direct forcing, momrhs += spread(Ut - interp(v0)) -- build-defined (the
        reference only promises an IBM, THEORY_GUIDE.md:130-132), the rule of NSSetImmersedBoundary in include/fluca_host.h."""
        assert abs(g.kappa - dt / rho) < 1e-15 * max(1.0, g.kappa)
        self.g, self.dt, self.rho, self.mu, self.velocity, self.pressure, self.ibm = g, dt, rho, mu, velocity, pressure, ibm
        self.S = g.assemble_S()
        self.L = g.assemble_momentum(0.0, 0.0, 1.0)
        self.krtol, self.ortol, self.omaxit = krylov_rtol, outer_rtol, outer_maxit
        self.nullspace = BC_PRESSURE_OUTLET not in g.bc
        self.step, self.t, self.phalf = 0, 0.0, None
        n = g.n
        self.cshape = (n[2], n[1], n[0])
        self.fshape = [(n[2], n[1], g.nf[0]), (n[2], g.nf[1], n[0]), (g.nf[2], n[1], n[0])]

    # -- boundary data -------------------------------------------------------------------------------------------
    def _wall(self, b, t):
        """(3, n2, n1) wall velocity on boundary b (in-plane axes in x,y,z order, the first fastest)"""
        g, ax, side = self.g, b // 2, b % 2
        a1, a2 = (1 if ax == 0 else 0), (1 if ax == 2 else 2)
        xc = [0.5 * (g.xf[d][1:] + g.xf[d][:-1]) if g.xc[d] is None else g.xc[d] for d in range(3)]
        X = np.empty((g.n[a2], g.n[a1], 3))
        X[..., ax] = g.xf[ax][-1 if side else 0]
        X[..., a1] = xc[a1][None, :]
        X[..., a2] = xc[a2][:, None]
        return np.asarray(self.velocity(b, t, X.reshape(-1, 3))).reshape(3, g.n[a2], g.n[a1])

    def _outlet(self, b, t):
        """(n2, n1) outlet pressure on boundary b: pressure(b, t, X) with X the face centres (npoints, 3)"""
        g, ax, side = self.g, b // 2, b % 2
        a1, a2 = (1 if ax == 0 else 0), (1 if ax == 2 else 2)
        xc = [0.5 * (g.xf[d][1:] + g.xf[d][:-1]) if g.xc[d] is None else g.xc[d] for d in range(3)]
        X = np.empty((g.n[a2], g.n[a1], 3))
        X[..., ax] = g.xf[ax][-1 if side else 0]
        X[..., a1] = xc[a1][None, :]
        X[..., a2] = xc[a2][:, None]
        return np.asarray(self.pressure(b, t, X.reshape(-1, 3))).reshape(g.n[a2], g.n[a1])

    def _layer(self, arr, ax, idx):
        """view of the layer `idx` along grid axis ax of an array shaped (k, j, i)"""
        sl = [slice(None)] * 3
        sl[2 - ax] = idx
        return arr[tuple(sl)]

    def _walls(self):
        return [b for b in range(6) if self.g.bc[b] == BC_VELOCITY]

    def step_once(self, v0, V0, p0):
        g, dt = self.g, self.dt
        cv = 0.5 * self.mu * dt / self.rho
        t = self.t
        N = g.ncell
        xc = [0.5 * (g.xf[d][1:] + g.xf[d][:-1]) if g.xc[d] is None else g.xc[d] for d in range(3)]
        W = g.apply_B(v0)
        momrhs = v0 + cv * self.L.mult(v0) - np.concatenate(g.apply_G(p0 if self.step == 0 else self.phalf))
        interprhs = [np.zeros(g.nface[d]) for d in range(3)]
        for b in self._walls():
            ax, side = b // 2, b % 2
            n, xf, c = g.n[ax], g.xf[ax], xc[ax]
            vb0, vb1 = self._wall(b, t), self._wall(b, t + dt)
            if side == 0:   # cnlinearcart3d.c:698-701
                h1, h2, h3, hc = c[0] - xf[0], c[1] - c[0], c[2] - c[0], xf[1] - xf[0]
            else:           # :726-729
                h1, h2, h3, hc = xf[n] - c[n - 1], c[n - 1] - c[n - 2], c[n - 1] - c[n - 3], xf[n] - xf[n - 1]
            cl = 2.0 * (h2 + h3) / (h1 * (h1 + h2) * (h1 + h3))
            sgn = 0.5 if side else -0.5
            for q in range(3):
                self._layer(W[q * 3 + ax].reshape(self.fshape[ax]), ax, -1 if side else 0)[...] = vb0[q]          # :1788
                cells = momrhs[q * N:(q + 1) * N].reshape(self.cshape)
                self._layer(cells, ax, -1 if side else 0)[...] += (cv * cl * (vb0[q] + vb1[q])                     # L: :701, twice (:2985, :2998)
                                                                   - dt * sgn * (vb1[q] * vb0[ax] + vb0[q] * vb1[ax]) / hc)   # C: :1338, :2991
            self._layer(interprhs[ax].reshape(self.fshape[ax]), ax, -1 if side else 0)[...] = vb1[ax]               # :2178, :3003-3005
        # PRESSURE_OUTLET: boundary-condition vector of G in momrhs (:2976-2984 with :257-259, :285-287 -- NOT scaled by
        # dt/rho, as written) and the Rhie-Chow boundary terms of interprhs (:3013-3044)
        tq, tp = (t if self.step == 0 else t - 0.5 * dt), t + 0.5 * dt
        wG = np.zeros(3 * N)
        for b in [b for b in range(6) if g.bc[b] == BC_PRESSURE_OUTLET]:
            ax, side = b // 2, b % 2
            n, xf, c = g.n[ax], g.xf[ax], xc[ax]
            pq, pp = self._outlet(b, tq), self._outlet(b, tp)
            h1, h2 = (xf[n] - c[n - 1], c[n - 1] - c[n - 2]) if side else (c[0] - xf[0], c[1] - c[0])
            cg = (1.0 if side else -1.0) * h2 / (h1 * (h1 + h2))
            cells = momrhs[ax * N:(ax + 1) * N].reshape(self.cshape)
            self._layer(cells, ax, -1 if side else 0)[...] -= cg * pq
            wc = wG[ax * N:(ax + 1) * N].reshape(self.cshape)
            self._layer(wc, ax, -1 if side else 0)[...] += g.kappa * cg * (pq - pp)
            self._layer(interprhs[ax].reshape(self.fshape[ax]), ax, -1 if side else 0)[...] += g.kappa * g.gst_bc_coeff(ax, side) * (pq - pp)
        if np.any(wG != 0.0):
            Tw = g.apply_T(wG)
            interprhs = [interprhs[d] - Tw[d] for d in range(3)]
        if self.ibm is not None:
            ib = self.ibm
            U = g.ibm_interp(ib["kind"], ib["X"], v0.reshape(3, N))
            F = (0.0 if ib.get("Ut") is None else np.asarray(ib["Ut"]).reshape(3, -1)) - U
            momrhs = momrhs + g.ibm_spread(ib["kind"], ib["X"], ib["dV"], F).ravel()
        A = g.assemble_momentum(1.0, dt, -cv, V0, W)

        def pcapply(fv, fV, fp):      # abfpc.c:71-101
            vs, _ = A.solve(fv, ksp=KSP_BCGS, pc=PC_JACOBI, nullspace=False, rtol=self.krtol, maxit=2000, history=False)
            Vs = g.apply_T(vs, fV)
            bS = g.rhs(*Vs, contrhs=fp)
            if fp is None:
                self.b_inf = float(np.abs(bS).max())   # || b ||_inf of the step's first Schur solve, b = -D V* (SURVEY 8d's scale for || D V ||_inf)
            ps, _ = self.S.solve(bS, ksp=getattr(self, "S_ksp", KSP_CG), nullspace=self.nullspace, rtol=self.krtol, maxit=20000, history=False)
            Gst = g.apply_gst(ps)
            return vs - np.concatenate(g.apply_G(ps)), [Vs[d] - Gst[d] for d in range(3)], ps

        def jmult(v, V, p):           # cnlinearcart3d.c:2885-2941
            kGp = np.concatenate(g.apply_G(p))
            Tw = g.apply_T(v + kGp)
            kGst = g.apply_gst(p)
            return A.mult(v) + kGp, [V[d] - Tw[d] + kGst[d] for d in range(3)], -g.rhs(*V)

        fnorm = np.sqrt(momrhs @ momrhs + sum(a @ a for a in interprhs))
        xv, xV, xp = pcapply(momrhs, interprhs, None)
        its = 1
        while True:
            jv, jV, jp = jmult(xv, xV, xp)
            rv, rV, rp = momrhs - jv, [interprhs[d] - jV[d] for d in range(3)], -jp
            rn = np.sqrt(rv @ rv + sum(a @ a for a in rV) + rp @ rp)
            if rn <= self.ortol * fnorm or its >= self.omaxit:
                break
            dv, dV, dp_ = pcapply(rv, rV, rp)
            xv, xV, xp = xv + dv, [xV[d] + dV[d] for d in range(3)], xp + dp_
            its += 1
        if self.step == 0:
            p, self.phalf = p0 + 2.0 * xp, p0 + xp
        else:
            p, self.phalf = self.phalf + 1.5 * xp, self.phalf + xp
        self.step += 1
        self.t += dt
        return xv, xV, p, dict(outer_its=its, rnorm=rn)

"""Host-side mirror of the reference's Poisson call sites on top of the C-ABI (device memory through torch).

`Poisson` plays the role of PC_ABF's Schur-complement half (fluca/src/ns/utils/abfpc/abfpc.c): set-up builds S from the
grid + NS boundary conditions, `.solve` is KSPSolve(kspS), `.rhs` / `.project` are the MatMult chains around it.
"""
import atexit
import ctypes as C
import weakref

import numpy as np
import torch

from . import capi
from .capi import check, lib


_LIVE = weakref.WeakSet()


@atexit.register
def _close_all():
    # handles kept alive until interpreter shutdown (e.g. by a traceback) must go before the HIP runtime does
    for h in list(_LIVE):
        try:
            h.close()
        except Exception:
            pass


def _ptr(t):
    if t is None:
        return None
    assert t.is_cuda and t.dtype == torch.float64 and t.is_contiguous(), "need a contiguous fp64 device tensor"
    return C.c_void_p(t.data_ptr())


def default_decomp(n, ranks, rank):
    """DMStag's default ownership ranges (MeshCartGetOwnershipRanges, cart.c:420-430)."""
    d = capi.fl_decomp()
    check(lib.fl_decomp_default((C.c_int64 * 3)(*n), (C.c_int * 3)(*ranks), int(rank), C.byref(d)), "fl_decomp_default")
    return d


class KspOptions:
    """The -ns_abf_schur_ksp_* / -ns_abf_schur_pc_type knobs, PETSc defaults."""

    def __init__(self, **kw):
        self.o = capi.fl_ksp_opts()
        lib.fl_ksp_opts_default(C.byref(self.o))
        for k, v in kw.items():
            if not hasattr(self.o, k):
                raise TypeError(f"unknown KSP option {k}")
            setattr(self.o, k, v)


class Poisson:
    def __init__(self, n, xf, bc, kappa, xc=None, decomp=None, device=0):
        if not torch.cuda.is_available():
            raise RuntimeError("fluca_amd needs a HIP device; there is no CPU path")
        self.n = tuple(int(v) for v in n)
        self.device = torch.device("cuda", device)
        self._xf = [np.ascontiguousarray(a, dtype=np.float64) for a in xf]
        self._xc = [None if (xc is None or xc[d] is None) else np.ascontiguousarray(xc[d], dtype=np.float64) for d in range(3)]
        g = capi.fl_grid()
        for d in range(3):
            assert self._xf[d].shape == (self.n[d] + 1,)
            g.n[d] = self.n[d]
            g.xf[d] = self._xf[d].ctypes.data
            g.xc[d] = None if self._xc[d] is None else self._xc[d].ctypes.data
        self.bc = tuple(int(b) for b in bc)
        self.kappa = float(kappa)
        self.decomp = decomp
        h = C.c_void_p()
        check(lib.fl_poisson_create(C.byref(g), (C.c_int * 6)(*self.bc), self.kappa,
                                    None if decomp is None else C.byref(decomp), device, C.byref(h)), "fl_poisson_create")
        self.h = h
        sz = (C.c_int64 * 4)()
        check(lib.fl_poisson_sizes(self.h, sz))
        self.ncell, self.nface = sz[0], (sz[1], sz[2], sz[3])
        self._cb = None
        self._children = []   # handles that borrow this one (Momentum): closed first
        _LIVE.add(self)
        # the library works on its own HIP stream; order it against torch's current stream around every call
        self.stream = torch.cuda.Stream(device=self.device)
        self._ext_stream = False
        check(lib.fl_poisson_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)))

    @classmethod
    def uniform(cls, n, box, bc, kappa, **kw):
        xf, xc = [], []
        for d in range(3):
            lo, hi = box[d]
            h = (hi - lo) / n[d]
            xf.append(lo + np.arange(n[d] + 1, dtype=np.float64) * h)       # DMStagSetUniformCoordinatesProduct
            xc.append(lo + (np.arange(n[d], dtype=np.float64) + 0.5) * h)
        return cls(n, xf, bc, kappa, xc=xc, **kw)

    def close(self):
        for ref in getattr(self, "_children", []):
            child = ref()
            if child is not None:
                child.close()
        self._children = []
        if getattr(self, "h", None):
            lib.fl_poisson_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """Run on a caller-managed torch stream: the caller orders it against other streams (bench.py does)."""
        self.stream = stream
        self._ext_stream = True
        check(lib.fl_poisson_set_stream(self.h, C.c_void_p(stream.cuda_stream)))

    def _pre(self):
        if not self._ext_stream:
            self.stream.wait_stream(torch.cuda.current_stream(self.device))

    def _post(self):
        if not self._ext_stream:
            torch.cuda.current_stream(self.device).wait_stream(self.stream)

    def synchronize(self):
        check(lib.fl_poisson_synchronize(self.h))

    def empty(self, n=None):
        return torch.empty(self.ncell if n is None else n, dtype=torch.float64, device=self.device)

    def tune_placement(self, max_tries=1):
        """Placement of the solver vectors inside one arena (see fl_poisson_tune_placement) -> (probe ms with all vectors in
        one physical block, probe ms at the chosen place); (0, 0) for handles too small to be placed."""
        out = (C.c_double * 2)()
        self._pre()
        check(lib.fl_poisson_tune_placement(self.h, int(max_tries), out), "fl_poisson_tune_placement")
        self._post()
        return out[0], out[1]

    def vector_bytes(self):
        """Device memory behind the handle's padded solver vectors (fl_poisson_vector_bytes)."""
        out = C.c_int64()
        check(lib.fl_poisson_vector_bytes(self.h, C.byref(out)), "fl_poisson_vector_bytes")
        return out.value

    # ---- MatMult(S) -------------------------------------------------------------------------------------------
    def apply(self, x, y=None):
        y = self.empty() if y is None else y
        self._pre()
        check(lib.fl_poisson_apply(self.h, _ptr(x), _ptr(y)), "fl_poisson_apply")
        self._post()
        return y

    def diagonal(self):
        d = self.empty()
        self._pre()
        check(lib.fl_poisson_diagonal(self.h, _ptr(d)), "fl_poisson_diagonal")
        self._post()
        return d

    # ---- KSPSolve(kspS, b, x) ---------------------------------------------------------------------------------
    def solve(self, b, x=None, opts=None, history=False, **kw):
        opts = opts or KspOptions(**kw)
        o = opts.o
        x = self.empty() if x is None else x
        hist = None
        if history:
            hist = np.full(o.maxit + 1, np.nan)
            o.history = hist.ctypes.data_as(C.POINTER(C.c_double))
            o.nhistory = hist.size
        st = capi.fl_ksp_stats()
        self._pre()
        check(lib.fl_poisson_solve(self.h, _ptr(b), _ptr(x), C.byref(o), C.byref(st)), "fl_poisson_solve")
        self._post()
        info = dict(iters=st.iters, reason=st.reason, rnorm0=st.rnorm0, rnorm=st.rnorm, seconds=st.seconds,
                    kernel_ms=st.kernel_ms, kernel_launches=st.kernel_launches, kernel2_ms=st.kernel2_ms, kernel2_launches=st.kernel2_launches)
        if history:
            info["history"] = hist[:st.iters + 1].copy()
            o.history = None
            o.nhistory = 0
        return x, info

    # ---- PCApply_ABF pieces ------------------------------------------------------------------------------------
    def rhs(self, Vx, Vy, Vz, contrhs=None, b=None):
        b = self.empty() if b is None else b
        self._pre()
        check(lib.fl_poisson_rhs(self.h, _ptr(Vx), _ptr(Vy), _ptr(Vz), _ptr(contrhs), _ptr(b)), "fl_poisson_rhs")
        self._post()
        return b

    def project(self, p, v=(None, None, None), V=(None, None, None)):
        self._pre()
        check(lib.fl_poisson_project(self.h, _ptr(p), _ptr(v[0]), _ptr(v[1]), _ptr(v[2]), _ptr(V[0]), _ptr(V[1]), _ptr(V[2])),
              "fl_poisson_project")
        self._post()

    def gst_bc(self, boundary, pb, V):
        self._pre()
        check(lib.fl_poisson_gst_bc(self.h, int(boundary), _ptr(pb), _ptr(V)), "fl_poisson_gst_bc")
        self._post()

    def pressure_update(self, first, dp, p0, phalf, p):
        self._pre()
        check(lib.fl_pressure_update(self.h, int(bool(first)), _ptr(dp), _ptr(p0), _ptr(phalf), _ptr(p)), "fl_pressure_update")
        self._post()

    # ---- multi-GPU ---------------------------------------------------------------------------------------------
    def comm_init_rccl(self, id_bytes, rank, nranks):
        buf = (C.c_char * capi.UNIQUE_ID_BYTES).from_buffer_copy(id_bytes)
        check(lib.fl_poisson_comm_init_rccl(self.h, buf, rank, nranks), "fl_poisson_comm_init_rccl")

    def comm_info(self):
        """What the handle's communicator is (fl_poisson_comm_info): transport 0 none / 1 RCCL / 2 host callbacks, rank and nranks as the
        communicator itself reports them, neighbours, messages and bytes sent per ghost exchange of one cell vector."""
        ci = capi.fl_comm_info()
        check(lib.fl_poisson_comm_info(self.h, C.byref(ci)), "fl_poisson_comm_info")
        return {k: getattr(ci, k) for k, _ in capi.fl_comm_info._fields_}

    def comm_init_host(self, exchange, allreduce, rank, nranks):
        """exchange(list of (peer, sendtag, recvtag, send_ndarray|None, recv_ndarray|None)); allreduce(ndarray) in place."""
        self._cb = host_transport_callbacks(exchange, allreduce)
        check(lib.fl_poisson_comm_init_host(self.h, self._cb[0], self._cb[1], None, rank, nranks), "fl_poisson_comm_init_host")


def host_transport_callbacks(exchange, allreduce):
    """The two C callbacks of fl_poisson_comm_init_host around Python functions (keep the returned pair alive)."""
    def _x(ctx, n, peer, stag, rtag, send, recv, nbytes):
        try:
            msgs = []
            for a in range(n):
                cnt = nbytes[a] // 8
                s = None if not send[a] else np.ctypeslib.as_array(C.cast(send[a], C.POINTER(C.c_double)), (cnt,))
                r = None if not recv[a] else np.ctypeslib.as_array(C.cast(recv[a], C.POINTER(C.c_double)), (cnt,))
                msgs.append((peer[a], stag[a], rtag[a], s, r))
            exchange(msgs)
            return 0
        except Exception:  # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1

    def _r(ctx, vals, n):
        try:
            allreduce(np.ctypeslib.as_array(vals, (n,)))
            return 0
        except Exception:  # pragma: no cover
            import traceback
            traceback.print_exc()
            return 1

    return capi.EXCHANGE_FN(_x), capi.ALLREDUCE_FN(_r)


def rccl_unique_id():
    buf = (C.c_char * capi.UNIQUE_ID_BYTES)()
    check(lib.fl_comm_unique_id(buf), "fl_comm_unique_id")
    return bytes(buf)


class Momentum:
    """The momentum block A = I + dt C - (mu dt / 2 rho) L of NSFormJacobian (cnlinearcart3d.c:2930-2941), matrix-free.

    Shares grid / BCs / stream / communicator with a `Poisson`; velocity vectors are component-major (3*ncell).
    """

    def __init__(self, poisson):
        self.p = poisson
        h = C.c_void_p()
        poisson._pre()
        check(lib.fl_momentum_create(poisson.h, C.byref(h)), "fl_momentum_create")
        poisson._post()
        self.h = h
        poisson._children.append(weakref.ref(self))

    def close(self):
        if getattr(self, "h", None):
            lib.fl_momentum_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_state(self, dt, rho, mu, V0, W, v0=None):
        """V0: 3 face tensors (face-normal velocity of the previous step); W: 9 face tensors, W[c*3+d] = v0interp_c on d-faces.
        v0 (3 * cells, optional): the cell-centred velocity W was interpolated from (W = B v0 + vbc) -- fl_momentum_set_state_v0."""
        assert len(V0) == 3 and len(W) == 9
        for d in range(3):
            assert V0[d].numel() == self.p.nface[d]
            for c in range(3):
                assert W[c * 3 + d].numel() == self.p.nface[d]
        a = (C.c_void_p * 3)(*[t.data_ptr() for t in V0])
        b = (C.c_void_p * 9)(*[t.data_ptr() for t in W])
        for t in list(V0) + list(W):
            _ptr(t)
        self.p._pre()
        if v0 is None:
            check(lib.fl_momentum_set_state(self.h, float(dt), float(rho), float(mu), a, b), "fl_momentum_set_state")
        else:
            assert v0.numel() == 3 * self.p.ncell
            check(lib.fl_momentum_set_state_v0(self.h, float(dt), float(rho), float(mu), a, b, _ptr(v0)), "fl_momentum_set_state_v0")
        self.p._post()

    def set_coefficients(self, cI, cC, cL):
        self.p._pre()
        check(lib.fl_momentum_set_coefficients(self.h, float(cI), float(cC), float(cL)), "fl_momentum_set_coefficients")
        self.p._post()

    def apply(self, v, y=None):
        y = self.p.empty(3 * self.p.ncell) if y is None else y
        assert v.numel() == 3 * self.p.ncell
        self.p._pre()
        check(lib.fl_momentum_apply(self.h, _ptr(v), _ptr(y)), "fl_momentum_apply")
        self.p._post()
        return y

    def diagonal(self):
        d = self.p.empty(3 * self.p.ncell)
        self.p._pre()
        check(lib.fl_momentum_diagonal(self.h, _ptr(d)), "fl_momentum_diagonal")
        self.p._post()
        return d

    def rowsum(self):
        d = self.p.empty(3 * self.p.ncell)
        self.p._pre()
        check(lib.fl_momentum_rowsum(self.h, _ptr(d)), "fl_momentum_rowsum")
        self.p._post()
        return d

    def gershgorin(self):
        """max_i sum_{j != i} |a_ij| / |a_ii|: the radius of the Gershgorin disc of the Jacobi-scaled operator around 1."""
        g = C.c_double()
        self.p._pre()
        check(lib.fl_momentum_gershgorin(self.h, C.byref(g)), "fl_momentum_gershgorin")
        self.p._post()
        return g.value

    def chebyshev_interval(self):
        """(emin, emax) FL_KSP_CHEBYSHEV uses on this state when none is given."""
        a, b = C.c_double(), C.c_double()
        self.p._pre()
        check(lib.fl_momentum_chebyshev_interval(self.h, C.byref(a), C.byref(b)), "fl_momentum_chebyshev_interval")
        self.p._post()
        return a.value, b.value

    def set_ainv_types(self, schur=0, upper=0):
        """PCABFSetSchurComplementAinvType / PCABFSetUpperTriangularAinvType: 0 ID, 1 DIAG, 2 ROWSUM"""
        check(lib.fl_abf_set_ainv_types(self.h, int(schur), int(upper)), "fl_abf_set_ainv_types")

    def schur_apply(self, p):
        y = self.p.empty()
        self.p._pre()
        check(lib.fl_abf_schur_apply(self.h, _ptr(p), _ptr(y)), "fl_abf_schur_apply")
        self.p._post()
        return y

    def solve(self, b, x=None, opts=None, history=False, **kw):
        kw.setdefault("type", capi.KSP_BCGS)
        opts = opts or KspOptions(**kw)
        o = opts.o
        x = self.p.empty(3 * self.p.ncell) if x is None else x
        hist = None
        if history:
            hist = np.full(o.maxit + 2, np.nan)
            o.history = hist.ctypes.data_as(C.POINTER(C.c_double))
            o.nhistory = hist.size
        st = capi.fl_ksp_stats()
        self.p._pre()
        check(lib.fl_momentum_solve(self.h, _ptr(b), _ptr(x), C.byref(o), C.byref(st)), "fl_momentum_solve")
        self.p._post()
        info = dict(iters=st.iters, reason=st.reason, rnorm0=st.rnorm0, rnorm=st.rnorm, seconds=st.seconds)
        if history:
            info["history"] = hist[:st.iters + 1].copy()
            o.history = None
            o.nhistory = 0
        return x, info

    def face_interp(self, v, rhs=(None, None, None)):
        """V_d = rhs_d + (T v)_d (abfpc.c:73-74)."""
        V = [self.p.empty(self.p.nface[d]) for d in range(3)]
        r = (C.c_void_p * 3)(*[None if t is None else t.data_ptr() for t in rhs])
        o = (C.c_void_p * 3)(*[t.data_ptr() for t in V])
        self.p._pre()
        check(lib.fl_momentum_face_interp(self.h, _ptr(v), r, o), "fl_momentum_face_interp")
        self.p._post()
        return V

    def abf_apply(self, momrhs, interprhs=(None, None, None), contrhs=None, momentum=None, schur=None):
        """PCApply_ABF (abfpc.c:48-111): returns v (3*ncell), V (3 face tensors), p, [stats kspA, stats kspS]."""
        mo = (momentum or KspOptions(type=capi.KSP_BCGS)).o
        so = (schur or KspOptions()).o
        v = self.p.empty(3 * self.p.ncell)
        pr = self.p.empty()
        V = [self.p.empty(self.p.nface[d]) for d in range(3)]
        r = (C.c_void_p * 3)(*[None if t is None else t.data_ptr() for t in interprhs])
        o = (C.c_void_p * 3)(*[t.data_ptr() for t in V])
        st = (capi.fl_ksp_stats * 2)()
        self.p._pre()
        check(lib.fl_abf_apply(self.h, C.byref(mo), C.byref(so), _ptr(momrhs), r, _ptr(contrhs), _ptr(v), o, _ptr(pr), st), "fl_abf_apply")
        self.p._post()
        info = [dict(iters=s.iters, reason=s.reason, rnorm0=s.rnorm0, rnorm=s.rnorm, seconds=s.seconds) for s in st]
        return v, V, pr, info

    def jacobian_mult(self, v, V, p):
        """MatMult of the block Jacobian (cnlinearcart3d.c:2885-2941): returns (fv, [fV], fp)."""
        fv = self.p.empty(3 * self.p.ncell)
        fp = self.p.empty()
        fV = [self.p.empty(self.p.nface[d]) for d in range(3)]
        a = (C.c_void_p * 3)(*[t.data_ptr() for t in V])
        b = (C.c_void_p * 3)(*[t.data_ptr() for t in fV])
        self.p._pre()
        check(lib.fl_abf_jacobian_mult(self.h, _ptr(v), a, _ptr(p), _ptr(fv), b, _ptr(fp)), "fl_abf_jacobian_mult")
        self.p._post()
        return fv, fV, fp

    def interp_faces(self, v, vbc=None, ends_only=False, out=None):
        """cnl->v0interp = B v (+ vbc): 9 face tensors, [c*3+d] = component c on the d-faces (cnlinearcart3d.c:2826-2829).
        ends_only: only the faces at the two ends of each axis are written (fl_momentum_interp_faces_ends; `out` may hand in the tensors)."""
        if out is None:
            out = [self.p.empty(self.p.nface[d]) for c in range(3) for d in range(3)]
        o = (C.c_void_p * 9)(*[t.data_ptr() for t in out])
        r = None if vbc is None else (C.c_void_p * 9)(*[None if t is None else t.data_ptr() for t in vbc])
        self.p._pre()
        if ends_only:
            check(lib.fl_momentum_interp_faces_ends(self.h, _ptr(v), r, o), "fl_momentum_interp_faces_ends")
        else:
            check(lib.fl_momentum_interp_faces(self.h, _ptr(v), r, o), "fl_momentum_interp_faces")
        self.p._post()
        return out

    def rhs(self, dt, rho, mu, v0, p=None, vbc=None, out=None):
        """Cell-wise part of momrhs (cnlinearcart3d.c:2976-2998): v0 + (mu dt/2 rho) L v0 - kappa G p (+ vbc)."""
        out = self.p.empty(3 * self.p.ncell) if out is None else out
        self.p._pre()
        check(lib.fl_momentum_rhs(self.h, float(dt), float(rho), float(mu), _ptr(v0), _ptr(p), _ptr(vbc), _ptr(out)), "fl_momentum_rhs")
        self.p._post()
        return out

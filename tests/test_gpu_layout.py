"""-m gpu: fl_layout_{from,to}_dmstag_{local,global} against an independent numpy enumeration of PETSc's DMStag orderings.

GLOBAL ordering (what PCApply_ABF's sub-vectors are; PETSc DMSetUp_Stag_3d): the rank's elements x fastest; inside a full element
the dofs in the order BACK faces, DOWN faces, LEFT faces, ELEMENT; behind the last element of a row on the last rank of a
non-periodic x axis a partial element with the LEFT faces only, behind the last row of a layer (non-periodic y) a partial row with
the DOWN faces only, behind the last layer (non-periodic z) a partial layer with the BACK faces only.  The enumeration below
walks that order element by element and dof by dof -- no closed-form offsets -- so it is independent of the kernel's index
arithmetic.  The DMs are the reference's: sdm (0,0,0,1), vdm (0,0,0,3), Sdm (0,0,1,0), Vdm (0,0,3,0) -- cart.c:88-116.
PETSc itself is not available here: the ordering is PETSc's documented one, unverified against the library."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import fluca_oracle as fo
from tests.gpu_common import dev

pytestmark = pytest.mark.gpu
V, PER = fo.BC_VELOCITY, fo.BC_PERIODIC
BACK, DOWN, LEFT, ELEM = 3, 2, 1, 0          # `what` of the C-ABI: 0 cells, 1 x-faces (LEFT), 2 y-faces (DOWN), 3 z-faces (BACK)


def enumerate_global(nel, extra, dof):
    """-> list of (what, comp, i, j, k) in the order of a DMStag global vector of one rank"""
    d2, d3 = dof[2], dof[3]
    out = []
    nx, ny, nz = nel
    for k in range(nz + (1 if extra[2] else 0)):
        for j in range(ny + (1 if extra[1] else 0)):
            for i in range(nx + (1 if extra[0] else 0)):
                px, py, pz = i == nx, j == ny, k == nz          # partial in x / y / z
                if not (px or py or pz):
                    out += [(BACK, c, i, j, k) for c in range(d2)] + [(DOWN, c, i, j, k) for c in range(d2)] + [(LEFT, c, i, j, k) for c in range(d2)]
                    out += [(ELEM, c, i, j, k) for c in range(d3)]
                elif px and not (py or pz):
                    out += [(LEFT, c, i, j, k) for c in range(d2)]
                elif py and not (px or pz):
                    out += [(DOWN, c, i, j, k) for c in range(d2)]
                elif pz and not (px or py):
                    out += [(BACK, c, i, j, k) for c in range(d2)]
                # elements partial in two or three directions only carry edge / vertex dofs: none here
    return out


def make(n, bc, decomp=None):
    from fluca_amd import capi
    from fluca_amd.poisson import Poisson, default_decomp
    box = [(0.0, 1.0)] * 3
    if decomp is None:
        return Poisson.uniform(n, box, bc, 1.0)
    ranks, rank = decomp
    return Poisson.uniform(n, box, bc, 1.0, decomp=default_decomp(n, ranks, rank))


def shapes(P):
    from fluca_amd import capi
    s = (C.c_int64 * 4)()
    capi.check(capi.lib.fl_poisson_sizes(P.h, s))
    return [int(v) for v in s]


CASES = [
    ((6, 5, 4), [V] * 6, None),                              # extra face on every axis
    ((6, 5, 4), [PER, PER, V, V, PER, PER], None),           # periodic x and z: no partial elements there
    ((7, 4, 3), [V] * 6, ((2, 1, 1), 0)),                    # first of two ranks in x: no extra x face on this rank
    ((7, 4, 3), [V] * 6, ((2, 1, 1), 1)),                    # last rank in x
    ((5, 6, 4), [V, V, PER, PER, V, V], ((1, 1, 2), 1)),
]


@pytest.mark.parametrize("n,bc,decomp", CASES)
@pytest.mark.parametrize("dof", [(0, 0, 0, 1), (0, 0, 0, 3), (0, 0, 1, 0), (0, 0, 3, 0), (0, 0, 2, 1)])
def test_global_ordering_round_trip(n, bc, decomp, dof):
    from fluca_amd import capi
    P = make(n, bc, decomp)
    ncell, nfx, nfy, nfz = shapes(P)
    # owned elements of this rank and whether it owns an extra face per axis, from the library's own sizes
    if decomp is None:
        nel = list(n)
    else:
        from fluca_amd.poisson import default_decomp
        d = default_decomp(n, decomp[0], decomp[1])
        nel = [int(d.len[a]) for a in range(3)]
    extra = [nfx // (nel[1] * nel[2]) > nel[0], nfy // (nel[0] * nel[2]) > nel[1], nfz // (nel[0] * nel[1]) > nel[2]]
    order = enumerate_global(nel, extra, dof)
    ent = C.c_int64()
    dofc = (C.c_int * 4)(*dof)
    capi.check(capi.lib.fl_dmstag_global_entries(P.h, dofc, C.byref(ent)))
    assert ent.value == len(order)
    rng = np.random.default_rng(1)
    glob = rng.standard_normal(len(order))
    gd = dev(glob)
    ext = {ELEM: nel, LEFT: [nel[0] + extra[0], nel[1], nel[2]], DOWN: [nel[0], nel[1] + extra[1], nel[2]], BACK: [nel[0], nel[1], nel[2] + extra[2]]}
    for what, nd in ((ELEM, dof[3]), (LEFT, dof[2]), (DOWN, dof[2]), (BACK, dof[2])):
        for comp in range(nd):
            e = ext[what]
            want = np.full(e[0] * e[1] * e[2], np.nan)
            for pos, (w, c, i, j, k) in enumerate(order):
                if w == what and c == comp:
                    want[(k * e[1] + j) * e[0] + i] = glob[pos]
            assert not np.isnan(want).any()                              # every item of the library array has exactly one home
            out = torch.empty(want.size, dtype=torch.float64, device="cuda")
            capi.check(capi.lib.fl_layout_from_dmstag_global(P.h, dofc, what, comp, C.c_void_p(gd.data_ptr()), C.c_void_p(out.data_ptr())))
            P.synchronize()
            assert np.array_equal(out.cpu().numpy(), want)
    # scatter everything back into a poisoned vector: every entry is written exactly once
    back = torch.full((len(order),), float("nan"), dtype=torch.float64, device="cuda")
    for what, nd in ((ELEM, dof[3]), (LEFT, dof[2]), (DOWN, dof[2]), (BACK, dof[2])):
        for comp in range(nd):
            e = ext[what]
            tmp = torch.empty(e[0] * e[1] * e[2], dtype=torch.float64, device="cuda")
            capi.check(capi.lib.fl_layout_from_dmstag_global(P.h, dofc, what, comp, C.c_void_p(gd.data_ptr()), C.c_void_p(tmp.data_ptr())))
            capi.check(capi.lib.fl_layout_to_dmstag_global(P.h, dofc, what, comp, C.c_void_p(tmp.data_ptr()), C.c_void_p(back.data_ptr())))
    P.synchronize()
    assert np.array_equal(back.cpu().numpy(), glob)
    P.close()


@pytest.mark.parametrize("n,bc,decomp", CASES)
def test_local_array_round_trip(n, bc, decomp):
    """DMStagVecGetArray layout: arr[k][j][i][slot] over a ghosted box (width-1 ghosts where a neighbour or a periodic image
    exists, the partial element at the high end of a non-periodic axis), here for Vdm + one element dof: 10 slots."""
    from fluca_amd import capi
    from fluca_amd.poisson import default_decomp
    P = make(n, bc, decomp)
    if decomp is None:
        lo, nel, ranks, coord = [0, 0, 0], list(n), (1, 1, 1), (0, 0, 0)
    else:
        d = default_decomp(n, decomp[0], decomp[1])
        lo, nel, ranks, coord = [int(d.lo[a]) for a in range(3)], [int(d.len[a]) for a in range(3)], decomp[0], [int(d.coord[a]) for a in range(3)]
    per = [bc[2 * a] == PER for a in range(3)]
    gstart, gsize = [], []
    for a in range(3):
        first, last = coord[a] == 0, coord[a] == ranks[a] - 1
        glo = lo[a] - (0 if (first and not per[a]) else 1)
        ghi = lo[a] + nel[a] + 1                                   # one ghost / partial element above in every case
        gstart.append(glo)
        gsize.append(ghi - glo)
    entries = 10
    D = capi.fl_dmstag_local()
    for a in range(3):
        D.gstart[a], D.gsize[a], D.start[a] = gstart[a], gsize[a], lo[a]
    D.entries = entries
    rng = np.random.default_rng(2)
    arr = rng.standard_normal((gsize[2], gsize[1], gsize[0], entries))
    ad = dev(arr.ravel())
    ncell, nfx, nfy, nfz = shapes(P)
    extra = [nfx // (nel[1] * nel[2]) > nel[0], nfy // (nel[0] * nel[2]) > nel[1], nfz // (nel[0] * nel[1]) > nel[2]]
    o = [lo[a] - gstart[a] for a in range(3)]
    copy = arr.copy()
    for what, slot in ((ELEM, 9), (LEFT, 6), (LEFT, 8), (DOWN, 3), (BACK, 1)):
        e = [nel[0] + (what == LEFT and extra[0]), nel[1] + (what == DOWN and extra[1]), nel[2] + (what == BACK and extra[2])]
        want = arr[o[2]:o[2] + e[2], o[1]:o[1] + e[1], o[0]:o[0] + e[0], slot]
        out = torch.empty(want.size, dtype=torch.float64, device="cuda")
        capi.check(capi.lib.fl_layout_from_dmstag_local(P.h, C.byref(D), what, slot, C.c_void_p(ad.data_ptr()), C.c_void_p(out.data_ptr())))
        P.synchronize()
        assert np.array_equal(out.cpu().numpy().reshape(want.shape), want)
        # INSERT a marked copy back: the owned entries change, ghosts and other slots do not
        capi.check(capi.lib.fl_layout_to_dmstag_local(P.h, C.byref(D), what, slot, C.c_void_p((out * 2).data_ptr()), C.c_void_p(ad.data_ptr())))
        copy[o[2]:o[2] + e[2], o[1]:o[1] + e[1], o[0]:o[0] + e[0], slot] *= 2
    P.synchronize()
    assert np.array_equal(ad.cpu().numpy().reshape(arr.shape), copy)
    # a box that does not contain the rank's items is refused
    D.gsize[0] = nel[0] - 1
    out = torch.empty(ncell, dtype=torch.float64, device="cuda")
    assert capi.lib.fl_layout_from_dmstag_local(P.h, C.byref(D), 0, 0, C.c_void_p(ad.data_ptr()), C.c_void_p(out.data_ptr())) == -62
    P.close()


def test_vertex_and_edge_dofs_are_refused():
    from fluca_amd import capi
    P = make((4, 4, 4), [V] * 6)
    ent = C.c_int64()
    assert capi.lib.fl_dmstag_global_entries(P.h, (C.c_int * 4)(1, 0, 0, 1), C.byref(ent)) == -56
    P.close()


def test_pcapply_dataflow_on_dmstag_vectors():
    """The call sequence of contrib/abfpc_hip.c::PCApply_ABF_HIP on numpy-built DMStag GLOBAL vectors (Vstar on Sdm, vstar on
    vdm, contrhs / p on sdm): de-interleave, Srhs = contrhs - D V*, p = S^-1 Srhs, v = v* - kappa G p, V = V* - kappa Gst p,
    re-interleave -- against the oracle's PCApply_ABF stage formulas arranged in the same ordering (abfpc.c:75-101)."""
    from fluca_amd import capi
    from tests.gpu_common import CAVITY, CAVITY_BOX, host, make_pair
    n = (12, 9, 7)
    P, g = make_pair(n, CAVITY, kappa=0.4)
    S = g.assemble_S()
    rng = np.random.default_rng(5)
    Vs = [rng.standard_normal(nf) for nf in g.nface]
    vs = [rng.standard_normal(g.ncell) for _ in range(3)]
    cr = rng.standard_normal(g.ncell)
    cr -= g.rhs(*Vs, contrhs=cr).mean()              # Srhs = contrhs - D V* sums to zero: compatible with the pure Neumann problem
    dofS, dofv = (0, 0, 1, 0), (0, 0, 0, 3)
    extra = [True, True, True]
    ordS, ordv = enumerate_global(list(n), extra, dofS), enumerate_global(list(n), extra, dofv)
    ext = {LEFT: [n[0] + 1, n[1], n[2]], DOWN: [n[0], n[1] + 1, n[2]], BACK: [n[0], n[1], n[2] + 1]}

    def to_S(fields):       # three face arrays -> Sdm ordering
        out = np.empty(len(ordS))
        for pos, (w, c, i, j, k) in enumerate(ordS):
            e = ext[w]
            out[pos] = fields[w - 1][(k * e[1] + j) * e[0] + i]
        return out

    def to_v(fields):       # three cell arrays -> vdm ordering (components of a cell adjacent)
        return np.stack(fields, axis=1).ravel()
    assert [t[:2] for t in ordv[:3]] == [(ELEM, 0), (ELEM, 1), (ELEM, 2)]
    gV, gv = dev(to_S(Vs)), dev(to_v(vs))
    cS, cv = (C.c_int * 4)(*dofS), (C.c_int * 4)(*dofv)
    ptr = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    dV = [torch.empty(nf, dtype=torch.float64, device="cuda") for nf in g.nface]
    dv = [torch.empty(g.ncell, dtype=torch.float64, device="cuda") for _ in range(3)]
    for d in range(3):
        capi.check(capi.lib.fl_layout_from_dmstag_global(P.h, cS, 1 + d, 0, ptr(gV), ptr(dV[d])))
        capi.check(capi.lib.fl_layout_from_dmstag_global(P.h, cv, 0, d, ptr(gv), ptr(dv[d])))
    P.synchronize()
    torch.cuda.current_stream().wait_stream(P.stream)
    srhs = P.rhs(*dV, contrhs=dev(cr))
    p, info = P.solve(srhs, rtol=1e-10, maxit=2000)
    P.project(p, v=dv, V=dV)
    oV, ov = torch.full_like(gV, float("nan")), torch.full_like(gv, float("nan"))
    P.stream.wait_stream(torch.cuda.current_stream())
    for d in range(3):
        capi.check(capi.lib.fl_layout_to_dmstag_global(P.h, cS, 1 + d, 0, ptr(dV[d]), ptr(oV)))
        capi.check(capi.lib.fl_layout_to_dmstag_global(P.h, cv, 0, d, ptr(dv[d]), ptr(ov)))
    P.synchronize()
    # oracle
    b = g.rhs(*Vs, contrhs=cr)
    po, _ = S.solve(b, rtol=1e-10, maxit=2000)
    Gst, Gc = g.apply_gst(po), g.apply_G(po)
    wantV = to_S([Vs[d] - Gst[d] for d in range(3)])
    wantv = to_v([vs[d] - Gc[d] for d in range(3)])
    assert info["reason"] == 2
    assert np.abs(host(oV) - wantV).max() <= 1e-7 * np.abs(wantV).max()
    assert np.abs(host(ov) - wantv).max() <= 1e-7 * np.abs(wantv).max()
    P.close()

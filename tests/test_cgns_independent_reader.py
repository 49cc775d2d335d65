"""The files libfluca_cgns.so writes, opened by an INDEPENDENT reader (tests/cgns_sids_reader.py: the published CGNS SIDS-to-HDF5
rules through `h5dump -x`, no knowledge of fluca_cgns.c) and compared with what the reference's call sequence must produce:
cartcgns.c:8-118 (Base, Zone, vertex coordinates), :293-401 (FlowSolution<step> with the cell fields and the three face-centred
UserDefinedData nodes and their GridLocation), flucacgns.c:22-70 (TimeIterValues / TimeValues, FlowSolutionPointers,
SimulationType).  libcgns itself is absent from the image: this checks conformance to the standard's file mapping, not cg_open."""
import os

import numpy as np
import pytest

from fluca_amd import build as flbuild
from tests import cgns_sids_reader as sids
from tests.test_cgns_layout import H5DUMP, write_file

pytestmark = pytest.mark.skipif(not (flbuild.have_hdf5() and os.path.exists(H5DUMP)), reason="no HDF5 C library / tools in this image")


@pytest.mark.parametrize("ranks,periodic", [((1, 1, 1), (0, 0, 0)), ((2, 1, 2), (0, 0, 1))])
def test_file_read_by_the_standards_rules(tmp_path, ranks, periodic):
    N = (5, 4, 3)
    steps, times = [0, 3, 6], [0.0, 0.3, 0.6]
    path = str(tmp_path / "series.cgns")
    xf, data = write_file(path, N, periodic, ranks, steps, times)
    S = sids.structured_solution(sids.read(path, H5DUMP))
    assert S["format"].startswith("IEEE_LITTLE") and 3.0 <= S["version"] < 5.0
    assert S["cells"] == N and S["vertices"] == tuple(n + 1 for n in N)
    # tensor-product vertex coordinates in Fortran order: CoordinateX varies along the FIRST CGNS index = the LAST HDF5 axis
    for axis, name in enumerate(("CoordinateX", "CoordinateY", "CoordinateZ")):
        a = S["coords"][name]
        want = xf[axis].reshape([-1 if d == 2 - axis else 1 for d in range(3)])
        assert np.array_equal(a, np.broadcast_to(want, a.shape)), name
    # one FlowSolution per output step (+ the CellInfo solution of cartcgns.c:94-116), all at cell centres
    assert sorted(S["solutions"]) == sorted(["CellInfo"] + [f"FlowSolution{s}" for s in steps])
    for s in steps:
        fs = S["solutions"][f"FlowSolution{s}"]
        cells, faces = data[s]
        assert fs["location"] == "CellCenter"
        assert sorted(fs["arrays"]) == sorted(cells)
        for name, want in cells.items():
            a = fs["arrays"][name]
            assert a.dtype == "R8" and a.data.shape == (N[2], N[1], N[0]) and np.array_equal(a.data, want), name    # (N0,N1,N2) reversed
        assert sorted(fs["user"]) == ["IFaceCenteredSolution", "JFaceCenteredSolution", "KFaceCenteredSolution"]
        for l, (node, loc) in enumerate((("IFaceCenteredSolution", "IFaceCenter"), ("JFaceCenteredSolution", "JFaceCenter"), ("KFaceCenteredSolution", "KFaceCenter"))):
            u = fs["user"][node]
            assert u["location"] == loc                                  # cartcgns.c:246-291: one GridLocation per face direction
            a = u["arrays"]["FaceNormalVelocity"]
            shape = [N[2], N[1], N[0]]
            shape[2 - l] += 1
            assert a.dtype == "R8" and a.data.shape == tuple(shape) and np.array_equal(a.data, faces[l]), node
    # time series bookkeeping (flucacgns.c:22-70)
    assert S["nsteps"] == len(steps) and np.array_equal(S["times"], times)
    assert S["pointers"]["FlowSolutionPointers"] == [f"FlowSolution{s}" for s in steps]
    assert S["pointers"]["FlowSolutionCellInfoPointers"] == ["CellInfo"] * len(steps)
    assert S["simulation_type"] == "TimeAccurate"
    rank = S["solutions"]["CellInfo"]["arrays"]["Rank"]
    assert rank.dtype == "I4" and rank.data.shape == (N[2], N[1], N[0]) and set(np.unique(rank.data)) == set(range(ranks[0] * ranks[1] * ranks[2]))

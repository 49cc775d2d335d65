"""An INDEPENDENT reader of CGNS/HDF5 files: walks a file by the rules of the published CGNS standard (SIDS + the "SIDS-to-HDF5"
file mapping), knowing nothing about fluca_amd/host/fluca_cgns.c -- neither its reader nor its node list.

Rules used (CGNS file mapping manual, HDF5 section):
  * every CGNS node is an HDF5 group; its attributes "name", "label" (the SIDS type, e.g. "Zone_t"), "type" (MT, I4, I8, R4, R8,
    C1, ...) describe it; its data, if any, live in the dataset " data"; the root group carries the datasets " format" and
    " hdf5version" and the child CGNSLibraryVersion_t;
  * arrays are stored in Fortran index order, so the HDF5 dimensions are the CGNS dimensions REVERSED: a CGNS array
    A(i, j, k) of extents (N0, N1, N2) is an HDF5 dataset of shape (N2, N1, N0) -- which numpy then indexes as A[k, j, i];
  * character data are C1 arrays (one byte per character; a list of 32-character names is C1 (32, n));
  * a structured Zone_t holds its size as I8 (index_dim, 3): vertex sizes, cell sizes, boundary-vertex sizes.
The file is read through `h5dump -x` (the HDF5 tools that come with the C library; h5py is not installed)."""
import os
import subprocess
import xml.etree.ElementTree as ET

import numpy as np

NS = "{http://hdfgroup.org/HDF5/XML/schema/HDF5-File.xsd}"


class Node:
    def __init__(self, name, label, dtype, data, children):
        self.name, self.label, self.dtype, self.data, self.children = name, label, dtype, data, children

    def child(self, name):
        hits = [c for c in self.children if c.name == name]
        assert len(hits) == 1, (self.name, name, [c.name for c in self.children])
        return hits[0]

    def by_label(self, label):
        return [c for c in self.children if c.label == label]

    def text(self):
        """C1 data as a string (or a list of strings for a 2-D C1 array)"""
        assert self.dtype == "C1"
        a = np.asarray(self.data, dtype=np.uint8)
        if a.ndim == 1:
            return bytes(a).decode().rstrip("\x00 ")
        return [bytes(r).decode().rstrip("\x00 ") for r in a]


def _attr_text(el, name):
    for a in el.findall(NS + "Attribute"):
        if a.get("Name") == name:
            raw = a.find(NS + "Data").find(NS + "DataFromFile").text
            return raw.strip().strip('"')
    return None


def _dataset(el):
    dims = [int(d.get("DimSize")) for d in el.iter(NS + "Dimension")]
    raw = el.find(NS + "Data").find(NS + "DataFromFile").text.split()
    isfloat = el.find(NS + "DataType").find(NS + "AtomicType").find(NS + "FloatType") is not None
    a = np.array([float(v) for v in raw]) if isfloat else np.array([int(v) for v in raw], dtype=np.int64)
    return a.reshape(dims) if dims else a


def _node(el):
    data = None
    for d in el.findall(NS + "Dataset"):
        if d.get("Name") == " data":
            data = _dataset(d)
    children = [_node(g) for g in el.findall(NS + "Group")]
    return Node(_attr_text(el, "name"), _attr_text(el, "label"), _attr_text(el, "type"), data, children)


def read(path, h5dump):
    xml = subprocess.run([h5dump, "-x", "-m", "%.17g", path], capture_output=True, text=True, check=True).stdout
    root_el = ET.fromstring(xml).find(NS + "RootGroup")
    extra = {d.get("Name"): _dataset(d) for d in root_el.findall(NS + "Dataset")}
    root = _node(root_el)
    root.file_datasets = extra
    return root


def structured_solution(root):
    """-> dict with everything a SIDS-conformant reader can say about a single-zone structured time series:
    cell counts, vertex coordinates (as 1-D arrays when the grid is a tensor product), per-step FlowSolution arrays by
    GridLocation, time values, solution pointers."""
    out = {}
    fmt = bytes(np.asarray(root.file_datasets[" format"], dtype=np.uint8)).decode().rstrip("\x00")
    out["format"] = fmt
    ver = root.by_label("CGNSLibraryVersion_t")
    assert len(ver) == 1 and ver[0].dtype == "R4"
    out["version"] = float(np.ravel(ver[0].data)[0])
    bases = root.by_label("CGNSBase_t")
    assert len(bases) == 1
    base = bases[0]
    assert base.dtype == "I4" and list(np.ravel(base.data)) == [3, 3]          # cell dimension, physical dimension
    zones = base.by_label("Zone_t")
    assert len(zones) == 1
    zone = zones[0]
    assert zone.dtype == "I8" and zone.data.shape == (3, 3)                    # CGNS (index_dim, 3) reversed: rows = vertex, cell, boundary
    out["vertices"], out["cells"], bnd = [tuple(int(v) for v in r) for r in zone.data]
    assert bnd == (0, 0, 0) and tuple(v - 1 for v in out["vertices"]) == out["cells"]
    zt = zone.by_label("ZoneType_t")
    assert len(zt) == 1 and zt[0].text() == "Structured"
    gc = zone.by_label("GridCoordinates_t")
    assert len(gc) == 1 and gc[0].name == "GridCoordinates"
    coords = {}
    for c in gc[0].by_label("DataArray_t"):
        assert c.dtype == "R8" and c.data.shape == tuple(reversed(out["vertices"]))
        coords[c.name] = c.data
    out["coords"] = coords
    sols = {}
    for fs in zone.by_label("FlowSolution_t"):
        loc = fs.by_label("GridLocation_t")
        entry = {"location": loc[0].text() if loc else "Vertex", "arrays": {a.name: a for a in fs.by_label("DataArray_t")}, "user": {}}
        for ud in fs.by_label("UserDefinedData_t"):
            uloc = ud.by_label("GridLocation_t")
            entry["user"][ud.name] = {"location": uloc[0].text() if uloc else None, "arrays": {a.name: a for a in ud.by_label("DataArray_t")}}
        sols[fs.name] = entry
    out["solutions"] = sols
    bid = base.by_label("BaseIterativeData_t")
    if bid:
        assert bid[0].dtype == "I4"
        out["nsteps"] = int(np.ravel(bid[0].data)[0])
        tv = [a for a in bid[0].by_label("DataArray_t") if a.name == "TimeValues"]
        assert len(tv) == 1 and tv[0].dtype == "R8"
        out["times"] = np.ravel(tv[0].data)
    zid = zone.by_label("ZoneIterativeData_t")
    if zid:
        out["pointers"] = {a.name: a.text() for a in zid[0].by_label("DataArray_t")}
    st = base.by_label("SimulationType_t")
    out["simulation_type"] = st[0].text() if st else None
    return out

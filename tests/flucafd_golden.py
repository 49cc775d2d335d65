"""Parser for the reference's FlucaFD golden stdout files (fluca/tests/fd/output/*.out, copied as data)."""
import os
import re

_COL = re.compile(r"col\[(\d+)\]: (.*), loc=(\w+), c=(\S+), v=(\S+)$")


def parse(path):
    """-> (header, [dict(i=, j=, k=, loc=, c=, v_text=, v=)])"""
    rows = []
    with open(path) as fh:
        lines = [l.rstrip("\n") for l in fh]
    header = lines[0]
    ncols = int(lines[1].split("=")[1])
    for l in lines[2:]:
        m = _COL.search(l)
        assert m, l
        idx = {kv.split("=")[0].strip(): int(kv.split("=")[1]) for kv in m.group(2).split(",")}
        vt = m.group(5)
        rows.append(dict(idx, loc=m.group(3), c=m.group(4), v_text=vt, v=float(vt)))
    assert len(rows) == ncols
    return header, rows


def fmt_g(v):
    """PetscPrintf("%g") followed by PETSc's habit of printing a trailing '.' for integral reals."""
    s = "%g" % v
    if re.fullmatch(r"-?\d+", s):
        s += "."
    return s

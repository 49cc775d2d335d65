"""Which source files a profiled kernel is compiled from, and their git blob hashes -- so that a counter pass quoted by bench.py
(profiles/pmc_*.json, `roofline.traffic`) can be tied to the kernel that ran: the pass records the hashes of the day it was taken,
bench.py compares them with the tree's and prints `traffic_stale`.  Measurement plumbing, not product code."""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_C = "fluca_amd/csrc/"
# every file whose text reaches the kernel's code object (the kernel's own file, the headers with its device functions); the compiler
# flags are recorded beside them under the key "hipcc flags"
KERNEL_SOURCES = {
    "k_cg_A": [_C + "fl_kernels.hip", _C + "fl_stencil.h", _C + "fl_device.h"],
    "k_cg_Bq": [_C + "fl_kernels.hip", _C + "fl_stencil.h", _C + "fl_device.h"],
    "k_cheb2": [_C + "fl_cheb2.hip", _C + "fl_stencil.h", _C + "fl_device.h"],
    "k_mom3": [_C + "fl_mom_tile3.h", _C + "fl_momentum.hip", _C + "fl_device.h"],
    "k_mom2": [_C + "fl_mom_tile.h", _C + "fl_momentum.hip", _C + "fl_device.h"],
}


def blob_sha(path):
    """git hash-object of a file of the tree (sha1 of "blob <size>\\0" + content): equals `git rev-parse HEAD:<path>` for a committed file."""
    with open(os.path.join(ROOT, path), "rb") as fh:
        data = fh.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def source_hashes(kernel):
    from . import build
    out = {p: blob_sha(p) for p in KERNEL_SOURCES[kernel]}
    out["hipcc flags"] = " ".join(build.FLAGS)
    return out


def stale(kernel, recorded):
    """True when the recorded hashes are missing or differ from the tree's: the counter pass is not of this kernel."""
    if not isinstance(recorded, dict) or not recorded:
        return True
    now = source_hashes(kernel)
    return any(recorded.get(p) != h for p, h in now.items())

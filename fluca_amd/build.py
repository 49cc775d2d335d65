"""Builds libflucahip.so (HIP kernels + C-ABI) in-tree with hipcc for gfx950.

No fallback: if hipcc is missing or the compile fails this raises.  Incremental by CONTENT: lib/manifest.json records, per target, a hash over the
text of everything it was built from and the command line; a target whose recorded hash differs from today's is rebuilt, whatever the time stamps
say.  fl_version() reports the hash over all sources (source_id).
"""
import hashlib
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
# the product lives in lib/; a build with FL_KBENCH_VARIANTS=1 (superseded kernels and experiment switches compiled in, fl_knobs.h) goes to
# lib_kbench/ and never replaces it -- point FLUCA_LIB_DIR at that directory to load it (tools/kbench.py, tools/experiments/)
KBENCH = bool(os.environ.get("FL_KBENCH_VARIANTS"))
LIBDIR = os.path.join(HERE, "lib_kbench" if KBENCH else "lib")
LIB = os.path.join(LIBDIR, "libflucahip.so")
SOURCES = ["fl_coeff.cpp", "fl_kernels.hip", "fl_api.hip", "fl_ksp.hip", "fl_cheb2.hip", "fl_layout.hip", "fl_ibm.hip", "fl_momentum.hip", "fl_mg.hip", "fl_schur_var.hip"]
HEADERS = ["fl_internal.h", "fl_knobs.h", "fl_handle.h", "fl_device.h", "fl_stencil.h", "fl_mom_tile.h", "fl_mom_tile3.h", os.path.join("..", "..", "include", "fluca_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (["-DFL_KBENCH_VARIANTS"] if KBENCH else []) + [f"-D{d}" for d in os.environ.get("FL_DEFINES", "").split()] + ([f"-DFL_MOM_WPE={int(os.environ['FL_MOM_WPE'])}"] if os.environ.get("FL_MOM_WPE") else []) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]
MANIFEST = os.path.join(LIBDIR, "manifest.json")


def _sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()


def _load_manifest(path=None):
    try:
        with open(path or MANIFEST) as fh:
            return json.load(fh)
    except (OSError, ValueError):
        return {}


def _fingerprint(deps, cmd_words):
    """What a target was built FROM: the content hashes of its inputs and the command line -- not their modification times (a `git checkout`
    can put an older file with a newer time stamp, or a newer file with an older one, under an object that then looks up to date)."""
    h = hashlib.sha256()
    for d in deps:
        h.update(os.path.basename(d).encode() + b"\0" + (_sha(d) if os.path.exists(d) else "absent").encode() + b"\0")
    # WHERE the tree lies is not an input: the GPU box runs the same files under another path (and /root/repo is a symlink to it there); a
    # fingerprint that carried the path would rebuild every library on the box -- under the feet of the process that has them loaded
    root = os.path.normpath(os.path.join(HERE, ".."))
    real = os.path.realpath(root)
    h.update(" ".join(w.replace(real, "$ROOT").replace(root, "$ROOT") for w in cmd_words).encode())
    return h.hexdigest()


def _stale(target, deps, cmd_words=(), manifest=None):
    """True when `target` is missing or was built from other inputs than today's (recorded fingerprint differs)."""
    if not os.path.exists(target):
        return True
    m = _load_manifest() if manifest is None else manifest
    return m.get(os.path.relpath(target, LIBDIR)) != _fingerprint(deps, cmd_words)


def _record(target, deps, cmd_words=()):
    m = _load_manifest()
    m[os.path.relpath(target, LIBDIR)] = _fingerprint(deps, cmd_words)
    with open(MANIFEST, "w") as fh:
        json.dump(m, fh, indent=0, sort_keys=True)


def _mapped(path):
    """True when this process has `path` mapped (Linux)."""
    try:
        real = os.path.realpath(path)
        with open("/proc/self/maps") as fh:
            return any(line.rstrip().endswith(real) or line.rstrip().endswith(path) for line in fh)
    except OSError:
        return False


def _link(cmd, target):
    """Run a link command whose output is `target` (after -o) into a temporary name and rename it over the target: a process that has the old file
    mapped (a test session that calls build() again) keeps the old inode instead of having its code pages rewritten under it."""
    if _mapped(target):
        # (the rename keeps the old pages alive, but the next dlopen of the name -- a dependent library, another ctypes handle -- would bring a SECOND copy
        # of the library with its own globals into the process: handles made by one and destroyed by the other end in a segmentation fault)
        raise RuntimeError(f"{target} is out of date with respect to the sources AND loaded in this process: rebuild it first "
                           "(`python -m fluca_amd.build`, or tests/conftest.py, which builds before anything is loaded); it is not replaced under a live process")
    tmp = target + ".tmp%d" % os.getpid()
    i = cmd.index("-o")
    assert cmd[i + 1] == target
    subprocess.check_call(cmd[:i + 1] + [tmp] + cmd[i + 2:])
    os.replace(tmp, target)


def source_id():
    """12 hex digits over every source and header of libflucahip.so and the compiler flags: what fl_version() reports, so that a log from the GPU
    box says which sources the loaded library came from."""
    files = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    return _fingerprint(files, FLAGS)[:12]


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
        cmd = [HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]
        if force or _stale(o, [s] + hdrs, cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            _record(o, [s] + hdrs, cmd)
        objs.append(o)
    # the build id: a translation unit of its own, regenerated whenever any source differs
    sid = source_id() + ("+kbench" if KBENCH else "")
    idsrc, idobj = os.path.join(LIBDIR, "fl_build_id.cpp"), os.path.join(LIBDIR, "fl_build_id.cpp.o")
    text = f'extern "C" const char *fl_build_id(void) {{ return "{sid}"; }}\n'
    if not os.path.exists(idsrc) or open(idsrc).read() != text:
        with open(idsrc, "w") as fh:
            fh.write(text)
    cmd = [os.environ.get("CXX", "g++"), "-O1", "-fPIC", "-c", idsrc, "-o", idobj]
    if force or _stale(idobj, [idsrc], cmd):
        subprocess.check_call(cmd)
        _record(idobj, [idsrc], cmd)
    objs.append(idobj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-ldl"]
    if force or _stale(LIB, objs, cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        _link(cmd, LIB)
        _record(LIB, objs, cmd)
    build_host(force, verbose)
    return LIB


HOST_LIB = os.path.join(LIBDIR, "libfluca_host.so")


def build_host(force=False, verbose=False):
    """The C host mirror (gcc): links against libflucahip.so through its C-ABI only."""
    src = os.path.join(HERE, "host", "fluca_host.c")
    hdrs = [os.path.normpath(os.path.join(HERE, "..", "include", h)) for h in ("fluca_host.h", "fluca_host_impl.h", "fluca_hip.h")]
    cmd = [os.environ.get("CC", "gcc"), "-std=gnu99", "-O2", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB, src,
           "-L" + LIBDIR, "-lflucahip", "-Wl,-rpath,$ORIGIN", "-lm", "-ldl", "-pthread"]
    if force or _stale(HOST_LIB, [src, LIB] + hdrs, cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        _link(cmd, HOST_LIB)
        _record(HOST_LIB, [src, LIB] + hdrs, cmd)
    build_cgns(force, verbose)
    build_example(force, verbose)
    build_example(force, verbose, name="cavity_flow_3d")
    build_example(force, verbose, name="flow_configs")
    return HOST_LIB


CGNS_LIB = os.path.join(LIBDIR, "libfluca_cgns.so")
HDF5_ROOT = os.environ.get("HDF5_ROOT", "/opt/conda")


def have_hdf5():
    return os.path.exists(os.path.join(HDF5_ROOT, "include", "hdf5.h")) and os.path.exists(os.path.join(HDF5_ROOT, "lib", "libhdf5.so"))


def build_cgns(force=False, verbose=False):
    """The CGNS-layout field dump (gcc + libhdf5).  Optional: without an HDF5 installation the library is not built and
    fluca_amd.hostapi.load_cgns() says so; nothing else depends on it."""
    src = os.path.join(HERE, "host", "fluca_cgns.c")
    if not have_hdf5():
        if verbose:
            print(f"no HDF5 under {HDF5_ROOT}: libfluca_cgns.so not built", flush=True)
        return None
    hdrs = [os.path.normpath(os.path.join(HERE, "..", "include", h)) for h in ("fluca_cgns.h", "fluca_host.h", "fluca_host_impl.h", "fluca_hip.h")]
    if True:
        # RUNPATH (new dtags), not RPATH: the HDF5 directory is searched for this library's direct dependencies only
        cmd = [os.environ.get("CC", "gcc"), "-std=gnu99", "-O2", "-fPIC", "-shared", "-Wall", "-o", CGNS_LIB, src,
               "-I" + os.path.join(HDF5_ROOT, "include"), "-L" + LIBDIR, "-lfluca_host", "-lflucahip",
               "-L" + os.path.join(HDF5_ROOT, "lib"), "-lhdf5", "-Wl,--enable-new-dtags",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(HDF5_ROOT, "lib"), "-lm"]
        if force or _stale(CGNS_LIB, [src, HOST_LIB] + hdrs, cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            _link(cmd, CGNS_LIB)
            _record(CGNS_LIB, [src, HOST_LIB] + hdrs, cmd)
    return CGNS_LIB


EXAMPLE = os.path.join(LIBDIR, "cavity_pressure_step")


def build_example(force=False, verbose=False, name="cavity_pressure_step"):
    """examples/<name>.c: the reference's cavity drivers against the C host mirror (no Python at run time)."""
    root = os.path.normpath(os.path.join(HERE, ".."))
    src = os.path.join(root, "examples", name + ".c")
    EXAMPLE = os.path.join(LIBDIR, name)
    cgns = os.path.exists(CGNS_LIB) and name == "cavity_flow_3d"   # the driver with -ns_view_solution / -ns_monitor_solution
    cmd = [os.environ.get("CC", "gcc"), "-std=gnu99", "-O2", "-Wall", "-o", EXAMPLE, src, "-I" + os.path.join(root, "include"),
           "-L" + LIBDIR] + (["-DFLUCA_HAVE_CGNS", "-lfluca_cgns"] if cgns else []) + ["-lfluca_host", "-lflucahip", "-lm",
           "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
    deps = [src, HOST_LIB, LIB] + ([CGNS_LIB] if cgns else []) + [os.path.join(root, "include", h) for h in ("fluca_host.h", "fluca_hip.h", "fluca_cgns.h")]
    if os.path.exists(src) and (force or _stale(EXAMPLE, deps, cmd)):
        if verbose:
            print(" ".join(cmd), flush=True)
        _link(cmd, EXAMPLE)
        _record(EXAMPLE, deps, cmd)
    return EXAMPLE


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

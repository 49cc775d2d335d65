"""Builds libflucahip.so (HIP kernels + C-ABI) in-tree with hipcc for gfx950.

No fallback: if hipcc is missing or the compile fails this raises.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libflucahip.so")
SOURCES = ["fl_coeff.cpp", "fl_kernels.hip", "fl_api.hip", "fl_ksp.hip", "fl_cheb2.hip", "fl_layout.hip", "fl_ibm.hip", "fl_momentum.hip", "fl_mg.hip"]
HEADERS = ["fl_internal.h", "fl_handle.h", "fl_device.h", "fl_stencil.h", "fl_mom_tile.h", "fl_mom_tile3.h", os.path.join("..", "..", "include", "fluca_hip.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = (["-DFL_KBENCH_VARIANTS"] if os.environ.get("FL_KBENCH_VARIANTS") else []) + [f"-D{d}" for d in os.environ.get("FL_DEFINES", "").split()] + ([f"-DFL_MOM_WPE={int(os.environ['FL_MOM_WPE'])}"] if os.environ.get("FL_MOM_WPE") else []) + ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-result"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.normpath(os.path.join(CSRC, h)) for h in HEADERS]
    objs = []
    for s in srcs:
        o = os.path.join(LIBDIR, os.path.basename(s) + ".o")
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC] + FLAGS + (["-x", "hip"] if s.endswith(".hip") else []) + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-Wl,-rpath,/opt/rocm/lib", "-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_host(force, verbose)
    return LIB


HOST_LIB = os.path.join(LIBDIR, "libfluca_host.so")


def build_host(force=False, verbose=False):
    """The C host mirror (gcc): links against libflucahip.so through its C-ABI only."""
    src = os.path.join(HERE, "host", "fluca_host.c")
    hdrs = [os.path.normpath(os.path.join(HERE, "..", "include", h)) for h in ("fluca_host.h", "fluca_host_impl.h", "fluca_hip.h")]
    if force or _stale(HOST_LIB, [src, LIB] + hdrs):
        cmd = [os.environ.get("CC", "gcc"), "-std=gnu99", "-O2", "-fPIC", "-shared", "-Wall", "-o", HOST_LIB, src,
               "-L" + LIBDIR, "-lflucahip", "-Wl,-rpath,$ORIGIN", "-lm", "-ldl", "-pthread"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
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
    if force or _stale(CGNS_LIB, [src, HOST_LIB] + hdrs):
        # RUNPATH (new dtags), not RPATH: the HDF5 directory is searched for this library's direct dependencies only
        cmd = [os.environ.get("CC", "gcc"), "-std=gnu99", "-O2", "-fPIC", "-shared", "-Wall", "-o", CGNS_LIB, src,
               "-I" + os.path.join(HDF5_ROOT, "include"), "-L" + LIBDIR, "-lfluca_host", "-lflucahip",
               "-L" + os.path.join(HDF5_ROOT, "lib"), "-lhdf5", "-Wl,--enable-new-dtags",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + os.path.join(HDF5_ROOT, "lib"), "-lm"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return CGNS_LIB


EXAMPLE = os.path.join(LIBDIR, "cavity_pressure_step")


def build_example(force=False, verbose=False, name="cavity_pressure_step"):
    """examples/<name>.c: the reference's cavity drivers against the C host mirror (no Python at run time)."""
    root = os.path.normpath(os.path.join(HERE, ".."))
    src = os.path.join(root, "examples", name + ".c")
    EXAMPLE = os.path.join(LIBDIR, name)
    cgns = os.path.exists(CGNS_LIB) and name == "cavity_flow_3d"   # the driver with -ns_view_solution / -ns_monitor_solution
    if os.path.exists(src) and (force or _stale(EXAMPLE, [src, HOST_LIB, LIB] + ([CGNS_LIB] if cgns else []))):
        cmd = [os.environ.get("CC", "gcc"), "-std=gnu99", "-O2", "-Wall", "-o", EXAMPLE, src, "-I" + os.path.join(root, "include"),
               "-L" + LIBDIR] + (["-DFLUCA_HAVE_CGNS", "-lfluca_cgns"] if cgns else []) + ["-lfluca_host", "-lflucahip", "-lm",
               "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,/opt/rocm/lib"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return EXAMPLE


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))

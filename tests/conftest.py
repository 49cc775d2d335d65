import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The libraries are brought up to date with the tree HERE, before any test module loads them: several fixtures call build() again, and a build
    # that finds a stale library after the process has loaded it would put a second copy of it into the process (round 5: a header edited after the
    # last build, every library rebuilt on the GPU box in the middle of the run, a segmentation fault 390 tests later).  A no-op when up to date.
    from fluca_amd import build as _flbuild
    _flbuild.build()
    if os.environ.get("FLUCA_TEST_BACKTRACE") == "1":
        # a crash inside a native library prints its C call stack (tests/plugins/segv_trace.c); off by default: faulthandler owns the signals otherwise
        import ctypes
        import subprocess
        src, out = os.path.join(ROOT, "tests", "plugins", "segv_trace.c"), os.path.join(ROOT, "tests", "plugins", "libsegv_trace.so")
        if not os.path.exists(out) or os.path.getmtime(out) < os.path.getmtime(src):
            subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-fPIC", "-shared", "-o", out, src])
        config._segv_trace = ctypes.CDLL(out)
        path = os.environ.get("FLUCA_TEST_BACKTRACE_FILE", os.path.join(ROOT, "gpurun_out", "segv_trace.txt"))
        os.makedirs(os.path.dirname(path), exist_ok=True)
        assert config._segv_trace.segv_trace_install(path.encode()) == 0


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")

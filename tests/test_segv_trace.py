"""tests/plugins/segv_trace.c (FLUCA_TEST_BACKTRACE=1): a crash inside a native library leaves its C call stack and the load addresses of the
repository's libraries in a file -- the tool that showed two copies of libflucahip.so in one test process (round 5)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_a_segmentation_fault_leaves_a_native_backtrace(tmp_path):
    src, so = os.path.join(ROOT, "tests", "plugins", "segv_trace.c"), str(tmp_path / "libsegv_trace.so")
    subprocess.check_call(["gcc", "-std=gnu99", "-O1", "-g", "-Wall", "-Werror", "-fPIC", "-shared", "-o", so, src])
    out = tmp_path / "trace.txt"
    code = ("import ctypes, sys\n"
            f"L = ctypes.CDLL({so!r})\n"
            f"assert L.segv_trace_install({str(out)!r}.encode()) == 0\n"
            "ctypes.string_at(8)\n")                      # reads address 8: SIGSEGV inside libc / ctypes
    r = subprocess.run([sys.executable, "-X", "faulthandler=0", "-c", code], capture_output=True, text=True, timeout=60)
    assert r.returncode == -11, (r.returncode, r.stderr[-300:])          # the default action still runs after the handler
    text = out.read_text()
    assert "native backtrace" in text and "libsegv_trace.so" in text and "mappings of the repository's libraries" in text
    assert sum(1 for line in text.splitlines() if "[0x" in line) >= 3    # frames below the handler: the faulting call chain

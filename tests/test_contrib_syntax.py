"""The reference-side binding contrib/abfpc_hip.c parses against the real include/fluca_hip.h and a declarations-only stand-in for the
PETSc names it uses (tools/check_contrib.sh; what that proves and what it does not: INTEGRATION.md section 2)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_contrib_binding_parses():
    out = subprocess.run([os.path.join(ROOT, "tools", "check_contrib.sh")], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr


def test_the_check_catches_a_wrong_call(tmp_path):
    """the check is not vacuous: one argument fewer in a library call makes it fail"""
    src = open(os.path.join(ROOT, "contrib", "abfpc_hip.c")).read()
    bad = src.replace("fl_poisson_solve(hip->flh, hip->d_Srhs, hip->d_p, &opts, &stats)", "fl_poisson_solve(hip->flh, hip->d_Srhs, hip->d_p, &opts)")
    assert bad != src
    d = tmp_path / "contrib"
    d.mkdir()
    (d / "abfpc_hip.c").write_text(bad)
    host = open(os.path.join(ROOT, "tools", "contrib_check", "abfpc_host.c")).read().replace('#include "../../contrib/abfpc_hip.c"', f'#include "{d / "abfpc_hip.c"}"')
    (tmp_path / "host.c").write_text(host)
    out = subprocess.run(["gcc", "-std=gnu99", "-fsyntax-only", "-Werror=implicit-function-declaration", "-I" + os.path.join(ROOT, "include"),
                          "-I" + os.path.join(ROOT, "tools", "contrib_check"), str(tmp_path / "host.c")], capture_output=True, text=True)
    assert out.returncode != 0 and "too few arguments" in out.stderr

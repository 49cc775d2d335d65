"""fluca_amd/provenance.py: a counter pass (profiles/pmc_*.json) is tied to the sources of the kernel it counted."""
import json
import os
import subprocess

from fluca_amd import provenance as pv

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_blob_sha_is_git_hash_object():
    path = "fluca_amd/csrc/fl_stencil.h"
    try:
        want = subprocess.check_output(["git", "hash-object", path], cwd=ROOT, text=True).strip()
    except (OSError, subprocess.CalledProcessError):
        want = None
    got = pv.blob_sha(path)
    assert len(got) == 40 and (want is None or got == want)


def test_stale_when_a_source_or_a_flag_differs():
    now = pv.source_hashes("k_cg_A")
    assert set(pv.KERNEL_SOURCES["k_cg_A"]) < set(now) and "hipcc flags" in now
    assert pv.stale("k_cg_A", now) is False
    assert pv.stale("k_cg_A", None) and pv.stale("k_cg_A", {})
    for key in now:
        other = dict(now)
        other[key] = "0" * 40
        assert pv.stale("k_cg_A", other), key
        del other[key]
        assert pv.stale("k_cg_A", other), key


def test_committed_passes_carry_their_sources():
    for k in ("k_cg_A", "k_cg_Bq", "k_cheb2", "k_mom3"):
        d = json.load(open(os.path.join(ROOT, "profiles", f"pmc_{k}.json")))
        assert d["hbm_bytes_per_launch"] > 0 and set(pv.KERNEL_SOURCES[k]) <= set(d["sources_at_profiling"]), k

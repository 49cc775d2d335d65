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


def test_build_is_incremental_by_content_not_by_time_stamp(tmp_path, monkeypatch):
    """fluca_amd/build.py rebuilds a target when the TEXT of an input differs from what the target was built from -- a header whose time stamp
    went backwards (git checkout of an older commit over a newer object) must still trigger the rebuild, and touching a file must not."""
    import os
    import time
    from fluca_amd import build
    monkeypatch.setattr(build, "LIBDIR", str(tmp_path))
    monkeypatch.setattr(build, "MANIFEST", str(tmp_path / "manifest.json"))
    hdr, src, obj = tmp_path / "a.h", tmp_path / "a.c", tmp_path / "a.o"
    hdr.write_text("#define A 1\\n")
    src.write_text("int a = A;\\n")
    cmd = ["cc", "-c", str(src)]
    assert build._stale(str(obj), [str(src), str(hdr)], cmd)                 # no target yet
    obj.write_text("object")
    build._record(str(obj), [str(src), str(hdr)], cmd)
    assert not build._stale(str(obj), [str(src), str(hdr)], cmd)
    old = time.time() - 86400
    hdr.write_text("#define A 2\\n")
    os.utime(hdr, (old, old))                                               # changed text, time stamp a day OLDER than the object
    assert os.path.getmtime(hdr) < os.path.getmtime(obj)
    assert build._stale(str(obj), [str(src), str(hdr)], cmd)
    build._record(str(obj), [str(src), str(hdr)], cmd)
    os.utime(hdr, None)                                                     # touched, same text: nothing to do
    assert not build._stale(str(obj), [str(src), str(hdr)], cmd)
    assert build._stale(str(obj), [str(src), str(hdr)], cmd + ["-O2"])      # another command line is another target


def test_the_loaded_library_was_built_from_the_sources_in_the_tree():
    """fl_version() carries the hash over every source, header and flag the library was compiled from (lib/fl_build_id.cpp); it must be the tree's --
    otherwise the tests are exercising a stale binary (run `python -m fluca_amd.build`)."""
    from fluca_amd import build, capi
    v = capi.lib.fl_version().decode()
    assert build.source_id() in v, (v, build.source_id())


def test_fingerprints_do_not_depend_on_where_the_tree_lies(tmp_path, monkeypatch):
    """The GPU box runs the snapshot under another path: nothing may look stale there (a rebuild would rewrite libraries the test process has
    loaded).  Same inputs, same command, another root -> the same fingerprint."""
    import os
    from fluca_amd import build
    root = os.path.normpath(os.path.join(build.HERE, ".."))
    src = os.path.join(build.CSRC, "fl_coeff.cpp")
    a = build._fingerprint([src], ["hipcc", "-c", src, "-o", os.path.join(root, "fluca_amd", "lib", "x.o")])
    monkeypatch.setattr(build, "HERE", "/somewhere/else/repo/fluca_amd")
    b = build._fingerprint([src], ["hipcc", "-c", src.replace(root, "/somewhere/else/repo"), "-o", "/somewhere/else/repo/fluca_amd/lib/x.o"])
    assert a == b


def test_a_library_this_process_has_loaded_is_never_replaced(tmp_path):
    """build._link refuses to rename a fresh link over a library that is mapped in the calling process (round 5: a stale libflucahip.so rebuilt by a
    test fixture on the GPU box put two copies of the library into the test process)."""
    import ctypes

    import pytest
    from fluca_amd import build
    src, so = tmp_path / "t.c", tmp_path / "libt.so"
    src.write_text("int t_answer(void) { return 42; }\n")
    cmd = ["gcc", "-shared", "-fPIC", "-o", str(so), str(src)]
    build._link(cmd, str(so))                                    # not loaded yet: linked through a temporary name
    assert so.exists() and not build._mapped(str(so))
    lib = ctypes.CDLL(str(so))
    assert lib.t_answer() == 42 and build._mapped(str(so))
    with pytest.raises(RuntimeError, match="loaded in this process"):
        build._link(cmd, str(so))
    assert [p.name for p in tmp_path.iterdir() if ".tmp" in p.name] == []

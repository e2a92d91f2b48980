"""The build records what it built (r06): `__graft_entry__.build_hip()` reads registers, spills, scratch bytes per lane and LDS per workgroup of
every kernel instantiation out of the code objects' own metadata (tools/kernel_resources.py) and writes them next to the library; the GPU validation run copies that record into
profiles/validated_build.json.  These tests hold the numbers the documentation quotes to what the compiler produced: a plain benchmark kernel that starts using scratch memory --
the r04 thermal kernel lost 18 % to 392 B/lane of it -- fails HERE, on a machine without a GPU, instead of drifting away from a hand-typed comment."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VALIDATED = os.path.join(ROOT, "profiles", "validated_build.json")
CURRENT = os.path.join(ROOT, "petlion.jl_amd", "libpetlion_hip.so.resources.json")
PLAIN = "k_integrate<0: plain>"
# (variant id, model, configs): the kernels the BASELINE configurations run
BENCH = {"v0": ("LCO isothermal", "C2, C4"), "v3": ("NMC + SEI", "C5"), "v4": ("LCO thermal", "C3")}
# scratch bytes per lane a plain benchmark kernel may have: none for the isothermal / SEI kernels; the thermal kernel sits at the 512-register limit and is held
# to what the validated binary has -- a regression beyond it fails
SCRATCH_MAX = {"v0": 0, "v3": 0, "v4": 64}
LDS_MAX = 40960        # four cells per CU


def _check(kern, where):
    for v, (model, cfgs) in BENCH.items():
        r = kern[v][PLAIN]
        assert r["private_segment_fixed_size"] <= SCRATCH_MAX[v], "%s: plain kernel of %s (%s) has %d B/lane of scratch" % (where, model, cfgs, r["private_segment_fixed_size"])
        assert r["group_segment_fixed_size"] <= LDS_MAX, (where, v, r["group_segment_fixed_size"])
    assert kern["v0"][PLAIN]["vgpr_spill_count"] <= 32, (where, kern["v0"][PLAIN])          # (spills into AGPRs -- one v_accvgpr each way --, never into scratch: asserted above)
    ds = kern["v0"].get("_ds_ops")
    if ds:      # the backend's DS merging is off (buildflags.NO_LSO): seen in the object, not believed from the command line (clang's front end says it ignores the feature)
        assert ds["merged_two_address"] < 0.1 * ds["plain"], (where, ds)
    for v, ks in kern.items():                          # every instantiation of every built-in variant: four cells per CU (two per SIMD for the small cells)
        for name, r in ks.items():
            if name.startswith("_"):
                continue
            assert r["group_segment_fixed_size"] <= LDS_MAX, (where, v, name, r["group_segment_fixed_size"])


def test_validated_build_records_its_kernels():
    d = json.load(open(VALIDATED))
    assert "kernel_resources" in d, "profiles/validated_build.json carries no kernel record: re-run the GPU validation script (tools/gpu/r06_final.sh)"
    assert d["kernel_resources"]["build_info"] == d["build_info"], "the kernel record is that of another binary than the validated one"
    _check(d["kernel_resources"]["kernels"], "validated build")


def test_current_build_keeps_the_plain_kernels_out_of_scratch():
    if not os.path.exists(CURRENT):
        pytest.skip("no build record next to the library (the library was not built by build_hip() of this tree)")
    d = json.load(open(CURRENT))
    if "error" in d.get("kernels", {}):
        pytest.skip("llvm tools unavailable when the library was built: %s" % d["kernels"]["error"])
    _check(d["kernels"], "current build")


def test_source_comments_quote_the_record_not_numbers():
    """petlion_kernels.h used to carry a hand-typed register table that said "0 B/lane" next to a binary with 28: the header now points at the record"""
    txt = open(os.path.join(ROOT, "petlion.jl_amd", "csrc", "petlion_kernels.h")).read()
    assert "validated_build.json" in txt and "B/lane of scratch in the integrate kernel" not in txt.split("namespace pl")[1][:2500]


def test_fallback_markers_carry_the_build_identity_and_keep_a_failed_variant_out_of_union_libraries(pkg, tmp_path, monkeypatch):
    """a grid library that fails the kernel self-test is rebuilt with the fall-back flags and registered after the failed one ("latest registration wins").
    A union library of the built-in flags built LATER for a second variant on the same grid used to carry the failed variant too, was registered after the fall-back library
    and shadowed it; and the marker was keyed on a file name, so it never expired.  Now the marker is per (grid, variant), carries the flag table's hash and the sources' age,
    and grids.library neither returns nor builds a built-in-flags library that holds a marked variant it was not asked for."""
    import json
    import time
    grids, buildflags = pkg.grids, pkg.buildflags
    monkeypatch.setattr(grids, "GRID_DIR", str(tmp_path))
    g = grids.grid_tuple(6, 5, 8, 12, 10, 10, None)
    assert not grids.needs_fallback(g, 8)
    grids.mark_fallback(g, 8, "selftest: V differs")
    assert grids.needs_fallback(g, 8) and not grids.needs_fallback(g, 0)
    # an existing union library of the built-in flags that holds the marked variant must not be handed out for variant 0 ...
    tag = grids.defines(g)[0]
    stem = os.path.join(str(tmp_path), "libplh_%s_v0_8" % tag)
    open(stem + ".so", "w").write("x")
    json.dump({"grid": list(g), "variants": [0, 8], "suffix": "", "machine_licm": False, "extra_flags": [], "fallback_variants": []}, open(stem + ".json", "w"))
    future = time.time() + 3600
    os.utime(stem + ".so", (future, future))
    asked = []

    class Stop(Exception):
        pass

    def fake_popen(cmd, **kw):
        asked.append(cmd)
        raise Stop()
    monkeypatch.setattr(buildflags, "popen", fake_popen)
    with pytest.raises(Stop):
        grids.library(g, [0])                       # ... it starts a build of a library without variant 8 instead
    assert any(("-DPL_VARIANT=0" in c) for c in asked) and not any(("-DPL_VARIANT=8" in c) for c in asked)
    # the marked variant itself, asked for with the built-in flags (a caller that ignores the marker), still gets what it asks for
    assert grids.library(g, [8]) == stem + ".so"
    # a marker written under another flag table has expired
    monkeypatch.setattr(buildflags, "table_repr", lambda: "another table")
    assert not grids.needs_fallback(g, 8)


def test_cached_libraries_are_fresh_by_content_not_by_age(pkg, tmp_path, monkeypatch):
    """r06: a grid / closure library is valid for the device sources it was built from, by CONTENT (manifest["src_hash"]).  Until now the files' ages decided: a `git checkout`
    of an unchanged source file made every cached library look stale, and the first GPU test that needed one rebuilt it on the box (397 s for six tests)."""
    import json
    import time
    grids = pkg.grids
    lib, man = str(tmp_path / "libplh_x.so"), str(tmp_path / "libplh_x.json")
    open(lib, "w").write("x")
    json.dump({"variants": [0], "src_hash": grids._sources_hash()}, open(man, "w"))
    past = time.time() - 10 * 365 * 86400
    os.utime(lib, (past, past))                                   # older than every source file: still fresh, the content matches
    assert grids.is_fresh(lib, man)
    json.dump({"variants": [0], "src_hash": "0" * 16}, open(man, "w"))
    future = time.time() + 3600
    os.utime(lib, (future, future))                               # newer than every source file: still stale, the content differs
    assert not grids.is_fresh(lib, man)
    json.dump({"variants": [0]}, open(man, "w"))                  # a manifest written before r06: the age rule decides once, and the hash is recorded
    assert grids.is_fresh(lib, man) and json.load(open(man))["src_hash"] == grids._sources_hash()
    os.utime(lib, (past, past))
    json.dump({"variants": [0]}, open(man, "w"))
    assert not grids.is_fresh(lib, man)

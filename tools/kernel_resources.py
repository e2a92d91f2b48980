#!/usr/bin/env python3
"""What the compiler actually built: registers, spills, scratch and LDS of every kernel instantiation of every variant object, read from the code objects' metadata notes
(llvm-readelf --notes on the gfx950 code object unbundled from the .hip_fatbin section) -- no GPU needed.

   python tools/kernel_resources.py [--json out.json] [variant ids ...]

__graft_entry__.build_hip() calls resources_of_build() after linking and writes petlion.jl_amd/_build/kernel_resources.json; the GPU validation run copies it into
profiles/validated_build.json, tests/test_build_records.py reads it (a plain benchmark kernel with scratch fails the suite), and the source comments / DESIGN.md quote THAT
file instead of hand-typed numbers (the header said "0 B/lane" while the shipped thermal kernel had 28)."""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"
TARGET = "hipv4-amdgcn-amd-amdhsa--gfx950"
FEATURE_NAMES = {0: "plain", 1: "stops", 3: "tables", 7: "closures", 23: "general row", 31: "refine", 33: "sensitivities"}
KEYS = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")


def demangle_kernel(name):
    """(kernel, feature set or None) from the mangled name of pl::k_*<ModelT<...>[, F]>"""
    m = re.match(r"_ZN\w*?(\d+)(k_[a-z_]+)I", name)
    kern = m.group(2) if m else name
    f = None
    if kern == "k_integrate":
        mf = re.search(r"EELi(\d+)EEEv", name)
        f = int(mf.group(1)) if mf else None
    return kern, f


def resources_of_object(obj):
    """{kernel label: {vgpr_count, agpr_count, ..., private_segment_fixed_size (scratch, B/lane), group_segment_fixed_size (LDS, B/workgroup)}} of one variant object"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj], stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=" + TARGET, "--output=" + co], stderr=subprocess.DEVNULL)
        txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
    out, cur = {}, {}
    for ln in txt.splitlines():
        t = ln.strip()
        if ln.startswith("  - ."):                  # first key of a kernel's map (the argument lists are nested deeper)
            if cur.get("name"):
                _store(out, cur)
            cur = {}
            t = t[2:]
        elif not ln.startswith("    ."):            # keys of a kernel's map sit at this indent
            continue
        m = re.match(r"\.(\w+):\s+(\S+)", t)
        if m and (m.group(1) in KEYS or m.group(1) == "name"):
            cur[m.group(1)] = m.group(2) if m.group(1) == "name" else int(m.group(2))
    if cur.get("name"):
        _store(out, cur)
    return out


def ds_ops_of_object(obj):
    """(merged, plain) DS operations in the object's device code: ds_read2* / ds_write2* against the other ds_read* / ds_write* -- the visible effect of the backend's
    load / store merging being OFF (buildflags.NO_LSO; clang's front end claims to ignore the feature string, the backend honours it: a build with merging on has
    thousands of the two-address forms, one with it off a few dozen that the IR vectoriser formed)"""
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section", ".hip_fatbin=" + fat, obj], stderr=subprocess.DEVNULL)
        subprocess.check_call([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + fat, "--targets=" + TARGET, "--output=" + co], stderr=subprocess.DEVNULL)
        txt = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--mcpu=gfx950", co], capture_output=True, text=True).stdout
    merged = len(re.findall(r"\bds_(?:read|write)2(?:st64)?_b\d+", txt))
    plain = len(re.findall(r"\bds_(?:read|write)_b\d+", txt))
    return merged, plain


def _store(out, cur):
    kern, f = demangle_kernel(cur["name"])
    label = kern if f is None else "%s<%d: %s>" % (kern, f, FEATURE_NAMES.get(f, "?"))
    out[label] = {k: cur.get(k) for k in KEYS}


def resources_of_build(obj_dir=None, tag="libpetlion_hip", variants=None):
    obj_dir = obj_dir or os.path.join(ROOT, "petlion.jl_amd", "_build")
    res = {}
    for fn in sorted(os.listdir(obj_dir)):
        m = re.match(re.escape(tag) + r"_v(\d+)\.o$", fn)
        if m and (variants is None or int(m.group(1)) in variants):
            res["v%s" % m.group(1)] = resources_of_object(os.path.join(obj_dir, fn))
            if m.group(1) == "0":                      # the DS-merging switch, seen in the object (one variant says it for the flag table)
                mg, pl = ds_ops_of_object(os.path.join(obj_dir, fn))
                res["v0"]["_ds_ops"] = {"merged_two_address": mg, "plain": pl}
    return res


def main():
    a = sys.argv[1:]
    out = a[a.index("--json") + 1] if "--json" in a else None
    ids = [int(x) for x in a if x.isdigit()]
    res = resources_of_build(variants=ids or None)
    if out:
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    print("%-8s %-34s %5s %5s %6s %6s %8s %7s" % ("variant", "kernel", "VGPR", "AGPR", "vspill", "sspill", "scratch", "LDS"))
    for v in sorted(res, key=lambda s: int(s[1:])):
        for k, r in sorted(res[v].items()):
            if k == "_ds_ops":
                print("%-8s DS operations: %d two-address (merged) against %d plain" % (v, r["merged_two_address"], r["plain"]))
            if k.startswith("k_integrate"):
                print("%-8s %-34s %5d %5d %6d %6d %8d %7d" % (v, k, r["vgpr_count"], r["agpr_count"], r["vgpr_spill_count"], r["sgpr_spill_count"], r["private_segment_fixed_size"], r["group_segment_fixed_size"]))


if __name__ == "__main__":
    main()

"""Kernels for discretisations other than the built-in one (reference src/params.jl:119-136: petlion(...; N_p, N_s, N_n, N_r_p, N_r_n, N_a, N_z)).

The grid dimensions are compile-time constants of the device source (one lane per node, register-resident blocks, LDS arrays sized by the grid), so another grid is
another build of csrc/variant_tu.hip -- the counterpart of the reference generating and caching its functions per model (generate_functions.jl:44-94, the
`saved_models/` cache).  `library(grid, variant_id)` compiles (hipcc, gfx950; once, cached under petlion.jl_amd/_grids/) a shared library holding the requested
model variants for that grid; Model.__init__ registers it with plh_register_grid_library() and then creates the handle as usual.

Limits (static_asserts in csrc/dfn_cell.h / dfn_thermal.h): 2 <= N_p, N_s, N_n; N_p + N_s + N_n <= 48; 10 <= N_r_p, N_r_n <= 16 (they may differ); with temperature = true additionally
5 <= N_p <= N/2, 5 <= N_n < (N + 1)/2 for N = N_p + N_s + N_n (each electrode inside its own half of the twisted sweeps: the T rows of its last / first node reach back to a
second neighbour), 2 <= N_a, N_z, N_a + N_z <= 30, N_a + N + N_z <= 64.  No CPU fallback: without hipcc the build fails loudly."""
import json
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
GRID_DIR = os.path.join(HERE, "_grids")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
DEFAULT = (10, 10, 10, 10, 10, 10)


def variant_table():
    """rows of PL_VARIANT_LIST (csrc/plh_host.h): id -> (chemistry, sei, thermal, mixed, sd, tf, rxn, w2) as the C spellings"""
    txt = open(os.path.join(CSRC, "plh_host.h")).read()
    rows = {}
    for m in re.finditer(r"X\((\d+),\s*(\w+),\s*(\w+),\s*(\w+),\s*(\w+),\s*(\w+),\s*(\w+),\s*(\w+),\s*(\d+)\)", txt):
        rows[int(m.group(1))] = m.groups()[1:]
    return rows


def variant_id(chemistry, sei, thermal, mixed, sd, tf, rxn, w2):
    """id of the model variant with these options (the names are those of include/petlion_hip.h), None if it is not instantiated"""
    mixed = int(mixed)                                   # plh_model_desc.precision: 0 f64, 1 mixed, 2 f64 in the reference's operation order (spelled false / true / 2 in the table)
    want = ("PLH_CHEM_" + chemistry, "true" if sei else "false", "true" if thermal else "false", {0: "false", 1: "true"}.get(mixed, str(mixed)), "PLH_SD_" + sd, "PLH_TF_" + tf,
            "PLH_RXN_" + rxn, str(int(w2)))
    for k, row in variant_table().items():
        if tuple(row) == want:
            return k
    return None


def grid_tuple(N_p, N_s, N_n, N_r, N_a=10, N_z=10, N_r_n=None):
    """(N_p, N_s, N_n, N_r_p, N_a, N_z) when the two particle grids are equal (the common case: the tag and the cache names of r01-r03), with N_r_n appended when they differ"""
    g = (int(N_p), int(N_s), int(N_n), int(N_r), int(N_a), int(N_z))
    return g if N_r_n is None or int(N_r_n) == int(N_r) else g + (int(N_r_n),)


def grid7(grid):
    """(N_p, N_s, N_n, N_r_p, N_a, N_z, N_r_n) of a 6- or 7-entry grid tuple"""
    g = tuple(int(x) for x in grid)
    return g if len(g) == 7 else g + (g[3],)


def check(grid, thermal=False, sei=False):
    p, s, n, r, a, z, rn = grid7(grid)
    if sei and n < 3:
        raise ValueError("discretisation: aging = :SEI needs N_n >= 3 (the SOH row extrapolates j_s from three nodes, residuals.jl:278-297)")
    if min(p, s, n) < 2 or p + s + n > 48:
        raise ValueError("discretisation: 2 <= N_p, N_s, N_n and N_p + N_s + N_n <= 48 (one lane per node)")
    if not (10 <= r <= 16 and 10 <= rn <= 16):
        raise ValueError("discretisation: 10 <= N_r_p, N_r_n <= 16 (radial operator tables: tools/gen_radial_tables.py)")
    if thermal:
        ne, mid = p + s + n, (p + s + n) // 2
        if not (5 <= p <= mid and 5 <= n < ne - mid and a >= 2 and z >= 2 and a + z <= 30 and a + ne + z <= 64):
            raise NotImplementedError("discretisation with temperature = true: 5 <= N_p <= (N_p + N_s + N_n) / 2, 5 <= N_n < (N_p + N_s + N_n + 1) / 2 (each electrode inside its "
                                      "own half of the twisted block sweeps), 2 <= N_a, N_z, N_a + N_z <= 30, N_a + N_p + N_s + N_n + N_z <= 64 (one lane per temperature node)")


def defines(grid):
    p, s, n, r, a, z, rn = grid7(grid)
    tag = "g%d_%d_%d_%d_%d_%d" % (p, s, n, r, a, z) + ("_rn%d" % rn if rn != r else "")
    return tag, ["-DPL_NP=%d" % p, "-DPL_NS=%d" % s, "-DPL_NN=%d" % n, "-DPL_NR=%d" % r, "-DPL_NRN=%d" % rn, "-DPL_NA=%d" % a, "-DPL_NZ=%d" % z, "-Dpl=pl_" + tag]


def _source_files():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".hip"))) + [os.path.join(HERE, "..", "include", "petlion_hip.h"), os.path.join(HERE, "buildflags.py")]


def _sources_mtime():
    return max(os.path.getmtime(f) for f in _source_files())


_HASH_CACHE = {}


def _sources_hash():
    """content hash of the device sources, the header and the flag table: what a cached grid / closure library is valid for.  (r06: the age of the files decided until
    now -- a `git checkout` of an unchanged source file, or a copy of the tree that does not keep time stamps, made every cached library look stale and the first GPU test
    that needed one rebuilt it on the box.)"""
    import hashlib
    files = _source_files()
    sig = tuple((f, os.path.getmtime(f), os.path.getsize(f)) for f in files)
    if _HASH_CACHE.get("sig") != sig:
        h = hashlib.sha1()
        for f in files:
            h.update(os.path.basename(f).encode()); h.update(open(f, "rb").read())
        _HASH_CACHE["sig"], _HASH_CACHE["hash"] = sig, h.hexdigest()[:16]
    return _HASH_CACHE["hash"]


def is_fresh(lib, manifest):
    """the library at `lib` was built from the sources as they are: by content (manifest["src_hash"]); a manifest written before r06 has no hash -- the age rule decides
    once and the hash is recorded"""
    try:
        meta = json.load(open(manifest))
    except (OSError, ValueError):
        return False
    if not os.path.exists(lib):
        return False
    if "src_hash" in meta:
        return meta["src_hash"] == _sources_hash()
    if os.path.getmtime(lib) >= _sources_mtime():
        meta["src_hash"] = _sources_hash()
        try:
            json.dump(meta, open(manifest, "w"))
        except OSError:
            pass
        return True
    return False


def library(grid, variants, force=False, machine_licm=False, extra_flags=(), suffix=""):
    """path of the grid library holding (at least) `variants` for `grid`, building it if it is missing, stale, or lacks one of them.  The kernels are compiled with the flags
    of the built-in library (buildflags.variant_flags); machine_licm=True is the fall-back build of a library that failed the kernel self-test (api._selftest_new_grid_library),
    its files carry the suffix "_licm".  extra_flags / suffix: experiment builds (tools/experiments/)."""
    check(grid, thermal=any(variant_table()[v][2] == "true" for v in variants))
    tag, defs = defines(grid)
    os.makedirs(GRID_DIR, exist_ok=True)
    import fcntl
    with open(os.path.join(GRID_DIR, "%s.lock" % tag), "w") as lock:        # one builder per grid at a time (the ranks of a multi-process job ask for the same library)
        fcntl.flock(lock, fcntl.LOCK_EX)
        return _library_locked(grid, variants, force, tag, defs, machine_licm, list(extra_flags), suffix + ("_licm" if machine_licm else ""))


def _licm_marker(grid, vid):
    return os.path.join(GRID_DIR, "fallback_%s_v%d.json" % (defines(grid)[0], vid))


def _build_identity():
    """what a fall-back marker is valid for: the flag table and the device sources (a marker written under other flags or older sources expires)"""
    import hashlib
    from . import buildflags
    return {"flags": hashlib.sha1(buildflags.table_repr().encode()).hexdigest()[:12], "src_hash": _sources_hash()}


def mark_fallback(grid, vid, why):
    """remember that variant `vid` on `grid` failed the kernel self-test with the built-in flags on this machine: later processes go to its fall-back build directly, and no
    union library built with the built-in flags carries this variant any more (a later union library registered after the `_licm` one shadowed it)"""
    os.makedirs(GRID_DIR, exist_ok=True)
    json.dump(dict(_build_identity(), why=str(why)), open(_licm_marker(grid, vid), "w"))


def needs_fallback(grid, vid):
    try:
        m = json.load(open(_licm_marker(grid, vid)))
    except (OSError, ValueError):
        return False
    idn = _build_identity()
    return m.get("flags") == idn["flags"] and m.get("src_hash") == idn["src_hash"]


def _library_locked(grid, variants, force, tag, defs, machine_licm=False, extra_flags=(), suffix=""):
    # The file name carries the variant ids it holds: a library is never rebuilt in place under a path that plh_register_grid_library may already have dlopen'ed in this
    # process (registering a known path is a no-op) -- a second model on the same grid with another variant gets a NEW file holding the union, registered next to the first.
    import glob
    from . import buildflags
    have_best = []
    for manifest in sorted(glob.glob(os.path.join(GRID_DIR, "libplh_%s_v*%s.json" % (tag, suffix)))):
        lib = manifest[:-5] + ".so"
        meta = json.load(open(manifest))
        if not is_fresh(lib, manifest) or meta.get("suffix", "") != suffix:
            continue
        have = meta["variants"]
        if not machine_licm and not extra_flags and any(v not in variants and needs_fallback(grid, v) for v in have):
            continue          # (holds kernels of a variant that failed the self-test with these flags: registered after that variant's fall-back library it would shadow it)
        if not force and set(variants) <= set(have):
            return lib
        if len(have) > len(have_best):
            have_best = have
    if not machine_licm and not extra_flags:          # a variant that needs its fall-back build never rides along in a library of the built-in flags
        have_best = [v for v in have_best if v in variants or not needs_fallback(grid, v)]
    allv = sorted(set(have_best) | set(variants))
    stem = os.path.join(GRID_DIR, "libplh_%s_v%s%s" % (tag, "_".join(str(v) for v in allv), suffix))
    lib, manifest = stem + ".so", stem + ".json"
    src = os.path.join(CSRC, "variant_tu.hip")
    common = [HIPCC, "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wno-unused-value", "-Wno-pass-failed"] + defs + list(extra_flags)
    jobs, objs = [], []
    for v in allv:
        o = os.path.join(GRID_DIR, "%s%s_v%d.o" % (tag, suffix, v))
        objs.append(o)
        # the flags of the built-in kernels of this variant (buildflags.py: ONE table since r05; r04's grid libraries kept MachineLICM and late inlining for every variant
        # and ran 2 ... 21 % behind the built-in kernels for it).  A library that fails the self-test on the user's machine is rebuilt with machine_licm=True.
        jobs.append((v, o, buildflags.popen(common + buildflags.variant_flags(v, machine_licm=machine_licm) + ["-DPL_VARIANT=%d" % v, "-c", src, "-o", o], echo=False)))
    glue = os.path.join(GRID_DIR, "%s%s_glue.o" % (tag, suffix))
    gj = subprocess.Popen(common + ["-O2", "-DPL_GRID_GLUE", "-c", src, "-o", glue])
    default_sched = []
    for v, o, j in jobs:
        err = j.communicate()[1]
        if j.returncode:
            # hipcc died on this instantiation (the iterative scheduler is experimental upstream): ONCE more with the conservative flag set (buildflags.variant_flags)
            if buildflags.call(common + buildflags.variant_flags(v, machine_licm=True) + ["-DPL_VARIANT=%d" % v, "-c", src, "-o", o]):
                sys.stderr.write(err.decode(errors="replace")[-4000:])
                raise RuntimeError("hipcc failed building the kernels of discretisation %r" % (grid,))
            default_sched.append(v)
    if gj.wait():
        raise RuntimeError("hipcc failed building the kernels of discretisation %r" % (grid,))
    tmp = lib + ".tmp%d" % os.getpid()                      # (a forced rebuild replaces the file atomically: a process that has the old one mapped keeps its inode)
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-Wl,-Bsymbolic", glue] + objs + ["-o", tmp])
    os.replace(tmp, lib)
    json.dump({"grid": list(grid), "variants": allv, "suffix": suffix, "machine_licm": bool(machine_licm), "extra_flags": list(extra_flags), "fallback_variants": default_sched,
               "src_hash": _sources_hash()}, open(manifest, "w"))
    for o in objs + [glue]:
        os.remove(o)
    return lib

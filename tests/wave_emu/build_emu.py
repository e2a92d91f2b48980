"""TEST INFRASTRUCTURE ONLY: builds tests/wave_emu/libpetlion_emu.so = the product's device + host source compiled with g++
against the lock-step wave emulator (tests/wave_emu/hip/hip_runtime.h).  Used by the `-m "not gpu"` tests to exercise the
*device source* on a machine without a GPU.  The package never loads this library."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
SRC = os.path.join(ROOT, "petlion.jl_amd", "csrc", "petlion_hip.hip")
OUT = os.path.join(HERE, "libpetlion_emu.so")


def build(force=False, variant=None, extra=(), tag=""):
    """variant=<id>: a developer's quick build holding that one model variant (libpetlion_emu_v<id>.so, ~20 s instead of ~2 min); the tests use the full library"""
    out = OUT if variant is None else OUT[:-3] + "_v%d%s.so" % (variant, tag)            # (extra / tag: experiment builds, e.g. -DPL_OCC2)
    deps = [SRC] + [os.path.join(os.path.dirname(SRC), f) for f in os.listdir(os.path.dirname(SRC)) if f.endswith((".h", ".hip"))]
    deps.append(os.path.join(HERE, "hip", "hip_runtime.h"))
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    # -DPL_EXP_BRANCHY_PHI: the form of the register-resident BDF history update that the product's thermal variants 4 / 7 are built with since the end of r04
    # (petlion.jl_amd/buildflags.py THERMAL_SRC; it only exists for the models that keep history orders in registers, i.e. the thermal ones)
    cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-x", "c++", "-I", HERE, SRC, "-o", out, "-Wno-unused-variable", "-ldl", "-DPL_EXP_BRANCHY_PHI"]
    if variant is not None:
        cmd.append("-DPL_VARIANT=%d" % variant)
    cmd += list(extra)
    subprocess.check_call(cmd)
    return out


def build_grid(grid, variants, force=False):
    """emulator build of a grid library (petlion.jl_amd/grids.py builds the real one with hipcc): the listed variants of csrc/variant_tu.hip compiled for another
    discretisation, to be registered into libpetlion_emu.so with plh_register_grid_library"""
    import sys
    sys.path.insert(0, ROOT)
    import pkgload
    grids = pkgload.load().grids
    tag, defs = grids.defines(grid)
    out = os.path.join(HERE, "libplh_emu_%s_%s.so" % (tag, "_".join(str(v) for v in sorted(variants))))
    src = os.path.join(os.path.dirname(SRC), "variant_tu.hip")
    deps = [os.path.join(os.path.dirname(SRC), f) for f in os.listdir(os.path.dirname(SRC)) if f.endswith((".h", ".hip"))] + [os.path.join(HERE, "hip", "hip_runtime.h")]
    if not force and os.path.exists(out) and all(os.path.getmtime(out) >= os.path.getmtime(d) for d in deps):
        return out
    common = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-x", "c++", "-I", HERE, "-Wno-unused-variable", "-DPL_EXP_BRANCHY_PHI"] + defs      # (the source switches of the product's thermal builds: petlion.jl_amd/buildflags.py)
    objs = []
    jobs = []
    for v in sorted(variants):
        o = out + ".v%d.o" % v
        objs.append(o)
        jobs.append(subprocess.Popen(common + ["-DPL_VARIANT=%d" % v, "-c", src, "-o", o]))
    glue = out + ".glue.o"
    jobs.append(subprocess.Popen(common + ["-DPL_GRID_GLUE", "-c", src, "-o", glue]))
    if any(j.wait() for j in jobs):
        raise RuntimeError("g++ failed")
    subprocess.check_call(["g++", "-shared", "-fPIC", "-Wl,-Bsymbolic", glue] + objs + ["-o", out])
    for o in objs + [glue]:
        os.remove(o)
    return out


if __name__ == "__main__":
    print(build(force=True))

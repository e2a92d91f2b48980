#!/usr/bin/env python3
"""Builds (here, without a GPU: hipcc cross-compiles gfx950) the closure libraries bench.py's `general_path` attaches for the BASELINE configs, so that they travel to the GPU box
with the tree instead of costing GPU-minutes there.  The protocols are traced on a model of the test-only emulator build (tracing needs a handle for the state layout; the
programs and their digest do not depend on which build made them); the library is the product's: csrc/variant_tu.hip through hipcc.
   python tools/build_bench_closures.py [C2 C3 C4 C5]"""
import os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emu"))
import pkgload, build_emu
import bench
pkg = pkgload.load()
emu = build_emu.build()


def one(name):
    c = bench.CONFIGS[name]
    kw = dict(c["model"]); cath = kw.pop("cathode")
    p = pkg.petlion({"LCO": pkg.LCO, "NMC": pkg.NMC}[cath], _lib_path=emu, **kw)
    inp = getattr(pkg.configs, name.lower())(p, 4)
    out = []
    for case in bench.closure_cases(p, inp):
        runs, _ = pkg.make_protocol(p, [case] + inp["protocol"][1:], 4)
        out.append(pkg.closure_lib.library(p, runs))
    return name, out


if __name__ == "__main__":
    names = sys.argv[1:] or ["C2", "C3", "C4", "C5"]
    with ThreadPoolExecutor(2) as ex:
        for name, libs in ex.map(one, names):
            print(name, libs)

#!/usr/bin/env python3
"""A reproducible miscompiled instantiation (VERDICT r04 "next" 4 asked for one) -- and what made it go away.

AT COMMIT b316e8d: variant 16 (LCO, temperature, reference-order rows) compiled with the thermal flag set (early inlining, no DS merging, MachineLICM off) and
`-mllvm -amdgpu-sched-strategy=iterative-ilp`.  Its TABLE-input instantiation k_integrate<M16, GF_STOPS | GF_FUNC> starts every run with a garbage SOC and first step size (exit
flag 3 after 9 steps at t = 1.3e-16 s; the kernel self-test refuses the library) while the other six instantiations of the same translation unit, and all seven of the other
sixteen variants, reproduce their plain kernels; LLVM's default scheduler compiles it correctly, and so does the iterative one WITHOUT the no-DS-merging target feature.  The same
instantiation of variant 4 failed the same way under `max-ilp` and under `-misched-cluster=0` (tools/experiments/sched_search.py), and r03 had seen it on the isothermal model: what
the instantiation had that no other has was an OUTER do-while around consistent initialisation + step loop (check_reinitialization!), i.e. every per-run scalar carried around
two nested loops.  clang finds no uninitialised variable in the source (-Wuninitialized -Wsometimes-uninitialized -Wconditional-uninitialized), the wave emulator and the
default scheduler run the same source correctly: a code-generation failure at the register allocator's limit, not undefined behaviour.

SINCE THE NEXT COMMIT the step loop is ONE loop (the initialisation block is called at the top of an iteration when a function input asks for it): the same three command lines
all pass -- which is what this tool shows on the current tree.  To see the failure: `git checkout b316e8d -- petlion.jl_amd/csrc/dfn_integrate.h`, build, run.

    python tools/experiments/miscompile_repro.py build      (here; ~6 min: one translation unit, three command lines)
    python tools/experiments/miscompile_repro.py run        (on the GPU box)

hipcc 7.2.26015, AMD clang 22.0.0git 7b800a194662, gfx950.  Measured r05 (gpurun, one MI355X): before the restructuring "iterative-ilp": passed without the target feature,
FAILED with it ("the table input instantiation of lco_thermal does not reproduce the plain kernel"); default scheduler: passed.  After: all three pass."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
EXP = os.path.join(ROOT, "petlion.jl_amd", "_exp")
LIBS = {"iterative-ilp (the flag table)": os.path.join(EXP, "libplh_repro16_iter.so"), "default scheduler": os.path.join(EXP, "libplh_repro16_default.so"),
        "iterative-ilp, the no-DS-merging feature given twice (r05i)": os.path.join(EXP, "libplh_repro16_iterx.so")}
XCLANG = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]      # (part of the thermal flag set until r05i; the compiler says it ignores it)

CHILD = r'''
import sys
sys.path.insert(0, %r)
import numpy as np, torch, pkgload
pkg = pkgload.load()
try:
    p = pkg.petlion(pkg.LCO, temperature=True, precision="f64_reforder")          # (a library that is not the validated binary runs the kernel self-test here)
except RuntimeError as e:
    print("   kernel self-test:", str(e)[:300]); sys.exit(0)
print("   kernel self-test: passed")
Th = np.tile(p.theta_vector(), (32, 1))
def dirty():
    # what the failing kernel reads is a register (spill slot) it never wrote: the outcome depends on what the PREVIOUS kernel on the SIMD left there.  In the GPU suite that was
    # another variant's kernel; in a fresh process the plain kernel of the same variant leaves the right numbers behind and the bug hides.  An fp64 GEMM dirties every register file.
    a = torch.randn(4096, 4096, device="cuda", dtype=torch.float64); b = (a @ a).sum().item()
dirty()
base = pkg.simulate_ensemble(p, Th, [{"I": -1.0, "tf": 300.0}], SOC=1.0)
o = pkg.Opts(); o.tstops = [1e7]
stops = pkg.simulate_ensemble(p, Th, [{"I": -1.0, "tf": 300.0}], SOC=1.0, opts=o)       # (the order of tests/test_gpu_parity.py::test_every_kernel_instantiation_of_every_variant)
tab = pkg.simulate_ensemble(p, Th, [{"I": ([0.0, 1e7], [-1.0, -1.0]), "tf": 300.0}], SOC=1.0)
bad = int((tab.run_info["flag"][:, 0] != base.run_info["flag"][:, 0]).sum())
print("   cells whose table-input run ends with another flag than the plain run: %%d of %%d" %% (bad, len(Th)))
print("   plain kernel :", base.run_info[0, 0])
print("   table input  :", tab.run_info[0, 0])
print("   equal        :", bool(np.array_equal(base.run_info["flag"], tab.run_info["flag"]) and np.abs(base.run_info["t_end"] - tab.run_info["t_end"]).max() == 0.0))
''' % ROOT

if __name__ == "__main__":
    if sys.argv[1] == "build":
        import __graft_entry__ as g
        os.makedirs(EXP, exist_ok=True)
        os.environ["PETLION_SCHED_ALL"] = "1"
        print(g.build_hip(lib=LIBS["iterative-ilp (the flag table)"], variants=[16], extra_flags=["-DPL_REPRO16"]))
        print(g.build_hip(lib=LIBS["iterative-ilp, the no-DS-merging feature given twice (r05i)"], variants=[16], extra_flags=["-DPL_REPRO16"] + XCLANG))
        os.environ.pop("PETLION_SCHED_ALL")
        print(g.build_hip(lib=LIBS["default scheduler"], variants=[16], extra_flags=["-DPL_REPRO16=0"]))
        import pkgload
        bf = pkgload.load().buildflags
        os.environ["PETLION_SCHED_ALL"] = "1"
        print("failing object: hipcc --offload-arch=gfx950 -std=c++17 -fPIC " + " ".join(bf.variant_flags(16)) + " -DPL_VARIANT=16 -c petlion.jl_amd/csrc/variant_tu.hip")
    else:
        for name, lib in LIBS.items():
            print(name)
            subprocess.call([sys.executable, "-c", CHILD], env=dict(os.environ, PETLION_HIP_LIB=lib))

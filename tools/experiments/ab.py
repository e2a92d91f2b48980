#!/usr/bin/env python3
"""Same-box A/B of experiment builds (tools/experiments/): `build` here (no GPU: hipcc cross-compiles; the libraries travel with the snapshot), `run` on the GPU box.
   python tools/experiments/ab.py build NAME[,NAME...]        python tools/experiments/ab.py run NAME[,NAME...] [--reps K] [--tests]
A build = (variants, extra flags, perf configs, pytest -k).  `base` entries compile the tree as it is; the others add -D switches that select an alternative code path."""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
LATE = ["-DPL_DEV=__device__ inline", "-mllvm", "-amdgpu-function-calls=false"]
EARLY = ["-DPL_DEV=__device__ __forceinline__"]
NOLSO = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]
NOVEC = ["-mllvm", "-amdgpu-load-store-vectorizer=0"]
# (r04 note: -DPL_EXP_UNI, -DPL_EXP_SWEEP_REGS, -DPL_EXP_SEI_PRED and -DPL_EXP_ALL_PRED selected code paths that have since been adopted (sweep blocks in registers, SEI predictor
#  in registers) or removed (uni(), thermal predictor in registers): the entries that name them are the record of what was measured -- gpurun_out/r04k..r04m, DESIGN.md 5a -- and
#  now build the tree as it is)
NOLICM = ["-mllvm", "-disable-machine-licm"]          # r04: MachineLICM hoists (among others) the copies of the exp / log polynomial coefficients out of the step loop
UNI = ["-DPL_EXP_UNI"]                                # r04: wave-uniform doubles of the step loop handed to scalar registers (uni(), dfn_cell.h)
BUILDS = {
    "th_prod": ([4], EARLY + NOLSO + NOVEC, "c3", "c3_thermal"),
    "th_nolicm": ([4], EARLY + NOLSO + NOVEC + NOLICM, "c3", "c3_thermal"), "th_uni": ([4], EARLY + NOLSO + NOVEC + UNI, "c3", "c3_thermal"),
    "th_uni_nolicm": ([4], EARLY + NOLSO + NOVEC + UNI + NOLICM, "c3", "c3_thermal"),
    "iso_nolicm": ([0], LATE + NOLICM, "c2 c4", "c2_1024 or evaluators"), "iso_uni": ([0], LATE + UNI, "c2 c4", "c2_1024 or evaluators"),
    "iso_uni_nolicm": ([0], LATE + UNI + NOLICM, "c2 c4", "c2_1024 or evaluators"),
    "sei_pred_nolicm": ([3], LATE + UNI + NOLICM + ["-DPL_EXP_SEI_PRED"], "c5", "c5_nmc_sei"),
    "th_late_nolicm": ([4], LATE + NOLSO + NOVEC + UNI + NOLICM, "c3", "c3_thermal"), "iso_early_nolicm": ([0], EARLY + UNI + NOLICM, "c2 c4", "c2_1024 or evaluators"),
    "th_merge_nolicm": ([4], EARLY + UNI + NOLICM, "c3", "c3_thermal"),          # (DS merging back on, now that 16-byte register tuples are no longer scarce)
    # (the sweep's blocks in registers: adopted in the source after th_sweepregs, +2.5 %) combinations on top of it:
    "th_new": ([4], EARLY + NOLSO + NOVEC + NOLICM, "c3", "c3_thermal"), "th_new_merge": ([4], EARLY + NOLICM, "c3", "c3_thermal"),
    "th_new_late_merge": ([4], LATE + NOLICM, "c3", "c3_thermal"), "th_new_late": ([4], LATE + NOLSO + NOVEC + NOLICM, "c3", "c3_thermal"),
    "th_new_pred": ([4], EARLY + NOLSO + NOVEC + NOLICM + ["-DPL_EXP_ALL_PRED"], "c3", "c3_thermal"),          # predictor of the step in registers for the thermal model too
    "th_branchy_nolicm": ([4], EARLY + NOLSO + NOVEC + NOLICM + ["-DPL_EXP_BRANCHY_PHI"], "c3", "c3_thermal"), "th_branchy_merge": ([4], EARLY + NOLICM + ["-DPL_EXP_BRANCHY_PHI"], "c3", "c3_thermal"),
    "th_fence": ([4], EARLY + NOLSO + NOVEC + NOLICM + ["-DPL_PHASE_FENCES"], "c3", "c3_thermal"), "iso_fence": ([0], LATE + NOLICM + ["-DPL_PHASE_FENCES"], "c2 c4", "c2_1024 or evaluators"),
    "sei_fence": ([3], LATE + NOLICM + ["-DPL_PHASE_FENCES"], "c5", "c5_nmc_sei"), "iso_new": ([0], LATE + NOLICM, "c2 c4", "c2_1024 or evaluators"), "sei_new": ([3], LATE + NOLICM, "c5", "c5_nmc_sei"),
    "th_fb": ([4], EARLY + NOLSO + NOVEC + NOLICM + ["-DPL_PHASE_FENCES", "-DPL_EXP_BRANCHY_PHI"], "c3", "c3_thermal"),          # = the production thermal build since the end of r04
    "lgm_fb": ([14], EARLY + NOLSO + NOVEC + NOLICM + ["-DPL_PHASE_FENCES", "-DPL_EXP_BRANCHY_PHI"], "", ""),          # variant 14 with the flags of variants 4 / 7 (selftest_delta.py)
    "lgm_nolicm": ([14], EARLY + NOLSO + NOVEC + NOLICM, "", ""), "lgm_licm": ([14], EARLY + NOLSO + NOVEC, "", ""),          # (tools/dbg/selftest_delta.py)
    "sei_pred2": ([3], LATE + NOLICM + ["-DPL_EXP_SEI_PRED"], "c5", "c5_nmc_sei"), "iso_early2": ([0], EARLY + NOLICM, "c2 c4", "c2_1024 or evaluators"),
    "sei_nolicm": ([3], LATE + NOLICM, "c5", "c5_nmc_sei"), "sei_uni_nolicm": ([3], LATE + UNI + NOLICM, "c5", "c5_nmc_sei"),
    "th_base": ([4], EARLY, "c3", "c3_thermal"),
    "th_branchy": ([4], EARLY + ["-DPL_EXP_BRANCHY_PHI"], "c3", "c3_thermal"),          # r03's branching update of the register-resident BDF history
    "iso_base": ([0], LATE, "c2 c4", "c2_1024 or evaluators"),
    "sei_base": ([3], LATE, "c5", "c5_nmc_sei"),
    # the LDS array is shared by the four cells of a CU: ds_read2_b64 / ds_read2st64_b64 cost 8 array cycles where two ds_read_b64 cost 2 + 2 (MI355X_MICROARCH.md).  nolso = the
    # machine-level merging of DS accesses off (subtarget feature load-store-opt); novec = also the IR load/store vectoriser (no ds_read_b128 either)
    "th_nolso": ([4], EARLY + NOLSO, "c3", "c3_thermal"), "iso_nolso": ([0], LATE + NOLSO, "c2 c4", "c2_1024 or evaluators"), "sei_nolso": ([3], LATE + NOLSO, "c5", "c5_nmc_sei"),
    "th_novec": ([4], EARLY + NOLSO + NOVEC, "c3", "c3_thermal"), "iso_novec": ([0], LATE + NOLSO + NOVEC, "c2 c4", "c2_1024 or evaluators"),
    # round-3 final source (built from a git worktree of that commit into the same _exp/ directory: `git worktree add /tmp/r03 <commit>`; build only lists them for `run`)
    "sei_pred": ([3], LATE + ["-DPL_EXP_SEI_PRED"], "c5", "c5_nmc_sei"),          # r04: predictor of the step in registers for the SEI models (PRED_REGS)
    "th_r03": ([4], None, "c3", "c3_thermal"), "iso_r03": ([0], None, "c2 c4", "c2_1024 or evaluators"), "sei_r03": ([3], None, "c5", "c5_nmc_sei"),
}
# r06: the instruction / latency diet of the isothermal kernels (variant 0, production flag table): everything on, and one switch off at a time
BUILDS.update({
    "r06_all": ([0], [], "c2 c4", "c2_1024 or evaluators"), "r06_noflat": ([0], ["-DPL_NO_FLAT"], "c2 c4", "c2_1024 or evaluators"),
    "r06_novpad": ([0], ["-DPL_NO_VPAD"], "c2 c4", "c2_1024 or evaluators"), "r06_ieeediv": ([0], ["-DPL_IEEE_DIV"], "c2 c4", "c2_1024 or evaluators"),
    "r06_none": ([0], ["-DPL_NO_FLAT", "-DPL_NO_VPAD", "-DPL_IEEE_DIV"], "c2 c4", "c2_1024 or evaluators"),
    "r06c_all": ([0], [], "c2 c4", "c2_1024 or evaluators"), "r06c_lane1": ([0], ["-DPL_LANE_OPAQUE=1"], "c2 c4", "c2_1024 or evaluators"),
    "r06d_all": ([0], [], "c2 c4", "c2_1024 or evaluators"), "r06d_nocsdpp": ([0], ["-DPL_NO_CSDPP"], "c2 c4", "c2_1024 or evaluators"),
    "th_r06f": ([4], [], "c3", "c3_thermal"), "th_r06f_nothrowb": ([4], ["-DPL_NO_THROWB"], "c3", "c3_thermal"),
    "th_r06g": ([4], [], "c3", "c3_thermal"), "th_r06g_nostride2": ([4], ["-DPL_NO_STRIDE2T"], "c3", "c3_thermal"),          # one level of recursive doubling in the 4x4 sweeps of a solve
    "r05": ([0], None, "c2 c4", "c2_1024 or evaluators"),          # the r05 library as committed (copied to _exp/libplh_r05.so)
    "r06v": ([0], None, "c2 c4", "c2_1024 or evaluators"),
    "r06g_both": ([0], None, "c2 c4", "c2_1024 or evaluators"), "r06g_sei": ([3], None, "c5", "c5_nmc_sei"), "r06g_sei_ocml": ([3], None, "c5", "c5_nmc_sei"),          # + the sweep blocks formed together
    "r06f_exp": ([0], None, "c2 c4", "c2_1024 or evaluators"), "r06f_ocml": ([0], None, "c2 c4", "c2_1024 or evaluators"),          # lean exp / expm1 in the node passes (built in a worktree)
    "th_r06h_exp": ([4], None, "c3", "c3_thermal"), "th_r06h_ocml": ([4], None, "c3", "c3_thermal"),         # the validated r06 library (copied to _exp/libplh_r06v.so)
    "r06e_base": ([0], [], "c2 c4", "c2_1024 or evaluators"), "r06e_sw": ([0], ["-DPL_SWCACHE"], "c2 c4", "c2_1024 or evaluators"),          # blocks of the doubled sweeps kept in registers from the factorisation
})
for k in list(BUILDS):          # every build also exists with the previous-point copy kept (r03) or dropped
    pass


def lib_of(name):
    return os.path.join(ROOT, "petlion.jl_amd", "_exp", "libplh_%s.so" % name)


def build(name):
    v, flags, _, _ = BUILDS[name]
    if flags is None:
        return lib_of(name)
    return g.build_hip(extra_flags=flags, lib=lib_of(name), variants=v)


if __name__ == "__main__":
    what, names = sys.argv[1], sys.argv[2].split(",")
    os.makedirs(os.path.join(ROOT, "petlion.jl_amd", "_exp"), exist_ok=True)
    if what == "build":
        g._flags()                               # (import the package once before the pool: pkgload.load() is not re-entrant across threads)
        with ThreadPoolExecutor(4) as ex:
            print(list(ex.map(build, names)))
    else:
        reps = sys.argv[sys.argv.index("--reps") + 1] if "--reps" in sys.argv else "3"
        for rnd in range(2):                     # A B A B: the boxes drift by a percent within a minute
            for nm in names:
                env = dict(os.environ, PETLION_HIP_LIB=lib_of(nm))
                print("=== %s (round %d)" % (nm, rnd), flush=True)
                subprocess.call([sys.executable, os.path.join(ROOT, "tools", "perf_configs.py")] + BUILDS[nm][2].split() + ["--reps", reps], env=env)
        if "--tests" in sys.argv:
            for nm in names:
                env = dict(os.environ, PETLION_HIP_LIB=lib_of(nm))
                print("=== tests %s" % nm, flush=True)
                subprocess.call([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k", BUILDS[nm][3], "-p", "no:cacheprovider"], env=env)

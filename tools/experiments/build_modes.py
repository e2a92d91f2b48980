#!/usr/bin/env python3
"""Experiment builds of single variants under different inlining modes / flags, timed on the GPU box (it has the same hipcc):
     python tools/experiments/build_modes.py            (builds in parallel, then runs tools/perf_configs.py and the matching parity tests for each)
Modes:  late   = device functions `inline`, everything inlined by the AMDGPU always-inline pass at the end (-mllvm -amdgpu-function-calls=false)
        early  = device functions __forceinline__ (inlined by the AlwaysInliner before the optimisation pipeline)
        fcall  = early + the thermal factorisation as a real function (-DPL_FACTOR_CALL)"""
import os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
LATE = ["-DPL_DEV=__device__ inline", "-mllvm", "-amdgpu-function-calls=false"]
EARLY = ["-DPL_DEV=__device__ __forceinline__"]
BUILDS = [  # tag, variants, {variant: opt}, extra flags, perf config, pytest -k
    ("iso_late_O3", [0], {0: "-O3"}, LATE, "c2 c4", "c2_1024 or evaluators"),
    ("iso_early_O3", [0], {0: "-O3"}, EARLY, "c2 c4", "c2_1024 or evaluators"),
    ("th_late_O2", [4], {4: "-O2"}, LATE, "c3", "c3_thermal"),
    ("th_early_O2", [4], {4: "-O2"}, EARLY, "c3", "c3_thermal"),
    ("th_early_O3", [4], {4: "-O3"}, EARLY, "c3", "c3_thermal"),
    ("th_fcall_O2", [4], {4: "-O2"}, EARLY + ["-DPL_FACTOR_CALL"], "c3", "c3_thermal"),
    ("th_fcall_O3", [4], {4: "-O3"}, EARLY + ["-DPL_FACTOR_CALL"], "c3", "c3_thermal"),
    # fp64 division without the IEEE scale / fixup sequence (approximate-function lowering: rcp + two Newton steps + one correction); th_* the same for the thermal kernel
    ("iso_afn", [0], {0: "-O3"}, LATE + ["-fapprox-func"], "c2 c4", "c2_1024 or evaluators"),
    ("iso_afn_rcp", [0], {0: "-O3"}, LATE + ["-fapprox-func", "-freciprocal-math"], "c2 c4", "c2_1024 or evaluators"),
    ("th_afn", [4], {4: "-O3"}, EARLY + ["-fapprox-func"], "c3", "c3_thermal"),
    # the r01 failure mode -- device functions left to the inliner's heuristics (real s_swappc calls inside k_integrate)
    ("iso_calls_O2", [0], {0: "-O2"}, ["-DPL_DEV=__device__ inline"], "c2", "c2_1024 or evaluators or consistent"),
    ("iso_calls_O3", [0], {0: "-O3"}, ["-DPL_DEV=__device__ inline"], "c2", "c2_1024 or evaluators or consistent"),
    ("iso_calls_O2_noattr", [0], {0: "-O2"}, ["-DPL_DEV=__device__ inline", "-DPL_NO_WAVES_ATTR"], "c2", "c2_1024 or evaluators or consistent"),
    ("iso_calls_O2_noinl", [0], {0: "-O2"}, ["-DPL_DEV=__device__ __attribute__((noinline))"], "c2", "c2_1024 or evaluators or consistent"),
    # r03: same-box A/B baselines and the cost of the per-step previous-point copy (upper bound: the copy removed)
    ("sei_late_O3", [3], {3: "-O3"}, LATE, "c5", "c5_nmc_sei"),
    # r03: recursive doubling in the block sweeps of the solve (default) against the one-lane recurrence
    ("iso_stride1", [0], {0: "-O3"}, LATE + ["-DPL_EXP_NO_STRIDE2"], "c2 c4", "evaluators"),
    ("sei_stride1", [3], {3: "-O3"}, LATE + ["-DPL_EXP_NO_STRIDE2"], "c5", "evaluators"),
    # r03: floating-point flags that keep IEEE results for finite data (no reassociation): does the compiler find anything?
    ("iso_fz", [0], {0: "-O3"}, LATE + ["-fno-signed-zeros", "-fno-trapping-math"], "c2 c4", "evaluators"),
    ("sei_fz", [3], {3: "-O3"}, LATE + ["-fno-signed-zeros", "-fno-trapping-math"], "c5", "evaluators"),
    ("th_fz", [4], {4: "-O3"}, EARLY + ["-fno-signed-zeros", "-fno-trapping-math"], "c3", "evaluators"),
    ("iso_noprev", [0], {0: "-O3"}, LATE + ["-DPL_EXP_NO_PREV"], "c2 c4", "evaluators"),
    ("th_noprev", [4], {4: "-O3"}, EARLY + ["-DPL_EXP_NO_PREV"], "c3", "evaluators"),
    ("sei_noprev", [3], {3: "-O3"}, LATE + ["-DPL_EXP_NO_PREV"], "c5", "evaluators"),
]
sel = sys.argv[1:]
builds = [b for b in BUILDS if not sel or b[0] in sel]
os.makedirs(os.path.join(ROOT, "petlion.jl_amd", "_exp"), exist_ok=True)


def build(b):
    tag, variants, opts, flags, _, _ = b
    lib = os.path.join(ROOT, "petlion.jl_amd", "_exp", "libplh_%s.so" % tag)
    g.build_hip(force=True, extra_flags=flags, lib=lib, variants=variants, opt=opts)
    return lib


with ThreadPoolExecutor(4) as ex:
    libs = list(ex.map(build, builds))
for b, lib in zip(builds, libs):
    env = dict(os.environ, PETLION_HIP_LIB=lib)
    print("=== %s" % b[0], flush=True)
    subprocess.call([sys.executable, os.path.join(ROOT, "tools", "perf_configs.py")] + b[4].split() + ["--reps", "3"], env=env)
    subprocess.call([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-q", "-x", "-m", "gpu", "-k", b[5], "-p", "no:cacheprovider"], env=env)

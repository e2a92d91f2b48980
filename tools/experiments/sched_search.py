#!/usr/bin/env python3
"""r05: does another instruction scheduler help a kernel that runs ONE wavefront per SIMD (nothing hides a latency but the schedule itself)?

    python tools/experiments/sched_search.py build          (here: hipcc cross-compiles; the libraries travel with the snapshot)
    python tools/experiments/sched_search.py run            (on the GPU box: bench.py --config C2 / C4 with variant 0, C3 with variant 4, per library)

One experiment library per flag set (variants 0 = LCO isothermal and 4 = LCO thermal, production flags + the set).  Prints per library the bench value and the kernel time; the
result decides whether a set joins petlion.jl_amd/buildflags.py."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
EXP = os.path.join(ROOT, "petlion.jl_amd", "_exp")
M = "-mllvm"
# batch 2 (after iterative-ilp joined the flag table: "base2" is the production flag set); a value that starts with "sched=" replaces the strategy
SETS2 = {
    "base2": [],
    "iter_minreg": ["sched=iterative-minreg"],
    "iter_maxocc": ["sched=iterative-maxocc"],
    "llvm_default": ["sched="],
    "no_cluster": [M, "-misched-cluster=0"],
    "no_postsched": [M, "-enable-post-misched=0"],
    "no_lowocc_resched": [M, "-amdgpu-disable-clustered-low-occupancy-reschedule"],
    "aa_sched": [M, "-enable-aa-sched-mi"],
    "antidep_all": [M, "-break-anti-dependencies=all"],
}
# batch 3: do the switches measured under the default scheduler keep their sign under the iterative one?  ("PL_DEV=" in a set makes build_hip take the set as the
# WHOLE flag list of both variants: tools say which config is meaningful)
EARLY = ["-DPL_DEV=__device__ __forceinline__"]
LATE = ["-DPL_DEV=__device__ inline", M, "-amdgpu-function-calls=false"]
NOLICM = [M, "-disable-machine-licm"]
ITER = [M, "-amdgpu-sched-strategy=iterative-ilp"]
NODS = [M, "-amdgpu-load-store-vectorizer=0", "-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]
SETS3 = {
    "base3": [],
    "iso_early": EARLY + NOLICM + ITER,                                   # C2 / C4
    "iso_fences": ["-DPL_PHASE_FENCES"],                                 # C2 / C4 (added to the table's flags)
    "iso_branchy": ["-DPL_EXP_BRANCHY_PHI"],                             # C2 / C4
    "iso_licm_on": LATE + ITER,                                          # C2 / C4
    "iso_nods": LATE + NOLICM + ITER + NODS,                             # C2 / C4
    "th_no_fences": EARLY + NODS + ["-DPL_EXP_BRANCHY_PHI"] + NOLICM + ITER,          # C3
    "th_no_branchy": EARLY + NODS + ["-DPL_PHASE_FENCES"] + NOLICM + ITER,            # C3
    "th_ds_merge": EARLY + ["-DPL_PHASE_FENCES", "-DPL_EXP_BRANCHY_PHI"] + NOLICM + ITER,   # C3
    "th_licm_on": EARLY + NODS + ["-DPL_PHASE_FENCES", "-DPL_EXP_BRANCHY_PHI"] + ITER,      # C3
}
# batch 4: the isothermal candidates of batch 3 once more, with variant 3 (NMC + SEI: C5) in the library
SETS4 = {
    "base4": [],
    "iso4_nods": LATE + NOLICM + ITER + NODS,
    "iso4_early_nods": EARLY + NOLICM + ITER + NODS,
    "iso4_nolsv": LATE + NOLICM + ITER + [M, "-amdgpu-load-store-vectorizer=0"],
    "iso4_nolso": LATE + NOLICM + ITER + ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt"],
}
NOLSO = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]
SETS4.update({
    "iso5_early_nolso": EARLY + NOLICM + ITER + NOLSO,
    "iso5_nolso_fences": LATE + NOLICM + ITER + NOLSO + ["-DPL_PHASE_FENCES"],
    "iso5_nolso_branchy": LATE + NOLICM + ITER + NOLSO + ["-DPL_EXP_BRANCHY_PHI"],
})
VARIANTS = {n: [0, 3] for n in SETS4}
# batch 6: the thermal set with one of the two DS-merging switches at a time
TH = ["-DPL_PHASE_FENCES", "-DPL_EXP_BRANCHY_PHI"]
SETS6 = {
    "base6": [],
    "th6_nolso_only": EARLY + NOLSO + TH + NOLICM + ITER,
    "th6_nolsv_only": EARLY + [M, "-amdgpu-load-store-vectorizer=0"] + TH + NOLICM + ITER,
}
# batch 7: the thermal kernels' source switches on the isothermal ones, now that both share early inlining
SETS7 = {
    "base7": [],
    "iso7_fences": EARLY + NOLSO + NOLICM + ITER + ["-DPL_PHASE_FENCES"],
    "iso7_branchy": EARLY + NOLSO + NOLICM + ITER + ["-DPL_EXP_BRANCHY_PHI"],
    "iso7_both": EARLY + NOLSO + NOLICM + ITER + TH,
}
SETS4.update(SETS7)
VARIANTS.update({n: [0, 3] for n in SETS7})
# batch 8: two switches of the r02 search (tools/flag_search.sh) again, on top of the final table
SETS8 = {"base8": [], "t8_noslp": ["-fno-slp-vectorize"], "t8_ifcvt": [M, "-amdgpu-early-ifcvt=1"]}
SETS4.update(SETS8)
SETS4.update(SETS6)
VARIANTS.update({n: [4] for n in SETS6})
SETS = {
    "base": [],
    "max_ilp": [M, "-amdgpu-sched-strategy=max-ilp"],
    "max_clause": [M, "-amdgpu-sched-strategy=max-memory-clause"],
    "iter_ilp": [M, "-amdgpu-sched-strategy=iterative-ilp"],
    "bias0": [M, "-amdgpu-schedule-metric-bias=0"],
    "trackers": [M, "-amdgpu-use-amdgpu-trackers"],
    "relaxed": [M, "-amdgpu-schedule-relaxed-occupancy"],
    "no_unclustered": [M, "-amdgpu-disable-unclustered-high-rp-reschedule"],
}
SETS.update(SETS2)
SETS.update(SETS3)
SETS.update(SETS4)


def lib(name):
    return os.path.join(EXP, "libplh_sched_%s.so" % name)


def build(names):
    import __graft_entry__ as g
    os.makedirs(EXP, exist_ok=True)
    for n in names:
        try:
            fl = [f for f in SETS[n] if not f.startswith("sched=")]
            st = [f[6:] for f in SETS[n] if f.startswith("sched=")]
            os.environ.pop("PETLION_SCHED_STRATEGY", None)
            if st:
                os.environ["PETLION_SCHED_STRATEGY"] = st[0]
            print(n, g.build_hip(extra_flags=fl, lib=lib(n), variants=VARIANTS.get(n, [0, 4])), flush=True)
        except Exception as e:          # (a flag set the compiler rejects or dies on is a result too)
            print(n, "FAILED", repr(e)[:300], flush=True)


def run(names, configs):
    rows = []
    for n in names:
        if not os.path.exists(lib(n)):
            continue
        for c in configs:
            e = dict(os.environ, PETLION_HIP_LIB=lib(n))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--config", c, "--no-cpu-baseline"], env=e, capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                rows.append(dict(set=n, config=c, value=d["value"], kernel_ms=d["roofline"]["kernel_ms_avg"]))
            except Exception:
                rows.append(dict(set=n, config=c, error=(r.stderr or r.stdout)[-300:]))
            print(json.dumps(rows[-1]), flush=True)
    return rows


if __name__ == "__main__":
    names = [a for a in sys.argv[2:] if a in SETS] or list(SETS)
    if sys.argv[1] == "build":
        build(names)
    else:
        rows = run(names, [a for a in sys.argv[2:] if a.startswith("C")] or ["C2", "C4", "C3"])
        out = os.path.join(ROOT, "gpurun_out", "r05s"); os.makedirs(out, exist_ok=True)
        json.dump(rows, open(os.path.join(out, "sched_search.json"), "w"), indent=1)

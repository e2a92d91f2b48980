"""ONE table of hipcc flags for every translation unit of csrc/variant_tu.hip -- the built-in library (__graft_entry__.build_hip), the per-grid libraries compiled at first use
(grids.py) and the per-protocol closure libraries (closure_lib.py).

r04 had three copies that had drifted: the built-in variants dropped MachineLICM (+2 ... 21 %), the user-compiled libraries kept it, and a compiled closure then ran at 0.78 of the
plain kernel where the interpreted one ran at 0.88.  Since r05 every builder asks `variant_flags(v)`; a library that fails the kernel self-test on the user's
machine is rebuilt ONCE with `machine_licm=True` (the flag set every such library passed under in r04) and the fallback is reported (api._selftest_new_grid_library).

Measured history of each switch: DESIGN.md 5a.
"""
from __future__ import annotations

OPT = "-O3"
# every device function is inlined into its kernel; WHEN differs:
#   late  (the conservative fall-back set of the isothermal / SEI kernels; their production set until r05): functions are `inline`, the AMDGPU always-inline pass merges them after the
#         function-level optimisations
#   early (every kernel since r05; thermal kernels since r04): functions are __forceinline__, merged before the optimisation pipeline -- the thermal kernels then no longer depend on the optimisation level
# (r06: -DPL_OCML_EXP with every late-inlining build -- the library's exp / expm1 instead of pl_exp / pl_expm1 (dfn_cell.h).  hipcc 7.2 fails on variant 8 with them
#  ("Illegal instruction detected: Operand has incorrect register class.  V_CMP_NE_U32_e32 0, $src_shared_base"), whichever way their results are returned; the late-inlining
#  set is also the conservative fall-back, which should not carry this round's new code either)
LATE_INLINE = ["-mllvm", "-amdgpu-function-calls=false", "-DPL_OCML_EXP"]
EARLY_INLINE = ["-DPL_DEV=__device__ __forceinline__"]
# MachineLICM off (r04): the pre-RA loop-invariant code motion hoists whatever is invariant in the ONE step loop every device function is inlined into -- above all the register
# copies of the exp / log polynomial coefficients of the thermal node pass -- to the top of the kernel and keeps it live across all phases (thermal: 392 B/lane of scratch with
# it, 0 without; C3 +18.6 %, C5 +4.6 %, C2 / C4 +1.8 %)
NO_MACHINE_LICM = ["-mllvm", "-disable-machine-licm"]
# The backend's DS merging (SILoadStoreOptimizer) off, for every kernel: plain 8-byte DS accesses.  r04 had measured it together with the IR load/store vectoriser switch on the
# thermal kernels (+4 %); r05 under the iterative scheduler, one switch at a time: thermal C3 +2.5 % with the backend switch ALONE over both (the vectoriser switch alone: -0.5 %),
# isothermal see ISO_FLAGS below.
# (clang answers the -target-feature with "'-load-store-opt' is not a recognized feature for this target (ignoring feature)" -- the FRONT END does not know it; the string still
#  reaches the function's target-features and the AMDGPU backend does honour it (FeatureEnableLoadStoreOpt): the objects differ by 65 kB)
NO_LSO = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]
# compiler fences at the phase boundaries (+2.9 % C3) and the branching update of the register-resident BDF history (+2.5 %), thermal variants only
THERMAL_SRC = ["-DPL_PHASE_FENCES", "-DPL_EXP_BRANCHY_PHI"]
# r05: the GCN iterative scheduler with the ILP strategy instead of the default max-occupancy one.  Every kernel runs ONE wavefront per SIMD (waves_per_eu(1, 1); two for the small
# cells): occupancy is not a goal, and nothing hides an LDS / DPP latency but the instruction order itself.  Measured (tools/experiments/sched_search.py, one MI355X): C2 +1.8 %,
# C4 +2.7 %, C3 +5.2 %; max-ilp +0.4 % on the isothermal kernel and a MISCOMPILED thermal one (the kernel self-test caught it), max-memory-clause / metric-bias 0 / relaxed
# occupancy / no unclustered reschedule +-0.3 %, the AMDGPU register-pressure trackers -5 %
SCHED_DEFAULT = "iterative-ilp"


def sched_flags():
    """(experiment builds select another strategy through PETLION_SCHED_STRATEGY -- the option may only be given once on a command line; "" = LLVM's default)"""
    import os
    st = os.environ.get("PETLION_SCHED_STRATEGY", SCHED_DEFAULT)
    return ["-mllvm", "-amdgpu-sched-strategy=" + st] if st else []


# variants built with LLVM's default scheduler (none at present).  Until the step loop of cell_simulate lost its outer re-initialisation loop (dfn_integrate.h, r05) variant 16's
# table-input instantiation came out of the iterative scheduler with a garbage SOC and first step -- tools/experiments/miscompile_repro.py reproduces it at commit b316e8d and shows
# the same command line passing on the restructured source; the mechanism stays for the next instantiation that needs it
DEFAULT_SCHED_VARIANTS: set = set()
# variants that keep MachineLICM in the built-in library (none at present; the mechanism stays for the next register-allocation miscompile of one instantiation)
KEEP_MACHINE_LICM: set = set()


def is_thermal(v):
    from . import grids
    return grids.variant_table()[v][2] == "true"


# r05, isothermal / SEI kernels under the iterative scheduler (tools/experiments/sched_search.py batches 4 / 5, variants 0 and 3, one box): early inlining + the backend's DS
# merging off: C2 +3.6 %, C4 +3.1 %, C5 +5.0 % over late inlining (late + no DS merging +2.1 / +1.7 / +2.9 %; early + both vectoriser switches +2.1 / +2.4 / +3.1 %; the IR
# vectoriser switch ALONE with late inlining: a variant-0 library that fails the kernel self-test).  The conservative set (machine_licm=True) keeps late inlining.
ISO_FLAGS = EARLY_INLINE + NO_LSO
# variant 8 (quadratic solid diffusion, two cells per SIMD): hipcc 7.2 segfaults in code generation with early inlining under the iterative scheduler -- it keeps the r05 late-inlining set
ISO_LATE_VARIANTS = {8}


def variant_flags(v, machine_licm=False, thermal=None, default_sched=False):
    """compile flags (after the common ones: arch, std, -fPIC, warnings, grid / namespace defines) of variant v's translation unit.
    default_sched=True: LLVM's default scheduler instead of the iterative one.  machine_licm=True is the CONSERVATIVE set -- late inlining for the isothermal kernels, MachineLICM
    on, LLVM's default scheduler: what every builder retries ONE object with when hipcc dies on it (the iterative scheduler is marked experimental upstream: it segfaults on the
    (2, 2, 2, 10) grid's isothermal kernel and on variant 8 with early inlining, r05) and what the self-test fall-back build uses"""
    th = is_thermal(v) if thermal is None else thermal
    fl = (EARLY_INLINE + NO_LSO + THERMAL_SRC) if th else (list(LATE_INLINE) if (machine_licm or v in ISO_LATE_VARIANTS) else list(ISO_FLAGS))
    if not (machine_licm or v in KEEP_MACHINE_LICM):
        fl = fl + NO_MACHINE_LICM
    import os
    excluded = v in DEFAULT_SCHED_VARIANTS and not os.environ.get("PETLION_SCHED_ALL")          # (PETLION_SCHED_ALL=1: tools/experiments/miscompile_repro.py)
    return fl + ([] if (default_sched or machine_licm or excluded) else sched_flags()) + [OPT]


def table_repr():
    """what enters the build-identity hash (plh_build_info): the whole table"""
    return repr((OPT, LATE_INLINE, EARLY_INLINE, ISO_FLAGS, sorted(ISO_LATE_VARIANTS), NO_MACHINE_LICM, NO_LSO, THERMAL_SRC, sched_flags(), sorted(DEFAULT_SCHED_VARIANTS), sorted(KEEP_MACHINE_LICM)))


# ---- the compiler's answer to NO_LSO ----
# clang's front end prints "'-load-store-opt' is not a recognized feature for this target (ignoring feature)" several times per compile (no warning class: it cannot be switched
# off) although the backend honours the feature string -- hundreds of lines in a GPU test log that compiles grid / closure libraries.  Every builder
# starts hipcc through popen(): that one line is dropped from the compiler's stderr, everything else passes through; that the switch DID take effect is checked where it can be
# seen, in the object (tools/kernel_resources.py counts the merged DS operations of the variant-0 kernels into <lib>.resources.json: a build in which the backend ignored the
# feature has thousands of ds_read2 / ds_write2, one that honoured it a few dozen).
_NOISE = "is not a recognized feature for this target (ignoring feature)"


class popen:
    """subprocess.Popen of a compiler job whose stderr is passed on without the NO_LSO noise line; wait() / returncode / communicate() as the callers use them"""
    def __init__(self, cmd, echo=True, **kw):
        import subprocess
        import threading
        self.p = subprocess.Popen(cmd, stderr=subprocess.PIPE, **kw)
        self.kept, self.echo = [], echo
        self.t = threading.Thread(target=self._pump, daemon=True)
        self.t.start()

    def _pump(self):
        import sys
        for raw in self.p.stderr:
            ln = raw.decode(errors="replace")
            if _NOISE in ln:
                continue
            self.kept.append(ln)
            if self.echo:
                sys.stderr.write(ln)

    def wait(self):
        rc = self.p.wait()
        self.t.join()
        return rc

    def communicate(self):
        self.wait()
        return b"", "".join(self.kept).encode()

    @property
    def returncode(self):
        return self.p.returncode


def call(cmd, **kw):
    return popen(cmd, **kw).wait()

#!/usr/bin/env python3
"""Config C5's precision question on the device ("run in fp64 and fp32, report max rel. deviation of V(t), SOH(t_end), film(t_end)").
A pure fp32 integrator is not viable (tools/fp32_study.py: fp32 residual rounding moves states by 2e-4, time and SOH need fp64), so the candidate is
mixed precision: the LDS-resident factors of the Newton matrix in fp32, everything else fp64.  This script runs C5 (NMC + SEI, GITT) with the product
library and with a study build whose factors are rounded to fp32 where they are stored (-DPL_FP32_FACTORS), and compares.

usage (here):     python tools/fp32_factor_study.py build
      (GPU box):  python tools/fp32_factor_study.py run [n_cells] > gpurun_out/fp32_factor_study.md"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB32 = os.path.join(ROOT, "petlion.jl_amd", "libpetlion_hip_f32factors.so")
if sys.argv[1] == "build":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-pass-failed",
                           "-mllvm", "-amdgpu-function-calls=false", "-DPL_FP32_FACTORS", os.path.join(ROOT, "petlion.jl_amd", "csrc", "petlion_hip.hip"), "-o", LIB32])
    sys.exit(0)
sys.path.insert(0, ROOT)
import pkgload
pkg = pkgload.load()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024


def u(seed, n, k):      # the counter-based generator of the C5 spec
    x = (np.uint64(seed) ^ (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(k)) + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9); x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB); x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) / 2.0 ** 53


proto = []
for _ in range(20):
    proto += [{"I": 1.0, "tf": 180.0}, {"I": "rest", "tf": 7200.0}]
res = {}
for tag, lib in (("fp64", None), ("fp32 factors", LIB32)):
    p = pkg.petlion(pkg.NMC, aging="SEI", _lib_path=lib)
    keys = ["D_sp", "D_sn", "k_p", "k_n"]
    Th = pkg.theta_matrix(p, n, {k: p.θ[k] * 2.0 ** (2 * u(5, n, j) - 1) for j, k in enumerate(keys)})
    ens = pkg.simulate_ensemble(p, Th, proto, SOC=0.0, max_points=4096)
    res[tag] = ens
    ind = p.ind
a, b = res["fp64"], res["fp32 factors"]
ok = (a.run_info["flag"] >= 0).all(axis=1) & (b.run_info["flag"] >= 0).all(axis=1)
print("# C5 with fp32 Newton-matrix factors (device study)\n")
print("`python tools/fp32_factor_study.py run %d` on 1x MI355X: %d NMC + SEI cells, 20 x (1C pulse 180 s, rest 7200 s), C5 parameter jitter.\n" % (n, n))
print("| | fp64 factors (product) | fp32-rounded factors |\n|---|---|---|")
for f, nm in (("n_steps", "steps / cell"), ("n_newton", "Newton iterations / cell"), ("n_jac", "Jacobians / cell"), ("n_convfail", "Newton failures / cell"), ("n_errfail", "error-test failures / cell")):
    print("| %s | %.1f | %.1f |" % (nm, a.counters[f].mean(), b.counters[f].mean()))
print("| cells with a solver error | %d | %d |" % ((a.run_info["flag"] < 0).any(axis=1).sum(), (b.run_info["flag"] < 0).any(axis=1).sum()))
print("| exit flags equal to fp64 | - | %d of %d cells |" % ((a.run_info["flag"] == b.run_info["flag"]).all(axis=1).sum(), n))
dV = np.abs(a.run_info["V"][ok] - b.run_info["V"][ok]) / np.abs(a.run_info["V"][ok])
soh = lambda e: e.Y[:, ind["SOH"]][:, 0]
film = lambda e: e.Y[:, ind["film"]]
print("\nmax relative deviation over the %d cells that finish in both runs: V at the end of each of the 40 runs %.2e; SOH(t_end) %.2e (loss 1 - SOH: %.2e relative); "
      "film(t_end) %.2e; final state vector (section-wise, tests/parity.state_rel_err) %.2e." % (
          ok.sum(), dV.max(), (np.abs(soh(a) - soh(b))[ok] / soh(a)[ok]).max(), (np.abs(soh(a) - soh(b))[ok] / (1 - soh(a)[ok])).max(),
          (np.abs(film(a) - film(b))[ok] / np.abs(film(a)[ok]).max()).max(),
          max(__import__("tests.parity", fromlist=["x"]).state_rel_err(b.Y[i], a.Y[i]) for i in np.flatnonzero(ok)[:256])))
same = np.ones(n, bool)
for f in ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail"):
    same &= a.counters[f] == b.counters[f]
sm = same & ok
if sm.any():
    dVs = np.abs(a.run_info["V"][sm] - b.run_info["V"][sm]) / np.abs(a.run_info["V"][sm])
    print("\n%d of %d cells take bit-identical solver decisions in both runs (all counters equal); over those: V at run ends %.2e, SOH(t_end) %.2e, "
          "film(t_end) %.2e.  The larger figures above come from the cells whose step sequences decorrelate (reltol = 1e-3 level, like any perturbation)." % (
              sm.sum(), n, dVs.max(), (np.abs(soh(a) - soh(b))[sm] / soh(a)[sm]).max(), (np.abs(film(a) - film(b))[sm] / np.abs(film(a)[sm]).max()).max()))
print("\nkernel time: fp64 %.2f ms, fp32-rounded factors %.2f ms (same arithmetic, rounding only: this measures convergence, not an fp32 kernel)." % (a.kernel_ms, b.kernel_ms))

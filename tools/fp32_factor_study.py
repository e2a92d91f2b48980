#!/usr/bin/env python3
"""Config C5's precision question on the device ("run in fp64 and fp32, report max rel. deviation of V(t), SOH(t_end), film(t_end)").
A pure fp32 integrator is not viable (tools/fp32_study.py: fp32 residual rounding moves states by 2e-4, time and SOH need fp64), so the reduced-precision
instantiation is MIXED (plh_model_desc.precision = PLH_PREC_MIXED): the LDS-resident factors of the Newton matrix are stored in fp32, everything else fp64.
This script runs C5 (8(d) inputs: NMC + SEI, GITT, seed 5) with both instantiations of the product library and compares.

usage (GPU box):  python tools/fp32_factor_study.py run [n_cells] > gpurun_out/fp32_factor_study.md"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pkgload
pkg = pkgload.load()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024

res = {}
for tag, prec in (("fp64", "f64"), ("fp32 factors", "mixed")):
    p = pkg.petlion(pkg.NMC, aging="SEI", precision=prec)
    cfg = pkg.configs.c5(p, n)
    pkg.simulate_ensemble(p, cfg["theta"], cfg["protocol"], SOC=cfg["SOC"], max_points=cfg["max_points"])
    ens = pkg.simulate_ensemble(p, cfg["theta"], cfg["protocol"], SOC=cfg["SOC"], max_points=cfg["max_points"])
    res[tag] = ens
    res[tag + " lds"] = p.lds_bytes
    ind = p.ind
a, b = res["fp64"], res["fp32 factors"]
ok = (a.run_info["flag"] >= 0).all(axis=1) & (b.run_info["flag"] >= 0).all(axis=1)
print("# C5 with fp32 Newton-matrix factors (device study)\n")
print("`python tools/fp32_factor_study.py run %d` on 1x MI355X: %d NMC + SEI cells, 20 x (1C pulse 180 s, rest 7200 s), C5 parameter jitter.\n" % (n, n))
print("| | precision = f64 | precision = mixed (fp32 factor storage) |\n|---|---|---|")
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
print("\nkernel time: f64 %.2f ms, mixed %.2f ms (%.1f %% faster); LDS per cell: f64 %d B, mixed %d B." % (a.kernel_ms, b.kernel_ms, 100 * (a.kernel_ms / b.kernel_ms - 1), res["fp64 lds"], res["fp32 factors lds"]))

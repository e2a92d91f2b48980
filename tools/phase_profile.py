#!/usr/bin/env python3
"""Per-phase shader-cycle breakdown of k_integrate (profiling build libpetlion_hip_prof.so, -DPL_PHASE_TIMERS).
usage (GPU box): python tools/phase_profile.py [n_cells]"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pkgload
pkg = pkgload.load()
import __graft_entry__ as g
detail = 1 if "--detail" in sys.argv else 2 if "--detail2" in sys.argv else 3 if "--detail3" in sys.argv else 0
args = [a for a in sys.argv[1:] if not a.startswith("--")]
which = args[1] if len(args) > 1 else "iso"
variant = {"iso": 0, "thermal": 4, "sei": 3, "iso2": 13}[which]
lib = g.build_hip(extra_flags=["-DPL_PHASE_TIMERS"] + (["-DPL_PHASE_DETAIL=%d" % detail] if detail else []), lib=os.path.join(ROOT, "petlion.jl_amd", "_exp", "libplh_prof_%s%s.so" % (which, "_d%d" % detail if detail else "")), variants=[variant])
if "--build-only" in sys.argv:          # (here, without a GPU: the library travels to the GPU box with the snapshot)
    print(lib); sys.exit(0)
import torch
n = int(args[0]) if args else 1024
if which == "thermal":
    p = pkg.petlion(pkg.LCO, temperature=True, _lib_path=lib)
    kw = dict(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
    proto, soc = [dict(I=4.0, **kw), dict(dT="hold", **kw), dict(V="hold", **kw)], 0.0
elif which == "sei":
    p = pkg.petlion(pkg.NMC, aging="SEI", _lib_path=lib)
    proto, soc = [{"I": 1.0, "tf": 180.0}, {"I": "rest", "tf": 7200.0}] * 3, 0.0
else:
    p = pkg.petlion(pkg.LCO, waves_per_cell=2 if which == "iso2" else 1, _lib_path=lib)       # iso2: the timers are wave 0's
    proto, soc = [{"I": -1.0}], 1.0
Th = torch.from_numpy(pkg.theta_matrix(p, n)).cuda()
for _ in range(3):
    ens = pkg.simulate_ensemble(p, Th, proto, SOC=soc, device=True, max_points=1024)
names = ["residual", "jac+factor", "solve", "newton-vec", "step-ctl", "init", "output", "TOTAL"]
if detail == 2:
    names = ["solve a: particle partial solutions", "solve b: fold into node rhs", "solve c: block sweeps", "solve d,e: border + node-local back-substitution", "solve f: particle update", "residual: node pass", "residual: particle rows", "TOTAL"]
elif detail == 3:
    names = ["set_coeffs", "test_error", "complete_step", "get_solution", "SOC + save point", "check_stop", "previous point + rest", "TOTAL"]
elif detail:
    names = ["jac: node pass", "jac: particle rows", "factor: resolvents+collectors", "factor: node-local elimination", "factor: block sweep", "factor: control row+border", "-", "TOTAL"]
cyc = ens.counters["cyc"].astype(np.float64)
c = ens.counters
print("kernel %.3f ms for %d cells; per cell: steps %.0f res %.0f jac %.0f solves %.0f" % (ens.kernel_ms, n, c["n_steps"].mean(), c["n_res"].mean(), c["n_jac"].mean(), c["n_solve"].mean()))
tot = cyc[:, 7].mean()
for k, nm in enumerate(names):
    print("  %-50s %10.0f cyc  %5.1f%%" % (nm, cyc[:, k].mean(), 100 * cyc[:, k].mean() / tot))
if detail == 2:
    print("  per solve:", ", ".join("%s %.0f" % (nm.split(":")[0], cyc[:, k].mean() / c["n_solve"].mean()) for k, nm in enumerate(names[:5])), "; per residual:", ", ".join("%.0f" % (cyc[:, k].mean() / c["n_res"].mean()) for k in (5, 6)))
elif detail == 3:
    print("  per step:", ", ".join("%s %.0f" % (nm, cyc[:, k].mean() / c["n_steps"].mean()) for k, nm in enumerate(names[:7])))
elif detail:
    print("  per Jacobian refresh:", ", ".join("%s %.0f" % (nm, cyc[:, k].mean() / c["n_jac"].mean()) for k, nm in enumerate(names[:6])))
else:
  print("  per call: residual %.0f  jac+factor %.0f  solve %.0f" % (cyc[:, 0].mean() / (c["n_res"].mean() - c["n_jac"].mean()), cyc[:, 1].mean() / c["n_jac"].mean(), cyc[:, 2].mean() / c["n_newton"].mean()))
print("  flags per run:", [dict(zip(*np.unique(ens.run_info["flag"][:, k], return_counts=True))) for k in range(ens.run_info.shape[1])])

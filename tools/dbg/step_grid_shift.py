"""DESIGN.md section 5, link 2: the step grid of a C4 cell (cell 164*256) on the device vs the oracle: t_k ratio constant (= h0 ratio) through the start-up phase;
band of the oracle under fd_perturb.  usage: python tools/dbg/step_grid_shift.py  (wave-emulator build, CPU)"""
import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests')); sys.path.insert(0, os.path.join(ROOT, 'tests', 'wave_emu'))
import pkgload, parity, build_emu
from oracle import oracle as O
pkg = pkgload.load()
p = pkg.petlion(pkg.LCO, _lib_path=build_emu.build())
cell = 164 * 256
Th = pkg.configs.sweep_theta(p, np.array([cell]), 4)
runs = parity.runs_to_oracle(O, p, pkg, [{"I": -1.0}])
ens = pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0)
ro = O.simulate("lco_iso", Th[0], 1.0, runs)
print("flags", ens.run_info[0,0]["flag"], ro["runs"][0]["flag"], "iters", ens.run_info[0,0]["iterations"], ro["runs"][0]["iterations"], "t_end", ens.run_info[0,0]["t_end"], ro["runs"][0]["t_end"])
print("counters", {f: (int(ens.counters[0][f]), ro["counters"][f]) for f in ("n_steps","n_res","n_jac","n_newton","n_errfail","n_convfail")})
print("state err emu vs orc", parity.state_rel_err(ens.Y[0], ro["Y"]))
n = int(ens.n_pts[0]); print("n pts", n, len(ro["t"]))
dt = ens.t[0,:n] - ro["t"][:n]
print("t diffs rel:", np.abs(dt[1:]/ro["t"][1:n]).max(), "last 5 t:", ro["t"][-5:], "dt last", dt[-5:])
for eps in (2.2e-16, 1e-15, 1e-14):
    b = []
    for seed in range(1, 9):
        rk = O.simulate("lco_iso", Th[0], 1.0, runs, opts=O.default_opts(fd_perturb=eps, perturb_seed=seed))
        b.append(parity.state_rel_err(rk["Y"], ro["Y"]))
    print("eps", eps, ["%.1e" % x for x in b])
print("k, t_orc, rel dt:")
for k in range(1, n):
    print(k, "%.6f %.3e" % (ro["t"][k], dt[k]/ro["t"][k]), end=" | ")
    if k % 4 == 0: print()
print()
for refine in (1,):
    o = pkg.Opts(); o.refine = refine
    e2 = pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0, opts=o)
    r2 = O.simulate("lco_iso", Th[0], 1.0, runs, opts=O.default_opts(refine=refine))
    print("refine", refine, "state err", parity.state_rel_err(e2.Y[0], r2["Y"]), "h0 rel", (e2.t[0,1]-r2["t"][1])/r2["t"][1], " max rel dt", np.abs((e2.t[0,1:n]-r2["t"][1:n])/r2["t"][1:n]).max())

#!/usr/bin/env python3
"""EVERY cell of the BASELINE configurations at their full sizes against the oracle, at the reference's default tolerances (GPU box; the oracle on all usable host cores):
     python tools/full_ensemble_parity.py [C2 C3 C4 C5] > gpurun_out/full_ensemble_parity.md
C2 1024 cells, C3 4096, C4 65 536, C5 8192 (NMC + SEI, 20-pulse GITT).  Per configuration: exit flags and run-end times of every run, the integrator's counters (identical
decisions = steps, residuals, Jacobians, Newton iterations, error-test and convergence failures all equal), and the end-state deviation per state section relative to the
section's scale (tests/parity.state_rel_err).  What deviations above 1e-6 mean at these tolerances, and why they vanish at tight ones: DESIGN.md 5."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload, parity
from oracle import oracle as O
import torch
O.build()
pkg = pkgload.load()
CASES = {"C2": (lambda: pkg.petlion(pkg.LCO), "c2", 1024), "C3": (lambda: pkg.petlion(pkg.LCO, temperature=True), "c3", 4096),
         "C4": (lambda: pkg.petlion(pkg.LCO), "c4", 65536), "C5": (lambda: pkg.petlion(pkg.NMC, aging="SEI"), "c5", 8192)}
cores = len(os.sched_getaffinity(0))
try:
    q, per = open("/sys/fs/cgroup/cpu.max").read().split()
    if q != "max":
        cores = max(1, min(cores, int(int(q) / int(per))))
except Exception:
    pass
print("# Every cell of C2 - C5 against the oracle, default tolerances (reltol 1e-3 / abstol 1e-6)\n")
print("`python tools/full_ensemble_parity.py` on one MI355X box; oracle on %d host threads.\n" % cores)
print("| config | cells | kernel | oracle wall | flags equal (every run) | identical decisions | run-end times max rel | end state: median / p90 / p99 / max | cells > 1e-6 | > 1e-4 | > 1e-3 |")
print("|---|---|---|---|---|---|---|---|---|---|---|")
for name in (sys.argv[1:] or list(CASES)):
    mk, cfgname, n = CASES[name]
    p = mk()
    cfg = getattr(pkg.configs, cfgname)(p, n)
    Th = np.ascontiguousarray(cfg["theta"])
    ens = pkg.simulate_ensemble(p, torch.from_numpy(Th).cuda(), cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"])
    torch.cuda.synchronize()
    Yd = ens.Y.cpu().numpy(); info = ens.run_info; cnt = ens.counters
    runs = parity.runs_to_oracle(O, p, pkg, cfg["protocol"])
    keys = ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail")

    def one(i):
        ro = O.simulate(p.variant, Th[i], cfg["SOC"], runs, max_out=8)
        fl = all(int(info[i, k]["flag"]) == rr["flag"] for k, rr in enumerate(ro["runs"]))
        dt = max(abs(float(info[i, k]["t_end"]) - rr["t_end"]) / max(1.0, rr["t_end"]) for k, rr in enumerate(ro["runs"]))
        same = all(int(cnt[i][f]) == ro["counters"][f] for f in keys)
        return fl, same, dt, parity.state_rel_err(Yd[i], ro["Y"])
    t0 = time.time()
    with ThreadPoolExecutor(cores) as ex:
        res = list(ex.map(one, range(n), chunksize=64))
    wall = time.time() - t0
    fl = np.array([r[0] for r in res]); same = np.array([r[1] for r in res]); dt = np.array([r[2] for r in res]); err = np.array([r[3] for r in res])
    print("| %s | %d | %.2f ms | %.0f s | %d | %d (%.1f %%) | %.1e | %.1e / %.1e / %.1e / %.1e | %d | %d | %d |"
          % (name, n, ens.kernel_ms, wall, fl.sum(), same.sum(), 100.0 * same.mean(), dt.max(), np.median(err), np.percentile(err, 90), np.percentile(err, 99), err.max(),
             (err > 1e-6).sum(), (err > 1e-4).sum(), (err > 1e-3).sum()), flush=True)
    ok = same & fl
    if ok.any():
        print("|  | | | | | of which: | | identical-decision cells %.1e / %.1e / %.1e / %.1e | | | |" % (np.median(err[ok]), np.percentile(err[ok], 90), np.percentile(err[ok], 99), err[ok].max()), flush=True)

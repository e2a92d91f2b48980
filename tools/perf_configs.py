#!/usr/bin/env python3
"""Kernel time / throughput of the BASELINE.json configs that are not the bench line (C3, C4 shard, C5 shard) on one GPU.
   python tools/perf_configs.py [c3|c4|c5|all] [--reps K]"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
g.build_hip()
import pkgload
pkg = pkgload.load()


def splitmix_u(seed, n, k):     # counter-based uniform numbers
    x = (np.uint64(seed) ^ (np.arange(n, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15)) ^ np.uint64(k))
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) / float(1 << 53)


def run(name, p, Th, proto, soc, reps, max_points):
    Thd = torch.from_numpy(np.ascontiguousarray(Th)).cuda()
    ms = []
    for r in range(reps + 1):
        ens = pkg.simulate_ensemble(p, Thd, proto, SOC=soc, device=True, max_points=max_points)
        torch.cuda.synchronize()
        if r > 0:
            ms.append(ens.kernel_ms)
    fl = ens.run_info["flag"]
    n = Th.shape[0]
    steps = ens.counters["n_steps"].astype(float)
    print("%-3s cells %6d  kernel %9.2f ms  -> %9.0f trajectories/s ; steps/cell mean %.0f max %.0f ; newton/cell %.0f ; flags(last run) %s ; errors %d"
          % (name, n, np.mean(ms), n / (np.mean(ms) * 1e-3), steps.mean(), steps.max(), ens.counters["n_newton"].mean(),
             dict(zip(*np.unique(fl[:, -1], return_counts=True))), int((fl < 0).sum())), flush=True)
    return ens


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("which", nargs="?", default="all"); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cells", type=int, default=0)
    a = ap.parse_args()
    if a.which in ("c3", "all"):
        p = pkg.petlion(pkg.LCO, temperature=True)
        n = a.cells or 4096
        Th = pkg.theta_matrix(p, n, {"T_amb": 298.15 + 5 * (splitmix_u(3, n, 0) - 0.5), "h_cell": 2.0 ** (2 * splitmix_u(3, n, 1) - 1)})
        kw = dict(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
        run("C3", p, Th, [dict(I=4.0, **kw), dict(dT="hold", **kw), dict(V="hold", **kw)], 0.0, a.reps, 512)
    if a.which in ("c4", "all"):
        p = pkg.petlion(pkg.LCO)
        n = a.cells or 8192
        keys = ["D_sp", "D_sn", "D_p", "D_s", "D_n", "k_p", "k_n"]
        Th = pkg.theta_matrix(p, n, {k: p.θ[k] * 2.0 ** (2 * splitmix_u(4, n, j) - 1) for j, k in enumerate(keys)})
        run("C4", p, Th, [{"I": -1.0}], 1.0, a.reps, 256)
    if a.which in ("c5", "all"):
        p = pkg.petlion(pkg.NMC, aging="SEI")
        n = a.cells or 1024
        keys = ["D_sp", "D_sn", "k_p", "k_n"]         # the NMC system has no D_p / D_s / D_n parameters (D_eff(c_e, T) closure)
        Th = pkg.theta_matrix(p, n, {k: p.θ[k] * 2.0 ** (2 * splitmix_u(5, n, j) - 1) for j, k in enumerate(keys)})
        proto = []
        for _ in range(20):
            proto += [{"I": 1.0, "tf": 180.0}, {"I": "rest", "tf": 7200.0}]
        run("C5", p, Th, proto, 0.0, a.reps, 4096)


if __name__ == "__main__":
    main()

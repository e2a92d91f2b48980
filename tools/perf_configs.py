#!/usr/bin/env python3
"""Kernel time / throughput of the BASELINE.json configs on one GPU (kernel time by the library's HIP events, Theta resident).
   python tools/perf_configs.py [c2|c3|c4|c5|all ...] [--reps K] [--cells N] [--precision f64|mixed] [--waves 1|2]
PETLION_HIP_LIB=<path> selects an experiment build of the library (tools/experiments/)."""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import __graft_entry__ as g
if not os.environ.get("PETLION_HIP_LIB"):
    g.build_hip()
import pkgload
pkg = pkgload.load()
MODELS = {"c2": dict(c=pkg.LCO), "c3": dict(c=pkg.LCO, temperature=True), "c4": dict(c=pkg.LCO), "c5": dict(c=pkg.NMC, aging="SEI")}
CELLS = {"c2": 1024, "c3": 4096, "c4": 8192, "c5": 1024}


def run(name, p, cfg, reps):
    Thd = torch.from_numpy(np.ascontiguousarray(cfg["theta"])).cuda()
    ms = []
    for r in range(reps + 1):
        ens = pkg.simulate_ensemble(p, Thd, cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"])
        torch.cuda.synchronize()
        if r > 0:
            ms.append(ens.kernel_ms)
    fl = ens.run_info["flag"]
    n = cfg["theta"].shape[0]
    steps = ens.counters["n_steps"].astype(float)
    print("%-3s cells %6d  kernel %9.3f ms  -> %9.0f trajectories/s ; steps/cell mean %.0f max %.0f ; newton/cell %.0f ; flags(last run) %s ; errors %d ; LDS %d B/cell"
          % (name.upper(), n, np.mean(ms), n / (np.mean(ms) * 1e-3), steps.mean(), steps.max(), ens.counters["n_newton"].mean(),
             dict(zip(*np.unique(fl[:, -1], return_counts=True))), int((fl < 0).sum()), p.lds_bytes), flush=True)
    return ens


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("which", nargs="*", default=["all"]); ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--cells", type=int, default=0); ap.add_argument("--precision", default="f64"); ap.add_argument("--waves", type=int, default=1)
    a = ap.parse_args()
    which = list(MODELS) if "all" in a.which else a.which
    for w in which:
        mk = dict(MODELS[w]); c = mk.pop("c")
        p = pkg.petlion(c, precision=a.precision, waves_per_cell=a.waves, **mk)
        run(w, p, getattr(pkg.configs, w)(p, a.cells or CELLS[w]), a.reps)


if __name__ == "__main__":
    main()

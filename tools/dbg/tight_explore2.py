"""exploration (GPU): does the DEVICE complete the C3 protocol at tolerances where the oracle runs into maxiters in the dT = hold leg?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pkgload, parity
from oracle import oracle as O
pkg = pkgload.load()
pt = pkg.petlion(pkg.LCO, temperature=True)
cfg = pkg.configs.c3(pt, 4096)
Th = cfg["theta"][::512]
runs = parity.runs_to_oracle(O, pt, pkg, cfg["protocol"])
ref = None
for tol in ((1e-3, 1e-6), (1e-5, 1e-7), (1e-6, 1e-8), (1e-7, 1e-9), (1e-8, 1e-10)):
    o = pkg.Opts(); o.reltol, o.abstol = tol; o.maxiters = 200000
    ens = pkg.simulate_ensemble(pt, Th, cfg["protocol"], SOC=cfg["SOC"], opts=o, max_points=200000)
    print("device tol", tol, "flags", [list(map(int, f)) for f in ens.run_info["flag"][:3]], "steps", ens.counters["n_steps"][:4], "errfail", ens.counters["n_errfail"][:4], "convfail", ens.counters["n_convfail"][:4], "kernel ms %.1f" % ens.kernel_ms, flush=True)
    print("   t_end", ens.run_info["t_end"][0], "I end leg2", ens.run_info["I"][0, 1])
    ro = O.simulate(pt.variant, Th[0], cfg["SOC"], runs, opts=O.default_opts(reltol=tol[0], abstol=tol[1], maxiters=200000), max_out=300000)
    print("   oracle flags", [r["flag"] for r in ro["runs"]], "t_end", [r["t_end"] for r in ro["runs"]], ro["counters"]["n_steps"], ro["counters"]["n_errfail"], ro["counters"]["n_convfail"], flush=True)

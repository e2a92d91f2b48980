"""exploration (GPU): one C5 cell through tests/parity.tight_compare, leg by leg"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pkgload, parity
from oracle import oracle as O
pkg = pkgload.load()
p5 = pkg.petlion(pkg.NMC, aging="SEI")
cfg = pkg.configs.c5(p5, 8192)
c = int(sys.argv[1]) if len(sys.argv) > 1 else 1536
for tol in (dict(reltol=1e-8, abstol=1e-10), dict(reltol=3e-8, abstol=3e-10)):
    for nleg in (40,):
        try:
            r = parity.tight_compare(pkg, p5, O, cfg["theta"][c], cfg["SOC"], cfg["protocol"][:nleg], sample_dt=300.0, max_points=80000, tol=tol)
        except parity.RunFails as e:
            print(tol, "RunFails", e); continue
        print(tol, "traj %.2e V %.2e n=%d steps %s" % (r["traj"], r["V"], r["n_times"], r["steps"]))
        for k, l in enumerate(r["legs"]):
            print("   leg %2d flags %d/%d t_end %.6f %.6f end %.2e" % (k, l[0], l[1], l[2], l[3], l[4]))

#!/usr/bin/env python3
"""C5 at reltol 1e-8: per cell, the device against the oracle (tests/parity.tight_compare) -- where the largest deviation sits and how it moves with the rounding of the
block sweeps (run under PETLION_HIP_LIB=<experiment build> for the one-lane recurrence).  Also the device against ITSELF at 3e-9 (its own global error at 1e-8)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload, parity
from oracle import oracle as O
O.build()
pkg = pkgload.load()
p = pkg.petlion(pkg.NMC, aging="SEI")
cfg = pkg.configs.c5(p, 8192)
cells = [int(c) for c in sys.argv[1:]] or list(range(0, 8192, 256))
for c in cells:
    for tol in ({"reltol": 1e-8, "abstol": 1e-10}, {"reltol": 3e-8, "abstol": 3e-10}):
        try:
            r = parity.tight_compare(pkg, p, O, cfg["theta"][c], cfg["SOC"], cfg["protocol"], sample_dt=300.0, max_points=80000, tol=tol)
            print("cell %5d reltol %.0e: traj %.3e at %s  V %.2e  end %.2e  steps %s" % (c, tol["reltol"], r["traj"], r["worst"], r["V"], max(l[4] for l in r["legs"]), r["steps"]), flush=True)
        except parity.RunFails as e:
            print("cell %5d reltol %.0e: %s fails" % (c, tol["reltol"], e.who), flush=True)

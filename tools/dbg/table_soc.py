"""debug: a table-input run started at SOC = 0.5 (GPU)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload
pkg = pkgload.load()
p = pkg.petlion(pkg.LCO)
th = p.theta_vector()
tt = np.linspace(0, 10, 201)
for name, proto, soc in (("P_sin table", [{"P": (tt, 29.23 * np.sin(tt)), "tf": 10.0}], 0.5), ("I const", [{"I": -0.1, "tf": 10.0}], 0.5), ("I table", [{"I": ([0.0, 1e7], [-1.0, -1.0]), "tf": 100.0}], 1.0),
                         ("I table tstops", [{"I": ([0.0, 1e7], [-1.0, -1.0]), "tf": 100.0}], 1.0)):
    o = pkg.Opts()
    if "tstops" in name: o.tstops = [50.0]
    ens = pkg.simulate_ensemble(p, th[None, :], proto, SOC=soc, opts=o)
    print(name, ens.run_info[0, 0], int(ens.n_pts[0]), ens.SOC[0, :3] if getattr(ens, "SOC", None) is not None else None)

import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pkgload
pkg = pkgload.load()
p = pkg.petlion(pkg.LCO)
n = 1024
rates = np.linspace(0.2, 5.0, n)
ens = pkg.simulate_ensemble(p, pkg.theta_matrix(p, n), [{"I": -rates}], SOC=1.0)
fl = ens.run_info["flag"][:, 0]
print(dict(zip(*np.unique(fl, return_counts=True))))
bad = ~np.isin(fl, (1, 3))
print(rates[bad][:10], fl[bad][:10], ens.run_info["t_end"][bad, 0][:10], ens.run_info["V"][bad, 0][:10], ens.run_info["SOC"][bad, 0][:10], ens.n_pts[bad][:10])

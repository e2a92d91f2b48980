import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pkgload
pkg = pkgload.load()
p = pkg.petlion(pkg.NMC, aging="SEI", _lib_path=os.environ.get("PLH_LIB"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
proto = []
for _ in range(20):
    proto += [{"I": 1.0, "tf": 180.0}, {"I": "rest", "tf": 7200.0}]
ens = pkg.simulate_ensemble(p, pkg.theta_matrix(p, n), proto, SOC=0.0, max_points=4096)
print("flags cell0", ens.run_info["flag"][0].tolist())
print("iters cell0", ens.run_info["iterations"][0].tolist())
print("t_end cell0", np.round(ens.run_info["t_end"][0], 3).tolist()[:8])
c = ens.counters[0]
print({k: int(c[k]) for k in ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail")})

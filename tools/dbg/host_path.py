import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pkgload
pkg = pkgload.load()
p = pkg.petlion(pkg.LCO)
n = 1024
Th = pkg.theta_matrix(p, n)
for k in range(8):
    t0 = time.perf_counter()
    ens = pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0, max_points=256)
    dt = 1e3 * (time.perf_counter() - t0); km = ens.kernel_ms
    if len(sys.argv) > 1: del ens            # bench-like: nothing of the previous call is kept alive
    print("call %d: %.2f ms (kernel %.2f ms)" % (k, dt, km))

import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests']
import pkgload
pkg=pkgload.load()
p=pkg.petlion(pkg.LCO, temperature=True)
kw = dict(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
proto=[dict(I=4.0, **kw), dict(dT="hold", **kw), dict(V="hold", **kw)]
for n in (256, 512, 768, 1536, 3072):
    Th=pkg.theta_matrix(p,n)
    ens = pkg.simulate_ensemble(p, Th, proto, SOC=0.0, max_points=1024)
    fl=ens.run_info["flag"]
    print(n, "flags:", [dict(zip(*np.unique(fl[:, k], return_counts=True))) for k in range(3)], "cells identical:", bool((ens.Y == ens.Y[0]).all()), "kernel %.2f ms" % ens.kernel_ms)

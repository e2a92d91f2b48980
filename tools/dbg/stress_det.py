import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests']
import pkgload
pkg=pkgload.load()
import torch
which = sys.argv[1] if len(sys.argv)>1 else "thermal"
if which=="thermal":
    p=pkg.petlion(pkg.LCO, temperature=True)
    kw = dict(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
    proto=[dict(I=4.0, **kw), dict(dT="hold", **kw), dict(V="hold", **kw)]; soc=0.0
elif which=="sei":
    p=pkg.petlion(pkg.NMC, aging="SEI"); proto=[{"I": 1.0, "tf": 180.0}, {"I": "rest", "tf": 3600.0}]*4; soc=0.0
else:
    p=pkg.petlion(pkg.LCO); proto=[{"I":-1.0}]; soc=1.0
n=int(sys.argv[2]) if len(sys.argv)>2 else 4096
Th=pkg.theta_matrix(p,n)
ref=None; bad=0
for rep in range(int(sys.argv[3]) if len(sys.argv)>3 else 10):
    junk=torch.randn(30_000_000, device="cuda")
    ens = pkg.simulate_ensemble(p, Th, proto, SOC=soc, max_points=2048)
    if ref is None: ref=ens.Y[0].copy()
    nb=int((ens.Y != ref).any(axis=1).sum()); bad+=nb
    print(which, "rep",rep,"cells differing from cell 0 of launch 0:", nb, "flags", dict(zip(*np.unique(ens.run_info["flag"][:,-1], return_counts=True))), flush=True)
print("TOTAL BAD", bad)

import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests']
import pkgload, parity
from oracle import oracle as O
pkg=pkgload.load()
p=pkg.petlion(pkg.NMC)
Th = pkg.theta_matrix(p, 16, {"D_sp": p.θ["D_sp"] * np.linspace(0.6, 1.6, 16)})
for kw in ({}, dict(init_step=1e-2)):
    o = pkg.Opts(); o.init_step = kw.get("init_step", 0.0)
    ens = pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0, opts=o)
    for i in range(16):
        ro = O.simulate("nmc_iso", Th[i], 1.0, parity.runs_to_oracle(O, p, pkg, [{"I": -1.0}]), opts=O.default_opts(**kw))
        n = int(ens.n_pts[i]); m = min(n, len(ro["t"]))
        dV = np.abs(ens.V[i,:m]-ro["V"][:m])
        print(kw, i, n, len(ro["t"]), "iters", int(ens.run_info[i,0]["iterations"]), ro["runs"][0]["iterations"], "dV max %.2e at %d (t=%.1f)" % (dV.max(), dV.argmax(), ro["t"][dV.argmax()]), "dt %.2e" % np.abs(ens.t[i,:m]-ro["t"][:m]).max(), "state %.2e" % parity.state_rel_err(ens.Y[i], ro["Y"]))

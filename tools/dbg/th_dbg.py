import sys, os, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/wave_emu']
import pkgload, parity
from oracle import oracle as O
pkg=pkgload.load()
emu = len(sys.argv)>1 and sys.argv[1]=="emu"
if emu:
    import build_emu
    p=pkg.petlion(pkg.LCO, temperature=True, _lib_path=build_emu.build())
else:
    p=pkg.petlion(pkg.LCO, temperature=True)
th=p.theta_vector(); N=p.N.tot
kw = dict(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1/20)
proto=[dict(I=4.0, **kw), dict(dT="hold", **kw), dict(V="hold", **kw)]
ro1=O.simulate(p.variant, th, 0.0, parity.runs_to_oracle(O,p,pkg,proto[:1]))
Y1=ro1["Y"]; t1=ro1["runs"][0]["t_end"]; soc1=ro1["runs"][0]["SOC"]
# leg 2 from the oracle's state, both implementations
lib,h=p._lib,p._h
for val in (0.0,):
    rc,Yo,YPo,ito=O.init_consistent(p.variant, th, Y1.copy(), 2, val)
    Y,YP=Y1.copy(),np.zeros(N); st,it=np.zeros(1,np.int32),np.zeros(1,np.int32)
    lib.plh_init_consistent(h,1,th.ctypes.data,2,val,1e-3,Y.ctypes.data,YP.ctypes.data,st.ctypes.data,it.ctypes.data,0,None)
    print("init dT: I dev %.10f orc %.10f rel %.2e ; iters %d %d" % (Y[-1], Yo[-1], abs(Y[-1]-Yo[-1])/Yo[-1], it[0], ito))
for okw in ({}, dict(jac_every_step=1)):
    o=pkg.Opts(); o.jac_every_step=bool(okw.get("jac_every_step",0))
    ens=pkg.simulate_ensemble(p, th[None,:], proto[:2], SOC=0.0, opts=o)
    ro=O.simulate(p.variant, th, 0.0, parity.runs_to_oracle(O,p,pkg,proto[:2]), opts=O.default_opts(**okw))
    k1=ro["runs"][0]["iterations"]; kd=int(ens.run_info[0,0]["iterations"])
    print(okw, "dev", [int(x) for x in ens.run_info[0]["iterations"]], ["%.6f"%x for x in ens.run_info[0]["t_end"]], "orc", [r["iterations"] for r in ro["runs"]], ["%.6f"%r["t_end"] for r in ro["runs"]])
    print("   t dev", np.round(ens.t[0,kd-1:kd+8],6), "\n   t orc", np.round(ro["t"][k1-1:k1+8],6))
    print("   I dev", ens.I[0,kd-1:kd+6], "\n   I orc", ro["I"][k1-1:k1+6])
    n=int(ens.n_pts[0]); print("   last t dev", np.round(ens.t[0,n-4:n],4), "orc", np.round(ro["t"][-4:],4)); print("   last V dev", ens.V[0,n-4:n], "orc", ro["V"][-4:])

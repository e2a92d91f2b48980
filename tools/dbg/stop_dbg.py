import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests','/root/repo/tests/wave_emu']
import pkgload, parity
from oracle import oracle as O
pkg=pkgload.load()
emu = len(sys.argv)>1 and sys.argv[1]=="emu"
if emu:
    import build_emu; p=pkg.petlion(pkg.LCO, aging="SEI", _lib_path=build_emu.build())
else: p=pkg.petlion(pkg.LCO, aging="SEI")
th=p.theta_vector()
cases = [("V_min", [{"I": -2.0, "V_min": 3.6}], 1.0), ("V_max", [{"I": 1.0, "V_max": 3.95}], 0.3), ("SOC_min", [{"I": -1.0, "SOC_min": 0.6}], 1.0),
  ("SOC_max", [{"I": 1.0, "SOC_max": 0.5}], 0.2), ("c_s_n_max", [{"I": 2.0, "c_s_n_max": 0.6}], 0.2), ("I_max", [{"V": 4.05, "I_max": 2.5, "tf": 600.0}], 0.6),
  ("c_e_min", [{"I": -3.0, "c_e_min": 600.0}], 1.0), ("eta", [{"I": 3.0, "η_plating_min": 0.02}], 0.2)]
for name, proto, soc in cases:
  for kw in ({}, dict(init_step=1e-2)):
    o=pkg.Opts(); o.init_step=kw.get("init_step",0.0)
    ens = pkg.simulate_ensemble(p, th[None, :], proto, SOC=soc, opts=o)
    ro = O.simulate(p.variant, th, soc, parity.runs_to_oracle(O, p, pkg, proto), opts=O.default_opts(**kw))
    N=p.N.tot
    print(name, kw, int(ens.run_info[0,0]["flag"]), ro["runs"][0]["flag"], int(ens.run_info[0,0]["iterations"]), ro["runs"][0]["iterations"], "t %.6f %.6f"%(ens.run_info[0,0]["t_end"], ro["runs"][0]["t_end"]),
          [(n,"%.1e"%(np.abs(ens.Y[0][a:e]-ro["Y"][a:e]).max()/(np.abs(ro["Y"][a:e]).max()+1e-300))) for n,a,e in parity.sections_for(N)])

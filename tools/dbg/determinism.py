import sys, numpy as np
sys.path[:0]=['/root/repo','/root/repo/tests']
import pkgload, parity
pkg=pkgload.load()
import torch
for kw in ({}, {"aging":"SEI"}, {"temperature":True}):
    p=pkg.petlion(pkg.LCO, **kw)
    cases = [("V_min", [{"I": -2.0, "V_min": 3.6}], 1.0), ("V_max", [{"I": 1.0, "V_max": 3.95}], 0.3), ("SOC_max", [{"I": 1.0, "SOC_max": 0.5}], 0.2),
      ("c_s_n_max", [{"I": 2.0, "c_s_n_max": 0.6}], 0.2), ("I_max", [{"V": 4.05, "I_max": 2.5, "tf": 600.0}], 0.6), ("I_min", [{"I": 1.0, "tf": 600.0}, {"V": "hold", "I_min": 0.3}], 0.3),
      ("c_e_min", [{"I": -3.0, "c_e_min": 600.0}], 1.0), ("eta", [{"I": 3.0, "η_plating_min": 0.02}], 0.2)]
    n=512
    Th=pkg.theta_matrix(p,n)
    for name, proto, soc in cases:
        Ys=[]
        for rep in range(3):
            junk = torch.randn(50_000_000, device="cuda")    # stir the allocator / caches between launches
            ens = pkg.simulate_ensemble(p, Th, proto, SOC=soc)
            Ys.append(ens.Y.copy())
            del junk
        same_cells = all((Y == Y[0]).all() for Y in Ys)
        same_runs = all((Y == Ys[0]).all() for Y in Ys)
        worst = max(np.abs(Y - Ys[0][0]).max() for Y in Ys)
        print(kw, name, "cells identical:", same_cells, "launches identical:", same_runs, "max abs dev %.3e" % worst)

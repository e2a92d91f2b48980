import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import __graft_entry__ as g
g.build_hip()
import pkgload
pkg = pkgload.load()
p1 = pkg.petlion(pkg.LCO); p2 = pkg.petlion(pkg.LCO, waves_per_cell=2)
cfg = pkg.configs.c4(p1, 512)
Thd = torch.from_numpy(np.ascontiguousarray(cfg["theta"])).cuda()
e1 = pkg.simulate_ensemble(p1, Thd, cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"])
e2 = pkg.simulate_ensemble(p2, Thd, cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"])
torch.cuda.synchronize()
Y1 = e1.Y.cpu().numpy(); Y2 = e2.Y.cpu().numpy()
print("flags equal", (e1.run_info["flag"] == e2.run_info["flag"]).all(), "steps equal", (e1.counters["n_steps"] == e2.counters["n_steps"]).mean())
d = np.abs(Y1 - Y2).max(axis=1) / np.abs(Y1).max(axis=1)
print("max rel dev W2 vs W1", d.max(), "median", np.median(d))

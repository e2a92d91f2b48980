"""exploration (GPU): the dT = :hold leg at reltol 1e-6: step sequences of device and oracle on cells where the device does not complete"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import pkgload, parity
from oracle import oracle as O
pkg = pkgload.load()
pt = pkg.petlion(pkg.LCO, temperature=True)
cfg = pkg.configs.c3(pt, 4096)
Th = cfg["theta"][::32]
runs = parity.runs_to_oracle(O, pt, pkg, cfg["protocol"])
o = pkg.Opts(); o.reltol, o.abstol = 1e-6, 1e-8; o.maxiters = 6000
ens = pkg.simulate_ensemble(pt, Th, cfg["protocol"], SOC=cfg["SOC"], opts=o, max_points=6010)
bad = np.flatnonzero((ens.run_info["flag"] < 0).any(axis=1))
print("failing cells (of 128, stride 32):", bad[:20])
for i in bad[:3]:
    ro = O.simulate(pt.variant, Th[i], cfg["SOC"], runs, opts=O.default_opts(reltol=1e-6, abstol=1e-8, maxiters=6000), max_out=6010)
    k1d, k1o = int(ens.run_info[i, 0]["iterations"]), ro["runs"][0]["iterations"]
    n = int(ens.n_pts[i])
    td, to = ens.t[i, k1d:n] - ens.t[i, k1d], ro["t"][k1o:k1o + ro["runs"][1]["iterations"]] - ro["t"][k1o]
    Id, Io = ens.I[i, k1d:n], ro["I"][k1o:k1o + ro["runs"][1]["iterations"]]
    m = min(len(td), len(to))
    same = np.flatnonzero(np.abs(td[:m] - to[:m]) > 1e-9 * (1 + to[:m]))
    first = same[0] if len(same) else m
    print("cell", i, "leg-1 iterations dev/orc", k1d, k1o, "leg-2 points dev/orc", len(td), len(to), "flags dev", [int(f) for f in ens.run_info[i]["flag"]], "orc", [r["flag"] for r in ro["runs"]], "first differing step", first)
    lo = max(0, first - 3)
    for j in range(lo, min(m, first + 12)):
        print("   %4d  t %.9f %.9f   I %.10f %.10f" % (j, td[j], to[j], Id[j], Io[j]))
    print("   device last 5 t:", td[-5:], "dt:", np.diff(td[-6:]))

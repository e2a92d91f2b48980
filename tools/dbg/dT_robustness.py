"""exploration (GPU): how many C3 cells complete the dT = :hold leg at reltol 1e-6 / abstol 1e-8 on the device, plain and with iterative refinement of every solve"""
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
for tol in ((1e-7, 1e-9), (3e-8, 3e-10), (1e-8, 1e-10)):
    for ref in (0, 1):
        o = pkg.Opts(); o.reltol, o.abstol = tol; o.maxiters = 60000; o.refine = ref
        ens = pkg.simulate_ensemble(pt, Th, cfg["protocol"], SOC=cfg["SOC"], opts=o, max_points=60010)
        fl = ens.run_info["flag"]
        ok = (fl >= 0).all(axis=1)
        print("device tol", tol, "refine", ref, "complete %d / %d" % (ok.sum(), len(Th)), "median steps of complete cells", int(np.median(ens.counters["n_steps"][ok])) if ok.any() else -1,
              "convfail median", int(np.median(ens.counters["n_convfail"])), "errfail median", int(np.median(ens.counters["n_errfail"])), "kernel %.0f ms" % ens.kernel_ms, flush=True)
    for ref in ():
        okc = 0; t0 = time.time()
        for i in range(0, len(Th), 8):
            ro = O.simulate(pt.variant, Th[i], cfg["SOC"], runs, opts=O.default_opts(reltol=tol[0], abstol=tol[1], maxiters=20000, refine=ref), max_out=20010)
            okc += min(r["flag"] for r in ro["runs"]) >= 0
        print("oracle tol", tol, "refine", ref, "complete %d / %d" % (okc, len(range(0, len(Th), 8))), "%.0f s" % (time.time() - t0), flush=True)

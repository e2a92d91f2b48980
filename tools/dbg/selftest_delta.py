#!/usr/bin/env python3
"""What the kernel self-test (api.selftest) compares, printed instead of asserted: plain vs stop-times instantiation of one variant on the 1C discharge.
   PETLION_HIP_LIB=<lib> python tools/dbg/selftest_delta.py lgm50_thermal|lco_thermal|lco"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import pkgload
pkg = pkgload.load()
which = sys.argv[1] if len(sys.argv) > 1 else "lgm50_thermal"
lib = os.environ.get("PETLION_HIP_LIB")
os.environ["PETLION_SKIP_SELFTEST"] = "1"
mk = {"lgm50_thermal": lambda: pkg.petlion(pkg.NMC_LGM50, _lib_path=lib), "lco_thermal": lambda: pkg.petlion(pkg.LCO, temperature=True, _lib_path=lib), "lco": lambda: pkg.petlion(pkg.LCO, _lib_path=lib)}
try:
    p = mk[which]()
except RuntimeError as e:          # (the self-test of the model's creation: go on with the handle-less path below is not possible -- report and stop)
    print("creation raised:", str(e)[:200]); sys.exit(0)
for tf in (100.0, 1000.0):
    Th = np.tile(p.theta_vector(), (2, 1))
    base = pkg.simulate_ensemble(p, Th, [{"I": -1.0, "tf": tf}], SOC=1.0)
    o = pkg.Opts(); o.tstops = [1e7]
    e = pkg.simulate_ensemble(p, Th, [{"I": -1.0, "tf": tf}], SOC=1.0, opts=o)
    print("%s tf %g: flags %s / %s ; |dt_end| %.3e |dSOC| %.3e |dV| %.3e ; steps %s / %s newton %s / %s ; max |dY|/|Y| %.3e ; n_pts %s / %s" % (
        which, tf, base.run_info["flag"][:, 0], e.run_info["flag"][:, 0], np.abs(e.run_info["t_end"] - base.run_info["t_end"]).max(), np.abs(e.run_info["SOC"] - base.run_info["SOC"]).max(),
        np.abs(e.run_info["V"] - base.run_info["V"]).max(), base.counters["n_steps"], e.counters["n_steps"], base.counters["n_newton"], e.counters["n_newton"],
        (np.abs(np.asarray(e.Y) - np.asarray(base.Y)) / (np.abs(np.asarray(base.Y)) + 1e-30)).max(), base.n_pts, e.n_pts))
    k = int(min(base.n_pts[0], e.n_pts[0]))
    dv = np.abs(np.asarray(e.V[0, :k]) - np.asarray(base.V[0, :k])); dt = np.abs(np.asarray(e.t[0, :k]) - np.asarray(base.t[0, :k]))
    first = int(np.argmax((dv > 0) | (dt > 0))) if ((dv > 0) | (dt > 0)).any() else -1
    print("   first saved point that differs: %d of %d ; dV there %.3e dt %.3e ; max dV %.3e" % (first, k, dv[first] if first >= 0 else 0.0, dt[first] if first >= 0 else 0.0, dv.max()))
try:
    pkg.selftest(p); print("selftest of every instantiation: passed")
except RuntimeError as e:
    print("selftest:", str(e)[:160])

#!/usr/bin/env python3
"""Developer's refactoring guard (no GPU): the device source of ONE model variant on the wave emulator, evaluators + a protocol, saved to / compared with a snapshot.
   python tools/dev/regress.py thermal|iso|sei save|check [tag]
`save` writes /tmp/plh_regress_<model>_<tag>.npz from the current source; `check` rebuilds and reports the largest deviation per quantity (a change that only re-orders
floating-point operations shows 1e-16 .. 1e-12 and identical counters; a behavioural change shows up in the counters)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "wave_emu")]
import build_emu, pkgload, parity
pkg = pkgload.load()
from oracle import oracle as O
O.build()
model, what = sys.argv[1], sys.argv[2]
tag = sys.argv[3] if len(sys.argv) > 3 else "base"
VAR = {"iso": 0, "sei": 3, "thermal": 4}[model]
lib = build_emu.build(variant=VAR)
kw = dict(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
if model == "thermal":
    p = pkg.petlion(pkg.LCO, temperature=True, _lib_path=lib); soc = 0.0
    proto = [dict(I=4.0, **kw), dict(dT="hold", **kw), dict(V="hold", **kw)]
    modes = ((0, 3.0), (1, 3.9), (2, 0.01), (3, 80.0))
elif model == "sei":
    p = pkg.petlion(pkg.NMC, aging="SEI", _lib_path=lib); soc = 0.0
    proto = [{"I": 1.0, "tf": 180.0}, {"I": "rest", "tf": 7200.0}] * 2
    modes = ((0, 1.0), (1, 3.9))
else:
    p = pkg.petlion(pkg.LCO, _lib_path=lib); soc = 1.0
    proto = [{"I": -1.0, "tf": 1500.0}, {"V": "hold", "tf": 300.0}, {"P": -60.0, "tf": 300.0}]
    modes = ((0, -1.0), (1, 3.9), (3, -80.0), (4, 0.05))
th = p.theta_vector()
N = p.N.tot
n = 3
Y, YP = parity.realistic_states(O, th, n, variant=p.variant)
Th = np.tile(th, (n, 1)); Th[:, p.θ_keys.index("D_sp")] *= np.linspace(0.5, 2.0, n); Th = np.ascontiguousarray(Th)
res = {}
L, h = p._lib, p._h
b0 = np.random.default_rng(1).standard_normal((n, N))
for mode, val in modes:
    F = np.zeros((n, N)); assert L.plh_residual(h, n, Th.ctypes.data, Y.ctypes.data, YP.ctypes.data, mode, val, F.ctypes.data, 0, None) == 0
    nz = np.zeros((n, len(p.jac_pattern(mode)[1]))); assert L.plh_jacobian(h, n, Th.ctypes.data, Y.ctypes.data, YP.ctypes.data, 0.37, mode, nz.ctypes.data, 0, None) == 0
    x = b0.copy(); assert L.plh_linear_solve(h, n, Th.ctypes.data, Y.ctypes.data, YP.ctypes.data, 0.37, mode, x.ctypes.data, 0, None) == 0
    res["F%d" % mode], res["J%d" % mode], res["x%d" % mode] = F, nz, x
o = pkg.Opts()
ens = pkg.simulate_ensemble(p, Th[:2], proto, SOC=soc, opts=o)
res["Yend"] = ens.Y; res["YPend"] = ens.YP
ri = ens.run_info
res["flag"] = ri["flag"].astype(float); res["iters"] = ri["iterations"].astype(float); res["t_end"] = ri["t_end"]; res["V"] = ri["V"]; res["I"] = ri["I"]; res["SOC"] = ri["SOC"]
for k in ("n_steps", "n_res", "n_jac", "n_solve", "n_newton", "n_errfail", "n_convfail"):
    res["cnt_" + k] = ens.counters[k].astype(float)
path = "/tmp/plh_regress_%s_%s.npz" % (model, tag)
if what == "save":
    np.savez(path, **res); print("saved", path, {k: v.tolist() for k, v in res.items() if k.startswith("cnt_")})
else:
    ref = np.load(path)
    bad = False
    for k in res:
        a, b = res[k], ref[k]
        if k in ("flag", "iters") or k.startswith("cnt_"):
            same = np.array_equal(a, b); bad |= not same
            print("%-12s %s %s" % (k, "same" if same else "DIFFERENT", "" if same else (a.tolist(), b.tolist())))
        else:
            sc = np.abs(b).max(axis=-1, keepdims=True) + 1e-300
            d = (np.abs(a - b) / sc).max()
            bad |= not (d < ((1e-10 if k[0] == 'x' else 1e-13) if k[0] in 'FJx' and k[1:].isdigit() else 3e-5))      # evaluators to rounding; trajectories: rounding amplified by the hold legs
            print("%-12s max rel dev %.2e" % (k, d))
    print("REGRESSION" if bad else "ok")

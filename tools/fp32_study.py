#!/usr/bin/env python3
"""C5's fp32-vs-fp64 question, answered on the equations (CPU, numpy) before spending a device instantiation on it.

For states along the C5 protocol (NMC + SEI aging, GITT pulse) it measures what IEEE single precision does to the three quantities the
integrator depends on, in the units the integrator itself uses (IDA's weighted RMS norm with reltol 1e-3 / abstol 1e-6, Newton tolerance 0.33):
  (1) residual evaluation in pure fp32 (every operation rounded to float) -> the Newton correction that rounding error alone produces
  (2) the Newton linear solve in fp32 (dense LU) with and without fp64 iterative refinement
  (3) representability of the time grid and of the slow aging states (film, SOH) in fp32
usage: python tools/fp32_study.py   (needs only the oracle; no GPU)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O
from oracle import dfn_model as dm
from oracle.codegen import VARIANTS


class F32Ops(dm.FloatOps):
    f = np.float32
    def sqrt(self, x): return np.sqrt(self.f(x))
    def exp(self, x): return np.exp(self.f(x))
    def sinh(self, x): return np.sinh(self.f(x))
    def atan(self, x): return np.arctan(self.f(x))
    def pow(self, x, y): return np.power(self.f(x), self.f(y))
    def const(self, v): return self.f(v)


def main():
    variant = "nmc_iso_sei"
    model = dm.Model(**VARIANTS[variant])
    m = O.meta(variant)
    keys = m["theta_keys"]
    th = O.theta_vector(variant)
    thd = dict(model.theta); thd.update(dict(zip(keys, th)))
    th32 = {k: np.float32(v) for k, v in thd.items()}
    rtol, atol = 1e-3, 1e-6
    rows = []
    for tf in (60.0, 170.0):                       # inside the first 1C pulse of the GITT protocol
        ro = O.simulate(variant, th, 0.0, [dict(mode=O.MODE_I, value=1.0, tf=tf)])
        Y, YP = ro["Y"], ro["YP"]
        N = len(Y)
        F64 = np.array(dm.residual(model, dm.FloatOps(), list(Y), list(YP), thd, dm.MODE_I, 1.0), dtype=np.float64)
        with np.errstate(all="ignore"):
            F32 = np.array([float(v) for v in dm.residual(model, F32Ops(), [np.float32(v) for v in Y], [np.float32(v) for v in YP], th32, dm.MODE_I, np.float32(1.0))])
        cj = 1.0 / 5.0                              # a typical BDF coefficient (h ~ 5 s)
        cp, ri, nz = O.jacobian(variant, th, Y, YP, cj, O.MODE_I, 1.0)
        J = np.zeros((N, N))
        for c in range(N):
            J[ri[cp[c]:cp[c + 1]], c] = nz[cp[c]:cp[c + 1]]
        ewt = 1.0 / (rtol * np.abs(Y) + atol)
        wrms = lambda v: float(np.sqrt(np.mean((v * ewt) ** 2)))
        d_round = np.linalg.solve(J, F32 - F64)     # Newton correction caused by fp32 rounding of the residual alone
        # (2) linear solve in fp32
        b = F64 + 1e-3 * np.abs(F64).max() * np.random.default_rng(0).standard_normal(N)
        x64 = np.linalg.solve(J, b)
        with np.errstate(all="ignore"):
            x32 = np.linalg.solve(J.astype(np.float32), b.astype(np.float32)).astype(np.float64)
            r = b - J @ x32
            x32r = x32 + np.linalg.solve(J.astype(np.float32), r.astype(np.float32)).astype(np.float64)   # one fp64-residual refinement step
        rows.append((tf, wrms(d_round), np.abs(d_round / (np.abs(Y) + 1e-30)).max(), wrms(x32 - x64) / max(wrms(x64), 1e-300), wrms(x32r - x64) / max(wrms(x64), 1e-300),
                     np.linalg.cond(J)))
    print("state  | WRMS of the Newton correction produced by fp32 residual rounding (Newton converges at 0.33) | max rel. state change | fp32 LU rel. error | + 1 refinement | cond(J)")
    for r in rows:
        print("t=%4.0fs | %.3e | %.3e | %.3e | %.3e | %.2e" % r)
    # (3) representability
    t_end = 20 * (180 + 7200.0)
    print("time grid: ulp_fp32(t = %.0f s) = %.3g s (IDA restarts every run with steps of 1e-2..1e-6 s) ; ulp_fp64 = %.3g s" % (t_end, np.spacing(np.float32(t_end)), np.spacing(t_end)))
    ro = O.simulate(variant, th, 0.0, [dict(mode=O.MODE_I, value=1.0, tf=180.0)])
    soh = ro["Y"][240]
    print("SOH after one pulse = 1 - %.3e ; ulp_fp32(1) = %.3e -> %.1f ulps per pulse ; film = %.3e m, film increment per 5 s step ~ %.1e (relative %.1e vs eps_fp32 %.1e)"
          % (1 - soh, np.spacing(np.float32(1.0)), (1 - soh) / np.spacing(np.float32(1.0)), ro["Y"][230:240].max(), ro["Y"][230:240].max() * 5 / 180, 5 / 180, np.finfo(np.float32).eps))


if __name__ == "__main__":
    main()

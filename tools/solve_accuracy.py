#!/usr/bin/env python3
"""Which linear solver is closer to the truth?  

Newton matrix J = dF/dY + cj dF/dYP of a realistic state, right-hand sides b; three solutions of J x = b:
  * the device's structured solver (plh_linear_solve: particle resolvent + node-local elimination + twisted block-Thomas + border),
    run through the wave-emulator build of the device source (CPU) or, with --gpu, through the HIP library;
  * the oracle's KLU-style sparse LU (oracle/ida_oracle.c);
  * "truth": dense Gaussian elimination with partial pivoting in 80-bit extended precision (numpy longdouble, eps 1.1e-19) on the oracle's
    Jacobian entries, followed by two steps of iterative refinement in extended precision.
Errors are reported per state section relative to that section's max |x| (the parity metric of tests/parity.py).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emu"))


def ld_solve(A, b):
    """dense LU with partial pivoting in long double + 2 refinement steps; A, b long double"""
    n = A.shape[0]
    LU = A.copy(); piv = np.arange(n)
    for k in range(n):
        p = k + int(np.argmax(np.abs(LU[k:, k])))
        if p != k:
            LU[[k, p]] = LU[[p, k]]; piv[[k, p]] = piv[[p, k]]
        LU[k + 1:, k] /= LU[k, k]
        LU[k + 1:, k + 1:] -= np.outer(LU[k + 1:, k], LU[k, k + 1:])

    def sub(r):
        y = r[piv].copy()
        for k in range(n):
            y[k + 1:] -= LU[k + 1:, k] * y[k]
        for k in range(n - 1, -1, -1):
            y[k] /= LU[k, k]
            y[:k] -= LU[:k, k] * y[k]
        return y
    x = sub(b)
    for _ in range(2):
        x = x + sub(b - A @ x)
    return x


def dense_from_csc(N, cp, ri, nz):
    A = np.zeros((N, N), dtype=np.longdouble)
    for c in range(N):
        A[ri[cp[c]:cp[c + 1]], c] = nz[cp[c]:cp[c + 1]]
    return A


def main():
    import pkgload
    import parity
    from oracle import oracle as O
    pkg = pkgload.load()
    gpu = "--gpu" in sys.argv
    refine = int(os.environ.get("REFINE", "0"))
    rows = []
    for variant_kw, name in ((dict(), "lco_iso"), (dict(aging="SEI"), "lco_iso_sei"), (dict(temperature=True), "lco_thermal")):
        if gpu:
            p = pkg.petlion(pkg.LCO, **variant_kw)
        else:
            import build_emu
            p = pkg.petlion(pkg.LCO, _lib_path=build_emu.build(), **variant_kw)
        th = p.theta_vector(); N = p.N.tot
        Y, YP = parity.realistic_states(O, th, 3, variant=p.variant)
        rng = np.random.default_rng(1)
        for mode, val in ((0, -1.0), (1, 3.9)):
            for cj in (0.37, 25.0):
                for i in range(3):
                    b = rng.standard_normal(N)
                    cp, ri, nz = O.jacobian(p.variant, th, Y[i], YP[i], cj, mode, val)
                    A = dense_from_csc(N, cp, ri, nz)
                    xt = ld_solve(A, b.astype(np.longdouble)).astype(np.float64)
                    xo = O.linear_solve(p.variant, th, Y[i], YP[i], cj, b, mode, val, **({"refine": refine} if refine else {}))
                    xd = b.copy()[None, :].copy()
                    Thm = np.ascontiguousarray(th[None, :]); Yi = np.ascontiguousarray(Y[i][None, :]); YPi = np.ascontiguousarray(YP[i][None, :])
                    if refine:
                        rc = p._lib.plh_linear_solve_refined(p._h, 1, Thm.ctypes.data, Yi.ctypes.data, YPi.ctypes.data, cj, mode, xd.ctypes.data, refine, 0, None)
                    else:
                        rc = p._lib.plh_linear_solve(p._h, 1, Thm.ctypes.data, Yi.ctypes.data, YPi.ctypes.data, cj, mode, xd.ctypes.data, 0, None)
                    assert rc == 0
                    ed = eo = edo = 0.0
                    for _, a, e in parity.sections_for(N):
                        s = np.abs(xt[a:e]).max() + 1e-300
                        ed = max(ed, np.abs(xd[0, a:e] - xt[a:e]).max() / s); eo = max(eo, np.abs(xo[a:e] - xt[a:e]).max() / s)
                        edo = max(edo, np.abs(xd[0, a:e] - xo[a:e]).max() / s)
                    rows.append((name, mode, cj, i, ed, eo, edo))
    print("| variant | mode | cj | state | device vs truth | oracle LU vs truth | device vs oracle |\n|---|---|---|---|---|---|---|")
    for r in rows:
        print("| %s | %d | %g | %d | %.1e | %.1e | %.1e |" % r)
    a = np.array([[r[4], r[5], r[6]] for r in rows])
    print("\nmax: device %.1e, oracle %.1e, device-vs-oracle %.1e ; median: device %.1e, oracle %.1e" % (a[:, 0].max(), a[:, 1].max(), a[:, 2].max(), np.median(a[:, 0]), np.median(a[:, 1])))


if __name__ == "__main__":
    main()

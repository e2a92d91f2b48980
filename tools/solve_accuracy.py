#!/usr/bin/env python3
"""Which linear solver is closer to the truth?  

Newton matrix J = dF/dY + cj dF/dYP of realistic states, random right-hand sides; three solutions of J x = b:
  * the device's structured solver (plh_linear_solve[_refined]: particle resolvent + node-local elimination + twisted block-Thomas + border) through the
    wave-emulator build of the device source (CPU) or, with --gpu, through the HIP library;
  * the oracle's KLU-style sparse LU (oracle/ida_oracle.c);
  * "truth": dense LU with partial pivoting in 80-bit extended precision + refinement (tests/parity.py: ld_solve) on the oracle's Jacobian entries.
Errors per state section relative to the section's max |x| (the parity metric).  REFINE=n in the environment: n refinement steps in both solvers.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emu"))


def main():
    import pkgload
    import parity
    from oracle import oracle as O
    pkg = pkgload.load()
    kw = {}
    if "--gpu" not in sys.argv:
        import build_emu
        kw["_lib_path"] = build_emu.build()
    for refine in (0, 1):
        print("\n### %d refinement step(s)\n\n| variant | mode | cj | state | device vs truth | oracle LU vs truth | device vs oracle |\n|---|---|---|---|---|---|---|" % refine)
        allr = []
        for mk, name in ((dict(), "lco_iso"), (dict(aging="SEI"), "lco_iso_sei"), (dict(temperature=True), "lco_thermal")):
            p = pkg.petlion(pkg.LCO, **mk, **kw)
            for r in parity.solver_accuracy_rows(p, O, refine=refine):
                print("| %s | %d | %g | %d | %.1e | %.1e | %.1e |" % ((name,) + r)); allr.append((name,) + r)
        for fam in ("iso", "thermal"):
            a = np.array([r[4:] for r in allr if ("thermal" in r[0]) == (fam == "thermal")])
            print("\n%s: max device %.1e, oracle %.1e, device-vs-oracle %.1e ; median device %.1e, oracle %.1e" % (fam, a[:, 0].max(), a[:, 1].max(), a[:, 2].max(), np.median(a[:, 0]), np.median(a[:, 1])))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""The every-cell two-sample statistics of tests/test_gpu_ensemble.py for C3 (4096 cells, CC-CT-CV), for both device evaluation orders against each oracle evaluation order
(VERDICT r04 "next" 1).  Prints, asserts nothing:   python tools/experiments/c3_two_sample.py [--cells N] [--out file.json]
   device precision f64 (flux / difference form) and f64_reforder (the generated code's order)  x  oracle lco_thermal (generated order, notebook-pinned), lco_thermal_tdiff
   (T rows on differences), lco_thermal_quiet (T rows and Phi_s rows on differences)."""
import argparse
import json
import os
import sys

import numpy as np
import torch  # noqa: F401  (before the HIP library is loaded: torch brings its own ROCm runtime, and initialising it after libpetlion_hip.so's fails)

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pkgload  # noqa: E402
from oracle import oracle as O  # noqa: E402
import test_gpu_ensemble as T  # noqa: E402

ap = argparse.ArgumentParser(); ap.add_argument("--cells", type=int, default=4096); ap.add_argument("--out", default=None); ap.add_argument("--config", default="c3")
a = ap.parse_args()
pkg = pkgload.load(); O.build()
out = []
mk = dict(c3=dict(temperature=True), c4=dict(), c2=dict(), c5=dict(aging="SEI"))[a.config]
variants = dict(c3=("lco_thermal", "lco_thermal_tdiff", "lco_thermal_quiet"), c4=("lco_iso", "lco_iso_quiet"), c2=("lco_iso", "lco_iso_quiet"), c5=("nmc_iso_sei", "nmc_iso_sei_quiet"))[a.config]
for prec in ("f64", "f64_reforder") if a.config != "c5" else ("f64",):
    p = pkg.petlion(pkg.NMC if a.config == "c5" else pkg.LCO, precision=prec, **mk)
    cfg = getattr(pkg.configs, a.config)(p, a.cells)
    for variant in variants:
        r = T.two_sample(pkg, O, p, cfg, np.arange(a.cells), "%s device %s vs oracle %s" % (a.config.upper(), prec, variant), variant=variant, check=False)
        out.append(r["stats"])
if a.out:
    json.dump(out, open(a.out, "w"), indent=1)

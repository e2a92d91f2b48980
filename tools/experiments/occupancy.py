#!/usr/bin/env python3
"""What is a second resident wavefront per SIMD worth on THIS code?  (VERDICT r04 "next" 3a.)

Every built-in kernel is compiled with amdgpu_waves_per_eu(1, 1): one cell per SIMD, 512 registers per lane.  The 301-state models need 37 kB of LDS per cell, so four cells per
CU are all that fit anyway; the small variants would fit more and still run four.  This tool prices the lever on them:

    python tools/experiments/occupancy.py build          (here: hipcc cross-compiles, the libraries travel with the snapshot)
    python tools/experiments/occupancy.py run [--cells N] [--reps K]      (on the GPU box)

  * variant 8 (LCO, quadratic solid diffusion: 121 states, 20.7 kB LDS -> 7 cells per CU) and variant 9 (polynomial: 141 states, 22.1 kB -> 7) on the default grid,
  * variant 0 on the (2, 2, 2, 10) grid (67 states, 12.2 kB -> 13 cells per CU by LDS),
each built with the production flags (A) and with -DPL_WAVES_PER_EU=2 (B: 256 registers per lane, up to 8 waves per CU) and, for the small grid, =3 (C: 168 registers).
Reported per build: kernel time of a C4-style sweep (the seven-parameter jitter, 1C discharge), trajectories/s, and that the exit flags / step counts are those of build A."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
import pkgload  # noqa: E402

EXP = os.path.join(ROOT, "petlion.jl_amd", "_exp")
SMALL = (2, 2, 2, 10, 10, 10)


def libs():
    return {"builtin_w1": (os.path.join(EXP, "libplh_occ_w1.so"), []), "builtin_w2": (os.path.join(EXP, "libplh_occ_w2.so"), ["-DPL_WAVES_PER_EU=2"])}


def build():
    pkg = pkgload.load()
    os.makedirs(EXP, exist_ok=True)
    out = {}
    for name, (lib, fl) in libs().items():
        out[name] = g.build_hip(extra_flags=fl, lib=lib, variants=[0, 8, 9])
    for w in (1, 2, 3):
        out["grid_w%d" % w] = pkg.grids.library(SMALL, [0], extra_flags=["-DPL_WAVES_PER_EU=%d" % w] if w > 1 else [], suffix="_w%d" % w if w > 1 else "")
    print(json.dumps(out, indent=1))


def run(cells, reps):
    import subprocess
    # one process per library: a library is loaded once per process (PETLION_HIP_LIB), and a grid library registers under its own path
    rows = []
    for name, (lib, _) in libs().items():
        for sd in ("quadratic", "polynomial", "Fickian"):
            rows.append(_child(dict(PETLION_HIP_LIB=lib), dict(solid_diffusion=sd), None, cells, reps, "%s %s" % (name, sd)))
    for w in (1, 2, 3):
        rows.append(_child(dict(PETLION_HIP_LIB=libs()["builtin_w1"][0]), dict(N_p=2, N_s=2, N_n=2), w, cells, reps, "grid (2,2,2,10) waves_per_eu %d" % w))
    print(json.dumps(rows, indent=1))


def _child(env, mkw, grid_w, cells, reps, label):
    import subprocess
    code = r'''
import sys, json, os
sys.path.insert(0, %r)
import numpy as np, torch, pkgload
pkg = pkgload.load()
mkw = %r; grid_w = %r
gl = None
if grid_w is not None:
    gl = pkg.grids.library(%r, [0], extra_flags=["-DPL_WAVES_PER_EU=%%d" %% grid_w] if grid_w > 1 else [], suffix="_w%%d" %% grid_w if grid_w > 1 else "")
p = pkg.petlion(pkg.LCO, _grid_lib=gl, **mkw) if gl else pkg.petlion(pkg.LCO, **mkw)
cfg = pkg.configs.c4(p, %d)
Th = torch.from_numpy(np.ascontiguousarray(cfg["theta"])).cuda()
ms = []
for r in range(%d + 1):
    ens = pkg.simulate_ensemble(p, Th, cfg["protocol"], SOC=1.0, device=True, max_points=256, YP=False)
    torch.cuda.synchronize()
    if r: ms.append(float(ens.kernel_ms))
fl = ens.run_info["flag"][:, 0]
print("OCC " + json.dumps(dict(label=%r, lds=int(p.lds_bytes), states=int(p.N.tot), cells=%d, kernel_ms=float(np.mean(ms)), traj_per_s=%d / (np.mean(ms) * 1e-3),
      flags={int(k): int(v) for k, v in zip(*np.unique(fl, return_counts=True))}, steps_mean=float(ens.counters["n_steps"].mean()), newton_mean=float(ens.counters["n_newton"].mean()))))
''' % (ROOT, mkw, grid_w, SMALL, cells, reps, label, cells, cells)
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True)
    for line in r.stdout.splitlines():
        if line.startswith("OCC "):
            rec = json.loads(line[4:]); print(rec, flush=True); return rec
    print(label, "FAILED", r.stdout[-400:], r.stderr[-1200:], flush=True)
    return dict(label=label, error=r.stderr[-300:])


if __name__ == "__main__":
    ap = argparse.ArgumentParser(); ap.add_argument("what"); ap.add_argument("--cells", type=int, default=8192); ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    build() if a.what == "build" else run(a.cells, a.reps)

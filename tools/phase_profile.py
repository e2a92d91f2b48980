#!/usr/bin/env python3
"""Per-phase shader-cycle breakdown of k_integrate (profiling build libpetlion_hip_prof.so, -DPL_PHASE_TIMERS).
usage (GPU box): python tools/phase_profile.py [n_cells]"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pkgload
pkg = pkgload.load()
import torch
lib = os.path.join(ROOT, "petlion.jl_amd", "libpetlion_hip_prof.so")
src = os.path.join(ROOT, "petlion.jl_amd", "csrc", "petlion_hip.hip")
csrc = os.path.dirname(src)
newest = max(os.path.getmtime(os.path.join(csrc, f)) for f in os.listdir(csrc))
if not os.path.exists(lib) or os.path.getmtime(lib) < newest:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-unused-value", "-Wno-pass-failed", "-mllvm", "-amdgpu-function-calls=false",
                           "-DPL_PHASE_TIMERS", src, "-o", lib])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
which = sys.argv[2] if len(sys.argv) > 2 else "iso"
if which == "thermal":
    p = pkg.petlion(pkg.LCO, temperature=True, _lib_path=lib)
    kw = dict(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
    proto, soc = [dict(I=4.0, **kw), dict(dT="hold", **kw), dict(V="hold", **kw)], 0.0
elif which == "sei":
    p = pkg.petlion(pkg.NMC, aging="SEI", _lib_path=lib)
    proto, soc = [{"I": 1.0, "tf": 180.0}, {"I": "rest", "tf": 7200.0}] * 3, 0.0
else:
    p = pkg.petlion(pkg.LCO, _lib_path=lib)
    proto, soc = [{"I": -1.0}], 1.0
Th = torch.from_numpy(pkg.theta_matrix(p, n)).cuda()
for _ in range(3):
    ens = pkg.simulate_ensemble(p, Th, proto, SOC=soc, device=True, max_points=1024)
names = ["residual", "jac+factor", "solve", "newton-vec", "step-ctl", "init", "output", "TOTAL"]
cyc = ens.counters["cyc"].astype(np.float64)
c = ens.counters
print("kernel %.3f ms for %d cells; per cell: steps %.0f res %.0f jac %.0f solves %.0f" % (ens.kernel_ms, n, c["n_steps"].mean(), c["n_res"].mean(), c["n_jac"].mean(), c["n_solve"].mean()))
tot = cyc[:, 7].mean()
for k, nm in enumerate(names):
    print("  %-11s %10.0f cyc  %5.1f%%" % (nm, cyc[:, k].mean(), 100 * cyc[:, k].mean() / tot))
print("  per call: residual %.0f  jac+factor %.0f  solve %.0f" % (cyc[:, 0].mean() / (c["n_res"].mean() - c["n_jac"].mean()), cyc[:, 1].mean() / c["n_jac"].mean(), cyc[:, 2].mean() / c["n_newton"].mean()))
print("  flags per run:", [dict(zip(*np.unique(ens.run_info["flag"][:, k], return_counts=True))) for k in range(ens.run_info.shape[1])])

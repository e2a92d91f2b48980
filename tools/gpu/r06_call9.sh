#!/bin/bash
# host-path facts: microbenchmark of the pieces + the blocking call with results dropped (bench pattern) vs kept (test pattern)
out=gpurun_out/${1:-r06i}; mkdir -p $out
hipcc --offload-arch=gfx950 -O2 -o /tmp/host_ubench tools/gpu/host_ubench.hip -lpthread > /dev/null 2>&1
timeout 300 /tmp/host_ubench > $out/host_ubench.txt 2>&1
timeout 600 python - > $out/host_patterns.txt 2>&1 <<'P'
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import pkgload; pkg = pkgload.load()
import torch
for name, n in (("C2", 1024), ("C4", 8192)):
    p = pkg.petlion(pkg.LCO)
    cfg = getattr(pkg.configs, name.lower())(p, n)
    Th = np.ascontiguousarray(cfg["theta"])
    Thd = torch.from_numpy(Th).cuda()
    for _ in range(3):
        e = pkg.simulate_ensemble(p, Thd, cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"]); torch.cuda.synchronize()
    kms = e.kernel_ms
    call = lambda: pkg.simulate_ensemble(p, Th, cfg["protocol"], SOC=cfg["SOC"], max_points=cfg["max_points"])
    call()
    ts = []
    for _ in range(7):
        t1 = time.perf_counter(); call(); ts.append(time.perf_counter() - t1)           # result dropped at once (bench.py)
    tk = []
    h = None
    for _ in range(7):
        t1 = time.perf_counter(); h = call(); tk.append(time.perf_counter() - t1)       # previous result alive during the call (the r06 test)
    print("%s: kernel %.3f ms, max_points %d; dropped: median %.3f ms (%.2f), kept: median %.3f ms (%.2f)" % (name, kms, cfg["max_points"], 1e3 * np.median(ts), kms / (1e3 * np.median(ts)), 1e3 * np.median(tk), kms / (1e3 * np.median(tk))))
P
cat $out/host_ubench.txt $out/host_patterns.txt

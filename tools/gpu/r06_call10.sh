#!/bin/bash
# where a blocking host call's time goes: the library's phase trace in a bare process and inside bench.py
out=gpurun_out/${1:-r06j}; mkdir -p $out
PLH_HOST_TRACE=1 timeout 600 python - > $out/trace_bare.txt 2>&1 <<'P'
import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import pkgload; pkg = pkgload.load()
for name, n in (("C2", 1024), ("C4", 8192)):
    p = pkg.petlion(pkg.LCO)
    cfg = getattr(pkg.configs, name.lower())(p, n)
    Th = np.ascontiguousarray(cfg["theta"])
    call = lambda: pkg.simulate_ensemble(p, Th, cfg["protocol"], SOC=cfg["SOC"], max_points=cfg["max_points"])
    call(); call()
    mp = cfg["max_points"]; N = p.N.tot
    ta = []
    for _ in range(5):          # the caller's own cycle: allocate the output arrays, touch every page, release them
        t1 = time.perf_counter(); arrs = [np.empty((n, mp)) for _ in range(4)] + [np.empty((n, N)) for _ in range(2)]
        for x in arrs: x.reshape(-1)[::512] = 0.0
        t2 = time.perf_counter(); del arrs, x; t3 = time.perf_counter(); ta.append((t2 - t1, t3 - t2))
    print("%s caller's cycle alone: allocate + touch %.3f ms, release %.3f ms (median of 5)" % (name, 1e3 * np.median([a for a, b in ta]), 1e3 * np.median([b for a, b in ta])), file=sys.stderr)
    for _ in range(4):
        t1 = time.perf_counter(); call(); dt = time.perf_counter() - t1
        print("%s python wall %.3f ms" % (name, 1e3 * dt), file=sys.stderr)
P
PLH_HOST_TRACE=1 timeout 900 python bench.py --config C2 --steps 100 --warmup 10 > $out/bench_c2.json 2> $out/trace_bench_c2.txt
PLH_HOST_TRACE=1 timeout 900 python bench.py --config C4 --steps 30 --warmup 5 > $out/bench_c4.json 2> $out/trace_bench_c4.txt
grep -v amdgpu.ids $out/trace_bare.txt | tail -20; grep "plh host call" $out/trace_bench_c2.txt | tail -8; grep "plh host call" $out/trace_bench_c4.txt | tail -8
python - <<P
import json
for c in ("c2","c4"):
    d=json.load(open("$out/bench_%s.json"%c)); print(c, d["value"], d["host_inclusive"]["synchronous_pageable"]["value"])
P
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "blocking_host_call or host" -s > $out/pytest_host.txt 2>&1; tail -8 $out/pytest_host.txt

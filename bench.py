#!/usr/bin/env python3
"""bench.py -- ensemble DFN trajectory throughput on N MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W [--config C2|C3|C4|C5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch = ONE plh_integrate launch that integrates this rank's shard of the ensemble through its whole
protocol (consistent initialisation + every BDF/Newton step + stop tests / back-interpolation).  The default workload at every N is config C2 of
BASELINE.json per GPU ("Batch of 1024 LCO isothermal 1C CC discharges (identical params) on one MI355X, fp64") -- weak scaling, independent cells,
contiguous blocks, no data-path collective.  --config selects C3 (4096 thermal cells, CC-CT-CV), C4 (8192 jittered cells per GPU, the shard of the
65 536-cell sweep) or C5 (1024 NMC + SEI cells, 20 GITT pulses); their inputs are SURVEY.md 8(d)'s (petlion.jl_amd/configs.py).

`value` is the rate with the parameters already resident in HBM (the contract of this benchmark: inputs in HBM when the timed region starts; the host-inclusive
rate of SURVEY 8(d)'s shape is `host_inclusive` in the same line and is never `value`).  The same JSON line also carries
  "roofline":       what actually bounds the kernel -- VALU ISSUE.  The cell state is LDS-resident, one wavefront per SIMD: the kernel moves 0.1 % of the bytes a streaming
                    implementation would, and what a wave spends its life on is issuing fp64 VALU instructions (4 cycles each on the 16-lane fp64 pipe of a gfx950 SIMD).
                    achieved = VALU-busy cycles per second over the launch = (cells x VALU-busy cycles per trajectory, from the committed rocprofv3 PMC pass of THIS binary and
                    workload: 4 x SQ_ACTIVE_INST_VALU, profiles/*_pmc.json) / the kernel's average duration measured live (HIP events on the launch stream);
                    peak = SIMDs occupied x 2.4 GHz (every cycle of every occupied SIMD issuing VALU); frac = achieved / peak.  Beside it: the s_waitcnt-parked, LDS and scalar
                    shares of the wave's cycles, the instruction counts per step, the effective shader clock, and "traffic" (HBM bytes per launch, PMC);
  "equivalent_streaming": SURVEY.md 8(d)'s algorithmic HBM bytes (byte model x the device counters) / the same kernel duration vs the 8 TB/s HBM3E peak -- the figure r01-r04
                    reported as "roofline".  It prices bytes this kernel never moves (the LDS-resident design exceeds the ceiling a perfect streaming implementation of the
                    north star would have, frac > 1 on C2): kept under its honest name, it bounds nothing;
  "cpu_baseline":   the oracle (plain-C port of the reference path) on one host core and on all usable cores, bounded sample of the same workload;
  "host_inclusive": SURVEY 8(d)'s measurement shape -- parameters start in (pinned) host memory, per-cell summaries and sampled outputs end there --
                    as a double-buffered PLH_HOST_ASYNC pipeline: median over >= 10 calls;
  N > 1 only, "ensemble_run": the C4 sweep (8192 cells per GPU) through plh_ensemble_run (RCCL scatter -> integrate -> gather inside the C ABI),
                    block and cyclic partitions, with the per-rank kernel times (load imbalance).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
W = 8                            # fp64 word
# SURVEY.md 8(d) byte-model constants per config: states N, theta entries P, nnz(J) Z, nnz(L+U) L, algebraic block N_alg / Z_alg, scalars per saved point
CONFIGS = {
    "C2": dict(N=301, P=35, Z=2139, L=4335, NA=71, ZA=245, SC=4, cells=1024, variant="lco_iso", model=dict(cathode="LCO"),
               text="BASELINE.json configs[1] (C2): batch of %d LCO isothermal 1C CC discharges (identical params) per GPU, 301 DAEs/cell, reltol 1e-3 / abstol 1e-6, SOC 1 -> SOC_min"),
    "C3": dict(N=351, P=56, Z=2883, L=7761, NA=71, ZA=245, SC=5, cells=4096, variant="lco_thermal", model=dict(cathode="LCO", temperature=True),
               text="BASELINE.json configs[2] (C3): %d LCO cells with temperature = true per GPU, CC-CT-CV fast charge (I = 4C -> dT = hold at 40 C -> V = hold), T_amb / h_cell jitter (seed 3), 351 DAEs/cell"),
    "C4": dict(N=301, P=35, Z=2139, L=4335, NA=71, ZA=245, SC=4, cells=8192, variant="lco_iso", model=dict(cathode="LCO"),
               text="BASELINE.json configs[3] (C4): %d cells per GPU of the LCO parameter sweep (7 log-uniform factors, seed 4), 1C discharge, 301 DAEs/cell"),
    "C5": dict(N=322, P=39, Z=2269, L=4985, NA=81, ZA=314, SC=4, cells=1024, variant="nmc_iso_sei", model=dict(cathode="NMC", aging="SEI"),
               text="BASELINE.json configs[4] (C5): %d NMC cells with aging = SEI per GPU, GITT 20 x {1C 180 s ; rest 7200 s}, parameter jitter (seed 5), 322 DAEs/cell (fp64 leg)"),
}


def algorithmic_bytes(c, counters, n_pts):
    """sum over cells of the SURVEY 8(d) streaming model; init-Newton evaluations are costed at the algebraic-block sizes."""
    N, P, Z, L, NA, ZA = c["N"], c["P"], c["Z"], c["L"], c["NA"], c["ZA"]
    B_RES, B_JAC, B_FACT, B_SOLVE, B_STEP1 = W * (2 * N + P) + W * N, W * (2 * N + P) + W * Z, W * Z + W * L, W * (L + N) + W * N, 2 * W * N
    B_RES_A, B_JAC_A, B_FACT_A, B_SOLVE_A = W * (2 * N + P) + W * NA, W * (2 * N + P) + W * ZA, 2 * W * ZA, W * (ZA + 2 * NA)
    k = {f: counters[f].astype(np.float64) for f in counters.dtype.names if f != "cyc"}
    ni = k["n_init_iters"]
    n_init_runs = np.maximum(1.0, np.round((k["n_res"] - k["n_newton"] - ni) / 2.0))     # every (re)initialisation: its Newton residuals + R_diff + the shifted R_alg
    b = ((k["n_res"] - ni - 2.0 * n_init_runs) * B_RES + (k["n_jac"] - ni) * B_JAC + (k["n_fact"] - ni) * B_FACT + (k["n_solve"] - ni - n_init_runs) * B_SOLVE
         + k["sum_kp2"] * B_STEP1 + (ni + 2.0 * n_init_runs) * B_RES_A + ni * (B_JAC_A + B_FACT_A) + (ni + n_init_runs) * B_SOLVE_A + n_pts.astype(np.float64) * c["SC"] * W)
    return float(b.sum())


SIMDS = 256 * 4                  # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md, chip-level parameters)
CLOCK_PEAK_HZ = 2.4e9            # max shader clock, same table
FP64_ISSUE_CYCLES = 4            # a wave64 fp64 VALU instruction occupies the SIMD's 16-lane fp64 pipe for 4 cycles (78.6 TFLOP/s fp64 vector = 256 CU x 4 SIMD x 16 lanes x 2 x 2.4 GHz)


FP64_VECTOR_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (MI355X_MICROARCH.md): 256 CU x 4 SIMD x 16 lanes x 2 flop x 2.4 GHz


def work_figures(k, cells, n_local, secs):
    """r06: figures that reward doing LESS.  `frac` above is a busy fraction -- it rises when the kernel wastes VALU instructions; these do not:
         flop_frac                 64 lanes x (2 FMA + MUL + ADD + TRANS) fp64 instructions per second / the 78.6 TFLOP/s fp64 vector peak (work / peak; lanes that a wave
                                   masks off count as work here, `lane_utilisation` says how many are live)
         fp64_arith_share_of_valu  (FMA + MUL + ADD + TRANS) / SQ_INSTS_VALU: the share of the VALU stream that is arithmetic (r05: 0.515; the rest is moves, selects,
                                   DPP shifts, lane broadcasts, SGPR spill traffic, division scaffolding)
         lane_utilisation          SQ_THREAD_CYCLES_VALU / (64 x SQ_ACTIVE_INST_VALU): mean fraction of the 64 lanes active per VALU instruction"""
    names = ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")
    if not all(nm in k for nm in names) or "SQ_INSTS_VALU" not in k:
        return {"flop_frac": None, "fp64_arith_share_of_valu": None, "lane_utilisation": None}
    fma, mul, add, trans = (k[nm] / cells for nm in names)
    flops = 64.0 * (2.0 * fma + mul + add + trans) * n_local / secs
    return {"flop_frac": flops / (FP64_VECTOR_PEAK_TFLOPS * 1e12), "fp64_tflops": flops / 1e12, "flop_frac_what": "64 x (2 FMA + MUL + ADD + TRANS) fp64 VALU instructions per second / 78.6 TFLOP/s",
            "fp64_arith_share_of_valu": (fma + mul + add + trans) * cells / k["SQ_INSTS_VALU"],
            "lane_utilisation": (k["SQ_THREAD_CYCLES_VALU"] / (64.0 * k["SQ_ACTIVE_INST_VALU"])) if "SQ_THREAD_CYCLES_VALU" in k and "SQ_ACTIVE_INST_VALU" in k else None}


def issue_roofline(pmc, traffic, n_local, kavg_ms, klast_ms, c, ens, pkg, p):
    """the bound that bounds: VALU issue (module docstring).  Everything PMC comes from the committed pass of this binary and workload; the duration is measured live."""
    secs = kavg_ms * 1e-3
    waves_resident = min(n_local, SIMDS * max(1, (160 * 1024) // int(p.lds_bytes) // 4))      # cells resident at a time: one per SIMD (LDS: four per CU)
    simds_busy = min(n_local, SIMDS)
    peak = simds_busy * CLOCK_PEAK_HZ / 1e9                                                     # G VALU-busy cycles / s if every occupied SIMD issued VALU every cycle at 2.4 GHz
    out = {"bound": "valu_issue", "unit": "G VALU-busy cycles/s", "peak": peak,
           "peak_what": "%d occupied SIMDs x 2.4 GHz: every cycle of every SIMD that holds a cell issuing VALU (one wavefront per SIMD: LDS allows four cells per CU)" % simds_busy,
           "achieved": None, "frac": None, "traffic": traffic,
           "traffic_unit": "HBM bytes per launch (rocprofv3 PMC FETCH_SIZE x2 [gfx950 correction] + WRITE_SIZE, separate passes)",
           "hbm_utilisation": (traffic / secs / 1e9 / HBM_PEAK_GBPS) if traffic else None,
           "kernel": "k_integrate<%s>" % c["variant"], "kernel_ms_avg": kavg_ms, "kernel_ms_last_launch": klast_ms, "cells_resident_at_a_time": waves_resident}
    if not pmc or "counters" not in pmc:
        out["note"] = "no profiles/*_pmc.json for this workload: run tools/prof.sh (rocprofv3 --pmc passes) and tools/summarize_profile.py"
        return out
    k = pmc["counters"]; cells = float(pmc["cells_per_launch"])
    wave_cyc = 4.0 * k["SQ_WAVE_CYCLES"] / cells                       # shader cycles a wavefront (= a trajectory) lives; SQ_* cycle counters count quad-cycles
    steps = float(ens.counters["n_steps"].mean())
    if "SQ_ACTIVE_INST_VALU" in k:
        valu_busy = 4.0 * k["SQ_ACTIVE_INST_VALU"] / cells             # cycles of it the VALU is executing this wave's instructions (measured)
        how = "4 x SQ_ACTIVE_INST_VALU / cells"
    else:                                                                # older passes: instruction count x the fp64 issue cost (a lower bound on the 32-bit share, an upper on none)
        valu_busy = FP64_ISSUE_CYCLES * k["SQ_INSTS_VALU"] / cells
        how = "%d x SQ_INSTS_VALU / cells (no SQ_ACTIVE_INST_VALU in the pass)" % FP64_ISSUE_CYCLES
    achieved = n_local * valu_busy / secs / 1e9
    out.update({"achieved": achieved, "frac": achieved / peak,
                "valu_busy_cycles_per_trajectory": valu_busy, "valu_busy_from": how, "wave_cycles_per_trajectory": wave_cyc,
                "valu_busy_share_of_wave_cycles": valu_busy / wave_cyc,
                "valu_instructions_x4_share_of_wave_cycles": FP64_ISSUE_CYCLES * k["SQ_INSTS_VALU"] / cells / wave_cyc if "SQ_INSTS_VALU" in k else None,
                "s_waitcnt_parked_share": (k["SQ_WAIT_ANY"] / k["SQ_WAVE_CYCLES"]) if "SQ_WAIT_ANY" in k else None,
                "issue_stall_share": (k["SQ_WAIT_INST_ANY"] / k["SQ_WAVE_CYCLES"]) if "SQ_WAIT_INST_ANY" in k else None,
                "any_instruction_active_share": (k["SQ_ACTIVE_INST_ANY"] / k["SQ_WAVE_CYCLES"]) if "SQ_ACTIVE_INST_ANY" in k else None,
                "lds_busy_share": (k["SQ_ACTIVE_INST_LDS"] / k["SQ_WAVE_CYCLES"]) if "SQ_ACTIVE_INST_LDS" in k else None,
                "scalar_busy_share": (k["SQ_ACTIVE_INST_SCA"] / k["SQ_WAVE_CYCLES"]) if "SQ_ACTIVE_INST_SCA" in k else None,
                "instructions_per_step": {nm: k[key] / cells / steps for nm, key in (("valu", "SQ_INSTS_VALU"), ("salu", "SQ_INSTS_SALU"), ("lds", "SQ_INSTS_LDS")) if key in k},
                "fp64_instruction_mix_per_trajectory": {nm[14:].lower(): k[nm] / cells for nm in k if nm.startswith("SQ_INSTS_VALU_")} or None,
                "effective_clock_ghz": (wave_cyc / secs / 1e9) if n_local <= SIMDS else None,
                **work_figures(k, cells, n_local, secs),
                "pmc_file": pmc.get("file"), "pmc_binary_src": pmc.get("build_src"),
                "pmc_binary_matches": (pmc.get("build_src") in pkg.api.build_info(p)) if pmc.get("build_src") else None,
                "limiter": "one wavefront per SIMD issuing a dependent fp64 instruction stream: the VALU is busy for `valu_busy_share_of_wave_cycles` of the wave's life and the wave is "
                           "parked on s_waitcnt (LDS / DPP results of its own previous instructions) for `s_waitcnt_parked_share`; a second resident wave would fill the parked cycles "
                           "(LDS: 37 kB per cell allows four cells per CU), fewer instructions per step lower both"})
    return out


def usable_cores():
    """cores this process can actually run on: the affinity mask, capped by the cgroup CPU quota (a container on a 256-core host may own 16)"""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0]); per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
        except Exception:
            pass
    return n


def cpu_baseline(cfg_name, c, p, pkg, inp, seconds):
    """the oracle on one host core, then on every usable core, on a bounded sample of the same workload"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import parity
    from oracle import oracle as O
    runs = parity.runs_to_oracle(O, p, pkg, inp["protocol"])
    th = np.ascontiguousarray(inp["theta"][:64])
    t1 = time.perf_counter(); ok, _, _ = O.run_batch(c["variant"], th, inp["SOC"], runs, 4); per = (time.perf_counter() - t1) / 4
    n_cpu = max(8, int(seconds / per))
    t1 = time.perf_counter(); ok, tsum, _ = O.run_batch(c["variant"], th, inp["SOC"], runs, n_cpu); dt = time.perf_counter() - t1
    assert ok == n_cpu, (ok, n_cpu)
    one = {"value": n_cpu / dt, "unit": "trajectories/s", "cores": 1, "kind": "port",
           "sample": "%d trajectories of this workload (its first 64 parameter sets, cyclically) run back to back on one host core by the oracle (plain-C IDA/KLU-style port of "
                     "the reference path, oracle/ida_oracle.c); %.1f s; host reports %d logical cores; reference publishes 2.616 ms per 1C discharge on an unspecified laptop "
                     "(examples/getting_started.ipynb:183-192)" % (n_cpu, dt, os.cpu_count())}
    from concurrent.futures import ThreadPoolExecutor
    cores = usable_cores()
    chunk, deadline = max(2, int(0.25 / per)), time.perf_counter() + 0.6 * seconds

    def worker(_):
        done = 0
        while time.perf_counter() < deadline:            # time-bounded: chunks of ~0.25 s until the deadline (ctypes drops the GIL)
            assert O.run_batch(c["variant"], th, inp["SOC"], runs, chunk)[0] == chunk
            done += chunk
        return done
    t1 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        n_all = sum(ex.map(worker, range(cores)))
    dt_all = time.perf_counter() - t1
    allc = {"value": n_all / dt_all, "unit": "trajectories/s", "cores": cores, "kind": "port",
            "sample": "%d of the same trajectories on %d oracle threads (the cores this process may use: affinity / cgroup quota); %.1f s; %.1fx the one-core rate"
                      % (n_all, cores, dt_all, n_all / dt_all / one["value"])}
    return one, allc


def host_inclusive(pkg, p, inp, n_local, kernel_ms, calls=None):
    """wall time of the batched call with Theta starting in host memory and the per-cell summaries + sampled outputs (t, V per saved point) ending there:
    a depth-2 PLH_HOST_ASYNC pipeline over pinned buffers; per-call time = interval between successive completions, median of `calls` (>= 12, and enough for >= 0.5 s)."""
    import torch
    calls = calls or int(min(600, max(12, np.ceil(600.0 / max(kernel_ms, 1e-3)))))
    pipe = pkg.api.HostPipeline(p, n_local, inp["protocol"], SOC=inp["SOC"], max_points=inp["max_points"], depth=2)
    Th = inp["theta"]
    for k in range(4):                                   # warm-up: staging blocks, per-stream workspaces
        pipe.submit(k % 2, Th)
    pipe.wait(0); pipe.wait(1)
    torch.cuda.synchronize()
    done = []
    t0 = time.perf_counter()
    for k in range(calls + 2):
        slot = k % 2
        if k >= 2:
            b = pipe.wait(slot); done.append(time.perf_counter())
            assert (b["run_info"]["flag"] >= 0).all()
        if k < calls:
            pipe.submit(slot, Th)
    total = done[-1] - t0
    iv = np.diff(np.array([t0] + done))
    steady = iv[2:] if len(iv) > 4 else iv               # the first completions include the pipeline fill
    ms_med = 1e3 * float(np.median(steady))
    # the same call made synchronously through pageable memory (PLH_HOST: what an unprepared host does)
    pkg.simulate_ensemble(p, Th, inp["protocol"], SOC=inp["SOC"], max_points=inp["max_points"])
    ts, tin = [], []
    for _ in range(7):
        t1 = time.perf_counter(); e_ = pkg.simulate_ensemble(p, Th, inp["protocol"], SOC=inp["SOC"], max_points=inp["max_points"]); tin.append(e_.call_ms); del e_
        ts.append(time.perf_counter() - t1)          # (allocation of the outputs, the call, and the release of the result: the whole cycle of a caller that allocates per call)
    # the caller's side of that cycle alone, in the allocator state this process is in right now: the same arrays allocated, every page touched, released.  glibc serves
    # arrays of this size either from recycled heap pages (~0.1 ms for C2) or from fresh mappings that it unmaps again on release (~1.4 ms for C2: page faults + munmap),
    # depending on the history of the process -- which is why `value` below is bimodal between runs while the call itself is not
    tc = []
    n_, mp_, N_ = n_local, int(inp["max_points"]), p.N.tot
    for _ in range(5):
        t1 = time.perf_counter()
        arrs = [np.empty((n_, mp_)) for _ in range(4)] + [np.empty((n_, N_)) for _ in range(2)]
        for x_ in arrs:
            x_.reshape(-1)[::512] = 0.0
        del arrs, x_
        tc.append(time.perf_counter() - t1)
    pipe.close()
    return {"value": n_local / (ms_med * 1e-3), "unit": "trajectories/s", "ms_per_call_median": ms_med, "calls": calls, "aggregate_value": calls * n_local / total,
            "fraction_of_kernel_rate": kernel_ms / ms_med,
            "what": "H2D of Theta (pinned), kernel, D2H of run_info + counters + n_pts + t, V [%d points] per cell; two calls in flight on two streams (PLH_HOST_ASYNC)" % inp["max_points"],
            "synchronous_pageable": {"value": n_local / float(np.median(ts)), "ms_per_call_median": 1e3 * float(np.median(ts)), "fraction_of_kernel_rate": kernel_ms / (1e3 * float(np.median(ts))),
                                     "callers_allocation_cycle_ms": 1e3 * float(np.median(tc)),
                                     "inside_plh_integrate": {"value": n_local / (1e-3 * float(np.median(tin))), "ms_per_call_median": float(np.median(tin)), "fraction_of_kernel_rate": kernel_ms / float(np.median(tin)),
                                                              "what": "wall time of the plh_integrate call alone (staging H2D, kernel, copies back into the caller's arrays); the difference to the line above is the caller's "
                                                                      "own allocation and release of 13 MB (C2) ... 100 MB (C4) of output arrays per call"},
                                     "what": "the whole cycle of a caller that allocates per call: numpy arrays allocated, ONE blocking PLH_HOST call with all outputs (t, V, I, SOC, Y, YP, ...), the "
                                             "result released.  Bimodal between runs with the allocator's state (callers_allocation_cycle_ms); the call alone is inside_plh_integrate"}}


def predicted_scaling(pkg, p, n_gpus=8, per_gpu=8192, reps=3):
    """A PREDICTION, not a measurement, of the 8-GPU run of BASELINE configs[3] (65 536 jittered cells, 8192 per GPU): cells are independent and the data path has no collective
    ("limited by load imbalance and launch overhead, not bandwidth"), so the 8-GPU launch time is the slowest shard's kernel time.  The eight shards of each
    partition (PLH_PART_BLOCK: contiguous blocks; PLH_PART_CYCLIC: cell mod 8) are integrated ONE AFTER THE OTHER on this GPU; max / mean over the shards is the
    load-imbalance efficiency the 8-GPU run will show, 65 536 / max the predicted aggregate rate (before the scatter / gather, whose one-rank host-to-host cost is reported
    beside it: plh_ensemble_run through a one-rank communicator, no RCCL call)."""
    import torch
    n_tot = n_gpus * per_gpu
    Th_all = np.ascontiguousarray(pkg.configs.c4(p, n_tot)["theta"])
    out = {"what": "PREDICTION from one GPU: the %d shards of the %d-cell C4 sweep timed back to back on this GPU, block and cyclic partitions; not a SCALE record" % (n_gpus, n_tot)}
    for part in ("block", "cyclic"):
        ms = []
        for g_ in range(n_gpus):
            idx = np.arange(g_ * per_gpu, (g_ + 1) * per_gpu) if part == "block" else np.arange(g_, n_tot, n_gpus)
            Th = torch.from_numpy(np.ascontiguousarray(Th_all[idx])).cuda()
            t = []
            for r in range(reps + 1):
                ens = pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0, device=True, max_points=256, YP=False)
                torch.cuda.synchronize()
                if r:
                    t.append(float(ens.kernel_ms))
            ms.append(float(np.mean(t)))
        ms = np.array(ms)
        out[part] = {"shard_kernel_ms": [float(x) for x in ms], "max_over_mean": float(ms.max() / ms.mean()), "predicted_efficiency": float(ms.mean() / ms.max()),
                     "predicted_8gpu_trajectories_per_s": n_tot / (ms.max() * 1e-3), "single_gpu_trajectories_per_s_mean_shard": per_gpu / (ms.mean() * 1e-3)}
    # host-to-host cost of the C ABI's multi-GPU entry with one rank (scatter / gather degenerate to host staging): what every rank adds to its kernel time
    try:
        from petlion_jl_amd import distributed as pd
        comm = pd.RcclComm(p._lib, 1, 0, bytes(pd.RcclComm.unique_id(p._lib)), device=torch.cuda.current_device())
        Th = Th_all[:per_gpu]
        pd.ensemble_run_capi(comm, p, Th, [{"I": -1.0}], 1.0, n_cells=per_gpu, partition="block")
        t1 = time.perf_counter(); res = pd.ensemble_run_capi(comm, p, Th, [{"I": -1.0}], 1.0, n_cells=per_gpu, partition="block"); dt = time.perf_counter() - t1
        out["ensemble_run_one_rank"] = {"wall_ms": 1e3 * dt, "kernel_ms": float(res[3][0]), "host_overhead_ms": 1e3 * dt - float(res[3][0]), "cells": per_gpu}
        comm.close()
    except Exception as e:                                   # noqa: BLE001
        out["ensemble_run_one_rank"] = {"error": repr(e)[:200]}
    return out


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: re-run this command under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N` on a free local port"""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0"); env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    rc = subprocess.call(cmd, env=env)
    if rc != 0:
        raise SystemExit(rc)


def closure_cases(p, inp):
    """the first run of the config's protocol with its constant current written as a closure of (t, Y) and as a closure of the cell voltage (1e-9 C-rate per volt: the same
    workload to 1e-9, with two derivative programs in the Newton matrix)"""
    first = dict(inp["protocol"][0])
    Ival = float(first["I"])
    ps = p.ind["Φ_s"]
    clo = dict(first); clo["I"] = lambda t, Y, P_: Ival + 0.0 * t + 0.0 * Y[0]
    clo_y = dict(first); clo_y["I"] = lambda t, Y, P_: Ival + 1e-9 * (Y[ps.start] - Y[ps.stop - 1])
    return clo, clo_y


def general_path(pkg, p, inp, Theta, n_local, kernel_ms, reps=5):
    """what a call pays for the features beyond constant inputs -- each runs in the smallest k_integrate instantiation that has it (GenFlag, csrc/dfn_integrate.h):
    kernel time of the same workload with the feature switched on, relative to the plain kernel of the timed region"""
    import torch
    proto = inp["protocol"]
    first = dict(proto[0])
    if "I" not in first or isinstance(first["I"], str):
        return None
    Ival, tf = float(first["I"]), float(first.get("tf", 1e6))
    tab = dict(first); tab["I"] = ([0.0, 1e7], [Ival, Ival])                         # the same constant current as a table / as a closure of t and the state
    clo, clo_y = closure_cases(p, inp)
    cases = [("stop times (opts.tstops; one stop beyond every run, so that the step sequence is the plain one)", proto, dict(tstops=[1e7]), None),
             ("state dump (outputs = :all)", proto, {}, "all"),
             ("table input", [tab] + proto[1:], {}, None),
             ("closure input", [clo] + proto[1:], {}, None),
             ("closure of the state with its derivative in the Newton matrix (general control row)", [clo_y] + proto[1:], {}, None),
             ("refine = 1", proto, dict(refine=1), None),
             # the same two closures COMPILED into the kernels (closure_lib.py: hipcc once per closure set; the libraries of the four configs are built by
             # tools/build_bench_closures.py and travel with the tree) -- what the reference does with a user's closure (scalar_residual.jl:231-416)
             ("closure input, compiled (closure_lib)", [clo] + proto[1:], dict(_compile=True), None),
             ("closure of the state with its derivative in the Newton matrix, compiled (closure_lib)", [clo_y] + proto[1:], dict(_compile=True), None)]
    skeys = [k for k in pkg.configs.SWEEP_KEYS if k in p.θ_keys]
    if all(all(not callable(r[m]) and not isinstance(r[m], (tuple, list)) for m in ("I", "V", "P", "dT") if m in r) for r in proto):      # constant / :rest / :hold inputs (r05: :hold legs too)
        cases.append(("forward sensitivities dY/dtheta, dV/dtheta for %d parameters (%s): plh_integrate_sens" % (len(skeys), ", ".join(skeys)), proto, {}, None))
    out = {}
    for name, pr, okw, outputs in cases:
        o = pkg.Opts()
        if okw.pop("_compile", False):
            try:
                p.compile_closures(pr)          # (cached under petlion.jl_amd/_closures/; a fresh tree compiles it here: hipcc, ~30 s)
            except Exception as e:              # no compiler on this machine: the interpreted figures above stand
                out[name] = {"skipped": "closure library could not be built: %s" % str(e)[:200]}
                continue
        for k, v in okw.items():
            setattr(o, k, v)
        ms = []
        for r in range(reps + 1 if "sensitivities" not in name else 2):
            ens = pkg.simulate_ensemble(p, Theta, pr, SOC=inp["SOC"], device=True, opts=o, max_points=inp["max_points"], outputs=outputs, YP=False,
                                        sens=skeys if "sensitivities" in name else None)
            torch.cuda.synchronize()
            if r:
                ms.append(float(ens.kernel_ms))
        assert (ens.run_info["flag"] >= 0).all(), name
        if "compiled" in name:
            assert p._lib.plh_last_integrate_compiled(p._h) == 1, name
        out[name] = {"kernel_ms": float(np.mean(ms)), "trajectories_per_s": n_local / (np.mean(ms) * 1e-3), "vs_plain_kernel": kernel_ms / float(np.mean(ms))}
        del ens
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400, help="timed launches (default: 0.6 s of C2 launches; the kernel needs 1.5 ms)")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default="C2", choices=sorted(CONFIGS))
    ap.add_argument("--cells-per-gpu", type=int, default=0)
    ap.add_argument("--precision", default="f64", choices=("f64", "mixed", "f64_reforder"))
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target length of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--leg-timeout", type=int, default=240, help="N > 1: watchdog of the plh_ensemble_run leg (seconds)")
    ap.add_argument("--no-extras", action="store_true", help="skip the host-inclusive pipeline, the copy-bandwidth probe and (N > 1) the plh_ensemble_run leg")
    args = ap.parse_args()
    c = CONFIGS[args.config]

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: spawn the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1) through torch.distributed.run -- the same launch the
        # driver uses -- and relay rank 0's line
        return spawn_ranks(args.gpus)
    # ONE JSON line on stdout whatever the libraries underneath print (RCCL's version banner, the driver's notes): everything else goes to stderr
    real_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)

    def emit(obj):
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(obj) + "\n").encode())
    if args.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d, or unset WORLD_SIZE and let bench.py spawn its ranks" % (args.gpus, world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")
    ndev = torch.cuda.device_count()
    local_dev = local_rank % ndev            # normally identity; lets the multi-process path be exercised on a 1-GPU box
    torch.cuda.set_device(local_dev)
    dev = torch.device("cuda", local_dev)
    backend = os.environ.get("BENCH_BACKEND", "nccl")     # "nccl" == RCCL over xGMI; "gloo" only for testing the rank logic
    cdev = dev if backend == "nccl" else torch.device("cpu")   # device of the collective payloads
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    import __graft_entry__ as g
    if world > 1:                                  # one rank per node builds (a no-op when the in-tree library is current), the others wait
        if local_rank == 0:
            g.build_hip()
        dist.barrier()
    else:
        g.build_hip()
    import pkgload
    pkg = pkgload.load()
    from petlion_jl_amd import distributed as pd

    mk = dict(c["model"]); cathode = mk.pop("cathode")
    p = pkg.petlion(getattr(pkg, cathode), precision=args.precision, device=local_dev, **mk)
    n_local = args.cells_per_gpu or c["cells"]
    n_total = n_local * world
    # this rank's shard: the inputs are a counter-based function of the global cell index (petlion.jl_amd/configs.py), so no scatter is needed to build them
    make = getattr(pkg.configs, args.config.lower())
    inp = make(p, n_local) if args.config == "C2" else make(p, n_local, first=rank * n_local)
    Theta = torch.from_numpy(np.ascontiguousarray(inp["theta"])).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        # YP = False: the reference's default output set (opts.outputs = (:t, :V); var_keep.YP is off) -- Y_final, every per-step scalar, run_info and the counters are written
        return pkg.simulate_ensemble(p, Theta, inp["protocol"], SOC=inp["SOC"], device=True, stream=stream, max_points=inp["max_points"], YP=False)

    for _ in range(args.warmup):
        ens = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    # The launches are asynchronous on torch's current stream (device pointers in, device pointers out; the per-cell summaries are read after
    # the loop), so the K steps run back to back; e0/e1 are recorded on that same stream around them.
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(args.steps):
        ens = step()
    e1.record()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    elapsed = float(el.item())

    # ---- per-cell results: correctness guard + counters (outside the timed region) ----
    flags = ens.run_info["flag"]
    assert (flags >= 0).all(), "solver failure in %d cells" % int((flags < 0).sum())
    if args.config == "C2":
        assert (flags[:, 0] == 3).all(), "every C2 cell must end on SOC_min (flag 3), got %r" % np.unique(flags)
        assert np.abs(ens.run_info["t_end"][:, 0] - 3600.0).max() < 1e-5
    bytes_launch = algorithmic_bytes(c, ens.counters, ens.n_pts.cpu().numpy())
    kavg_ms = e0.elapsed_time(e1) / args.steps     # average launch duration over the timed region (HIP events on the launch stream)
    klast_ms = float(ens.kernel_ms)                # the library's own event pair around the last launch
    kms = torch.tensor([kavg_ms], dtype=torch.float64, device=cdev)
    if world > 1:
        allk = [torch.zeros_like(kms) for _ in range(world)]
        dist.all_gather(allk, kms)
        rank_kernel_ms = [float(x.item()) for x in allk]
    else:
        rank_kernel_ms = [kavg_ms]

    # PMC counters cannot be collected from inside the timed process: instruction counts, VALU-busy cycles and HBM traffic per launch are those committed under profiles/ for
    # this exact workload and binary (the same command under rocprofv3 --pmc in separate passes: tools/prof.sh -> tools/summarize_profile.py -> profiles/<round>_<config>_pmc.json).
    # Instruction counts are a property of (binary, workload), not of the run: the file records the source hash of the binary it was taken from and a mismatch is reported.
    pmc, traffic = None, None
    try:
        import glob
        cands = []
        for f in glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")):
            tj = json.load(open(f))
            if tj.get("cells_per_launch") == n_local and tj.get("workload") == args.config and tj.get("precision", "f64") == args.precision:
                cands.append((os.path.basename(f), tj))
        if cands:
            pmc = sorted(cands)[-1][1]; pmc["file"] = "profiles/" + sorted(cands)[-1][0]
            traffic = pmc.get("hbm_bytes_per_launch")
        else:                                              # r04 and earlier: traffic only
            for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic*.json")), reverse=True):
                tj = json.load(open(f))
                if tj.get("cells_per_launch") == n_local and tj.get("workload") == args.config and tj.get("precision", "f64") == args.precision:
                    traffic = float(tj["hbm_bytes_per_launch"]); break
    except Exception:
        pmc, traffic = None, None

    out = None
    if rank == 0:
        traj_s = n_total * args.steps / elapsed
        achieved = bytes_launch / (kavg_ms * 1e-3) / 1e9
        out = {
            "metric": "DFN full-discharge trajectories/sec (ensemble)" if args.config in ("C2", "C4") else "DFN protocol trajectories/sec (ensemble)",
            "value": traj_s, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": {"f64": "f64", "mixed": "f64 (fp32 storage of the Newton-matrix factors)", "f64_reforder": "f64 (finite-volume rows in the reference's operation order)"}[args.precision],
            "data": "synthetic",
            "config": {"workload": c["text"] % n_local, "cells_per_gpu": n_local, "cells_total": n_total, "sharding": "independent cells, contiguous blocks, no data-path collective",
                       "outputs": "per cell: t, V, I, SOC (and T_avg) at every saved point, Y_final, run_info, counters (YP_final not requested: the reference keeps YP only with var_keep.YP)",
                       "steps_per_trajectory": float(ens.counters["n_steps"].mean()), "newton_iters_per_trajectory": float(ens.counters["n_newton"].mean()),
                       "rank_kernel_ms": rank_kernel_ms},
            "roofline": issue_roofline(pmc, traffic, n_local, kavg_ms, klast_ms, c, ens, pkg, p),
            "equivalent_streaming": {"what": "SURVEY.md 8(d) algorithmic bytes (streaming byte model x the device counters) / kernel duration vs the HBM3E peak: the figure r01-r04 "
                                             "reported as roofline.  The LDS-resident kernel never moves these bytes (see roofline.traffic); a value above 1 means it beats the ceiling a "
                                             "perfect HBM-streaming implementation of the north star's design would have.  Not a utilisation of anything.",
                                     "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                                     "algorithmic_bytes_per_launch": bytes_launch, "algorithmic_bytes_per_trajectory": bytes_launch / n_local},
        }

    # ---- N > 1: the C4 sweep through the C ABI's own multi-GPU entry (RCCL scatter -> integrate -> gather), block and cyclic partitions ----
    # The timed region above is complete and `out` holds the line; this extra leg must not be able to lose it: exceptions are caught on every rank (the ranks agree on
    # success through an all_reduce), and a watchdog emits the line without the leg if a collective hangs.
    if world > 1 and not args.no_extras and backend == "nccl":
        import threading

        def give_up():
            if rank == 0:
                out["ensemble_run"] = {"error": "plh_ensemble_run leg did not finish within %d s (watchdog)" % args.leg_timeout}
                emit(out)
            os._exit(0)
        dog = threading.Timer(args.leg_timeout, give_up); dog.daemon = True; dog.start()
        ens_run, comm = {}, None

        def all_ok(ok):
            f = torch.tensor([1 if ok else 0], device=cdev); dist.all_reduce(f, op=dist.ReduceOp.MIN); return bool(f.item())
        try:
            p4 = p if args.config in ("C2", "C4") and args.precision == "f64" else pkg.petlion(pkg.LCO, device=local_dev)
            uid = torch.zeros(128, dtype=torch.uint8, device=cdev)
            if rank == 0:
                uid = torch.frombuffer(bytearray(pd.RcclComm.unique_id(p4._lib)), dtype=torch.uint8).to(cdev)
            dist.broadcast(uid, src=0)
            err = None
            try:
                comm = pd.RcclComm(p4._lib, world, rank, bytes(uid.cpu().numpy().tobytes()), device=local_dev)
            except Exception as e:                                   # noqa: BLE001
                err = "plh_comm_create: %r" % (e,)
            if not all_ok(err is None):
                raise RuntimeError(err or "plh_comm_create failed on another rank")
            n4 = 8192 * world
            Th4 = pkg.configs.c4(p4, n4)["theta"] if rank == 0 else None
            for part in ("block", "cyclic"):
                pd.ensemble_run_capi(comm, p4, Th4, [{"I": -1.0}], 1.0, n_cells=n4, partition=part)          # warm-up (RCCL channels, staging blocks)
                dist.barrier(); t1 = time.perf_counter()
                res = pd.ensemble_run_capi(comm, p4, Th4, [{"I": -1.0}], 1.0, n_cells=n4, partition=part)
                dist.barrier(); dt = time.perf_counter() - t1
                if rank == 0:
                    info, cnt, _, ms = res
                    ens_run[part] = {"wall_ms": 1e3 * dt, "trajectories_per_s_host_to_host": n4 / dt, "rank_kernel_ms": [float(x) for x in ms],
                                     "kernel_ms_spread": float(ms.max() / ms.min()), "steps_per_cell_mean": float(cnt["n_steps"].mean()),
                                     "all_cells_finished": bool(np.isin(info["flag"][:, 0], (1, 3)).all())}
            if rank == 0:
                ens_run["what"] = ("plh_ensemble_run over %d ranks: %d C4 cells (8192 per GPU) from rank 0's host memory -- ncclBroadcast of the shape, grouped ncclSend/ncclRecv scatter of "
                                   "Theta, one plh_integrate per rank, gather of run_info / counters / Y_final to rank 0; wall time includes the host-side permutation and copies" % (world, n4))
        except Exception as e:                                       # noqa: BLE001
            ens_run["error"] = repr(e)
        finally:
            dog.cancel()
            try:
                if comm is not None:
                    comm.close()
            except Exception:                                        # noqa: BLE001
                pass
        if rank == 0:
            out["ensemble_run"] = ens_run

    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            one, allc = cpu_baseline(args.config, c, p, pkg, inp, args.cpu_seconds)
            out["cpu_baseline"] = one
            out["cpu_baseline_all_cores"] = allc
        if world == 1 and not args.no_extras:
            out["host_inclusive"] = host_inclusive(pkg, p, inp, n_local, kavg_ms)
            out["general_path"] = general_path(pkg, p, inp, Theta, n_local, kavg_ms)
            # like-for-like with r01-r03, whose timed launches also stored YP_final (and with it YP of the previous point at every step): the same workload with YP = True
            ms = []
            for r in range(6):
                e2 = pkg.simulate_ensemble(p, Theta, inp["protocol"], SOC=inp["SOC"], device=True, max_points=inp["max_points"], YP=True)
                torch.cuda.synchronize()
                if r:
                    ms.append(float(e2.kernel_ms))
            out["with_YP_final"] = {"kernel_ms": float(np.mean(ms)), "trajectories_per_s": n_local / (np.mean(ms) * 1e-3), "vs_timed_region": kavg_ms / float(np.mean(ms)),
                                    "what": "the timed region requests the reference's default output set (YP not kept: var_keep.YP is off); with YP_final requested the kernel also stores YP "
                                            "of the previous accepted point per step (the r01-r03 bench lines were measured this way)"}
            if args.config == "C4" and args.precision == "f64":
                out["predicted_scaling"] = predicted_scaling(pkg, p)
            # measured device-to-device copy bandwidth of this box (read + write bytes), the second peak SURVEY 8(d) asks to quote
            a = torch.empty(1 << 28, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
            b.copy_(a); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record(); torch.cuda.synchronize()
            out["roofline"]["measured_copy_peak"] = 10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        emit(out)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py -- ensemble DFN trajectory throughput on N MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W [--config C2|C3|C4|C5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch = ONE plh_integrate launch that integrates this rank's shard of the ensemble through its whole
protocol (consistent initialisation + every BDF/Newton step + stop tests / back-interpolation).  The default workload at every N is config C2 of
BASELINE.json per GPU ("Batch of 1024 LCO isothermal 1C CC discharges (identical params) on one MI355X, fp64") -- weak scaling, independent cells,
contiguous blocks, no data-path collective.  --config selects C3 (4096 thermal cells, CC-CT-CV), C4 (8192 jittered cells per GPU, the shard of the
65 536-cell sweep) or C5 (1024 NMC + SEI cells, 20 GITT pulses); their inputs are SURVEY.md 8(d)'s (petlion.jl_amd/configs.py).

`value` is the rate with the parameters already resident in HBM (the contract of this benchmark).  The same JSON line also carries
  "roofline":       algorithmic HBM bytes per launch (SURVEY.md 8(d) byte model x the device counters) / the integrate kernel's average duration
                    (HIP events on its stream) vs the 8 TB/s HBM3E peak -- the metric BASELINE.json names.  The kernel is LDS-resident: its measured HBM
                    traffic ("traffic", rocprofv3 PMC) is a fraction of a percent of the model bytes, so "frac" is an equivalent-streaming rate, not HBM
                    utilisation; "hbm_utilisation" is the real one and "limiter" names what bounds the kernel;
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
    ts = []
    for _ in range(5):
        t1 = time.perf_counter(); pkg.simulate_ensemble(p, Th, inp["protocol"], SOC=inp["SOC"], max_points=inp["max_points"]); ts.append(time.perf_counter() - t1)
    pipe.close()
    return {"value": n_local / (ms_med * 1e-3), "unit": "trajectories/s", "ms_per_call_median": ms_med, "calls": calls, "aggregate_value": calls * n_local / total,
            "fraction_of_kernel_rate": kernel_ms / ms_med,
            "what": "H2D of Theta (pinned), kernel, D2H of run_info + counters + n_pts + t, V [%d points] per cell; two calls in flight on two streams (PLH_HOST_ASYNC)" % inp["max_points"],
            "synchronous_pageable": {"value": n_local / float(np.median(ts)), "ms_per_call_median": 1e3 * float(np.median(ts)),
                                     "what": "one blocking PLH_HOST call at a time through freshly allocated pageable numpy arrays, all outputs (t, V, I, SOC, Y, YP, ...): "
                                             "staging H2D, kernel, D2H through the pinned bounce buffer, memcpy; its spread between runs (r01: 221 k vs 341 k traj/s) is the page-fault "
                                             "cost of first-touch output arrays, which depends on the allocator state of the calling process"}}


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
    if all(not isinstance(r.get("I", 0.0), str) or r["I"] == "rest" for r in proto) and all(set(r) & {"I", "V", "P"} for r in proto):
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
    ap.add_argument("--precision", default="f64", choices=("f64", "mixed"))
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

    # measured HBM traffic per launch: PMC counters cannot be collected from inside the timed process, so the value is the one committed under profiles/ for this
    # exact workload (same command under rocprofv3 --pmc, tools/prof.sh); null otherwise
    traffic = None
    try:
        import glob
        for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic*.json")), reverse=True):
            tj = json.load(open(f))
            if tj.get("cells_per_launch") == n_local and tj.get("workload") == args.config and tj.get("precision", "f64") == args.precision:
                traffic = float(tj["hbm_bytes_per_launch"]); break
    except Exception:
        traffic = None

    out = None
    if rank == 0:
        traj_s = n_total * args.steps / elapsed
        achieved = bytes_launch / (kavg_ms * 1e-3) / 1e9
        out = {
            "metric": "DFN full-discharge trajectories/sec (ensemble)" if args.config in ("C2", "C4") else "DFN protocol trajectories/sec (ensemble)",
            "value": traj_s, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64" if args.precision == "f64" else "f64 (fp32 storage of the Newton-matrix factors)",
            "data": "synthetic",
            "config": {"workload": c["text"] % n_local, "cells_per_gpu": n_local, "cells_total": n_total, "sharding": "independent cells, contiguous blocks, no data-path collective",
                       "outputs": "per cell: t, V, I, SOC (and T_avg) at every saved point, Y_final, run_info, counters (YP_final not requested: the reference keeps YP only with var_keep.YP)",
                       "steps_per_trajectory": float(ens.counters["n_steps"].mean()), "newton_iters_per_trajectory": float(ens.counters["n_newton"].mean()),
                       "rank_kernel_ms": rank_kernel_ms},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, from profiles/*_traffic*.json)",
                         "hbm_utilisation": (traffic / (kavg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                         "limiter": "dependent-instruction latency at one wavefront per SIMD (LDS-resident cell state; VALU-active / s_waitcnt shares in profiles/): 'achieved' and 'frac' "
                                    "price the SURVEY 8(d) streaming model's bytes, which this kernel never moves -- read them as an equivalent-streaming rate, 'hbm_utilisation' is the real one",
                         "kernel": "k_integrate<%s>" % c["variant"], "kernel_ms_avg": kavg_ms, "kernel_ms_last_launch": klast_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "algorithmic_bytes_per_trajectory": bytes_launch / n_local},
        }

    # ---- N > 1: the C4 sweep through the C ABI's own multi-GPU entry (RCCL scatter -> integrate -> gather), block and cyclic partitions ----
    # The timed region above is complete and `out` holds the line; this extra leg must not be able to lose it: exceptions are caught on every rank (the ranks agree on
    # success through an all_reduce), and a watchdog emits the line without the leg if a collective hangs.
    if world > 1 and not args.no_extras and backend == "nccl":
        import threading

        def give_up():
            if rank == 0:
                out["ensemble_run"] = {"error": "plh_ensemble_run leg did not finish within %d s (watchdog)" % args.leg_timeout}
                print(json.dumps(out), flush=True)
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
            # measured device-to-device copy bandwidth of this box (read + write bytes), the second peak SURVEY 8(d) asks to quote
            a = torch.empty(1 << 28, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
            b.copy_(a); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record(); torch.cuda.synchronize()
            out["roofline"]["measured_copy_peak"] = 10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

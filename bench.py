#!/usr/bin/env python3
"""bench.py -- ensemble DFN full-discharge throughput on N MI355X (BASELINE.json metric), one process per GPU.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

A "step" = one pass of the hot path over one batch = ONE plh_integrate launch that integrates this rank's shard of the
ensemble from t = 0 to the stop condition (consistent initialisation + every BDF/Newton step + stop/back-interpolation).
Workload at every N: config C2 of BASELINE.json per GPU ("Batch of 1024 LCO isothermal 1C CC discharges (identical params)
on one MI355X, fp64") -- weak scaling: 1024 cells per GPU, so N GPUs integrate N*1024 cells per step (the C4 sharding
pattern: independent cells, contiguous blocks, no data-path collective).
Parameters are resident in HBM before the timed region; RCCL is used only outside it (scatter of the parameter rows before,
gather of the per-cell summaries after) -- that is the whole communication the path has.

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), including
  "roofline":     algorithmic HBM bytes per launch (SURVEY.md 8(d) byte model x the device counters) / the integrate
                  kernel's average duration measured with HIP events on its stream, vs the 8 TB/s HBM3E peak;
  "cpu_baseline": the oracle (plain-C port of the reference path) timed on one host core on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CELLS_PER_GPU = 1024
HBM_PEAK_GBPS = 8000.0           # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
# SURVEY.md 8(d) byte model for C1/C2/C4: N=301, P=35, Z=2139, L=4335 (fp64, w = 8 B)
W, N_ST, P_TH, Z_NNZ, L_LU = 8, 301, 35, 2139, 4335
N_ALG, Z_ALG = 71, 245
B_RES = W * (2 * N_ST + P_TH) + W * N_ST
B_JAC = W * (2 * N_ST + P_TH) + W * Z_NNZ
B_FACT = W * Z_NNZ + W * L_LU
B_SOLVE = W * (L_LU + N_ST) + W * N_ST
B_STEP1 = 2 * W * N_ST                      # x (k+2) per step
B_RES_A, B_JAC_A, B_FACT_A, B_SOLVE_A = W * (2 * N_ST + P_TH) + W * N_ALG, W * (2 * N_ST + P_TH) + W * Z_ALG, 2 * W * Z_ALG, W * (Z_ALG + 2 * N_ALG)
B_PT = 4 * W                                # t, V, I, SOC per saved point


def algorithmic_bytes(counters, n_pts):
    """sum over cells of the SURVEY 8(d) model; init-Newton evaluations are costed at the algebraic-block sizes."""
    c = {k: counters[k].astype(np.float64) for k in counters.dtype.names}
    ni = c["n_init_iters"]
    n_res_main = c["n_res"] - ni - 2.0          # init: one R_alg per iteration + R_diff + the shifted R_alg
    n_jac_main = c["n_jac"] - ni
    n_fact_main = c["n_fact"] - ni
    n_solve_main = c["n_solve"] - ni - 1.0
    b = (n_res_main * B_RES + n_jac_main * B_JAC + n_fact_main * B_FACT + n_solve_main * B_SOLVE + c["sum_kp2"] * B_STEP1
         + (ni + 2.0) * B_RES_A + ni * (B_JAC_A + B_FACT_A) + (ni + 1.0) * B_SOLVE_A + n_pts.astype(np.float64) * B_PT)
    return float(b.sum())


def usable_cores():
    """cores this process can actually run on: the affinity mask, capped by the cgroup CPU quota (a container on a 256-core host may own 8)"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--cells-per-gpu", type=int, default=CELLS_PER_GPU)
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target length of the CPU-baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
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
    if world > 1:                                  # one rank per node builds (a no-op when the in-tree libraries are current), the others wait
        if local_rank == 0:
            g.build_hip()
        dist.barrier()
    else:
        g.build_hip()
    import pkgload
    pkg = pkgload.load()
    from petlion_jl_amd import distributed as pd

    p = pkg.petlion(pkg.LCO)
    n_local = args.cells_per_gpu
    n_total = n_local * world
    protocol = [{"I": -1.0}]                       # simulate(p, I=-1, SOC=1): full 1C discharge to the stop condition

    # ---- ensemble scatter (RCCL, outside the timed region): rank 0 owns Theta ----
    if world > 1:
        mine = torch.empty(n_local, len(p.θ_keys), dtype=torch.float64, device=cdev)
        try:
            if rank == 0:
                full = torch.from_numpy(pkg.theta_matrix(p, n_total)).to(cdev)
                dist.scatter(mine, [c.contiguous() for c in full.chunk(world, dim=0)], src=0)
            else:
                dist.scatter(mine, None, src=0)
        except (RuntimeError, NotImplementedError):      # a backend without scatter: broadcast the matrix, keep the own block
            full = torch.from_numpy(pkg.theta_matrix(p, n_total)).to(cdev) if rank == 0 else torch.empty(n_total, len(p.θ_keys), dtype=torch.float64, device=cdev)
            dist.broadcast(full, src=0)
            mine = full[rank * n_local:(rank + 1) * n_local].clone()
        Theta = mine.to(dev)
    else:
        Theta = torch.from_numpy(pkg.theta_matrix(p, n_local)).to(dev)
    stream = torch.cuda.current_stream().cuda_stream

    def step():
        return pkg.simulate_ensemble(p, Theta, protocol, SOC=1.0, device=True, stream=stream, max_points=256)

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

    # ---- per-cell results: correctness guard + counters (gather of summaries, outside the timed region) ----
    flags = ens.run_info["flag"][:, 0]
    assert (flags == 3).all(), "every C2 cell must end on SOC_min (flag 3), got %r" % np.unique(flags)
    assert np.abs(ens.run_info["t_end"][:, 0] - 3600.0).max() < 1e-5
    summ = pd.summarize(ens)
    if world > 1:
        mine_s = torch.from_numpy(summ).to(cdev)
        try:
            parts = [torch.empty_like(mine_s) for _ in range(world)] if rank == 0 else None
            dist.gather(mine_s, parts, dst=0)
        except (RuntimeError, NotImplementedError):      # a backend without gather
            parts = [torch.empty_like(mine_s) for _ in range(world)]
            dist.all_gather(parts, mine_s)
        if rank == 0:
            allsum = torch.cat(parts).cpu().numpy()
            assert (allsum[:, 0] == 3).all()
    bytes_launch = algorithmic_bytes(ens.counters, ens.n_pts.cpu().numpy())
    kavg_ms = e0.elapsed_time(e1) / args.steps     # average launch duration over the timed region (HIP events on the launch stream)
    klast_ms = float(ens.kernel_ms)                # the library's own event pair around the last launch

    # measured HBM traffic per launch: PMC counters cannot be collected from inside the timed process, so the value is the one
    # committed under profiles/ for this exact workload (same command under rocprofv3 --pmc, see tools/prof.sh); null otherwise
    traffic = None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_traffic.json")))
        if cands:
            tj = json.load(open(cands[-1]))
            if tj.get("cells_per_launch") == n_local and tj.get("workload") == "C2":
                traffic = float(tj["hbm_bytes_per_launch"])
    except Exception:
        traffic = None
    if rank == 0:
        traj_s = n_total * args.steps / elapsed
        out = {
            "metric": "DFN full-discharge trajectories/sec (ensemble)", "value": traj_s, "unit": "trajectories/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE.json configs[1] (C2): batch of %d LCO isothermal 1C CC discharges (identical params) per GPU, "
                                   "301 DAEs/cell, reltol 1e-3 / abstol 1e-6, SOC 1 -> SOC_min" % n_local,
                       "cells_per_gpu": n_local, "cells_total": n_total, "sharding": "independent cells, contiguous blocks, no data-path collective",
                       "steps_per_trajectory": float(ens.counters["n_steps"].mean()), "newton_iters_per_trajectory": float(ens.counters["n_newton"].mean())},
            "roofline": {"bound": "hbm", "achieved": bytes_launch / (kavg_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": bytes_launch / (kavg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": traffic,
                         "traffic_unit": "bytes per launch (rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, from profiles/*_traffic.json)",
                         "kernel": "k_integrate", "kernel_ms_avg": kavg_ms, "kernel_ms_last_launch": klast_ms, "algorithmic_bytes_per_launch": bytes_launch,
                         "algorithmic_bytes_per_trajectory": bytes_launch / n_local,
                         "note": "algorithmic bytes = SURVEY 8(d) streaming model; the kernel is LDS-resident, see DESIGN.md and profiles/ for measured HBM traffic"},
        }
        if world == 1 and not args.no_cpu_baseline:
            from oracle import oracle as O
            th = O.theta_vector("lco_iso")
            runs = [dict(mode=O.MODE_I, value=-1.0)]
            t1 = time.perf_counter(); O.run_batch("lco_iso", th, 1.0, runs, 50); per = (time.perf_counter() - t1) / 50
            n_cpu = max(100, int(args.cpu_seconds / per))
            t1 = time.perf_counter(); ok, tsum, _ = O.run_batch("lco_iso", th, 1.0, runs, n_cpu); dt = time.perf_counter() - t1
            assert ok == n_cpu and abs(tsum / n_cpu - 3600.0) < 1e-6
            out["cpu_baseline"] = {"value": n_cpu / dt, "unit": "trajectories/s", "cores": 1, "kind": "port",
                                   "sample": "%d of the same C2 trajectories (1C discharge, identical params), run back to back on one host core by the "
                                             "oracle (plain-C IDA/KLU-style port, oracle/ida_oracle.c); %.1f s; host has %d cores; reference publishes "
                                             "2.616 ms/trajectory on an unspecified laptop (examples/getting_started.ipynb:183-192)" % (n_cpu, dt, os.cpu_count())}
            # the same sample on every host core at once ("1 thread and all host cores"): one oracle context per thread, ctypes drops the GIL
            from concurrent.futures import ThreadPoolExecutor
            cores = usable_cores()
            chunk, deadline = max(10, int(0.25 / per)), time.perf_counter() + 0.6 * args.cpu_seconds

            def worker(_):
                done = 0
                while time.perf_counter() < deadline:            # time-bounded: chunks of ~0.25 s until the deadline
                    assert O.run_batch("lco_iso", th, 1.0, runs, chunk)[0] == chunk
                    done += chunk
                return done
            t1 = time.perf_counter()
            with ThreadPoolExecutor(cores) as ex:
                n_all = sum(ex.map(worker, range(cores)))
            dt_all = time.perf_counter() - t1
            out["cpu_baseline_all_cores"] = {"value": n_all / dt_all, "unit": "trajectories/s", "cores": cores, "kind": "port",
                                             "sample": "%d of the same trajectories on %d oracle threads (the cores this process may use: affinity / cgroup quota; "
                                                       "the machine reports %d logical cores); %.1f s; %.1fx the one-core rate" % (n_all, cores, os.cpu_count(), dt_all, n_all / dt_all / out["cpu_baseline"]["value"])}
            # the same launch through host pointers (PLH_HOST: H2D of Theta, D2H of every output array) -- the PCIe-inclusive rate, never `value`
            Th_host = pkg.theta_matrix(p, n_local)
            pkg.simulate_ensemble(p, Th_host, protocol, SOC=1.0, max_points=256)
            t1 = time.perf_counter()
            for _ in range(5):
                pkg.simulate_ensemble(p, Th_host, protocol, SOC=1.0, max_points=256)
            out["host_pointer_rate"] = {"value": 5 * n_local / (time.perf_counter() - t1), "unit": "trajectories/s",
                                        "note": "plh_integrate with PLH_HOST pointers: pageable-memory staging of Theta in and t/V/I/SOC[256]/Y/YP/run_info/counters out per call"}
            # measured device-to-device copy bandwidth of this box (read + write bytes), the second peak SURVEY 8(d) asks to quote
            a = torch.empty(1 << 28, dtype=torch.uint8, device=dev); b = torch.empty_like(a)
            b.copy_(a); torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                b.copy_(a)
            e1.record(); torch.cuda.synchronize()
            copy_gbps = 10 * 2 * a.numel() / (e0.elapsed_time(e1) * 1e-3) / 1e9
            out["roofline"]["measured_copy_peak"] = copy_gbps
            out["roofline"]["frac_of_measured_copy"] = out["roofline"]["achieved"] / copy_gbps
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

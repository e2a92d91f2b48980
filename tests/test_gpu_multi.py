"""The C ABI's multi-GPU entry (plh_comm_* / plh_ensemble_run: RCCL scatter -> integrate -> gather) on real devices.
One process per GPU; the 2-rank test needs two GPUs and skips on a one-GPU box (the driver's 8-GPU runs exercise it through bench.py)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ensemble_run_single_rank_through_rccl(hip_model, pkg):
    """a one-rank communicator created WITH an id initialises RCCL (ncclCommInitRank, ncclBroadcast on the stream): the collective code path on one GPU;
    results equal plh_integrate's, for both partitions"""
    from petlion_jl_amd import distributed as pd
    p = hip_model
    comm = pd.RcclComm(p._lib, 1, 0, pd.RcclComm.unique_id(p._lib))
    n = 777
    Th = pkg.configs.c4(p, n)["theta"]
    ref = pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0)
    for part in ("block", "cyclic"):
        info, cnt, Y, ms = pd.ensemble_run_capi(comm, p, Th, [{"I": -1.0}], 1.0, partition=part, want_Y=True)
        assert np.array_equal(Y, ref.Y) and np.array_equal(info["t_end"][:, 0], ref.run_info["t_end"][:, 0]) and np.array_equal(cnt["n_steps"], ref.counters["n_steps"])
        assert ms[0] > 0
    comm.close()


def test_ensemble_run_two_ranks_nccl(tmp_path):
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (one process per GPU)")
    idfile, outfile = str(tmp_path / "uid.bin"), str(tmp_path / "out.npz")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "mp_ensemble_worker.py"), str(r), "2", idfile, outfile], env=env) for r in range(2)]
    rcs = [q.wait(timeout=600) for q in procs]
    assert rcs == [0, 0], rcs
    d = np.load(outfile)
    for part in ("block", "cyclic"):
        assert np.array_equal(d[part + "_Y"], d["ref_Y"]) and np.array_equal(d[part + "_t_end"], d["ref_t_end"])      # bitwise: the shard a cell lands in does not matter
        assert (d[part + "_ms"] > 0).all() and len(d[part + "_ms"]) == 2


def test_bench_spawns_its_own_ranks_and_the_rank_logic_runs(tmp_path):
    """`python bench.py --gpus 2` WITHOUT a launcher spawns its two ranks itself (torch.distributed.run on 127.0.0.1) and rank 0 prints the one JSON line -- the path an
    8-GPU driver run takes if it starts bench.py plainly.  BENCH_BACKEND=gloo lets both ranks share the one GPU of this box (RCCL refuses two ranks per device); the
    RCCL path itself is test_ensemble_run_* above and the driver's multi-GPU run."""
    import json
    env = dict(os.environ, BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None); env.pop("RANK", None); env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--no-extras", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["cells_total"] == 2048 and len(d["config"]["rank_kernel_ms"]) == 2 and d["value"] > 0 and d["scaling"] == "weak"


def test_ensemble_run_per_cell_protocol_values(hip_model, pkg):
    """per-cell protocol arrays (value_cell / tf_cell) through plh_ensemble_run are indexed by the GLOBAL cell: with the cyclic partition of one rank the
    shard order equals the caller's, and the results must equal plh_integrate's on the same per-cell protocol"""
    from petlion_jl_amd import distributed as pd
    p = hip_model
    n = 96
    rates = -np.linspace(0.5, 3.0, n); tfs = np.linspace(300.0, 900.0, n)
    proto = [{"I": rates, "tf": tfs}]
    Th = pkg.configs.c4(p, n)["theta"]
    ref = pkg.simulate_ensemble(p, Th, proto, SOC=1.0)
    comm = pd.RcclComm(p._lib, 1, 0, pd.RcclComm.unique_id(p._lib))
    for part in ("block", "cyclic"):
        info, cnt, Y, ms = pd.ensemble_run_capi(comm, p, Th, proto, 1.0, partition=part, want_Y=True)
        assert np.array_equal(Y, ref.Y) and np.array_equal(info["t_end"][:, 0], ref.run_info["t_end"][:, 0]) and np.array_equal(info["I"][:, 0], rates)
    comm.close()


def _gloo_gpu_worker(rank, world, port, n_cells, out_dir):
    """one of `world` gloo ranks that all integrate their shard on the ONE GPU of the box"""
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import pkgload
    pkg = pkgload.load()
    from petlion_jl_amd import distributed as pd
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    p = pkg.petlion(pkg.LCO)
    Th = np.ascontiguousarray(pkg.configs.c4(p, n_cells)["theta"]) if rank == 0 else None
    for part in ("block", "cyclic"):
        summ, _ = pd.ensemble_run(p, Th, [{"I": -1.0}], 1.0, partition=part)
        if rank == 0:
            np.save(os.path.join(out_dir, "gathered_%s.npy" % part), summ)
    if rank == 0:
        np.save(os.path.join(out_dir, "serial.npy"), pd.summarize(pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0)))
    dist.destroy_process_group()


def test_eight_gloo_ranks_on_one_gpu_block_and_cyclic(tmp_path):
    """r06: EIGHT ranks (gloo: RCCL refuses two ranks per device) share the one GPU of the box and run the sharded C4 sweep through
    distributed.ensemble_run with both partitions of plh_ensemble_run (contiguous blocks; cell mod 8): scatter -> eight concurrent plh_integrate launches from eight
    processes -> gather in the caller's cell order, bit-equal to the serial launch of the same cells.  No hardware scaling curve is claimed from it."""
    import socket
    import torch
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    n = 8 * 160 + 5                                  # ragged shards
    mp.spawn(_gloo_gpu_worker, args=(8, port, n, str(tmp_path)), nprocs=8, join=True)
    serial = np.load(tmp_path / "serial.npy")
    assert serial.shape == (n, 8)
    for part in ("block", "cyclic"):
        g = np.load(tmp_path / ("gathered_%s.npy" % part))
        assert np.array_equal(g, serial), part

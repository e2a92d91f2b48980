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

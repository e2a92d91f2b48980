"""Multi-process ensemble sharding on CPU (gloo, world_size 2 and 3): scatter of parameter rows -> local integration ->
gather of per-cell summaries must reproduce the single-process result cell by cell.  The local integration is injected
(oracle on CPU) because this container has no GPU; on the GPU box the same code path runs with the HIP integrate and RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_cells, out_dir, partition="block"):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import pkgload
    pkg = pkgload.load()
    from petlion_jl_amd import distributed as pd
    from oracle import oracle as O
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)

    def local_integrate(Th, soc):
        rows = []
        for th in Th:
            ro = O.simulate("lco_iso", th, soc, [dict(mode=O.MODE_I, value=-1.0, tf=600.0)])
            r = ro["runs"][0]
            rows.append([r["flag"], r["iterations"], r["t_end"], r["V"], r["I"], r["SOC"], ro["counters"]["n_steps"], ro["counters"]["n_newton"]])
        return np.array(rows)

    Theta = None
    if rank == 0:
        th0 = O.theta_vector("lco_iso")
        Theta = np.tile(th0, (n_cells, 1))
        Theta[:, O.meta("lco_iso")["theta_keys"].index("D_sp")] *= np.linspace(0.5, 2.0, n_cells)
    summ, _ = pd.ensemble_run(None, Theta, [{"I": -1.0}], 1.0, local_integrate=local_integrate, partition=partition)
    if rank == 0:
        np.save(os.path.join(out_dir, "gathered_w%d.npy" % world), summ)
        np.save(os.path.join(out_dir, "serial.npy"), local_integrate(Theta, 1.0))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_cells,partition", [(2, 7, "block"), (3, 5, "block"), (3, 7, "cyclic")])
def test_scatter_integrate_gather_matches_serial(tmp_path, world, n_cells, partition):
    import torch.multiprocessing as mp
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_cells, str(tmp_path), partition), nprocs=world, join=True)
    g = np.load(tmp_path / ("gathered_w%d.npy" % world))
    s = np.load(tmp_path / "serial.npy")
    assert g.shape == (n_cells, 8)
    assert np.array_equal(g, s)            # same cells, same order, bitwise the same summaries


def test_shard_bounds_are_contiguous_and_balanced(pkg):
    from petlion_jl_amd import distributed as pd
    for n, w in ((65536, 8), (1024, 8), (7, 2), (5, 3), (3, 8)):
        off = pd.shard_bounds(n, w)
        sizes = np.diff(off)
        assert off[0] == 0 and off[-1] == n and sizes.max() - sizes.min() <= 1 and (sizes >= 0).all()

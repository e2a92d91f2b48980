"""Ensemble sharding over the GPUs of one node: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

Cells are independent for the whole trajectory, so the data path has NO collective: the only exchanges are
the scatter of parameter rows before and the gather of per-cell summaries after the integration -- exactly the
north-star's "RCCL only for the ensemble scatter/gather".  Payloads are tiny (65 536 cells x 35 parameters x 8 B = 18 MB;
summaries ~100 B/cell), so the per-link xGMI bandwidth is irrelevant; what matters is that the partition is static,
balanced and reproducible: contiguous blocks, rank r owns cells [r*n/G, (r+1)*n/G).

A single trajectory never spans GPUs.
"""
from __future__ import annotations

import numpy as np

SUMMARY_FIELDS = ("flag", "iterations", "t_end", "V", "I", "SOC", "n_steps", "n_newton")


def shard_bounds(n_cells, world_size):
    """start offsets of the contiguous, balanced partition (len = world_size + 1)."""
    base, rem = divmod(n_cells, world_size)
    sizes = [base + (1 if r < rem else 0) for r in range(world_size)]
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def summarize(ens):
    """[n_cells, len(SUMMARY_FIELDS)] float64 summary of the LAST run of every cell."""
    ri = ens.run_info[:, -1]
    c = ens.counters
    return np.stack([ri["flag"].astype(np.float64), ri["iterations"].astype(np.float64), ri["t_end"], ri["V"], ri["I"], ri["SOC"],
                     c["n_steps"].astype(np.float64), c["n_newton"].astype(np.float64)], axis=1)


def shard_cells(n_cells, world_size, rank, partition="block"):
    """global cell indices of rank's shard: contiguous balanced blocks (plh_ensemble_run's PLH_PART_BLOCK) or cell mod world_size == rank (PLH_PART_CYCLIC: a sweep
    whose cost varies smoothly with the cell index spreads evenly)"""
    if partition == "cyclic":
        return np.arange(rank, n_cells, world_size, dtype=np.int64)
    off = shard_bounds(n_cells, world_size)
    return np.arange(off[rank], off[rank + 1], dtype=np.int64)


def ensemble_run(p, Theta, protocol, SOC=1.0, *, group=None, device=None, local_integrate=None, partition="block"):
    """Scatter Theta ([n_cells, n_theta], significant on rank 0 only) over the ranks of `group` (partition: "block" or "cyclic", see shard_cells), integrate each
    shard on the rank's GPU, gather the per-cell summaries back to rank 0 IN THE CALLER'S CELL ORDER.

    Returns (summary [n_cells, 8] on rank 0 / None elsewhere, local EnsembleSolution-or-summary).
    `local_integrate(Theta_shard: np.ndarray, SOC) -> np.ndarray [m, 8]` replaces the GPU integration (tests use it to
    check the sharding logic on CPU ranks with the gloo backend).
    """
    import torch
    import torch.distributed as dist

    rank = dist.get_rank(group)
    world = dist.get_world_size(group)
    backend = dist.get_backend(group)
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    # 1. broadcast the ensemble shape, 2. scatter padded equal-size shards
    meta = torch.zeros(2, dtype=torch.int64, device=device)
    if rank == 0:
        Theta = np.ascontiguousarray(Theta, dtype=np.float64)
        meta[0], meta[1] = Theta.shape
    dist.broadcast(meta, src=0, group=group)
    n, P = int(meta[0]), int(meta[1])
    cells = [shard_cells(n, world, r, partition) for r in range(world)]
    m_max = max(len(c) for c in cells)
    mine = torch.zeros(m_max, P, dtype=torch.float64, device=device)
    if rank == 0:
        full = torch.from_numpy(Theta).to(device)
        chunks = []
        for r in range(world):
            c = torch.zeros(m_max, P, dtype=torch.float64, device=device)
            c[: len(cells[r])] = full[torch.from_numpy(cells[r]).to(device)]
            chunks.append(c)
        dist.scatter(mine, chunks, src=0, group=group)
    else:
        dist.scatter(mine, None, src=0, group=group)
    m = len(cells[rank])
    local = None
    if m > 0:
        if local_integrate is not None:
            summ = np.asarray(local_integrate(mine[:m].cpu().numpy(), SOC), dtype=np.float64)
        else:
            from .api import simulate_ensemble
            if mine.device.type != "cuda":                  # a CPU group (gloo) carries host tensors: move the shard to this rank's GPU (there is no CPU integrator)
                mine = mine.cuda()
            local = simulate_ensemble(p, mine[:m].contiguous(), protocol, SOC=SOC, device=True)
            summ = summarize(local)
    else:
        summ = np.zeros((0, len(SUMMARY_FIELDS)))
    # 3. gather summaries (padded to m_max rows)
    pad = torch.zeros(m_max, len(SUMMARY_FIELDS), dtype=torch.float64, device=device)
    pad[:m] = torch.from_numpy(summ).to(device)
    if rank == 0:
        parts = [torch.zeros_like(pad) for _ in range(world)]
        dist.gather(pad, parts, dst=0, group=group)
        out = np.zeros((n, len(SUMMARY_FIELDS)))
        for r in range(world):
            out[cells[r]] = parts[r][: len(cells[r])].cpu().numpy()
        return out, (local if local is not None else summ)
    dist.gather(pad, None, dst=0, group=group)
    return None, (local if local is not None else summ)


# ---- the C-ABI multi-GPU path (plh_comm_* / plh_ensemble_run: RCCL inside the library, what a Julia host binds) ----
class RcclComm:
    """one rank of a plh_comm communicator.  `unique_id` (128 bytes) comes from RcclComm.unique_id() on rank 0 and reaches the other ranks by the host's own
    means (here: a torch.distributed broadcast in bench.py; MPI / Distributed.jl in a Julia host)."""

    def __init__(self, lib, n_ranks, rank, unique_id=None, device=-1):
        import ctypes as C
        self.lib = lib
        h = C.c_void_p()
        from ._capi import check
        check(lib, lib.plh_comm_create(int(n_ranks), int(rank), unique_id, int(device), C.byref(h)), "plh_comm_create")
        self.h = h
        self.n_ranks, self.rank = int(n_ranks), int(rank)

    @staticmethod
    def unique_id(lib):
        import ctypes as C
        from ._capi import check
        buf = C.create_string_buffer(128)
        check(lib, lib.plh_comm_unique_id(buf), "plh_comm_unique_id")
        return buf.raw

    def close(self):
        if self.h:
            self.lib.plh_comm_destroy(self.h)
            self.h = None


def ensemble_run_capi(comm, p, Theta, protocol, SOC=1.0, *, n_cells=None, partition="block", opts=None, want_Y=False):
    """plh_ensemble_run: collective over `comm`; Theta ([n_cells, n_theta] numpy) and SOC are significant on rank 0 only (other ranks pass None and n_cells).
    Returns on rank 0 (run_info [n_cells, n_runs], counters [n_cells], Y_final or None, rank_ms [n_ranks]); None elsewhere."""
    import ctypes as C
    from . import _capi as cap
    from .api import _opts_struct, make_protocol
    root = comm.rank == 0
    n = int(Theta.shape[0] if root else n_cells)
    runs, _ = make_protocol(p, protocol, n)
    arr = (cap.Run * len(runs))(*runs)
    o = _opts_struct(opts or p.opts, p)
    part = {"block": cap.PART_BLOCK, "cyclic": cap.PART_CYCLIC}[partition]
    if root:
        Theta = np.ascontiguousarray(Theta, dtype=np.float64)
        soc = np.full(n, float(SOC)) if np.isscalar(SOC) else np.ascontiguousarray(SOC, dtype=np.float64)
        info = np.zeros((n, len(runs)), cap.RUN_INFO_DTYPE); cnt = np.zeros(n, cap.COUNTERS_DTYPE)
        Y = np.zeros((n, p.N.tot)) if want_Y else None
        ms = np.zeros(comm.n_ranks)
        cap.check(p._lib, p._lib.plh_ensemble_run(comm.h, p._h, n, cap.ptr(Theta), cap.ptr(soc), len(runs), arr, C.byref(o), part, cap.ptr(info), cap.ptr(cnt),
                                                  cap.ptr(Y), cap.ptr(ms)), "plh_ensemble_run")
        return info, cnt, Y, ms
    cap.check(p._lib, p._lib.plh_ensemble_run(comm.h, p._h, n, None, None, len(runs), arr, C.byref(o), part, None, None, None, None), "plh_ensemble_run")
    return None

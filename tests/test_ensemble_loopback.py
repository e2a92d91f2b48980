"""plh_ensemble_run with 2 and 3 RANKS, executed: the multi-rank control flow of the C ABI's ensemble entry -- shard offsets, ragged and cyclic
partitions, per-rank shards of the per-cell protocol arrays, the status agreements between the phases, the gather re-ordering, a failure on one rank at every phase --
runs here as 2 / 3 PROCESSES of the test-only wave-emulator build, whose transport is a file-backed loopback behind the same five call sites the product build serves
with RCCL (x_send / x_recv / x_bcast / x_allmin in csrc/petlion_hip.hip).  The RCCL transport itself: tests/test_gpu_multi.py (one rank on the one-GPU box, two ranks when
a second GPU exists) and the driver's multi-GPU bench."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "loopback_worker.py")


def run_ranks(tmp_path, world, n, extra=(), env_extra=None, timeout=900):
    idfile, outfile = str(tmp_path / "uid.bin"), str(tmp_path / "out")
    env = dict(os.environ, TMPDIR=str(tmp_path), **(env_extra or {}))
    procs = [subprocess.Popen([sys.executable, WORKER, str(r), str(world), idfile, outfile, str(n)] + list(extra), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [q.communicate(timeout=timeout)[0] for q in procs]
    assert [q.returncode for q in procs] == [0] * world, outs
    return [dict(np.load(outfile + ".%d.npz" % r, allow_pickle=True)) for r in range(world)]


@pytest.fixture(scope="module", autouse=True)
def emu_built():
    sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emu"))
    import build_emu
    build_emu.build()            # (once, before the ranks start: they would otherwise all try to build it)


@pytest.mark.parametrize("world,n", [(2, 5), (3, 7), (3, 2)])
def test_ranks_agree_with_the_serial_run(tmp_path, world, n):
    """ragged shards (5 cells on 2 ranks, 7 on 3) and a rank with NO cell (2 cells on 3 ranks), both partitions: the gathered results equal one plh_integrate call over the
    whole ensemble bit for bit, in the caller's cell order"""
    d = run_ranks(tmp_path, world, n)
    assert all(r[part + "_rc"] == 0 for r in d for part in ("block", "cyclic"))
    r0 = d[0]
    for part in ("block", "cyclic"):
        assert np.array_equal(r0[part + "_Y"], r0["ref_Y"]) and np.array_equal(r0[part + "_t_end"], r0["ref_t_end"]) and np.array_equal(r0[part + "_steps"], r0["ref_steps"])
        assert len(r0[part + "_ms"]) == world


def test_per_cell_protocol_arrays_are_sharded_by_global_cell(tmp_path):
    """value_cell / tf_cell: n_cells_total entries by GLOBAL cell on every rank; each rank integrates with its own shard (block: contiguous, cyclic: cell mod G)"""
    d = run_ranks(tmp_path, 3, 7, extra=["per_cell"])
    r0 = d[0]
    for part in ("block", "cyclic"):
        assert np.array_equal(r0[part + "_I"], -np.linspace(0.5, 2.0, 7)) and np.array_equal(r0[part + "_t_end"], r0["ref_t_end"]) and np.array_equal(r0[part + "_Y"], r0["ref_Y"])
    assert np.array_equal(r0["ref_t_end"], np.linspace(20.0, 40.0, 7))


@pytest.mark.parametrize("phase", [0, 1, 2, 3])
def test_a_failure_on_one_rank_returns_on_all_ranks(tmp_path, phase):
    """a LOCAL failure on rank 1 (arguments, shape, scatter preparation, its plh_integrate) is agreed on before the next collective: every rank returns non-zero, none is left
    waiting (the loopback's receive would time out after 120 s: the test would fail on its 60 s limit)"""
    d = run_ranks(tmp_path, 3, 5, env_extra={"PLH_TEST_FAIL": "1:%d" % phase}, timeout=60)
    for r, res in enumerate(d):
        assert res["block_rc"] == -1 and res["cyclic_rc"] == -1, (r, res)
        msg = str(res["block_err"])
        assert ("injected failure" in msg) if r == 1 else ("another rank failed" in msg), (r, msg)


def test_a_failure_inside_a_collective_phase_times_out_instead_of_hanging(tmp_path):
    """a failure INSIDE the gather cannot be agreed on any more: rank 1 aborts its communicator and returns; the others' receives give up after the transport's timeout (3 s
    here) -- nobody hangs, every rank reports a transport failure, and the aborted communicator refuses the second call"""
    d = run_ranks(tmp_path, 2, 4, env_extra={"PLH_TEST_FAIL": "1:4", "PLH_LOOPBACK_TIMEOUT_S": "3"}, timeout=120)
    assert all(res["block_rc"] == -1 and res["cyclic_rc"] == -1 for res in d)
    assert "injected" in str(d[1]["block_err"]) and "transport failure" in str(d[0]["block_err"])
    assert all("aborted" in str(res["cyclic_err"]) for res in d)

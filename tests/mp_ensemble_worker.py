"""worker of test_gpu_multi.py: one rank of a plh_comm communicator (one process per GPU).  usage: mp_ensemble_worker.py RANK WORLD IDFILE OUTFILE"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, idfile, outfile = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
import pkgload
pkg = pkgload.load()
from petlion_jl_amd import distributed as pd

p = pkg.petlion(pkg.LCO, device=rank)
if rank == 0:
    uid = pd.RcclComm.unique_id(p._lib)
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120, "rank 0 never published the communicator id"
        time.sleep(0.05)
    uid = open(idfile, "rb").read()
comm = pd.RcclComm(p._lib, world, rank, uid, device=rank)
n = 1000 + 3                                     # not a multiple of the world size: ragged shards
Th = pkg.configs.c4(p, n)["theta"] if rank == 0 else None
out = {}
for part in ("block", "cyclic"):
    res = pd.ensemble_run_capi(comm, p, Th, [{"I": -1.0}], 1.0, n_cells=n, partition=part, want_Y=True)
    if rank == 0:
        info, cnt, Y, ms = res
        out[part] = dict(t_end=info["t_end"][:, 0], flag=info["flag"][:, 0], Y=Y, steps=cnt["n_steps"], ms=ms)
if rank == 0:
    ref = pkg.simulate_ensemble(p, Th, [{"I": -1.0}], SOC=1.0)            # the same ensemble on rank 0's GPU alone
    np.savez(outfile, ref_Y=ref.Y, ref_t_end=ref.run_info["t_end"][:, 0], **{"%s_%s" % (k, f): v for k, d in out.items() for f, v in d.items()})
comm.close()

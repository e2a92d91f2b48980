"""worker of test_ensemble_loopback.py: one rank of a plh_comm communicator on the TEST-ONLY wave-emulator build, whose transport is the file-backed loopback of
csrc/petlion_hip.hip (no GPU, no RCCL).  usage: loopback_worker.py RANK WORLD IDFILE OUTFILE N_CELLS [per_cell]
Every rank writes OUTFILE.<rank>.npz with its return codes; rank 0 adds the gathered results and the serial reference."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests", "wave_emu")]
rank, world, idfile, outfile, n = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4], int(sys.argv[5])
per_cell = len(sys.argv) > 6 and sys.argv[6] == "per_cell"
import build_emu
import pkgload
pkg = pkgload.load()
from petlion_jl_amd import distributed as pd
from petlion_jl_amd._capi import PetlionHipError

p = pkg.petlion(pkg.LCO, _lib_path=build_emu.build())
if rank == 0:
    uid = pd.RcclComm.unique_id(p._lib)
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        assert time.time() - t0 < 120, "rank 0 never published the communicator id"
        time.sleep(0.02)
    uid = open(idfile, "rb").read()
comm = pd.RcclComm(p._lib, world, rank, uid)
rng = np.random.default_rng(5)
Th_all = pkg.theta_matrix(p, n, {"D_sp": p.θ["D_sp"] * 2.0 ** (2 * rng.random(n) - 1), "k_p": p.θ["k_p"] * 2.0 ** (2 * rng.random(n) - 1)})
Th = Th_all if rank == 0 else None
# a short protocol (the emulator runs one lane at a time); per_cell: a C-rate and a run length per cell -- value_cell / tf_cell are indexed by the GLOBAL cell
proto = [{"I": -np.linspace(0.5, 2.0, n), "tf": np.linspace(20.0, 40.0, n)}] if per_cell else [{"I": -1.0, "tf": 30.0}]
out = {}
for part in ("block", "cyclic"):
    try:
        res = pd.ensemble_run_capi(comm, p, Th, proto, 1.0, n_cells=n, partition=part, want_Y=True)
        out[part + "_rc"] = 0
    except PetlionHipError as e:
        res = None
        out[part + "_rc"] = -1
        out[part + "_err"] = str(e)
    if rank == 0 and res is not None:
        info, cnt, Y, ms = res
        out.update({part + "_t_end": info["t_end"][:, 0], part + "_flag": info["flag"][:, 0], part + "_I": info["I"][:, 0], part + "_Y": Y, part + "_steps": cnt["n_steps"], part + "_ms": ms})
if rank == 0:
    ref = pkg.simulate_ensemble(p, Th, proto, SOC=1.0)            # the same ensemble in one plh_integrate call
    out.update(ref_Y=ref.Y, ref_t_end=ref.run_info["t_end"][:, 0], ref_steps=ref.counters["n_steps"], ref_I=ref.run_info["I"][:, 0])
np.savez(outfile + ".%d.npz" % rank, **out)
comm.close()

#!/usr/bin/env python3
"""Writes petlion.jl_amd/selftest_golden.json: the KNOWN ANSWER of the kernel self-test (api.known_answer_check) for every built-in variant and every discretisation the test
suite builds -- run on the GPU with the binary that the committed GPU test run validates (profiles/validated_build.json names it):

    python tools/make_selftest_golden.py            (on the GPU box; the file is committed)

Per model: the digest of api.known_answer_protocol (2 cells, 3 runs -- 1C discharge 100 s, V hold 50 s, rest 50 s -- at reltol 1e-8 / abstol 1e-10): run-end t, V, I, SOC and
per state section (max |Y|, sum Y).  A later build (another compiler, other flags, a grid library compiled on the user's machine) must reproduce it to 2e-6."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402
import pkgload  # noqa: E402

pkg = pkgload.load()
g.build_hip()
MODELS = [("LCO", {}), ("NMC", {}), ("LCO", dict(aging="SEI")), ("NMC", dict(aging="SEI")), ("LCO", dict(temperature=True)),
          ("LCO", dict(precision="mixed")), ("NMC", dict(aging="SEI", precision="mixed")), ("LCO", dict(temperature=True, precision="mixed")),
          ("LCO", dict(solid_diffusion="quadratic")), ("LCO", dict(solid_diffusion="polynomial")), ("LCO", dict(thermodynamic_factor="nonlinear")),
          ("LCO", dict(rxn_p="MHC", rxn_n="MHC")), ("NMC_LGM50", dict(temperature=False)), ("NMC_LGM50", {}),
          ("LCO", dict(precision="f64_reforder")), ("LCO", dict(temperature=True, precision="f64_reforder"))]
VAR = {0: ("LCO", {}), 3: ("NMC", dict(aging="SEI")), 4: ("LCO", dict(temperature=True)), 8: ("LCO", dict(solid_diffusion="quadratic"))}
for grid, vids in g.TEST_GRIDS + g.TEST_GRIDS_SECOND + [((3, 2, 2, 10, 10, 10), [0])]:
    for v in vids:
        chem, kw = VAR[v]
        kw = dict(kw, N_p=grid[0], N_s=grid[1], N_n=grid[2], N_r_p=grid[3], N_a=grid[4], N_z=grid[5], N_r_n=grid[6] if len(grid) == 7 else grid[3])
        MODELS.append((chem, kw))
out = {}
for chem, kw in MODELS:
    p = pkg.petlion(getattr(pkg, chem), **kw)
    d = pkg.api.known_answer_digest(p)
    assert all(f >= 0 for row in d["flags"] for f in row), (p.variant, d["flags"])
    out[pkg.api._golden_key(p)] = d
    print(pkg.api._golden_key(p), d["V"], flush=True)
json.dump({"build_info": pkg.api.build_info(), "tolerance": pkg.api.KA_TOL, "protocol": "api.known_answer_protocol", "digests": out},
          open(os.path.join(ROOT, "petlion.jl_amd", "selftest_golden.json"), "w"), indent=1, ensure_ascii=False)
print("wrote petlion.jl_amd/selftest_golden.json: %d models" % len(out))

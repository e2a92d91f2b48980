#!/usr/bin/env python3
"""After the final GPU call (tools/gpu/r06_final.sh -> gpurun_out/<dir>): copy what it produced into the tracked places and write profiles/validated_build.json --
the identity of the binary the run validated, what the run consisted of, and WHAT THE COMPILER BUILT (tools/kernel_resources.py record of that binary).
   python tools/finalize_validation.py gpurun_out/r06v r06"""
import json, os, re, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, tag = sys.argv[1], sys.argv[2]
info = open(os.path.join(d, "build_info.txt")).read().strip().splitlines()[-1]
res = json.load(open(os.path.join(d, "libpetlion_hip.so.resources.json")))
assert res["build_info"] == info, (res["build_info"], info)
for f in os.listdir(os.path.join(d, "profiles")):
    shutil.copy(os.path.join(d, "profiles", f), os.path.join(ROOT, "profiles", f))
shutil.copy(os.path.join(d, "selftest_golden.json"), os.path.join(ROOT, "petlion.jl_amd", "selftest_golden.json"))
log = open(os.path.join(d, "pytest.log")).read()
keep = [l for l in log.splitlines() if "is not a recognized feature" not in l and "amdgpu" not in l.lower()[:20]]
open(os.path.join(ROOT, "profiles", "%s_gpu_pytest.log" % tag), "w").write("\n".join(l[:1500] for l in keep) + "\n")
summary = [l for l in log.splitlines() if re.search(r"\d+ passed", l)][-1].strip()
subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "bench_tables.py"), d, "%s_bench_lines.json" % tag], stdout=open(os.path.join(d, "tables.md"), "w"))
L = json.load(open(os.path.join(ROOT, "profiles", "%s_bench_lines.json" % tag)))
golden = json.load(open(os.path.join(ROOT, "petlion.jl_amd", "selftest_golden.json")))
smoke = open(os.path.join(d, "smoke.txt")).read().strip().splitlines()[-1]
json.dump({"build_info": info,
           "validated_by": "ONE gpurun call on one MI355X (tools/gpu/%s_final.sh, %s): tools/make_selftest_golden.py (petlion.jl_amd/selftest_golden.json: %d models), python -m pytest tests -m gpu -s "
                           "(profiles/%s_gpu_pytest.log: %s), smoke() (%s), tools/prof.sh C2..C5 (profiles/%s_c*_rocprofv3_summary.md, %s_c*_pmc.json, %s_c*_traffic.json), bench.py --config C2..C5 "
                           "(profiles/%s_bench_lines.json: %s trajectories/s) -- all on this binary"
                           % (tag, d, len(golden["digests"]), tag, summary, smoke[:120], tag, tag, tag, tag, ", ".join("%s %.0f" % (c, L[c]["value"]) for c in L)),
           "kernel_resources": res,
           "note": "petlion() runs the kernel self-test once per variant when the loaded library's plh_build_info() differs from build_info (DESIGN.md 5a); kernel_resources: registers, "
                   "spills, scratch bytes per lane (private_segment_fixed_size) and LDS per cell (group_segment_fixed_size) of every instantiation, from the code objects "
                   "(tools/kernel_resources.py); tests/test_build_records.py asserts on it"},
          open(os.path.join(ROOT, "profiles", "validated_build.json"), "w"), indent=1, sort_keys=True)
print("profiles/validated_build.json:", info, "|", summary)

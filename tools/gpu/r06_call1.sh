#!/bin/bash
# r06 call 1: phase profile of the r05 binary's isothermal kernel (coarse + three detail levels) and same-box baseline kernel times
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06a; mkdir -p $O
export TMPDIR=/tmp
for d in "" --detail --detail2 --detail3; do
  timeout 300 python tools/phase_profile.py 1024 iso $d > $O/phase_iso$d.txt 2>&1
done
cat $O/phase_iso*.txt | grep -v "amdgpu\|RCCL\|warning"
timeout 300 python tools/perf_configs.py c2 c4 --reps 5 > $O/perf_base.txt 2>&1; cat $O/perf_base.txt | grep -v warning

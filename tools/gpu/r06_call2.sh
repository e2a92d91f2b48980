#!/bin/bash
# r06 call 2: same-box A/B of the diet switches (variant 0), parity subset on the new build, phase profile of the new build
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06b; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/experiments/ab.py run r05,r06_all,r06_noflat,r06_novpad,r06_ieeediv --reps 5 > $O/ab.txt 2>&1
grep "===\|^C[24]" $O/ab.txt
PETLION_HIP_LIB=petlion.jl_amd/_exp/libplh_r06_all.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "c2_1024 or evaluators or c4" -p no:cacheprovider > $O/pytest_subset.txt 2>&1; tail -5 $O/pytest_subset.txt
for d in "" --detail --detail2 --detail3; do
  timeout 300 python tools/phase_profile.py 1024 iso $d > $O/phase_iso$d.txt 2>&1
done
cat $O/phase_iso*.txt | grep -v "amdgpu\|RCCL\|warning"

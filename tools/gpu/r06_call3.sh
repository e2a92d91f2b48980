#!/bin/bash
# r06 call 3: A/B of the second batch (set_coeffs / complete_step scalars, node-pass tables, sweep masks; opaque lane ids), parity subset, phase profile
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06c; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/experiments/ab.py run r05,r06_all,r06c_all,r06c_lane1 --reps 5 > $O/ab.txt 2>&1
grep "===\|^C[24]" $O/ab.txt
for L in r06c_all r06c_lane1; do
PETLION_HIP_LIB=petlion.jl_amd/_exp/libplh_$L.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "c2_1024 or evaluators or c4" -p no:cacheprovider > $O/pytest_subset_$L.txt 2>&1; tail -3 $O/pytest_subset_$L.txt
done
for d in "" --detail2 --detail3; do
  timeout 300 python tools/phase_profile.py 1024 iso $d > $O/phase_iso$d.txt 2>&1
done
cat $O/phase_iso*.txt | grep -v "amdgpu\|RCCL\|warning"

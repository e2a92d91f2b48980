#!/bin/bash
# r06 call 5: A/B of the row-broadcast particle phases (variant 0), parity subset
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06e; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/experiments/ab.py run r06c_all,r06d_all,r06d_nocsdpp --reps 5 > $O/ab.txt 2>&1
grep "===\|^C[24]" $O/ab.txt
PETLION_HIP_LIB=petlion.jl_amd/_exp/libplh_r06d_all.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "c2_1024 or evaluators or c4" -p no:cacheprovider > $O/pytest_subset.txt 2>&1; tail -5 $O/pytest_subset.txt

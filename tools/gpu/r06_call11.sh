#!/bin/bash
# r06 call 11: thermal kernel -- one level of recursive doubling in the 4x4 sweeps of a solve, A/B on variant 4 + parity subset
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06r}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/experiments/ab.py run th_r06g,th_r06g_nostride2 --reps 4 > $O/ab.txt 2>&1
grep "===\|^C[234]" $O/ab.txt
PETLION_HIP_LIB=petlion.jl_amd/_exp/libplh_th_r06g.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "c3_thermal or evaluators_thermal or thermal" -p no:cacheprovider > $O/pytest_subset.txt 2>&1; tail -5 $O/pytest_subset.txt

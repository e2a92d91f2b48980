#!/bin/bash
# r05 GPU call 2: the OCC2 layout (history orders >= 2 in global memory, six 301-state cells per CU, two waves per SIMD) against the production kernels; C3 two-sample table;
# the new GPU tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05b; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2; do
  PETLION_HIP_LIB=$PWD/petlion.jl_amd/_exp/libplh_base03.so timeout 600 python tools/perf_configs.py c2 c4 c5 --reps 3 >> $O/perf_base.txt 2>&1
  PETLION_HIP_LIB=$PWD/petlion.jl_amd/_exp/libplh_occ2.so timeout 600 python tools/perf_configs.py c2 c4 c5 --reps 3 >> $O/perf_occ2.txt 2>&1
done
PETLION_HIP_LIB=$PWD/petlion.jl_amd/_exp/libplh_occ2.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "c2_1024 or evaluators_residual or test_consistent or c4_parameter_sweep_subset or cc_cv_protocol or gitt_like" -p no:cacheprovider > $O/pytest_occ2.txt 2>&1
for f in perf_base perf_occ2 pytest_occ2; do echo "=== $f"; tail -14 $O/$f.txt | cut -c1-400; done

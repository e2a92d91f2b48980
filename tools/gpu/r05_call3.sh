#!/bin/bash
# r05 GPU call 3 (production library of the round's sources): occ4 perf, C3 two-sample table, the new GPU tests, sensitivity tests
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
export TMPDIR=/tmp
for r in 1 2; do for L in base03 occ4; do
  echo "== $L" >> $O/perf_occ4.txt
  PETLION_HIP_LIB=$PWD/petlion.jl_amd/_exp/libplh_$L.so timeout 600 python tools/perf_configs.py c2 c4 --reps 3 2>&1 | grep "^C" >> $O/perf_occ4.txt
done; done
PETLION_HIP_LIB=$PWD/petlion.jl_amd/_exp/libplh_occ4.so timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "c2_1024 or evaluators_residual or test_consistent or c4_parameter_sweep_subset or cc_cv_protocol or gitt_like" -p no:cacheprovider > $O/pytest_occ4.txt 2>&1
export PETLION_HIP_LIB=$PWD/petlion.jl_amd/libpetlion_hip.so      # (the library as built here: no rebuild on the GPU box)
timeout 600 python tools/perf_configs.py c2 c3 c4 c5 --reps 3 > $O/perf.txt 2>&1
timeout 900 python tools/experiments/c3_two_sample.py --out $O/c3_two_sample.json > $O/c3_two_sample.txt 2>&1
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_sensitivities.py tests/test_gpu_ensemble.py -q -m gpu -k "quiet or reference_order or stop_function or hold_leg or sens or every_cell_c3" -p no:cacheprovider > $O/pytest_new.txt 2>&1
for f in perf_occ4 pytest_occ4 perf c3_two_sample pytest_new; do echo "=== $f"; tail -16 $O/$f.txt | cut -c1-420; done

#!/bin/bash
# r05 GPU call 2b: where the single-wave penalty of the OCC2 layout comes from -- the same layout at 512 registers, the default layout at 256 registers
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05b; mkdir -p $O
for r in 1 2; do for L in base03 occ2 occ2_w1 base_r256; do
  echo "== $L" >> $O/perf_diag.txt
  PETLION_HIP_LIB=$PWD/petlion.jl_amd/_exp/libplh_$L.so timeout 600 python tools/perf_configs.py c2 c4 --reps 3 2>&1 | grep "^C" >> $O/perf_diag.txt
done; done
cat $O/perf_diag.txt | cut -c1-120

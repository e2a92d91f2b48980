#!/bin/bash
# Build experiments for the thermal sibling library (thermal variant only): one library per flag set, then C3 on the GPU box.
# usage: tools/flag_search_thermal.sh build | run
R=$(cd "$(dirname "$0")/.." && pwd)
declare -A F
F[O2]="-O2"
F[O1]="-O1"
F[O3]="-O3"
F[O2_u75]="-O2 -mllvm -unroll-threshold=75"
F[O2_u300]="-O2 -mllvm -unroll-threshold=300"
F[O2_noslp]="-O2 -fno-slp-vectorize"
F[O2_minreg]="-O2 -mllvm -amdgpu-sched-strategy=gcn-iterative-minreg"
F[O2_memclause]="-O2 -mllvm -amdgpu-sched-strategy=max-memory-clause"
F[O2_maxilp]="-O2 -mllvm -amdgpu-sched-strategy=max-ilp"
F[O2_nolicm]="-O2 -mllvm -disable-licm-promotion"
F[O2_nopre]="-O2 -mllvm -enable-pre=false"
F[O2_nogvnhoist]="-O2 -mllvm -enable-gvn-sink=false"
if [ "$1" = build ]; then
  mkdir -p $R/petlion.jl_amd/flagsearch
  for k in "${!F[@]}"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-function-calls=false -DPL_ONLY_THERMAL ${F[$k]} \
        $R/petlion.jl_amd/csrc/petlion_hip.hip -o $R/petlion.jl_amd/flagsearch/thermal_$k.so 2>&1 | grep -i " error\|unknown" ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait; ls $R/petlion.jl_amd/flagsearch
else
  for f in $R/petlion.jl_amd/flagsearch/thermal_*.so; do
    echo "$(basename $f .so): $(PETLION_HIP_LIB=$f python $R/tools/perf_configs.py c3 2>&1 | tail -1 | cut -c1-80)"
  done
fi

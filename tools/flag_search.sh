#!/bin/bash
# Build experiments: the LCO isothermal variant only, one library per compiler-flag set (here), then bench each on the GPU box.
# usage: tools/flag_search.sh build | run
R=$(cd "$(dirname "$0")/.." && pwd)
declare -A F
F[base]=""
F[noslp]="-fno-slp-vectorize"
F[ifcvt]="-mllvm -amdgpu-early-ifcvt=1"
F[nolsv]="-mllvm -amdgpu-load-store-vectorizer=0"
F[nopostsched]="-mllvm -enable-post-misched=0"
F[nodpp]="-mllvm -amdgpu-dpp-combine=false"
F[bias]="-mllvm -amdgpu-schedule-metric-bias=100"
F[noslp_ifcvt]="-fno-slp-vectorize -mllvm -amdgpu-early-ifcvt=1"
if [ "$1" = build ]; then
  mkdir -p $R/petlion.jl_amd/flagsearch
  for k in "${!F[@]}"; do
    ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-function-calls=false -DPL_ONLY_LCO_ISO ${F[$k]} \
        $R/petlion.jl_amd/csrc/petlion_hip.hip -o $R/petlion.jl_amd/flagsearch/lib_$k.so 2>&1 | grep -i " error\|unknown" ) &
    while [ $(jobs -r | wc -l) -ge 4 ]; do sleep 1; done
  done
  wait; ls $R/petlion.jl_amd/flagsearch
else
  for f in $R/petlion.jl_amd/flagsearch/lib_*.so; do
    k=$(basename $f .so); 
    for i in 1 2; do PETLION_HIP_LIB=$f python $R/bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$k', round(d['value']), round(d['roofline']['kernel_ms_avg'],4))"; done
  done
fi

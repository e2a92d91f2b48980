#!/bin/bash
# Robustness check: the GPU parity suite must pass for the same source built at -O1, -O2 and -O3 (device functions always inlined).
# usage (here): tools/opt_level_check.sh build   -> builds petlion.jl_amd/libpetlion_hip_O{1,2}.so (they travel with the snapshot)
#       (GPU box): tools/opt_level_check.sh run  -> runs pytest -m gpu against each
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
if [ "$1" = build ]; then
  for o in 1 2; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O$o -std=c++17 -fPIC -shared -Wno-unused-value -Wno-pass-failed -mllvm -amdgpu-function-calls=false \
      $R/petlion.jl_amd/csrc/petlion_hip.hip -o $R/petlion.jl_amd/libpetlion_hip_O$o.so &
  done
  wait
else
  for o in 1 2; do
    echo "== -O$o"; PETLION_HIP_NO_SIBLING=1 PETLION_HIP_LIB=$R/petlion.jl_amd/libpetlion_hip_O$o.so python -m pytest $R/tests -m gpu -q 2>&1 | tail -3
  done
fi

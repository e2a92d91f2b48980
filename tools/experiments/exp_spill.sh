#!/bin/bash
# usage: exp_spill.sh TAG "extra defines"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -std=c++17 -fPIC -Wno-unused-value -Wno-pass-failed -O3 -DPL_VARIANT=4 "-DPL_DEV=__device__ __forceinline__" $2 -Rpass-analysis=kernel-resource-usage -c /root/repo/petlion.jl_amd/csrc/variant_tu.hip -o /tmp/w/exp_$1.o > /tmp/w/exp_$1.log 2>&1
echo "$1: $(grep -A14 'Function Name: .*k_integrate.*Lb0EEEv13' /tmp/w/exp_$1.log | grep -E 'VGPRs Spill|ScratchSize' | sed 's/.*remark: //' | sed 's/\[-Rpass.*//' | tr '\n' ' ')"

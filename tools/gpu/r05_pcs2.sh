#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05pcs; mkdir -p $O; export TMPDIR=/tmp
( rocprofv3-avail info --pc-sampling 2>&1 | tail -30; rocprofv3-avail list --pc-sampling 2>&1 | tail -30 ) > $O/avail.txt 2>&1
cat $O/avail.txt | cut -c1-300
cd /tmp
for cfg in "stochastic cycles 1048576" "stochastic cycles 4096" "stochastic instructions 1048576" "host_trap time 1000" "host_trap time 1" "host_trap time 10000"; do
  set -- $cfg
  timeout 120 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $1 --pc-sampling-unit $2 --pc-sampling-interval $3 -d /tmp/pcs_t -o t -- python -c "import torch; print(torch.zeros(4).cuda().sum().item())" 2>&1 | grep -v "^W" | tail -2 | cut -c1-250
  echo "== $cfg rc=$?"
done

#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06f; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python tools/host_path_check.py > $O/host_path.txt 2>&1; grep -v "warning\|amdgpu\|RCCL" $O/host_path.txt | tail -8
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_host_api.py tests/test_c_host.py -q -m gpu -p no:cacheprovider -x > $O/pytest_host.txt 2>&1; tail -5 $O/pytest_host.txt

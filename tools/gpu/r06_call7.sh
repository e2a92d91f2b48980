#!/bin/bash
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06g; mkdir -p $O
export TMPDIR=/tmp
for d in "" --detail --detail2; do timeout 300 python tools/phase_profile.py 4096 thermal $d > $O/phase_thermal$d.txt 2>&1; done
cat $O/phase_thermal*.txt | grep -v "amdgpu\|RCCL\|warning"
timeout 2400 python -m pytest tests/test_gpu_multi.py tests/test_gpu_parity.py -q -m gpu -p no:cacheprovider -x -k "eight_gloo or build_from_source or blocking_host or closure" -s > $O/pytest_new.txt 2>&1; tail -12 $O/pytest_new.txt | cut -c1-300

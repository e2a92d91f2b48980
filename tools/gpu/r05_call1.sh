#!/bin/bash
# r05 GPU call 1: sanity + the hold-leg A/B + occupancy pricing (next 3a) + counter list for the issue roofline (next 2)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05a; mkdir -p $O
export TMPDIR=/tmp
( cd /tmp && rocprofv3 -L > $GRAFT_REPO_ROOT/$O/counters_list.txt 2>&1 )
timeout 600 python tools/perf_configs.py c2 c3 c4 c5 --reps 3 > $O/perf.txt 2>&1
timeout 1500 python tools/experiments/reforder_ab.py --cells 256 --case both --out $O/reforder_ab.json > $O/reforder_ab.txt 2>&1
timeout 900 python tools/experiments/c3_two_sample.py --out $O/c3_two_sample.json > $O/c3_two_sample.txt 2>&1
timeout 900 python tools/experiments/occupancy.py run > $O/occupancy.txt 2>&1
timeout 1200 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "quiet or reference_order or stop_function or hold_leg" -p no:cacheprovider > $O/pytest_new.txt 2>&1
for f in perf reforder_ab c3_two_sample occupancy pytest_new; do echo "=== $f"; tail -12 $O/$f.txt; done
grep -c . $O/counters_list.txt

#!/bin/bash
# r05 GPU call 6: the sensitivity tests with the refresh path + the accuracy test with its per-trajectory criterion
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O
timeout 1500 python -m pytest tests/test_sensitivities.py tests/test_gpu_tight.py -m gpu -q -s -p no:cacheprovider -k "sens or accuracy" > $O/pytest_sens.log 2>&1
tail -5 $O/pytest_sens.log; grep -n "factored\|corrector\|accuracy vs" $O/pytest_sens.log | cut -c1-700

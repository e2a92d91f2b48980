#!/bin/bash
# r05 GPU call 8: the C3 sensitivity test with its corrected criterion (on the validated binary) + flag batch 3 (do the r02-r04 switches keep their sign under the iterative scheduler?)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05n; mkdir -p $O
timeout 900 python -m pytest tests/test_sensitivities.py -m gpu -q -s -p no:cacheprovider -k "c3_protocol" > $O/pytest_sens_c3.log 2>&1
tail -2 $O/pytest_sens_c3.log; grep -n "C3 CC-CT-CV\|cells whose" $O/pytest_sens_c3.log | cut -c1-300
python -c "
import sys; sys.path.insert(0, '.')
import pkgload; print(pkgload.load().api.build_info())" > $O/build_info.txt 2>&1; tail -1 $O/build_info.txt
timeout 1800 python tools/experiments/sched_search.py run base3 iso_early iso_fences iso_branchy iso_licm_on iso_nods C2 C4 2>&1 | tail -14
timeout 1800 python tools/experiments/sched_search.py run base3 th_no_fences th_no_branchy th_ds_merge th_licm_on C3 2>&1 | tail -6

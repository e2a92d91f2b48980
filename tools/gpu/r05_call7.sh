#!/bin/bash
# r05 GPU call 7: which instantiations of which variants fail under the iterative scheduler (the test collects them all) + scheduler batch 2
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -p no:cacheprovider -k "every_kernel_instantiation" > $O/pytest_inst.log 2>&1
tail -3 $O/pytest_inst.log; grep -n "AssertionError" $O/pytest_inst.log | head -3 | cut -c1-3000
timeout 1800 python tools/experiments/sched_search.py run base2 iter_minreg iter_maxocc llvm_default no_cluster no_postsched no_lowocc_resched aa_sched antidep_all 2>&1 | tail -40

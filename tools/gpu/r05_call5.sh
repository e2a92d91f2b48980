#!/bin/bash
# r05 GPU call 5: every-cell statistics of C4 / C5 / C2 against the quiet oracle variants (thresholds of the new tests)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05e; mkdir -p $O
for c in c2 c4 c5; do
  n=$(python -c "print(dict(c2=1024, c4=8192, c5=8192)['$c'])")
  timeout 1500 python tools/experiments/c3_two_sample.py --config $c --cells $n --out $O/two_sample_$c.json > $O/two_sample_$c.txt 2>&1
  grep -v amdgpu $O/two_sample_$c.txt | cut -c1-600
done

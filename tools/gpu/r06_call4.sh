#!/bin/bash
# r06 call 4: the whole GPU suite on the full r06 library, smoke(), quick bench lines of the four configs
cd $GRAFT_REPO_ROOT; O=gpurun_out/r06d; mkdir -p $O
export TMPDIR=/tmp
timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider -rs -s > $O/pytest.log 2>&1
tail -15 $O/pytest.log | cut -c1-300
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
for C in C2 C3 C4 C5; do
  timeout 600 python bench.py --config $C --no-cpu-baseline --no-extras > $O/bench_$C.json 2> $O/bench_$C.err
  python -c "
import json,sys
d=json.loads(open('$O/bench_$C.json').read().strip().splitlines()[-1]); print('$C', d['value'], d['ms_per_step'], d['roofline'].get('frac'))" 2>&1 | tail -1
done

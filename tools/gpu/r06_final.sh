#!/bin/bash
# r06 final GPU call: known answers of the self-test from THIS binary, the whole GPU suite, smoke(), rocprofv3 profiles (kernel trace + PMC passes), bench lines C2..C5,
# the validated-build record (identity + what the compiler built)
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06v}; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/make_selftest_golden.py > $O/golden.txt 2>&1; tail -1 $O/golden.txt
cp petlion.jl_amd/selftest_golden.json $O/ 2>/dev/null
timeout 3300 python -m pytest tests -m gpu -q -s -p no:cacheprovider -rs > $O/pytest.log 2>&1
tail -8 $O/pytest.log | cut -c1-300
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
for C in C2 C3 C4 C5; do
  c=$(echo $C | tr A-Z a-z)
  timeout 1500 bash tools/prof.sh r06_$c --config $C > $O/prof_$c.txt 2>&1
  cells=$(python -c "import bench; print(bench.CONFIGS['$C']['cells'])")
  python tools/summarize_profile.py r06_$c r06_$c $C $cells > $O/summary_$c.txt 2>&1
done
mkdir -p $O/profiles; cp profiles/r06_* $O/profiles/ 2>/dev/null
for C in C2 C3 C4 C5; do
  timeout 900 python bench.py --config $C > $O/bench_$C.json 2> $O/bench_$C.err
  python -c "
import json
d=json.loads([l for l in open('$O/bench_$C.json') if l.startswith('{')][-1]); r=d['roofline']
print('$C', round(d['value']), d['ms_per_step'], 'frac', r.get('frac'), 'flop_frac', r.get('flop_frac'), 'arith', r.get('fp64_arith_share_of_valu'), 'valu/step', (r.get('instructions_per_step') or {}).get('valu'), 'parked', r.get('s_waitcnt_parked_share'), 'sync_host', d.get('host_inclusive',{}).get('synchronous_pageable',{}).get('value'))" 2>&1 | tail -1
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err; tail -c 300 $O/bench_default.json; echo
python -c "
import sys; sys.path.insert(0, '.')
import pkgload; print(pkgload.load().api.build_info())" > $O/build_info.txt 2>&1
cp petlion.jl_amd/libpetlion_hip.so.resources.json $O/ 2>/dev/null
find gpurun_out/prof_r06_* -name "*.csv" -size +2M -delete 2>/dev/null
find gpurun_out/prof_r06_* -name "*.db" -size +8M -delete 2>/dev/null
du -sh gpurun_out/prof_r06_* 2>/dev/null | tail -4

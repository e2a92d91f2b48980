#!/bin/bash
# r05 final GPU call: known answers of the self-test from THIS binary, the whole GPU suite, smoke(), rocprofv3 profiles (kernel trace + PMC passes), bench lines C2..C5
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05v; mkdir -p $O
export TMPDIR=/tmp
timeout 900 python tools/make_selftest_golden.py > $O/golden.txt 2>&1
cp petlion.jl_amd/selftest_golden.json $O/ 2>/dev/null
timeout 3000 python -m pytest tests -m gpu -q -s -p no:cacheprovider -rs > $O/pytest.log 2>&1
tail -5 $O/pytest.log
timeout 600 python __graft_entry__.py smoke > $O/smoke.txt 2>&1; tail -2 $O/smoke.txt
for C in C2 C3 C4 C5; do
  c=$(echo $C | tr A-Z a-z)
  timeout 1500 bash tools/prof.sh r05_$c --config $C > $O/prof_$c.txt 2>&1
  cells=$(python -c "import bench; print(bench.CONFIGS['$C']['cells'])")
  python tools/summarize_profile.py r05_$c r05_$c $C $cells > $O/summary_$c.txt 2>&1
done
mkdir -p $O/profiles; cp profiles/r05_* $O/profiles/ 2>/dev/null
for C in C2 C3 C4 C5; do
  timeout 900 python bench.py --config $C > $O/bench_$C.json 2> $O/bench_$C.err
  tail -c 600 $O/bench_$C.json; echo
done
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python -c "
import sys; sys.path.insert(0, '.')
import pkgload; print(pkgload.load().api.build_info())" > $O/build_info.txt 2>&1
# keep the merge-back small: the raw rocprofv3 outputs stay on the box except the databases the summaries were made from
du -sh gpurun_out/prof_r05_* 2>/dev/null | tail -4
find gpurun_out/prof_r05_* -name "*.csv" -size +2M -delete 2>/dev/null
# the miscompile reproducer (tools/experiments/miscompile_repro.py build ran before the snapshot)
[ -f petlion.jl_amd/_exp/libplh_repro16_iter.so ] && timeout 600 python tools/experiments/miscompile_repro.py run > $O/miscompile_repro.txt 2>&1
grep -v "amdgpu\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" $O/miscompile_repro.txt | cut -c1-400

#!/bin/bash
# instruction-cache counters of the bench kernels (GPU box): tools/pmc_icache.sh <tag> --config C2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmcic_$1
shift
mkdir -p $OUT
cd $R
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQC_TC_INST_REQ SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d $OUT/pmc -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/bench.log 2>&1
rocprofv3 --pmc SQ_IFETCH_LEVEL SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQC_ICACHE_BUSY_CYCLES -d $OUT/pmc2 -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras "$@" > $OUT/bench2.log 2>&1
python3 - <<PY
import sqlite3, collections, glob
for f in sorted(glob.glob("$OUT/**/*.db", recursive=True)):
    con = sqlite3.connect(f); cur = con.cursor()
    cur.execute("select * from counters_collection limit 1"); cols = [c[0] for c in cur.description]
    acc = collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        rec = dict(zip(cols, r))
        if "k_integrate" in str(rec.get("kernel_name", rec.get("name", ""))): acc[rec["counter_name"]].append(rec["value"])
    print({k: (sum(v) / len(v)) for k, v in acc.items()})
PY

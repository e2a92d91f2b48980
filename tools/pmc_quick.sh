#!/bin/bash
# usage: tools/pmc_quick.sh TAG "CTR1 CTR2 ..." -> prints per-launch averages for k_integrate
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/pmcq_$1; mkdir -p $OUT; cd $R
rocprofv3 --pmc $2 -d $OUT -o pmc -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline > $OUT/log.txt 2>&1
python3 - <<PY
import sqlite3, collections, glob
for f in glob.glob("$OUT/**/*.db", recursive=True):
    con=sqlite3.connect(f); cur=con.cursor()
    cur.execute("select * from counters_collection limit 1"); cols=[c[0] for c in cur.description]
    acc=collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        rec=dict(zip(cols,r))
        if "k_integrate" in str(rec.get("kernel_name", rec.get("name",""))): acc[rec["counter_name"]].append(rec["value"])
    print({k:(sum(v)/len(v)) for k,v in acc.items()})
PY

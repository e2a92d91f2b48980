#!/bin/bash
# instruction-cache behaviour of the integrate kernels: is the step loop (about 100 kB of code, four cells per CU at different program counters) missing in the I-cache?
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-r06y}; mkdir -p $O; export TMPDIR=/tmp
cd /tmp
rocprofv3-avail list 2>/dev/null | grep -i -E "icache|ifetch|SQC_|INST_LEVEL|SQ_WAIT_INST" | head -60 > $GRAFT_REPO_ROOT/$O/avail.txt
cd $GRAFT_REPO_ROOT
for C in C2 C3; do
  c=$(echo $C | tr A-Z a-z)
  rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $O/ic_$c -o pmc --output-format csv -- python bench.py --config $C --steps 2 --warmup 1 > $O/ic_$c.log 2>&1
  rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -d $O/if_$c -o pmc --output-format csv -- python bench.py --config $C --steps 2 --warmup 1 > $O/if_$c.log 2>&1
done
python - <<P
import csv, glob, collections
for d in sorted(glob.glob("$O/i[cf]_c*")):
    acc = collections.defaultdict(float); n = collections.Counter()
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_integrate" in r.get("Kernel_Name", ""):
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    print(d, {k: round(v / max(n[k], 1)) for k, v in acc.items()}, "launches", dict(n))
P
cat $O/avail.txt | head -30
find $O -name "*.db" -delete 2>/dev/null

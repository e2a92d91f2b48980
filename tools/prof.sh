#!/bin/bash
# profile the bench on the GPU box: kernel trace + separate PMC passes (never combined with sys/hip traces)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
shift
EXTRA="$@ --no-cpu-baseline --no-extras"     # e.g. tools/prof.sh r02_c3 --config C3
mkdir -p $OUT
cd $R
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python bench.py --steps 5 --warmup 2 $EXTRA > $OUT/bench_trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc1 -o pmc -- python bench.py --steps 2 --warmup 1 $EXTRA > $OUT/bench_pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM -d $OUT/pmc2 -o pmc -- python bench.py --steps 2 --warmup 1 $EXTRA > $OUT/bench_pmc2.log 2>&1
# r05: the issue roofline's inputs -- VALU / LDS / scalar busy cycles, and the fp64 instruction mix (each pass on its own: an unknown counter name fails only its pass)
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU -d $OUT/pmc5 -o pmc -- python bench.py --steps 2 --warmup 1 $EXTRA > $OUT/bench_pmc5.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT -d $OUT/pmc6 -o pmc -- python bench.py --steps 2 --warmup 1 $EXTRA > $OUT/bench_pmc6.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc7 -o pmc -- python bench.py --steps 2 --warmup 1 $EXTRA > $OUT/bench_pmc7.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc -- python bench.py --steps 2 --warmup 1 $EXTRA > $OUT/bench_pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc4 -o pmc -- python bench.py --steps 2 --warmup 1 $EXTRA > $OUT/bench_pmc4.log 2>&1
find $OUT -name "*.csv" | head -30
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do echo "== $f"; head -8 $f; done
python3 - <<PY
import csv, glob, collections
for d in ("pmc1","pmc2","pmc3","pmc4","pmc5","pmc6","pmc7"):
    for f in glob.glob("$OUT/%s/**/*counter_collection.csv" % d, recursive=True):
        acc = collections.defaultdict(lambda: [0.0,0])
        for r in csv.DictReader(open(f)):
            if "k_integrate" in r.get("Kernel_Name",""):
                acc[r["Counter_Name"]][0] += float(r["Counter_Value"]); acc[r["Counter_Name"]][1] += 1
        print(d, {k:(v[0]/max(1,v[1]), v[1]) for k,v in acc.items()})
PY

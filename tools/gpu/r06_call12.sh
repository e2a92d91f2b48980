#!/bin/bash
# why does the whole-cycle figure of bench.py's blocking host call differ between runs?  default flags against --steps 100, with the library's phase trace
out=gpurun_out/${1:-r06t}; mkdir -p $out
PLH_HOST_TRACE=1 timeout 900 python bench.py --config C2 > $out/bench_default.json 2> $out/trace_default.txt
PLH_HOST_TRACE=1 timeout 900 python bench.py --config C2 --steps 100 --warmup 10 > $out/bench_s100.json 2> $out/trace_s100.txt
for t in default s100; do echo "== $t"; grep "plh host call" $out/trace_$t.txt | grep "7.4 MB" | tail -4 | cut -c1-330; python - <<P
import json
d=json.loads([l for l in open("$out/bench_$t.json") if l.startswith("{")][-1]); sp=d["host_inclusive"]["synchronous_pageable"]
print(round(d["value"]), "cycle", round(sp["value"]), round(sp["ms_per_call_median"],3), "call", round(sp["inside_plh_integrate"]["ms_per_call_median"],3))
P
done

#!/usr/bin/env python3
"""Refresh the measured tables of DESIGN.md from one GPU run's bench lines (tools/bench_tables.py does the formatting):
       python tools/update_design_tables.py gpurun_out/<dir> [r05_bench_lines.json]
replaces, in place and nothing else: the bench table and the roofline table of section 6, the feature-cost table of section 3 and the two `predicted_scaling` partitions lines."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d, name = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "r05_bench_lines.json")
out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_tables.py"), d, name], capture_output=True, text=True).stdout
tabs = [t.strip("\n") for t in out.split("\n\n") if t.strip()]
bench_t, roof_t, feat_t, pred = tabs[0], tabs[1], tabs[2], tabs[3].split("\n")
p = os.path.join(ROOT, "DESIGN.md")
s = open(p).read()


def swap_table(s, new):
    head = new.split("\n")[0]
    a = s.index(head)
    b = s.index("\n\n", a)
    return s[:a] + new + s[b:]


for t in (bench_t, roof_t, feat_t):
    s = swap_table(s, t)
m = re.search(r"partitions: block: shard kernel ms .*?; cyclic: shard kernel ms .*?trajectories/s\.", s, re.S)          # (the prose may have been re-wrapped: tools/wrap_design.py)
assert m, "predicted_scaling line"
s = s[:m.start()] + "partitions: " + "; ".join(x.strip() for x in pred[:2]) + "." + s[m.end():]
open(p, "w").write(s)
print("DESIGN.md: tables of sections 3 and 6 from", d)

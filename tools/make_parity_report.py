#!/usr/bin/env python3
"""The measured lines of profiles/<round>_parity_report.md from a `pytest tests -m gpu -s` log:
     python tools/make_parity_report.py gpurun_out/r03p/pytest.log        (prints the block that goes between the ``` fences)"""
import re, sys
KEEP = re.compile(r"^\.*(accuracy vs reltol|C5 at 8192|lgm50_thermal:|C2 \+ C4|C3 CC|C3 model|C5 GITT|CC-CV and pulse|LCO isothermal, CC|   rungs skipped|   sections beyond|two waves per cell|closure with derivative|"
                  r"C4 subset|C4, 65 536|C5 mixed vs fp64|lco_iso|grid \(|C2:|C3|C4:|C5:|device vs quiet|CC -> V hold|default-tolerance sens|C5 at 1e-8|thermal device|LCO CC|LCO 2C|C4 cells|C4 shard|cells whose|   corrector|   differing|   cells with|default build)")
for line in open(sys.argv[1]):
    line = line.rstrip("\n")
    if KEEP.match(line):
        print(re.sub(r"^\.+", "", line))

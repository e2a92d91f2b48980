#!/usr/bin/env python3
"""Re-wrap the over-long PROSE lines of DESIGN.md (tables, headings and indented code blocks are left alone): python tools/wrap_design.py [file] [width]"""
import re, sys, textwrap
path = sys.argv[1] if len(sys.argv) > 1 else "DESIGN.md"
width = int(sys.argv[2]) if len(sys.argv) > 2 else 180
out = []
for ln in open(path).read().split("\n"):
    if len(ln) <= width + 10 or ln.startswith(("|", "#", "    ")) and not re.match(r"\s+\S", ln[:4] + "x") or ln.lstrip().startswith("|"):
        out.append(ln); continue
    if ln.startswith("    ") and not re.match(r"\s*(\*|\d+\.)\s", ln):        # code block
        out.append(ln); continue
    m = re.match(r"(\s*)((?:\*|\d+\.)\s+)?", ln)
    lead, mark = m.group(1), m.group(2) or ""
    body = ln[len(lead) + len(mark):]
    cont = lead + " " * len(mark)
    w = textwrap.wrap(body, width=width - len(cont), break_long_words=False, break_on_hyphens=False)
    out.append(lead + mark + w[0]); out.extend(cont + x for x in w[1:])
open(path, "w").write("\n".join(out))

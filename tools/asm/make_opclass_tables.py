#!/usr/bin/env python3
"""Regenerates profiles/r06_opclass_{iso,thermal}.md from the tree as it is (no GPU: hipcc cross-compiles, ~15 s per kernel): the ISA-inspection build of the integrate kernel
with the production flag table (tools/asm/isa.py), then the opcode-class table of the step loop (tools/asm/opclass.py), from the third `tic` marker -- the first one inside
the step loop -- to the end of the kernel."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NOLSO = ["-Xclang", "-target-feature", "-Xclang", "-load-store-opt"]
SCHED = ["-mllvm", "-disable-machine-licm", "-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]
ISO_TEXT = """From `python tools/asm/isa.py iso integrate <production flags>` + `python tools/asm/opclass.py /tmp/asm/iso_integrate.s --from %d --md` (tools/asm/make_opclass_tables.py; the ISA-inspection build: the production source and flag table plus -DPL_ASM_MARKS, whose markers are compiler barriers; static instruction counts of the code between consecutive markers IN FILE ORDER, from the first marker of the step loop to the end of the kernel; loops counted once, both arms of every branch counted).

Classes: f64 = v_fma/fmac/mul/add_f64; trans = v_rcp/rsq/sqrt_f64; divh = v_div_scale/fmas/fixup (IEEE-division scaffolding); f64x = other fp64 (ldexp, frexp, rndne, cvt, class); mov, cnd = v_cndmask, cmp, dpp = DPP moves and the fused v_fmac_f64_dpp of the particle phases, xlane = v_readlane/writelane (broadcasts and SGPR spill traffic), agpr = v_accvgpr copies, int; smov = s_mov_b32 (halves of fp64 literals), sexec = exec-mask manipulation, sbr = branches, salu = other scalar; lds, vmem, wait, nop."""
TH_TEXT = "As profiles/r06_opclass_iso.md, for the thermal variant (`python tools/asm/isa.py thermal integrate <production flags of the thermal variants>`, `tools/asm/opclass.py /tmp/asm/thermal_integrate.s --from %d --md`)."


def gen(model, title, flags, text):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm", "isa.py"), model, "integrate"] + flags, capture_output=True, text=True)
    res = [l for l in r.stdout.splitlines() if "load-store-opt" not in l][:6]
    asm = "/tmp/asm/%s_integrate.s" % model
    tics = [k for k, l in enumerate(open(asm), 1) if l.strip() == "; PLMARK tic"]
    first = tics[2]
    tab = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asm", "opclass.py"), asm, "--from", str(first), "--md"], capture_output=True, text=True).stdout
    out = os.path.join(ROOT, "profiles", "r06_opclass_%s.md" % model)
    open(out, "w").write("# Opcode classes of `k_integrate<%s, 0>`: the step loop (static counts per marked region, r06)\n\n%s\n\n%s\n\n%s" % (title, text % first, "\n".join("    " + l for l in res), tab))
    print(out, "from line", first, [l.strip() for l in res[:6]])


gen("iso", "lco_iso", ["-DPL_DEV=__device__ __forceinline__"] + NOLSO + SCHED, ISO_TEXT)
gen("thermal", "lco_thermal", NOLSO + ["-DPL_PHASE_FENCES", "-DPL_EXP_BRANCHY_PHI"] + SCHED, TH_TEXT)

#!/usr/bin/env python3
"""Opcode-class table of one kernel's assembly (the .s that tools/asm/isa.py leaves in /tmp/asm), per marked region IN FILE ORDER (every occurrence of a PLMARK
separately: the integrate kernel holds several inlined copies of the solve / node pass, and only the copies inside the step loop are hot).
   python tools/asm/opclass.py /tmp/asm/iso_integrate.s [--from LINE --to LINE] [--md]
Classes: fp64 arithmetic (fma / mul / add), transcendental (rcp, rsq, sqrt), IEEE-division helpers (div_scale / div_fmas / div_fixup), other fp64
(ldexp, frexp, rndne, cvt, class, min / max), v_mov (b32 / b64), v_cndmask, compares, DPP moves, readlane / writelane (SGPR spills and broadcasts), AGPR copies,
integer VALU; SALU: s_mov (literal halves of fp64 constants), exec-mask manipulation, branches, other; LDS; VMEM; waitcnt / nop."""
import collections, re, sys

CLASSES = ["f64", "trans", "divh", "f64x", "mov", "cnd", "cmp", "dpp", "xlane", "agpr", "int", "smov", "sexec", "sbr", "salu", "lds", "vmem", "wait", "nop"]


def classify(t):
    op = t.split()[0]
    if op.startswith("v_"):
        if "row_shr" in t or "row_shl" in t or "wave_shr" in t or "wave_shl" in t or "quad_perm" in t or "row_bcast" in t or "row_newbcast" in t: return "dpp"
        if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")): return "xlane"
        if op.startswith("v_accvgpr"): return "agpr"
        if op.startswith(("v_fma_f64", "v_fmac_f64", "v_mul_f64", "v_add_f64")): return "f64"
        if op.startswith(("v_rcp_f64", "v_rsq_f64", "v_sqrt_f64")): return "trans"
        if op.startswith(("v_div_scale", "v_div_fmas", "v_div_fixup")): return "divh"
        if op.startswith("v_cndmask"): return "cnd"
        if op.startswith("v_cmp"): return "cmp"
        if op.startswith(("v_mov_b32", "v_mov_b64")): return "mov"
        if "f64" in op: return "f64x"
        return "int"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "flat_", "buffer_", "scratch_", "s_load", "s_buffer_load", "s_store")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith("s_nop"): return "nop"
    if op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")): return "sbr"
    if op.startswith("s_mov_b32") or op.startswith("s_brev_b32"): return "smov"
    if "exec" in t or op.startswith(("s_and_saveexec", "s_or_saveexec", "s_andn2_saveexec", "s_xor_saveexec")): return "sexec"
    if op.startswith("s_"): return "salu"
    return "salu"


def regions(path, lo=0, hi=10 ** 9):
    cur, out = ("start", 1), []
    cnt = collections.Counter()
    for k, ln in enumerate(open(path), 1):
        t = ln.strip()
        if t.startswith("; PLMARK"):
            out.append((cur, cnt)); cur, cnt = (t[9:], k), collections.Counter()
            continue
        if k < lo or k > hi or not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        cnt[classify(t)] += 1
    out.append((cur, cnt))
    return out


def main():
    a = sys.argv[1:]
    path = a[0]
    lo = int(a[a.index("--from") + 1]) if "--from" in a else 0
    hi = int(a[a.index("--to") + 1]) if "--to" in a else 10 ** 9
    md = "--md" in a
    regs = [r for r in regions(path, lo, hi) if sum(r[1].values())]
    tot = collections.Counter()
    sep = " | " if md else " "
    head = ("%-26s" % "region (line)") + sep + sep.join("%5s" % c for c in CLASSES) + sep + "  VALU" + sep + " f64%"
    print(("| " if md else "") + head + (" |" if md else ""))
    if md:
        print("|" + "---|" * (len(CLASSES) + 3))

    def row(name, c):
        valu = sum(c[x] for x in ("f64", "trans", "divh", "f64x", "mov", "cnd", "cmp", "dpp", "xlane", "agpr", "int"))
        arith = c["f64"] + c["trans"]
        s = ("%-26s" % name[:26]) + sep + sep.join("%5d" % c[x] for x in CLASSES) + sep + "%6d" % valu + sep + ("%5.1f" % (100.0 * arith / valu) if valu else "    -")
        print(("| " if md else "") + s + (" |" if md else ""))
    for (name, line), c in regs:
        row("%s (%d)" % (name, line), c)
        tot.update(c)
    row("TOTAL", tot)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""ISA inspection of one kernel instantiation (no GPU needed: hipcc cross-compiles gfx950).
   python tools/asm/isa.py thermal|iso|sei integrate|solve|residual [extra hipcc flags ...]
Prints the compiler's resource lines and, per marked phase (the PL_TIC* / PL_TOC* positions, -DPL_ASM_MARKS), the instruction mix between consecutive markers:
VALU / DPP / LDS / VMEM (global + scratch) / SALU / waitcnt counts.  The assembly stays in /tmp/asm/<model>_<kernel>.s."""
import collections, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
MODELS = {"iso": ("0", "false", "false", []), "sei": ("1", "true", "false", []), "thermal": ("0", "false", "true", ["-DPL_DEV=__device__ __forceinline__"]),
          "lcosei": ("0", "true", "false", [])}
LATE = ["-mllvm", "-amdgpu-function-calls=false"]
KERN = {"residual": 0, "solve": 1, "integrate": 2}


def main():
    model, kern, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    chem, sei, th, inl = MODELS[model]
    os.makedirs("/tmp/asm", exist_ok=True)
    out = "/tmp/asm/%s_%s.s" % (model, kern)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-std=c++17", "-O3", "-Wno-unused-value", "-Wno-pass-failed", "-DPL_ASM_MARKS", "-DPL_ONE_CHEM=" + chem, "-DPL_ONE_SEI=" + sei,
           "-DPL_ONE_TH=" + th, "-DPL_ONE_KERNEL=%d" % KERN[kern], "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-I", os.path.join(ROOT, "petlion.jl_amd", "csrc"),
           os.path.join(ROOT, "tools", "asm", "one_kernel.hip"), "-o", out] + (inl or LATE) + extra
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        print(r.stderr[-3000:]); sys.exit(1)
    for ln in r.stderr.splitlines():
        m = re.search(r"remark:\s+(VGPRs|AGPRs|ScratchSize \[bytes/lane\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", ln)
        if m:
            print("  %-28s %s" % (m.group(1), m.group(2)))
    summarize(out)


LDS_CYC = {"ds_read_b32": 2, "ds_read_b64": 2, "ds_read_b128": 4, "ds_read_b96": 8, "ds_read2_b32": 4, "ds_read2_b64": 8, "ds_read2st64_b64": 8, "ds_read2st64_b32": 4,
           "ds_write_b32": 4, "ds_write_b64": 6, "ds_write_b128": 13, "ds_write2_b64": 13, "ds_write2st64_b64": 13, "ds_write2_b32": 6, "ds_write_b96": 10}


def classify(op):
    if op.startswith("v_") and "dpp" in op: return "dpp"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_permlane")): return "xlane"
    if op.startswith(("v_accvgpr",)): return "agpr"
    if op.startswith("v_"): return "valu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("scratch_"): return "scratch"
    if op.startswith(("global_", "flat_", "buffer_")): return "vmem"
    if op.startswith("s_waitcnt"): return "wait"
    if op.startswith(("s_load", "s_buffer_load")): return "smem"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_"): return "salu"
    return "other"


def summarize(path):
    cur, stats, order = "start", collections.OrderedDict(), []
    f64 = collections.Counter()
    for ln in open(path):
        t = ln.strip()
        if t.startswith("; PLMARK"):
            cur = t[9:]
            continue
        if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
            continue
        op = t.split()[0]
        # dpp shows up as a modifier on the line
        cl = "dpp" if ("row_shr" in t or "row_shl" in t or "wave_shr" in t or "wave_shl" in t or "quad_perm" in t or "row_bcast" in t) else classify(op)
        stats.setdefault(cur, collections.Counter())[cl] += 1
        if cl == "lds":          # LDS-array cycles per wave-instruction (MI355X_MICROARCH.md, LDS table): the array is shared by the CU's four single-wave cells
            stats[cur]["ldscyc"] += LDS_CYC.get(op, 4)
            stats[cur]["lds"] -= 0
    cols = ["valu", "dpp", "xlane", "agpr", "lds", "ldscyc", "vmem", "scratch", "smem", "salu", "branch", "wait"]
    print("  (static instruction counts in the code FOLLOWING each marker, up to the next one; loops are counted once)")
    print("  %-22s" % "after marker" + "".join("%8s" % c for c in cols) + "   total")
    for k, c in stats.items():
        print("  %-22s" % k[:22] + "".join("%8d" % c[x] for x in cols) + "   %5d" % (sum(c.values()) - c["ldscyc"]))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turns the rocprofv3 sqlite outputs under gpurun_out/prof_<tag>/ into small tracked text summaries under profiles/.
usage: python tools/summarize_profile.py <tag> [label]"""
import collections, glob, os, sqlite3, sys
tag = sys.argv[1]; label = sys.argv[2] if len(sys.argv) > 2 else tag
workload = sys.argv[3] if len(sys.argv) > 3 else "C2"                       # usage: summarize_profile.py <tag> [label] [workload] [cells] [precision]
cells = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
precision = sys.argv[5] if len(sys.argv) > 5 else "f64"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_record(workload):
    """the plain kernel of the workload's variant in the build record next to the library (tools/kernel_resources.py; the same figures go into profiles/validated_build.json)"""
    import json
    try:
        k = json.load(open(os.path.join(ROOT, "petlion.jl_amd", "libpetlion_hip.so.resources.json")))["kernels"][{"C2": "v0", "C4": "v0", "C3": "v4", "C5": "v3"}[workload]]["k_integrate<0: plain>"]
        return ("petlion.jl_amd/libpetlion_hip.so.resources.json: %d registers (%d of them AGPRs), %d VGPR spills, %d SGPR spills, %d B/lane of scratch, %d B of LDS per cell"
                % (k["vgpr_count"], k["agpr_count"], k["vgpr_spill_count"], k["sgpr_spill_count"], k["private_segment_fixed_size"], k["group_segment_fixed_size"]))
    except Exception as e:
        return "no build record next to the library (%r)" % (e,)


src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
out = os.path.join(ROOT, "profiles", "%s_rocprofv3_summary.md" % label)
L = ["# rocprofv3 summary `%s`" % label, "",
     "Command profiled (GPU box, 1x MI355X): `python bench.py --config %s%s --steps 5 --warmup 2 --no-cpu-baseline --no-extras` for the kernel trace," % (workload, "" if precision == "f64" else " --precision " + precision),
     "`--steps 2 --warmup 1` for each PMC pass (separate passes, never combined with sys/hip traces). Source: tools/prof.sh.", ""]
con = sqlite3.connect(os.path.join(src, "trace", "trace_results.db")); cur = con.cursor()
L += ["## `rocprofv3 --kernel-trace --stats` : top kernels", "", "| kernel | calls | total (us) | average (us) | % |", "|---|---|---|---|---|"]
for r in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels limit 6"):
    L.append("| `%s` | %d | %.1f | %.1f | %.2f |" % (r[0][:70], r[1], r[2] / 1e3 if r[2] > 1e6 else r[2], r[3] / 1e3 if r[3] > 1e5 else r[3], r[4]))
rows = cur.execute("select duration, vgpr_count, accum_vgpr_count, sgpr_count, lds_size, scratch_size, grid_x, workgroup_x from kernels where name like '%k_integrate%' order by start").fetchall()
d = [r[0] for r in rows]
L += ["", "`k_integrate`: %d dispatches, duration min/avg/max = %.1f / %.1f / %.1f us; grid %d x wg %d; arch VGPR %s, AGPR %s, SGPR %s, LDS %s B, scratch %s B/lane"
      % (len(d), min(d) / 1e3, sum(d) / len(d) / 1e3, max(d) / 1e3, rows[0][6], rows[0][7], rows[0][1], rows[0][2], rows[0][3], rows[0][4], rows[0][5]),
      "", "(register columns as rocprofv3 reports them; `accum_vgpr_count` reads 0 on this unified-file part.  What the compiler built, from the code object's own notes -- "
      + build_record(workload) + ")",
      "", "Per dispatch in launch order (us): " + ", ".join("%.0f" % (x / 1e3) for x in d) + " -- the first launches of a process run slower (clock ramp); "
      "the steady state (last three: %.0f us) is what `bench.py` times after its warm-up steps (`roofline.kernel_ms_avg`, HIP events on the launch stream)." % (sum(d[-3:]) / 3e3), ""]
L += ["## PMC passes (per k_integrate launch = %d cells = %d wavefronts; averages over the dispatches of the run)" % (cells, cells), "", "| counter | value per launch | per wavefront |", "|---|---|---|"]
vals = {}
for dbf in sorted(glob.glob(os.path.join(src, "pmc*", "*.db"))):
    con = sqlite3.connect(dbf); cur = con.cursor()
    cur.execute("select * from counters_collection limit 1"); cols = [c[0] for c in cur.description]
    acc = collections.defaultdict(list)
    for r in cur.execute("select * from counters_collection"):
        rec = dict(zip(cols, r))
        if "k_integrate" in str(rec.get("kernel_name", rec.get("name", ""))):
            acc[rec["counter_name"]].append(rec["value"])
    for k, v in acc.items():
        vals[k] = sum(v) / len(v)
for k in sorted(vals):
    L.append("| %s | %.4g | %.4g |" % (k, vals[k], vals[k] / cells))
if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    rd, wr = vals["FETCH_SIZE"] * 1024 * 2, vals["WRITE_SIZE"] * 1024     # KiB -> B; gfx950 FETCH_SIZE reads 1/2 (MI355X_MICROARCH.md, HBM)
    L += ["", "HBM traffic per launch (FETCH_SIZE x 1024 B x 2 [gfx950 correction for wide coalesced reads, MI355X_MICROARCH.md section HBM] + WRITE_SIZE x 1024 B, write side uncalibrated):",
          "read %.1f MB + write %.1f MB = %.1f MB per launch = %.1f kB per trajectory." % (rd / 1e6, wr / 1e6, (rd + wr) / 1e6, (rd + wr) / cells / 1e3)]
    import json
    json.dump({"hbm_bytes_per_launch": rd + wr, "read_bytes": rd, "write_bytes": wr, "cells_per_launch": cells, "workload": workload, "precision": precision,
               "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on `python bench.py --config %s --steps 2 --warmup 1 --no-cpu-baseline --no-extras`; " % workload +
                         "FETCH_SIZE x 1024 B x 2 (gfx950 correction, MI355X_MICROARCH.md HBM section) + WRITE_SIZE x 1024 B; see %s_rocprofv3_summary.md" % label},
              open(os.path.join(ROOT, "profiles", "%s_traffic.json" % label), "w"), indent=1)
# everything the bench line's issue roofline needs, per launch (bench.py issue_roofline): instruction counts and busy cycles are a property of (binary, workload)
try:
    import json, subprocess
    sys.path.insert(0, ROOT)
    info = ""
    try:
        import pkgload
        info = pkgload.load().api.build_info()
    except Exception:
        pass
    src_hash = info.split("src=")[-1] if "src=" in info else None
    json.dump({"workload": workload, "cells_per_launch": cells, "precision": precision, "build_info": info, "build_src": src_hash,
               "hbm_bytes_per_launch": (vals["FETCH_SIZE"] * 1024 * 2 + vals["WRITE_SIZE"] * 1024) if ("FETCH_SIZE" in vals and "WRITE_SIZE" in vals) else None,
               "kernel_us_avg": sum(d) / len(d) / 1e3, "kernel_us_steady": sum(d[-3:]) / 3e3, "counters": vals,
               "source": "rocprofv3 --pmc (separate passes, tools/prof.sh) on `python bench.py --config %s --steps 2 --warmup 1 --no-cpu-baseline --no-extras`; per-launch averages over the "
                         "k_integrate dispatches; SQ_*_CYCLES / SQ_ACTIVE_* / SQ_WAIT_* count quad-cycles (MI355X_MICROARCH.md)" % workload},
              open(os.path.join(ROOT, "profiles", "%s_pmc.json" % label), "w"), indent=1)
except Exception as e:
    print("pmc.json not written:", e)
if "SQ_ACTIVE_INST_VALU" in vals and "SQ_WAVE_CYCLES" in vals:
    L += ["", "VALU-busy share of the wave's cycles (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES): %.1f %%; LDS-busy %.1f %%; scalar-busy %.1f %%; VALU instructions x 4 cycles / wave cycles: %.1f %%."
          % (100 * vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"], 100 * vals.get("SQ_ACTIVE_INST_LDS", 0) / vals["SQ_WAVE_CYCLES"],
             100 * vals.get("SQ_ACTIVE_INST_SCA", 0) / vals["SQ_WAVE_CYCLES"], 100 * vals.get("SQ_INSTS_VALU", 0) / vals["SQ_WAVE_CYCLES"])]
if "SQ_WAVE_CYCLES" in vals:
    L += ["", "SQ_WAVE_CYCLES etc. count quad-cycles: %.3g shader cycles per wavefront; VALU-active fraction %.0f %%, s_waitcnt-parked fraction %.0f %%."
          % (4 * vals["SQ_WAVE_CYCLES"] / cells, 100 * vals.get("SQ_ACTIVE_INST_ANY", 0) / vals["SQ_WAVE_CYCLES"], 100 * vals.get("SQ_WAIT_ANY", 0) / vals["SQ_WAVE_CYCLES"])]
open(out, "w").write("\n".join(L) + "\n")
print(open(out).read())

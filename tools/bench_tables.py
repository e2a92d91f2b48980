#!/usr/bin/env python3
"""The tables of DESIGN.md section 6 / 3 from the bench lines of one GPU call:   python tools/bench_tables.py gpurun_out/r05d   (also writes profiles/r05_bench_lines.json)"""
import json, os, sys
d = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = {}
for C in ("C2", "C3", "C4", "C5"):
    L[C] = json.loads([l for l in open(os.path.join(d, "bench_%s.json" % C)) if l.startswith('{"metric')][0])
if len(sys.argv) > 2:
    json.dump(L, open(os.path.join(ROOT, "profiles", sys.argv[2]), "w"), indent=1)
k = lambda x: "%.1f k" % (x / 1e3) if x < 1e5 else "%.0f k" % (x / 1e3)
print("| config | cells | kernel | `value` (traj/s) | host-inclusive pipeline | with `YP_final` | oracle, 1 core / %d cores |" % L["C2"]["cpu_baseline_all_cores"]["cores"])
print("|---|---|---|---|---|---|---|")
for C, b in L.items():
    print("| %s | %d | %.3f ms | **%s** | %s | %s | %.0f / %s |" % (C, b["config"]["cells_per_gpu"], b["roofline"]["kernel_ms_avg"], k(b["value"]), k(b["host_inclusive"]["aggregate_value"]),
                                                              k(b["with_YP_final"]["trajectories_per_s"]), b["cpu_baseline"]["value"], k(b["cpu_baseline_all_cores"]["value"])))
print()
print("| config | `roofline.frac` (VALU issue) | VALU-busy / s_waitcnt-parked / LDS-busy / scalar-busy share of the wave's cycles | VALU / SALU / LDS instructions per step | fp64 FMA / MUL / ADD / transcendental per trajectory | HBM bytes per launch (utilisation) | `equivalent_streaming.frac` |")
print("|---|---|---|---|---|---|---|")
for C, b in L.items():
    r = b["roofline"]; m = r["fp64_instruction_mix_per_trajectory"] or {}; i = r["instructions_per_step"]
    print("| %s | **%.3f** (%.0f of %.0f G cycles/s) | %.3f / %.3f / %.3f / %.3f | %.0f / %.0f / %.0f | %.0f k / %.0f k / %.0f k / %.1f k | %.1f MB (%.2f %%) | %.2f |"
          % (C, r["frac"], r["achieved"], r["peak"], r["valu_busy_share_of_wave_cycles"], r["s_waitcnt_parked_share"], r["lds_busy_share"], r["scalar_busy_share"], i["valu"], i["salu"], i["lds"],
             m.get("fma_f64", 0) / 1e3, m.get("mul_f64", 0) / 1e3, m.get("add_f64", 0) / 1e3, m.get("trans_f64", 0) / 1e3, r["traffic"] / 1e6, 100 * r["hbm_utilisation"], b["equivalent_streaming"]["frac"]))
print()
names = list(L["C2"]["general_path"].keys())
print("| feature (kernel instantiation) | C2 | C3 | C4 | C5 |")
print("|---|---|---|---|---|")
for n in names + [x for x in L["C5"]["general_path"] if x not in names]:
    row = []
    for C in L:
        g = L[C]["general_path"]
        key = n if n in g else None          # (exact names only: matching on a prefix printed the 7-parameter sensitivity row twice)
        row.append("%.2f" % g[key]["vs_plain_kernel"] if key and "vs_plain_kernel" in g[key] else "--")
    print("| %s | %s |" % (n.split(":")[0][:110], " | ".join(row)))
ps = L["C4"].get("predicted_scaling")
if ps:
    print()
    for part in ("block", "cyclic"):
        q = ps[part]
        print("%s: shard kernel ms %s; max / mean %.4f; predicted efficiency %.3f; predicted 8-GPU rate %.2f M trajectories/s" % (part, ["%.2f" % x for x in q["shard_kernel_ms"]], q["max_over_mean"], q["predicted_efficiency"], q["predicted_8gpu_trajectories_per_s"] / 1e6))
    print("one-rank plh_ensemble_run:", ps["ensemble_run_one_rank"])

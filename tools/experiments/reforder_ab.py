"""A/B of the device's evaluation order against the oracle on :hold legs (VERDICT r04 "next" 1; DESIGN.md 5 "why a hold leg is not two exchangeable draws").

The claim under test: the device's flux-form Phi_e rows carry ~40x less evaluation rounding than the reference's matrix form, IDA's start-up order selection in a :hold leg reads
exactly that rounding, and THAT -- not a defect of the device integrator -- is why the device's error on a CV-hold leg is not distributed like the oracle's (r04: median ratio
1.49 on the isothermal CC -> CV hold, C3 median 4x the floor's).  The experiment that can falsify it: the SAME device code with the finite-volume rows evaluated in the reference's
operation order (precision = "f64_reforder", PLH_PREC_F64_REFORDER) against the notebook-pinned oracle variants (lco_iso, lco_thermal).

    python tools/experiments/reforder_ab.py [--emu] [--cells N] [--case iso|thermal|both] [--out file.json]

Per case and precision: device error / oracle error against the tight-tolerance oracle (median, range, mean log ratio), cells with the oracle's step count, cells with
identical counters.  --emu runs the device source on the test-only wave emulator (CPU, slow: ~2 s per isothermal cell)."""
import argparse
import json
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "wave_emu"))
import pkgload  # noqa: E402
import parity  # noqa: E402
from oracle import oracle as O  # noqa: E402

CNT = ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail")


def cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emu", action="store_true"); ap.add_argument("--cells", type=int, default=64); ap.add_argument("--case", default="both"); ap.add_argument("--out", default=None)
    a = ap.parse_args()
    pkg = pkgload.load(); O.build()
    lib = None
    if a.emu:
        import build_emu
        lib = build_emu.build()
    kw = dict(T_max=400.0, V_max=5.0, I_max=10.0, I_min=0.0, SOC_max=2.0)
    cases = []
    if a.case in ("iso", "both"):
        hold = [dict(I=2.0, tf=900.0, V_max=5.0), dict(V="hold", tf=600.0, V_max=5.0, I_min=0.0)]
        cases.append(("LCO isothermal, CC 900 s -> CV hold 600 s", dict(), lambda p: pkg.configs.sweep_theta(p, np.arange(a.cells), 4), hold, None, (1, 2)))
    if a.case in ("thermal", "both"):
        th_proto = [dict(I=4.0, tf=300.0, **kw), dict(dT="hold", tf=200.0, **kw), dict(V="hold", tf=300.0, **kw)]
        cases.append(("C3 model, CC 300 s -> CT hold 200 s -> CV hold 300 s", dict(temperature=True), lambda p: pkg.configs.c3(p, 4096)["theta"][::4096 // a.cells][:a.cells], th_proto, "lco_thermal_quiet", (1, 2, 3)))
    out = []
    for what, mkw, thf, proto_full, tight_variant, prefixes in cases:
        models = {prec: pkg.petlion(pkg.LCO, precision=prec, _lib_path=lib, **mkw) for prec in ("f64", "f64_reforder")}
        p0 = models["f64"]
        Th = np.ascontiguousarray(thf(p0))
        for npre in prefixes:
            proto = proto_full[:npre]
            runs = parity.runs_to_oracle(O, p0, pkg, proto)

            def one(i):
                ro = O.simulate(p0.variant, Th[i], 0.0, runs)
                rt = O.simulate(tight_variant or p0.variant, Th[i], 0.0, runs, opts=O.default_opts(maxiters=1000000, **parity.TIGHT), max_out=200000)
                rp = O.simulate(p0.variant, Th[i], 0.0, runs, opts=O.default_opts(fd_perturb=2.2e-16, res_perturb=2.2e-16, perturb_seed=1 + i % 7))
                rq = O.simulate(p0.variant + "_quiet", Th[i], 0.0, runs)            # the oracle with every cancelling stencil evaluated on differences (codegen.py)
                return ro, rt, rp, rq
            with ThreadPoolExecutor(cores()) as ex:
                both = list(ex.map(one, range(len(Th))))
            e_orc = np.array([parity.state_rel_err(b[0]["Y"], b[1]["Y"]) for b in both])
            quiet = [b[3] for b in both]; both = [b[:3] for b in both]
            rows = {"perturbed oracle": (np.array([parity.state_rel_err(rp["Y"], rt["Y"]) for _, rt, rp in both]),
                                         sum(rp["counters"]["n_steps"] == ro["counters"]["n_steps"] for ro, _, rp in both),
                                         sum(all(rp["counters"][f] == ro["counters"][f] for f in CNT) for ro, _, rp in both),
                                         np.array([parity.state_rel_err(rp["Y"], ro["Y"]) for ro, _, rp in both]))}
            devs = {}
            for prec, pm in models.items():
                ens = pkg.simulate_ensemble(pm, Th, proto, SOC=0.0)
                devs[prec] = ens
                e_dev = np.array([parity.state_rel_err(ens.Y[i], both[i][1]["Y"]) for i in range(len(Th))])
                rows["device " + prec] = (e_dev, sum(int(ens.counters[i]["n_steps"]) == both[i][0]["counters"]["n_steps"] for i in range(len(Th))),
                                          sum(all(int(ens.counters[i][f]) == both[i][0]["counters"][f] for f in CNT) for i in range(len(Th))),
                                          np.array([parity.state_rel_err(ens.Y[i], both[i][0]["Y"]) for i in range(len(Th))]))
            rows["quiet oracle"] = (np.array([parity.state_rel_err(rq["Y"], rt["Y"]) for (_, rt, _), rq in zip(both, quiet)]),
                                    sum(rq["counters"]["n_steps"] == ro["counters"]["n_steps"] for (ro, _, _), rq in zip(both, quiet)),
                                    sum(all(rq["counters"][f] == ro["counters"][f] for f in CNT) for (ro, _, _), rq in zip(both, quiet)),
                                    np.array([parity.state_rel_err(rq["Y"], ro["Y"]) for (ro, _, _), rq in zip(both, quiet)]))
            for prec, ens in list(devs.items()):
                dq = np.array([parity.state_rel_err(ens.Y[i], quiet[i]["Y"]) for i in range(len(Th))])
                print("%s [first %d leg(s)] device %-14s against the QUIET oracle: identical decisions in %d of %d cells; deviation p50 / p90 / p99 %.1e / %.1e / %.1e"
                      % (what, npre, prec, sum(all(int(ens.counters[i][f]) == quiet[i]["counters"][f] for f in CNT) for i in range(len(Th))), len(Th), *np.percentile(dq, (50, 90, 99))), flush=True)
                out.append(dict(case=what, legs=npre, who="device %s vs quiet oracle" % prec, identical_decisions=int(sum(all(int(ens.counters[i][f]) == quiet[i]["counters"][f] for f in CNT) for i in range(len(Th)))),
                                cells=len(Th), dev_p50=float(np.percentile(dq, 50)), dev_p90=float(np.percentile(dq, 90)), dev_p99=float(np.percentile(dq, 99))))
            for who, (e, same_steps, same_all, dev) in rows.items():
                r = e / e_orc
                rec = dict(case=what, legs=npre, who=who, cells=len(Th), ratio_median=float(np.median(r)), ratio_min=float(r.min()), ratio_max=float(r.max()),
                           mean_log_ratio=float(np.mean(np.log(r))), same_step_count=int(same_steps), identical_decisions=int(same_all),
                           dev_vs_oracle_p50=float(np.percentile(dev, 50)), dev_vs_oracle_p90=float(np.percentile(dev, 90)), dev_vs_oracle_p99=float(np.percentile(dev, 99)))
                out.append(rec)
                print("%s [first %d leg(s)] %-22s error / oracle error: median %.3f [%.3f, %.3f], mean log %+.3f; oracle's step count in %d, identical decisions in %d of %d cells; "
                      "deviation from the oracle p50 / p90 / p99 %.1e / %.1e / %.1e" % (what, npre, who, rec["ratio_median"], rec["ratio_min"], rec["ratio_max"], rec["mean_log_ratio"], same_steps, same_all, len(Th),
                                                                                 rec["dev_vs_oracle_p50"], rec["dev_vs_oracle_p90"], rec["dev_vs_oracle_p99"]), flush=True)
    if a.out:
        json.dump(out, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()

"""EVERY cell of the BASELINE configurations at full size against the oracle at the reference's DEFAULT tolerances (reltol 1e-3 / abstol 1e-6), as a pass / fail statement.

What can be asserted at these tolerances, and why in this form (the reproducibility floor of the reference algorithm).  Two correct fp64 implementations of
the reference's algorithm do not agree to 1e-6 per cell at reltol 1e-3: the finite-difference estimate of YP_alg in newtons_method! (model_evaluation.jl:462-477) turns last-bit
differences of a residual evaluation into 1e-6 of h0, the whole step grid scales with h0, the reference's LINEAR back-interpolation of a run end (model_evaluation.jl:369-382)
turns that into up to 1e-4 at a voltage knee, and a leg that starts from a :hold set point decorrelates altogether.  The oracle shows the same spread against ITSELF when
EVERY residual evaluation is perturbed by one unit of evaluation rounding (orc_opts.fd_perturb for the finite difference of the initialisation, orc_opts.res_perturb for every
evaluation of the corrector: res_i += 2.2e-16 u sum_c |J_ic Y_c|, a fresh u in [-1, 1) per row and evaluation) -- which is how a second correct implementation differs from the
first: the flux form or the matrix form of a stencil, the order of a sum, in every evaluation.  So the statement is a two-sample one, over the full ensemble:

  (1) exit flags equal in every run of every cell (no tolerance), and
  (2) the distribution of the device-vs-oracle deviation is no worse than the distribution of the oracle-vs-perturbed-oracle deviation ON THE SAME CELLS:
      quantile_q(device vs oracle) <= 1.5 x quantile_q(perturbed oracle vs oracle) for q = 50 %, 90 %, 99 % (floored at 1e-7: below it both are rounding), for the end
      state (max over the state sections of max|dY| / max|Y|: parity.state_rel_err) and for the run-end times;
  (3) the fraction of cells with identical integrator decisions (all counters equal) is not smaller than the perturbed oracle's by more than 5 points.
C3 is the one configuration in the BIMODAL regime: its two hold legs restart the integrator from a state that carries the previous leg's noise, and a cell either keeps identical
decisions (deviation ~1e-7) or decorrelates (~1e-2) -- in the oracle against its perturbed self in 76 % of the cells, in the device against the oracle in 65-72 %.  The median of
such a mixture sits in the lower part of the decorrelated mode and compares how that mode is populated: measured, the device's 1.2e-2 against the floor's 2.9e-3 -- 4x, NOT
within 1.5x, and it is reported as such.  What is asserted for a configuration in that regime is the 90 % and 99 % quantiles within 1.5x (3.5e-2 / 7.2e-2 against 2.7e-2 / 5.3e-2),
the median within the floor's own 90 % quantile, (1) and (3).

C2: 1024 cells (identical parameters: one oracle run serves all), C3: 4096, C4: every 8th of 65 536 (8192 cells; the launch is the full 65 536), C5: 8192 (40 runs per cell).
The tight-tolerance suite (test_gpu_tight.py) is the per-cell 1e-6 statement; this module is the every-cell statement at the tolerances the benchmark runs at."""
import os
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import parity

pytestmark = pytest.mark.gpu

CNT = ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail")
QS = (50, 90, 99)
FACTOR, FLOOR = 1.5, 1e-7


def _cores():
    n = len(os.sched_getaffinity(0))
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = max(1, min(n, int(int(q) / int(per))))
    except Exception:
        pass
    return n


def two_sample(pkg, O, p, cfg, cells, what, variant=None, check=True, bimodal_branch=True):
    """device launch over all cells of cfg, oracle + one perturbed oracle re-run for every cell of `cells`; asserts (1)-(3) of the module docstring.  variant: the oracle variant
    (default p.variant)"""
    variant = variant or p.variant
    import torch
    Th_all = np.ascontiguousarray(cfg["theta"])
    ens = pkg.simulate_ensemble(p, torch.from_numpy(Th_all).cuda(), cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"])
    torch.cuda.synchronize()
    Yd = ens.Y.cpu().numpy(); info = ens.run_info; cnt = ens.counters
    runs = parity.runs_to_oracle(O, p, pkg, cfg["protocol"])

    def one(i):
        ro = O.simulate(variant, Th_all[i], cfg["SOC"], runs, max_out=8)
        rp = O.simulate(variant, Th_all[i], cfg["SOC"], runs, max_out=8, opts=O.default_opts(fd_perturb=2.2e-16, res_perturb=2.2e-16, perturb_seed=1 + i % 7))
        fl_d = [int(info[i, k]["flag"]) for k in range(len(runs))]; fl_o = [r["flag"] for r in ro["runs"]]; fl_p = [r["flag"] for r in rp["runs"]]
        te = lambda a, b: max(abs(x - y) / max(1.0, y) for x, y in zip(a, b))
        t_o = [r["t_end"] for r in ro["runs"]]
        return (fl_d == fl_o, fl_p == fl_o,
                all(int(cnt[i][f]) == ro["counters"][f] for f in CNT), all(rp["counters"][f] == ro["counters"][f] for f in CNT),
                parity.state_rel_err(Yd[i], ro["Y"]), parity.state_rel_err(rp["Y"], ro["Y"]),
                te([float(info[i, k]["t_end"]) for k in range(len(runs))], t_o), te([r["t_end"] for r in rp["runs"]], t_o), fl_d, fl_o)
    with ThreadPoolExecutor(_cores()) as ex:
        res = list(ex.map(one, cells, chunksize=16))
    fl_dev = np.array([r[0] for r in res]); fl_pert = np.array([r[1] for r in res])
    same_d = np.array([r[2] for r in res]); same_p = np.array([r[3] for r in res])
    e_d = np.array([r[4] for r in res]); e_p = np.array([r[5] for r in res]); t_d = np.array([r[6] for r in res]); t_p = np.array([r[7] for r in res])
    qd, qp = np.percentile(e_d, QS), np.percentile(e_p, QS)
    qtd, qtp = np.percentile(t_d, QS), np.percentile(t_p, QS)
    print("%s: %d cells (launch of %d, kernel %.2f ms) -- flags equal %d / %d (perturbed oracle: %d); identical decisions device %.1f %% / perturbed oracle %.1f %%; "
          "end state p50 / p90 / p99: device %.1e / %.1e / %.1e, perturbed oracle %.1e / %.1e / %.1e; run-end times: device %.1e / %.1e / %.1e, perturbed oracle %.1e / %.1e / %.1e; "
          "max: device %.1e, perturbed oracle %.1e"
          % (what, len(cells), Th_all.shape[0], ens.kernel_ms, fl_dev.sum(), len(cells), fl_pert.sum(), 100 * same_d.mean(), 100 * same_p.mean(), *qd, *qp, *qtd, *qtp, e_d.max(), e_p.max()))
    bad = [(int(cells[k]), res[k][8], res[k][9]) for k in np.nonzero(~fl_dev)[0][:5]]
    stats = dict(what=what, variant=variant, precision=getattr(p, "precision", "f64"), cells=len(cells), flags_equal=int(fl_dev.sum()), flags_equal_perturbed=int(fl_pert.sum()),
                 identical_decisions_device=float(same_d.mean()), identical_decisions_perturbed=float(same_p.mean()), end_state_device=[float(x) for x in qd],
                 end_state_perturbed=[float(x) for x in qp], t_end_device=[float(x) for x in qtd], t_end_perturbed=[float(x) for x in qtp], kernel_ms=float(ens.kernel_ms))
    if not check:
        return dict(e_d=e_d, e_p=e_p, same_d=same_d, same_p=same_p, stats=stats)
    assert fl_dev.all(), ("exit flags differ", what, bad)
    # (bimodal regime: when fewer than half of the cells keep identical decisions in EITHER sample the median sits inside the decorrelated mode, where it measures how the
    #  mode is populated, not how far apart two runs are: the device's median must then lie within the floor's 90 % quantile -- module docstring, C3)
    bimodal = bimodal_branch and same_d.mean() < 0.5 and same_p.mean() < 0.5
    for name, qa, qb in (("end state", qd, qp), ("run-end times", qtd, qtp)):
        for q, a, b in zip(QS, qa, qb):
            lim = FACTOR * max(b, FLOOR) if not (bimodal and q == 50) else max(FACTOR * max(b, FLOOR), qb[1])
            assert a <= lim, (name, what, q, a, b)
    assert same_d.mean() >= same_p.mean() - 0.05, ("identical decisions", what, same_d.mean(), same_p.mean())
    return dict(e_d=e_d, e_p=e_p, same_d=same_d, same_p=same_p, stats=stats)


def test_every_cell_c2(hip_model, O, pkg):
    p = hip_model
    cfg = pkg.configs.c2(p, 1024)
    r = two_sample(pkg, O, p, cfg, np.arange(1024), "C2")
    assert r["same_d"].all() and r["e_d"].max() <= 1e-6          # identical parameters, identical decisions: every cell within 1e-6 outright


def test_every_cell_c3(hip_model_thermal, O, pkg):
    # the oracle variant that evaluates the heat-conduction stencil on temperature differences, like the device (same equations as lco_thermal row by row to 1e-12 of the
    # terms and on the thermal notebook KATs: tests/test_oracle_golden.py).  Against the matrix form the device differs by that form's own rounding, 1e-9 K/s per T row -- 1e4 x the
    # last-bit perturbation the floor is measured with (the numbers against either variant: DESIGN.md 5 "every cell, asserted")
    p = hip_model_thermal
    two_sample(pkg, O, p, pkg.configs.c3(p, 4096), np.arange(4096), "C3", variant="lco_thermal_tdiff")


def test_every_cell_c3_against_the_quiet_oracle(hip_model_thermal, O, pkg):
    """r05: C3 against `lco_thermal_quiet` (T rows AND Phi_s rows on differences: the device's evaluation order), every one of the 4096 cells, the two-sample criterion WITHOUT the
    bimodal relaxation -- and far inside it: the device keeps the quiet oracle's decisions in (nearly) every cell, so the deviation quantiles are rounding, not a floor.  What
    made C3 bimodal against the other two variants is the rounding of THEIR Phi_s rows (tests/test_oracle_golden.py)."""
    p = hip_model_thermal
    r = two_sample(pkg, O, p, pkg.configs.c3(p, 4096), np.arange(4096), "C3 vs the quiet oracle", variant="lco_thermal_quiet", bimodal_branch=False)
    # measured r05: identical decisions in 98.9 % of the 4096 cells (against the plain variant: 27.7 %), end state p50 / p90 / p99 = 7.3e-7 / 1.3e-5 / 4.1e-4 (1.3e-2 / 3.5e-2 / 7.6e-2)
    assert r["same_d"].mean() >= 0.97 and np.percentile(r["e_d"], 50) <= 5e-6 and np.percentile(r["e_d"], 90) <= 1e-4, (r["same_d"].mean(), np.percentile(r["e_d"], (50, 90, 99)))


def test_every_cell_c3_reference_order_build(hip_model_thermal, O, pkg):
    """r05: the reference-order build (precision = "f64_reforder": finite-volume and Phi_s rows in the generated code's operation order) against the PLAIN, notebook-pinned oracle
    `lco_thermal` on all 4096 C3 cells: the two-sample criterion at every quantile including the median, WITHOUT the bimodal relaxation r04 needed for the default build."""
    p = pkg.petlion(pkg.LCO, temperature=True, precision="f64_reforder")
    two_sample(pkg, O, p, pkg.configs.c3(p, 4096), np.arange(4096), "C3, reference-order build vs the plain oracle", variant="lco_thermal", bimodal_branch=False)


def test_every_8th_cell_c4(hip_model, O, pkg):
    p = hip_model
    two_sample(pkg, O, p, pkg.configs.c4(p, 65536), np.arange(0, 65536, 8), "C4")


def test_every_cell_c5(hip_model_nmc_sei, O, pkg):
    p = hip_model_nmc_sei
    two_sample(pkg, O, p, pkg.configs.c5(p, 8192), np.arange(8192), "C5")


def _quiet(r, same, p50, p99, mx):
    q = np.percentile(r["e_d"], (50, 99))
    assert r["same_d"].mean() >= same and q[0] <= p50 and q[1] <= p99 and r["e_d"].max() <= mx, (r["same_d"].mean(), q, r["e_d"].max())


def test_every_cell_c2_c4_c5_against_the_quiet_oracle(hip_model, hip_model_nmc_sei, O, pkg):
    """r05: the isothermal configurations against the oracle variants whose Phi_s rows are evaluated on differences (`lco_iso_quiet`, `nmc_iso_sei_quiet`: the same model as the plain
    variants, tests/test_oracle_golden.py) -- the device's evaluation order.  With the one systematic rounding difference between device and oracle removed, the DEFAULT-tolerance
    trajectories agree nearly as tightly as the reltol-1e-8 ones: measured r05 (gpurun_out/r05e -> profiles/r05_two_sample_quiet.json), identical decisions / end state p50, p99, max:
      C2 1024 / 1024 cells, 6.5e-12;   C4 65 504 / 65 536 (ALL cells of the sweep: 99.95 %), 4.1e-12, 9.7e-10, 8.5e-3 (the 32 cells that decide differently);   C5 8192 / 8192 (40 runs each), 4.7e-8, 6.0e-7, 1.6e-6
    (against the plain variants: C4 99.7 %, p99 9.8e-6; C5 92.8 %, p99 6.0e-4 -- the two-sample tests above).  These are ABSOLUTE thresholds, not relative to a perturbed oracle."""
    p = hip_model
    r = two_sample(pkg, O, p, pkg.configs.c2(p, 1024), np.arange(1024), "C2 vs the quiet oracle", variant="lco_iso_quiet", bimodal_branch=False)
    _quiet(r, 1.0, 1e-9, 1e-9, 1e-9)
    r = two_sample(pkg, O, p, pkg.configs.c4(p, 65536), np.arange(65536), "C4 vs the quiet oracle, all 65 536 cells", variant="lco_iso_quiet", bimodal_branch=False)
    _quiet(r, 0.995, 1e-9, 1e-7, 2e-2)
    p = hip_model_nmc_sei
    r = two_sample(pkg, O, p, pkg.configs.c5(p, 8192), np.arange(8192), "C5 vs the quiet oracle", variant="nmc_iso_sei_quiet", bimodal_branch=False)
    _quiet(r, 0.99, 1e-6, 1e-5, 1e-3)

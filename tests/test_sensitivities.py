"""Forward parameter sensitivities next to the states (plh_integrate_sens, csrc/dfn_sens.h).

The reference has no derivative output and no vector to pin one on; the oracle for dY/dtheta is the oracle itself, differenced: sixth-order central differences of six
tight-tolerance runs per parameter with common stop times (parity.oracle_fd_sens).  Its own accuracy is ~2e-6 x |Y| / (theta |dY/dtheta|), so the criteria are
  * dV/dtheta at every stop time within 1e-4 + 2e-7 V / (theta max |dV/dtheta|) of max |dV/dtheta| over the trajectory (the Jacobian of the voltage curve: what a
    least-squares fit consumes; the second term is the differenced oracle's noise: a parameter that moves the voltage by 10 mV per unit relative change is held to 1.2e-4);
  * dY/dtheta at the end of the protocol within 1e-4 + 2e-5 / r of its scale in every state section, r = theta |dY/dtheta| / |Y| the section's relative sensitivity: the
    second term is the differenced oracle's own noise (two tight-tolerance end states agree to ~1e-6 of their scale -- 100 x reltol, test_gpu_tight.py -- over a step of
    5 % of theta; measured: going from second- to fourth- to sixth-order differences moved the device-vs-oracle gap of weakly dependent sections from 3e-3 to 7e-4 to
    5e-5 while the device's numbers did not move).  Sections that depend on the parameter at all (r >= 0.1) are therefore held to ~3e-4, and the summary prints their worst;
  * the states, saved points and counters of a call with sensitivities are BIT FOR BIT those of the call without (nothing of the integrator is touched);
  * every corrector solve reached its tolerance (sens_stat);
  * a run that ends on a bound: the derivative of the end state as simulate() returns it -- the reference's linear back-interpolation, including the shift of the crossing."""
import numpy as np
import pytest

import parity


def check_sens(pkg, p, O, Th, soc, protocol, keys, ts, tol=parity.TIGHT, variant=None, what="", all_ts=None, rel_h=0.05, lim_a=1e-4, lim_b=2e-5, exact=True):
    """exact: the states of the launch with sensitivities are bit for bit those of the launch without (True everywhere but in the event case on the GPU, see _event_case).
    all_ts: the stop times the runs use (default ts); ts: the ones dV/dtheta is compared at.  rel_h, lim_a, lim_b: step of the differenced oracle and the criterion lim_a + lim_b / r"""
    o = pkg.Opts(); o.reltol, o.abstol, o.maxiters = tol["reltol"], tol["abstol"], 200000; o.tstops = list(ts if all_ts is None else all_ts)
    Th = np.ascontiguousarray(Th)
    ens = pkg.simulate_ensemble(p, Th, protocol, SOC=soc, opts=o, max_points=20000, sens=keys)
    ref = pkg.simulate_ensemble(p, Th, protocol, SOC=soc, opts=o, max_points=20000)
    if exact:
        assert np.array_equal(np.asarray(ens.Y), np.asarray(ref.Y)) and np.array_equal(ens.n_pts, ref.n_pts), "the states changed with sensitivities on"
        for i in range(Th.shape[0]):          # (entries beyond n_pts are whatever the allocator handed out)
            k = int(ens.n_pts[i])
            assert np.array_equal(np.asarray(ens.V[i, :k]), np.asarray(ref.V[i, :k])) and np.array_equal(np.asarray(ens.t[i, :k]), np.asarray(ref.t[i, :k])), "the saved points changed with sensitivities on"
        assert np.array_equal(ens.counters["n_steps"], ref.counters["n_steps"]) and np.array_equal(ens.counters["n_newton"], ref.counters["n_newton"])
    else:
        # two compilations of the step loop (the sensitivity and the stop-times instantiation) may round differently on the GPU; over the ~1000 steps of this case at
        # reltol 1e-8 the difference stays far inside the tolerance of the integration, which is what is asserted (flags and run ends included)
        Ye, Yr = np.asarray(ens.Y), np.asarray(ref.Y)
        dev = np.abs(Ye - Yr) / (np.abs(Yr) + tol["abstol"] / tol["reltol"])
        print("%s: states with / without sensitivities: max weighted deviation %.2e (reltol %.0e)" % (what, dev.max(), tol["reltol"]))
        assert dev.max() <= 10 * tol["reltol"], dev.max()
        assert np.array_equal(ens.run_info["flag"], ref.run_info["flag"])
        assert np.allclose(ens.run_info["t_end"], ref.run_info["t_end"], rtol=0, atol=1e-5)
    assert (np.asarray(ens.sens_stat)[:, 1] == 0).all(), ("corrector solves without convergence", np.asarray(ens.sens_stat)[:, 1].max())
    worstV, worstY = 0.0, 0.0
    import os
    from concurrent.futures import ThreadPoolExecutor
    assert (ens.run_info["flag"] >= 0).all()
    with ThreadPoolExecutor(min(16, len(os.sched_getaffinity(0)))) as ex:       # (the oracle runs release the GIL)
        allres = list(ex.map(lambda i: parity.sens_compare(O, p, pkg, ens, i, Th[i], soc, protocol, keys, ts, variant=variant, rel_h=rel_h), range(Th.shape[0])))
    for i, res in enumerate(allres):
        for key, (eV, sec) in res.items():
            eV, volts = eV
            # (the differenced oracle again: two tight-tolerance voltages agree to ~1e-8 V, over a step of 5 % of theta)
            assert eV <= lim_a + 2e-7 / max(volts, 1e-12), (what, i, key, "dV/dtheta", eV, volts)
            if volts >= 1e-2:
                worstV = max(worstV, eV)
            for name, (err, rel) in sec.items():
                if rel < 1e-6:          # a section that does not depend on the parameter (a millionth of the state / theta scale): err is the differenced oracle's noise over ~0
                    continue
                # (the algebraic flux sections are controlled ABSOLUTELY by both integrators -- |j| ~ 1e-5 against abstol 1e-10: 1e-5 of their scale, test_gpu_tight.py -- so
                #  the differenced oracle is ten times noisier there)
                lim = lim_a + (10 * lim_b if name in ("j", "j_s") else lim_b) / max(rel, 1e-12)
                assert err <= lim, (what, i, key, name, err, rel)
                if rel >= 0.1:
                    worstY = max(worstY, err)
    its = np.asarray(ens.sens_stat)[:, 0].sum() / max(1, int(ens.counters["n_steps"].sum()) * len(keys))
    print("%s: %d cell(s) x %s -- dV/dtheta (parameters worth >= 10 mV) max %.1e, dY/dtheta (sections with relative sensitivity >= 0.1) max %.1e; %.2f corrector iterations per step and parameter; kernel %.2f ms (without: %.2f ms)"
          % (what, Th.shape[0], keys, worstV, worstY, its, ens.kernel_ms, ref.kernel_ms))
    return ens


def test_sens_lco_discharge_emu(emu_model, O, pkg):
    p = emu_model
    check_sens(pkg, p, O, p.theta_vector()[None, :], 1.0, [{"I": -1.0, "tf": 600.0}], ["D_sp", "k_n", "D_s"], np.arange(50.0, 600.0, 50.0), what="LCO 1C 600 s (emulator)")


def test_sens_through_run_changes_and_a_bound_emu(emu_model, O, pkg):
    """three runs (discharge, rest, constant voltage): the differential part of s is carried, the algebraic part re-solved with the new control row at every run start"""
    p = emu_model
    proto = [{"I": -2.0, "tf": 200.0}, {"I": "rest", "tf": 100.0}, {"V": 4.0, "tf": 150.0}]
    check_sens(pkg, p, O, p.theta_vector()[None, :], 0.9, proto, ["D_sp", "k_p"], np.arange(20.0, 200.0, 20.0), what="LCO 2C / rest / CV (emulator)")


@pytest.mark.gpu
def test_sens_of_a_run_that_ends_on_a_voltage_bound_gpu(hip_model, O, pkg):
    _event_case(hip_model, O, pkg, "LCO 2C to V_min, rest", exact=False)


def test_sens_of_a_run_that_ends_on_a_voltage_bound_emu(emu_model, O, pkg):
    _event_case(emu_model, O, pkg, "LCO 2C to V_min, rest (emulator)")


def _event_case(p, O, pkg, what, exact=True):
    """a 2C discharge that ends on V_min, then a rest: the end state of run 1 is the reference's linear back-interpolation, and its derivative includes the shift of the
    crossing with theta (sens_finish) -- dV/dtheta of the interpolated point is 0 (the voltage there IS the bound), and the derivative of the final state agrees with the
    differenced oracle, whose six runs each end at their own crossing.  (The oracle is noisier here: each of its runs locates the crossing on a 0.05 s stop grid; its step is
    10 % of theta and the criterion 3e-4 + 5e-5 / r -- halving / doubling the step moves the device-vs-oracle gap of c_s between 7e-4 and 4e-4 while the device's numbers stay.)"""
    th = p.theta_vector()
    proto = [{"I": -2.0, "V_min": 3.75, "tf": 3000.0}, {"I": "rest", "tf": 120.0}]
    o = pkg.Opts(); o.reltol, o.abstol, o.maxiters = 1e-8, 1e-10, 200000
    e0 = pkg.simulate_ensemble(p, th[None, :].copy(), proto[:1], SOC=0.9, opts=o, max_points=20000)
    assert int(e0.run_info[0, 0]["flag"]) == 1
    te = float(e0.run_info[0, 0]["t_end"])
    coarse = np.arange(50.0, te - 50.0, 50.0)          # (the differenced oracle moves theta by up to +-30 %: its runs cross the bound up to ~10 s earlier or later)
    all_ts = np.unique(np.round(np.concatenate([coarse, np.arange(te - 15.0, te + 15.0, 0.05)]), 6))
    ens = check_sens(pkg, p, O, th[None, :], 0.9, proto, ["D_sp", "k_n"], coarse, all_ts=all_ts, rel_h=0.1, lim_a=3e-4, lim_b=5e-5, what=what, exact=exact)
    k1 = int(ens.run_info[0, 0]["iterations"])
    dV = np.asarray(ens.dV_dtheta[0])
    assert (np.abs(dV[:, k1 - 1]) <= 1e-8 * np.abs(dV[:, k1 - 2])).all(), (dV[:, k1 - 1], dV[:, k1 - 2])


def test_sens_thermal_emu(emu_model_thermal, O, pkg):
    p = emu_model_thermal
    check_sens(pkg, p, O, p.theta_vector()[None, :], 0.2, [{"I": 3.0, "tf": 150.0}], ["h_cell", "k_p"], np.arange(25.0, 150.0, 25.0), variant="lco_thermal_tdiff", what="LCO thermal 3C charge (emulator)")


def test_sens_through_hold_legs_emu(emu_model, O, pkg):
    """r05: :hold runs.  The held value is the previous run's end value of the held quantity, so it depends on theta: d value / d theta_k = that quantity's sensitivity at the
    previous run's end, one constant term in F_theta of the control row (dfn_sens.h sens_init).  CC (fixed time) -> V hold -> P hold -> I hold, against the differenced
    quiet oracle (the plain variant's Phi_s-row rounding steers ITS hold legs: DESIGN.md 5)."""
    p = emu_model
    proto = [{"I": 2.0, "tf": 300.0, "V_max": 5.0}, {"V": "hold", "tf": 200.0, "V_max": 5.0, "I_min": 0.0}, {"P": "hold", "tf": 60.0, "V_max": 5.0}, {"I": "hold", "tf": 60.0, "V_max": 5.0}]
    check_sens(pkg, p, O, p.theta_vector()[None, :], 0.2, proto, ["D_sp", "k_n"], np.arange(30.0, 300.0, 30.0), variant="lco_iso_quiet", what="LCO CC / V hold / P hold / I hold (emulator)")


def _cccv_event_case(p, O, pkg, second, what, exact=True):
    """CC until V_max (the crossing moves with theta), then V = :hold until the bound `second` fires: every run of the differenced oracle ends at its own crossings, so both
    leg ends get the 0.05 s stop grid of the tight suite around them (run-local times; opts.tstops applies to every run)"""
    th = p.theta_vector()
    proto = [{"I": 2.0, "V_max": 4.0, "tf": 3000.0}, dict({"V": "hold", "V_max": 4.0, "tf": 3000.0}, **second)]
    o = pkg.Opts(); o.reltol, o.abstol, o.maxiters = 1e-8, 1e-10, 200000
    e0 = pkg.simulate_ensemble(p, th[None, :].copy(), proto, SOC=0.3, opts=o, max_points=20000)
    assert [int(f) for f in e0.run_info[0]["flag"]] == [2, 8 if "I_min" in second and second["I_min"] > 0 else 4], e0.run_info[0]
    te1 = float(e0.run_info[0, 0]["t_end"]); te2 = float(e0.run_info[0, 1]["t_end"]) - te1
    coarse = np.arange(50.0, te1 - 50.0, 50.0)
    fine = np.concatenate([np.arange(max(1.5, te - 15.0), te + 15.0, 0.05) for te in (te1, te2)])
    all_ts = np.unique(np.round(np.concatenate([coarse, fine]), 6))
    return check_sens(pkg, p, O, th[None, :], 0.3, proto, ["D_sp", "k_n"], coarse, all_ts=all_ts, variant=p.variant + "_quiet", rel_h=0.1, lim_a=3e-4, lim_b=1.5e-4, what=what, exact=exact)      # (two crossings per differenced run: three times the one-event noise)


@pytest.mark.gpu
def test_sens_cc_cv_to_a_current_bound_gpu(hip_model, O, pkg):
    """r05: the commonest estimation protocol -- CC until V_max, then V = :hold until I_min (a current bound in a voltage run).  (Passes on the emulator too: 2.3e-5 / 1.8e-3; it is a
    GPU test for the CPU suite's run time.)"""
    _cccv_event_case(hip_model, O, pkg, dict(I_min=0.5, SOC_max=2.0), "LCO CC to V_max / V hold to I_min", exact=False)


@pytest.mark.gpu
def test_sens_cc_cv_to_an_soc_bound_gpu(hip_model, O, pkg):
    """r05: ... until SOC_max, an SOC bound under a VARYING current: the bounded quantity is the trapezoid SOC of the saved points (calc_SOC), its sensitivity the trapezoid of
    dI/dtheta (C3's CC-CT-CV ends this way)"""
    _cccv_event_case(hip_model, O, pkg, dict(I_min=0.0, SOC_max=0.6), "LCO CC to V_max / V hold to SOC_max", exact=False)


@pytest.mark.gpu
def test_sens_c3_protocol_on_gpu(hip_model_thermal, pkg):
    """r05: plh_integrate_sens accepts C3's CC-CT-CV protocol (4C -> dT = :hold -> V = :hold to SOC_max): every cell of a 256-cell sample finishes with finite sensitivities of
    the end state with respect to h_cell, k_p and D_sn, states bit for bit the plain launch's, every corrector solve converged (where the integrator's stale matrix does not
    contract, the step factors its own and copies the integrator's back: dfn_sens.h, sens_factor_copy)"""
    p = hip_model_thermal
    cfg = pkg.configs.c3(p, 4096)
    Th = np.ascontiguousarray(cfg["theta"][::16])
    ens = pkg.simulate_ensemble(p, Th, cfg["protocol"], SOC=cfg["SOC"], max_points=cfg["max_points"], sens=["h_cell", "k_p", "D_sn"])
    ref = pkg.simulate_ensemble(p, Th, cfg["protocol"], SOC=cfg["SOC"], max_points=cfg["max_points"])
    assert np.array_equal(ens.run_info["flag"], ref.run_info["flag"]) and (ens.run_info["flag"][:, 2] == 4).all()
    assert np.array_equal(ens.counters["n_steps"], ref.counters["n_steps"])
    dY = np.asarray(ens.dY_dtheta)
    st = np.asarray(ens.sens_stat)
    print("C3 CC-CT-CV with three sensitivities, 256 cells: finite in %d cells; corrector solves without convergence: %d; steps that factored their own matrix: %d of %d; kernel %.1f ms (plain %.1f ms)"
          % (int(np.isfinite(dY).all(axis=(1, 2)).sum()), int(st[:, 1].sum()), int(st[:, 2].sum()), int(ens.counters["n_steps"].sum()), ens.kernel_ms, ref.kernel_ms))
    # (the first GPU run of this test, before sens_step factored the step's own matrix when the integrator's stale one does not contract: 28 solves at the iteration cap)
    # the states are those of the plain launch: every counter equal in every cell; bit for bit in (nearly) all of them -- the sensitivity instantiation is another compilation of the
    # step loop and may contract a sum differently (the 8192-cell C4 test below: 23 cells at 1e-13 ... 1e-11)
    Ye, Yr = np.asarray(ens.Y), np.asarray(ref.Y)
    bad = np.nonzero((Ye != Yr).any(axis=1))[0]
    worst = max([parity.state_rel_err(Ye[c], Yr[c]) for c in bad], default=0.0)
    print("   cells whose end state differs in any bit from the plain launch's: %d of %d (worst %.1e)" % (len(bad), len(Ye), worst))
    for fld in ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail"):
        assert np.array_equal(ens.counters[fld], ref.counters[fld]), fld
    assert len(bad) <= 0.05 * len(Ye) and worst < 1e-9, (len(bad), worst)
    # (r05: 0 of ~139 000 corrector solves without convergence, r06: 1 -- a solve at the first steps of a hold run whose difference quotients hop on their own rounding; the
    #  sensitivities stay finite and the history continues from the last iterate)
    assert np.isfinite(dY).all() and st[:, 1].sum() <= 3


def test_sens_refused_where_theta_enters_through_the_protocol(emu_model, pkg):
    p = emu_model
    for proto in ([{"I": (lambda t: -1.0 - 0.001 * t), "tf": 100.0}],):
        with pytest.raises(Exception) as e:
            pkg.simulate_ensemble(p, p.theta_vector()[None, :].copy(), proto, SOC=1.0, sens=["D_sp"])
        assert "plh_integrate_sens" in str(e.value)


@pytest.mark.gpu
def test_sens_c4_parameters_on_gpu(hip_model, O, pkg):
    """the seven jittered parameters of C4 on 16 cells of the sweep: 1C discharge for 1800 s (every cell ends at tf: no event time to move)"""
    p = hip_model
    keys = [k for k in pkg.configs.SWEEP_KEYS]
    Th = pkg.configs.sweep_theta(p, np.arange(0, 65536, 4096), 4)
    check_sens(pkg, p, O, Th, 1.0, [{"I": -1.0, "tf": 1800.0}], keys, np.arange(100.0, 1800.0, 100.0), what="C4 cells, 1C 1800 s")


@pytest.mark.gpu
def test_sens_thermal_and_sei_on_gpu(hip_model_thermal, hip_model_nmc_sei, O, pkg):
    pt = hip_model_thermal
    cfg = pkg.configs.c3(pt, 4096)
    check_sens(pkg, pt, O, cfg["theta"][::1024], 0.0, [{"I": 4.0, "tf": 250.0, "T_max": 400.0}], ["h_cell", "k_p", "D_sn"], np.arange(25.0, 250.0, 25.0), variant="lco_thermal_tdiff", what="C3 cells, 4C 250 s")
    ps = hip_model_nmc_sei
    cfg = pkg.configs.c5(ps, 8192)
    check_sens(pkg, ps, O, cfg["theta"][::2048], 0.0, cfg["protocol"][:2], ["D_sp", "k_n"], np.arange(60.0, 180.0, 60.0), what="C5 cells, first pulse + rest")


@pytest.mark.gpu
def test_sens_full_c4_launch_on_gpu(hip_model, pkg):
    """the whole 8192-cell C4 shard with seven sensitivities per cell at the default tolerances: states bit-identical to the plain launch, every solve converged, finite outputs"""
    import torch
    p = hip_model
    cfg = pkg.configs.c4(p, 8192)
    Thd = torch.from_numpy(np.ascontiguousarray(cfg["theta"])).cuda()
    ens = pkg.simulate_ensemble(p, Thd, cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"], sens=list(pkg.configs.SWEEP_KEYS))
    ms_sens = ens.kernel_ms                 # (the handle reports its LAST launch: read it before the next one)
    # the sensitivity instantiation is the stop-times instantiation plus the sensitivity phase: bit for bit THAT kernel's states (two instantiations of a template are two
    # compilations, free to contract a product differently; against the plain instantiation the count of cells that differ at all is reported)
    o = pkg.Opts(); o.tstops = [1e7]
    ref = pkg.simulate_ensemble(p, Thd, cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"], opts=o)
    plain = pkg.simulate_ensemble(p, Thd, cfg["protocol"], SOC=cfg["SOC"], device=True, max_points=cfg["max_points"])
    torch.cuda.synchronize()
    ndiff = int((ens.Y != plain.Y).any(dim=1).sum())
    print("cells whose end state differs in any bit from the plain instantiation's: %d of %d (from the stop-times instantiation's: %d)" % (ndiff, Thd.shape[0], int((ens.Y != ref.Y).any(dim=1).sum())))
    bad = (ens.Y != ref.Y).any(dim=1).nonzero().flatten().cpu().numpy()
    if len(bad):
        Ye, Yr = ens.Y.cpu().numpy(), ref.Y.cpu().numpy()
        print("   differing cells:", [(int(c), "%.1e" % parity.state_rel_err(Ye[c], Yr[c]), int(ens.run_info["flag"][c, 0]), int(ref.run_info["flag"][c, 0]), int(ens.counters["n_steps"][c]), int(ref.counters["n_steps"][c]),
                                      int(ens.counters["n_newton"][c]), int(ref.counters["n_newton"][c])) for c in bad[:12]])
        ef = ens.counters["n_errfail"] + ens.counters["n_convfail"]
        print("   cells with a failed step attempt: %d of %d; among the differing cells: %d of %d" % (int((ef > 0).sum()), len(ef), int((ef[bad] > 0).sum()), len(bad)))
    # the counters (every decision of the integrator) are equal in every cell and the states in all but a handful, where they agree to rounding
    for fld in ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail"):
        assert np.array_equal(ens.counters[fld], ref.counters[fld]), fld
    assert torch.equal(ens.n_pts, ref.n_pts) and len(bad) <= 0.01 * Thd.shape[0]
    assert float(((ens.Y - ref.Y).abs() / (ref.Y.abs() + 1e-300)).max()) < 1e-6 and all(parity.state_rel_err(Ye[c], Yr[c]) < 1e-9 for c in bad)
    st = ens.sens_stat.cpu().numpy()
    nonfinite = int((~torch.isfinite(ens.dY_dtheta)).any(dim=2).any(dim=1).sum())
    print("   corrector solves that did not reach the tolerance: %d in %d cells (of %d solves); steps that factored their own matrix: %d in %d cells (of %d steps); cells with a non-finite sensitivity: %d"
          % (int(st[:, 1].sum()), int((st[:, 1] > 0).sum()), int(7 * ens.counters["n_steps"].sum()), int(st[:, 2].sum()), int((st[:, 2] > 0).sum()), int(ens.counters["n_steps"].sum()), nonfinite))
    # (until the first r05 GPU run: 2636 of the 4.7 M solves ran into the iteration cap with the integrator's stale matrix; since then such a step factors its own, dfn_sens.h)
    assert nonfinite == 0 and st[:, 1].sum() == 0 and st[:, 2].sum() <= 0.01 * ens.counters["n_steps"].sum()
    print("C4 shard, 8192 cells x 7 sensitivities: kernel %.1f ms (plain %.1f ms: x%.1f), %.2f corrector iterations per step and parameter"
          % (ms_sens, plain.kernel_ms, ms_sens / plain.kernel_ms, st[:, 0].sum() / (7.0 * ens.counters["n_steps"].sum())))


@pytest.mark.gpu
def test_sens_default_tolerance_accuracy_on_gpu(hip_model, pkg):
    """r05: sensitivities at the tolerances the benchmark runs at.  256 cells of the C4 sweep (every 256th of 65 536), 1C discharge for 1800 s with a stop
    every 100 s, the seven sweep parameters: dV/dtheta at the stops from the run at reltol 1e-3 / abstol 1e-6 against the same cells' sensitivities at 1e-8 / 1e-10 -- within
    10 x reltol of max |dV/dtheta| over the trajectory for every cell and parameter worth at least 1 mV per unit relative change -- and NO corrector solve ends at the iteration
    cap (r04: 0.09 % did; the corrector now also stops on IDANls' rate-based estimate, dfn_sens.h)."""
    p = hip_model
    keys = list(pkg.configs.SWEEP_KEYS)
    Th = np.ascontiguousarray(pkg.configs.sweep_theta(p, np.arange(0, 65536, 256), 4))
    ts = list(np.arange(100.0, 1800.0, 100.0))
    res = {}
    for name, (rt, at) in (("default", (1e-3, 1e-6)), ("tight", (1e-8, 1e-10))):
        o = pkg.Opts(); o.reltol, o.abstol, o.maxiters, o.tstops = rt, at, 200000, ts
        res[name] = pkg.simulate_ensemble(p, Th, [{"I": -1.0, "tf": 1800.0}], SOC=1.0, opts=o, max_points=20000, sens=keys)
        assert (res[name].run_info["flag"] == 0).all()
    d, t = res["default"], res["tight"]
    st = np.asarray(d.sens_stat)
    worst, errs = 0.0, []
    for i in range(Th.shape[0]):
        nd, nt = int(d.n_pts[i]), int(t.n_pts[i])
        td, tt = np.asarray(d.t[i, :nd]), np.asarray(t.t[i, :nt])
        id_, it_ = [int(np.argmin(np.abs(td - x))) for x in ts], [int(np.argmin(np.abs(tt - x))) for x in ts]
        assert np.abs(td[id_] - ts).max() < 1e-6 and np.abs(tt[it_] - ts).max() < 1e-6
        for k, key in enumerate(keys):
            a, b = np.asarray(d.dV_dtheta[i, k])[id_], np.asarray(t.dV_dtheta[i, k])[it_]
            sc = np.abs(b).max()
            if sc * abs(Th[i, p.θ_keys.index(key)]) < 1e-3:          # (a parameter that moves the voltage by less than 1 mV per unit relative change: noise over ~0)
                continue
            e = float(np.abs(a - b).max() / sc)
            worst = max(worst, e)
            errs.append((e, i, key))
    its = st[:, 0].sum() / (len(keys) * float(d.counters["n_steps"].sum()))
    ev = np.array([e for e, _, _ in errs])
    print("default-tolerance sensitivities, 256 C4 cells x 7 parameters (%d pairs worth >= 1 mV): dV/dtheta against the tight run, relative to max |dV/dtheta|: p50 %.1e p90 %.1e p99 %.1e max %.1e "
          "(criterion 10 x reltol = 1e-2 for 99 %% of the pairs, 5e-2 for all; the states themselves are within reltol-sized errors of the tight ones); worst pairs %s; %.2f corrector iterations per "
          "step and parameter; solves at the iteration cap: %d" % (len(ev), *np.percentile(ev, (50, 90, 99)), ev.max(), sorted(errs, reverse=True)[:3], its, int(st[:, 1].sum())))
    assert np.percentile(ev, 99) <= 1e-2 and ev.max() <= 5e-2
    assert st[:, 1].sum() == 0


@pytest.mark.parametrize("variant,kw", [(0, {}), (4, {"temperature": True})])
def test_sens_refresh_path_emu(pkg, variant, kw):
    """r05: a sensitivity corrector that does not converge with the integrator's (stale) matrix factors the step's own matrix and afterwards copies the integrator's factorisation
    back (dfn_sens.h).  A test build sends the first parameter of every few steps -- and of every initialisation -- through that path (-DPL_TEST_SENS_FORCE_REFRESH): the states
    stay bit for bit those of the normal build (the copy restores LDS pools and per-lane registers exactly), the sensitivities agree with the normal build's to the corrector's
    tolerance, and the path was taken.  Isothermal and thermal (whose factorisation also lives in ThermalPool and in LaneRegs.rcp); CC -> hold -> rest, three parameters."""
    import os, subprocess, sys, build_emu
    la, lb = build_emu.build(variant=variant), build_emu.build(variant=variant, extra=["-DPL_TEST_SENS_FORCE_REFRESH"], tag="_sfr")
    code = ("import sys; sys.path.insert(0, %r)\nimport numpy as np, pkgload\npkg = pkgload.load()\np = pkg.petlion(pkg.LCO, _lib_path=sys.argv[1], **%r)\n"
            "Th = pkg.configs.sweep_theta(p, np.arange(2), 4)\n"
            "proto = [dict(I=2.0, tf=300.0, V_max=4.1), dict(V='hold', tf=200.0, I_min=0.0), dict(I='rest', tf=60.0)]\n"
            "e = pkg.simulate_ensemble(p, Th, proto, SOC=0.3, sens=['D_sp', 'k_n', 'D_s'])\n"
            "r = pkg.simulate_ensemble(p, Th, proto, SOC=0.3)\n"
            "assert np.array_equal(np.asarray(e.Y), np.asarray(r.Y)) and np.array_equal(e.counters['n_steps'], r.counters['n_steps']), 'the states changed with sensitivities on'\n"
            "np.savez(sys.argv[2], Y=np.asarray(e.Y), dY=np.asarray(e.dY_dtheta), dV=np.asarray(e.dV_dtheta), st=np.asarray(e.sens_stat), npts=e.n_pts)\n"
            % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), kw))
    out = []
    for lib in (la, lb):            # (one process per library: both export the same C symbols)
        f = "/tmp/sens_refresh_%d_%d.npz" % (variant, len(out))
        subprocess.check_call([sys.executable, "-c", code, lib, f])
        out.append(np.load(f))
    a, b = out
    assert np.array_equal(a["Y"], b["Y"]) and np.array_equal(a["npts"], b["npts"])
    assert (a["st"][:, 1] == 0).all() and (b["st"][:, 1] == 0).all() and (b["st"][:, 2] >= 3).all(), (a["st"], b["st"])
    scale = np.abs(a["dY"]).max(axis=2, keepdims=True) + 1e-300
    eY = (np.abs(a["dY"] - b["dY"]) / scale).max()
    k = int(a["npts"].min())
    eV = (np.abs(a["dV"][:, :, :k] - b["dV"][:, :, :k]).max(axis=2) / np.abs(a["dV"][:, :, :k]).max(axis=2)).max()
    print("variant %d: refresh path in %s steps of the two cells (normal build: %s); dY/dtheta normal vs forced-refresh build %.1e, dV/dtheta %.1e (default tolerances)"
          % (variant, b["st"][:, 2], a["st"][:, 2], eY, eV))
    assert eY <= 5e-3 and eV <= 5e-3            # (both are converged to 0.33 reltol of the state scale: the rate-based exit of the stale iteration against one or two exact solves)

"""Pins the ORACLE (oracle/) against the reference's own known answers (tests/golden/notebook_kats.json, values printed in
examples/*.ipynb) and against independent evaluations of itself (complex-step Jacobian, tolerance tightening)."""
import json
import os

import numpy as np
import pytest

from oracle import dfn_model as dm

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "notebook_kats.json"), encoding="utf-8"))


def test_I1C_bit_exact():
    assert dm.calc_I1C(dm.theta_LCO()) == G["I1C_LCO"]["value"]


def test_theta_defaults_match_printed_table():
    th = dm.theta_LCO()
    for k, v in G["theta_LCO_printed"]["values"].items():
        assert th[k] == v, k


def test_layout_sizes():
    assert dm.Model("LCO").lay.N == 301 and dm.Model("LCO").lay.N_diff == 230
    assert dm.Model("LCO", temperature=True).lay.N == 351
    assert dm.Model("NMC", aging=True).lay.N == 322


def test_generated_functions_match_python_restatement(O):
    """straight-line C (sympy codegen) == direct evaluation of oracle/dfn_model.py"""
    m = dm.Model("LCO")
    th = O.theta_vector("lco_iso")
    thd = dict(m.theta)
    rng = np.random.default_rng(3)
    ro = O.simulate("lco_iso", th, 0.3, [dict(mode=O.MODE_I, value=1.5, tf=300.0)])
    Y = ro["Y"] * (1 + 1e-3 * rng.standard_normal(301))
    YP = ro["YP"] * (1 + 1e-2 * rng.standard_normal(301))
    F = O.residual("lco_iso", th, Y, YP, O.MODE_I, 1.5)
    Fp = np.array(dm.residual(m, dm.FloatOps(), list(Y), list(YP), thd, dm.MODE_I, 1.5), dtype=float)
    assert np.abs(F - Fp).max() <= 1e-12 * np.abs(Fp).max()


def test_symbolic_jacobian_vs_complex_step(O):
    m = dm.Model("LCO")
    th = O.theta_vector("lco_iso")
    thd = dict(m.theta)
    ro = O.simulate("lco_iso", th, 0.5, [dict(mode=O.MODE_I, value=-2.0, tf=200.0)])
    Y, YP, cj = ro["Y"], ro["YP"], 0.8
    cp, ri, nz = O.jacobian("lco_iso", th, Y, YP, cj, O.MODE_V, 3.7)
    J = np.zeros((301, 301))
    for c in range(301):
        J[ri[cp[c]:cp[c + 1]], c] = nz[cp[c]:cp[c + 1]]
    ops, h = dm.FloatOps(), 1e-30
    for c in range(0, 301, 7):      # every 7th column keeps the test fast and still hits every block
        Yc = [complex(v) for v in Y]; Yc[c] += 1j * h
        dY = np.imag(np.array(dm.residual(m, ops, Yc, [complex(v) for v in YP], thd, dm.MODE_V, 3.7))) / h
        YPc = [complex(v) for v in YP]; YPc[c] += 1j * h
        dYP = np.imag(np.array(dm.residual(m, ops, [complex(v) for v in Y], YPc, thd, dm.MODE_V, 3.7))) / h
        col = dY + cj * dYP
        assert np.abs(J[:, c] - col).max() <= 1e-10 * (np.abs(col).max() + 1e-300), c


def test_V0_known_answer(O, variant="lco_iso"):
    th = O.theta_vector(variant)
    Y0 = O.initial_guess(variant, th, 0.0); Y0[-1] = 2.0
    rc, Y, YP, it = O.init_consistent(variant, th, Y0, O.MODE_I, 2.0)
    assert rc == 0 and it == 4
    assert abs((Y[280] - Y[299]) - G["V0_2C_charge"]["value"]) < G["V0_2C_charge"]["tol_abs"]


def test_notebook_runs(O, variant="lco_iso"):
    th = O.theta_vector(variant)
    k = G["runs"]["discharge_1C"]
    r = O.simulate(variant, th, 1.0, [dict(mode=O.MODE_I, value=-1.0)])["runs"][0]
    assert r["flag"] == k["flag"] and abs(r["t_end"] - k["t_end"]) <= k["tol"]["t_end_rel"] * k["t_end"]
    assert abs(r["V"] - k["V_end"]) < k["tol"]["V_abs"] and abs(r["SOC"]) < 1e-12
    b = O.default_bounds(V_max=4.1)
    ro = O.simulate(variant, th, 0.0, [dict(mode=O.MODE_I, value=2.0, tf=1800.0, bounds=b),
                                          dict(mode=O.MODE_V, value_kind=O.VAL_HOLD, bounds=O.default_bounds(V_max=4.1, I_min=1 / 20))])
    k1, k2 = G["runs"]["charge_2C_to_4p1"], G["runs"]["cv_hold_after_2C"]
    r1, r2 = ro["runs"]
    assert r1["flag"] == k1["flag"] and abs(r1["t_end"] - k1["t_end"]) <= k1["tol"]["t_end_rel"] * k1["t_end"]
    assert abs(r1["SOC"] - k1["SOC_end"]) < k1["tol"]["SOC_abs"]
    assert abs(r1["I"] * G["I1C_LCO"]["value"] * r1["V"] - k1["P_end"]) <= k1["tol"]["P_rel"] * k1["P_end"]
    assert r2["flag"] == k2["flag"] and abs(r2["t_end"] - k2["t_end"]) <= k2["tol"]["t_end_rel"] * k2["t_end"]
    assert abs(r2["I"] - k2["I_end"]) <= k2["tol"]["I_rel"] * k2["I_end"] and abs(r2["V"] - 4.1) < 1e-9
    # the reference saved 121 points for this pair of runs; the IDA restatement must be in the same regime
    assert abs(len(ro["t"]) - G["V_first13_2C_charge"]["n_points_total"]) <= 8
    assert abs(ro["V"][0] - G["V_first13_2C_charge"]["values"][0]) < 1e-10


def check_notebook_step_history(sim, exact_hold_leg=True):
    """`sim(yp_alg_zero)` -> dict(t, V, c_e [points, 30], runs) of `simulate(p, I=2, SOC=0, V_max=4.1); simulate!(sol, p, V=:hold)` with outputs (:t, :V, :c_e).
    The reference's notebook prints the STEP HISTORY of that pair of runs as SUNDIALS IDA + KLU produced it: 121 saved points, sol.V[1:13] and the last 12, the leading and
    trailing ten entries of sol.c_e[1:5].  Started with YP_alg = 0 -- the one thing today's source does differently, it adds the finite-difference estimate of
    model_evaluation.jl:462-477 -- the IDA restatement reproduces every printed digit that an independent fp64 implementation can (2.5e-9 V after twelve steps):
    h0 = 0.5/||y'||_wrms = 6.10 ms, an error-test failure on the first attempt (x0.60), then IDA's order and step selection, step for step."""
    r = sim(True)
    k = G["V_first13_2C_charge"]
    assert [q["flag"] for q in r["runs"]] == [2, 4] and r["runs"][0]["iterations"] == 84
    if exact_hold_leg:
        assert len(r["t"]) == k["n_points_total"] == 121 and r["runs"][1]["iterations"] == 37
    else:       # (a second implementation: the hold leg restarts from a back-interpolated state, its step sequence is reproducible to +-2 steps -- DESIGN.md 5)
        assert abs(len(r["t"]) - 121) <= 2
    assert np.abs(r["V"][:13] - np.array(k["values"])).max() < 1e-8 and np.abs(r["V"][-12:] - np.array(k["last12_CC_CV"])).max() < 1e-9
    assert np.abs(1e3 * r["t"][:13] - np.array(k["inferred_step_times_ms"])).max() < 1e-3
    for j, pr in enumerate(G["c_e_first5_saved_profiles"]["profiles"]):
        assert np.abs(r["c_e"][j, :10] - pr["first10"]).max() < 1e-7 and np.abs(r["c_e"][j, 20:30] - pr["last10"]).max() < 1e-7, j
    kc, kv = G["runs"]["charge_2C_to_4p1"], G["runs"]["cv_hold_after_2C"]
    lim = 5e-3 if exact_hold_leg else 1.0
    assert abs(r["runs"][0]["t_end"] - kc["t_end"]) < 5e-3 and abs(r["runs"][1]["t_end"] - kv["t_end"]) < lim        # printed with two decimals: 1388.68 s, 2440.61 s
    assert abs(r["runs"][1]["I"] - kv["I_end"]) < (5e-5 if exact_hold_leg else 2e-3) and abs(r["runs"][1]["SOC"] - 1.0001) < 5e-5
    # with today's source (YP_alg estimated) the first step is 1.30 ms instead: another step history, the same solution within the integration tolerance
    r0 = sim(False)
    assert abs(len(r0["t"]) - 121) <= 8 and abs(r0["V"][0] - k["values"][0]) < 1e-10 and abs(1e3 * r0["t"][1] - 1.3035) < 1e-3


def test_step_history_of_the_2C_charge_notebook(O, variant="lco_iso", exact_hold_leg=True):
    """examples/model_inputs_and_outputs.ipynb cells 6-12 (lines 152-164, 236-240)"""
    th = O.theta_vector(variant)
    runs = [dict(mode=O.MODE_I, value=2.0, tf=1e6, bounds=O.default_bounds(V_max=4.1)), dict(mode=O.MODE_V, value_kind=O.VAL_HOLD, tf=1e6, bounds=O.default_bounds(V_max=4.1))]

    def sim(z):
        r = O.simulate(variant, th, 0.0, runs, opts=O.default_opts(exp_yp_alg_zero=int(z)), keep_Y=True)
        return dict(t=r["t"], V=r["V"], c_e=r["Y_all"][:, :30], runs=r["runs"])
    check_notebook_step_history(sim, exact_hold_leg)


def test_tolerance_tightening_converges(O):
    """accuracy is proven by tightening: the end state converges with first order in reltol or better"""
    th = O.theta_vector("lco_iso")
    V = []
    for rt in (1e-3, 1e-5, 1e-7):
        ro = O.simulate("lco_iso", th, 1.0, [dict(mode=O.MODE_I, value=-1.0, tf=3000.0)], opts=O.default_opts(reltol=rt, abstol=rt * 1e-3))
        assert ro["runs"][0]["flag"] == 0
        V.append(ro["runs"][0]["V"])
    assert abs(V[1] - V[2]) < 0.1 * abs(V[0] - V[2]) + 1e-7
    assert abs(V[0] - V[2]) < 2e-3


def test_rest_and_hold_semantics(O):
    th = O.theta_vector("lco_iso")
    runs = [dict(mode=O.MODE_I, value=1.0, tf=180.0), dict(mode=O.MODE_I, value_kind=O.VAL_REST, tf=600.0),
            dict(mode=O.MODE_I, value_kind=O.VAL_HOLD, tf=50.0)]
    ro = O.simulate("lco_iso", th, 0.0, runs)
    assert [r["flag"] for r in ro["runs"]] == [0, 0, 0]
    assert abs(ro["runs"][0]["t_end"] - 180.0) < 1e-9 and abs(ro["runs"][1]["t_end"] - 780.0) < 1e-6
    assert ro["runs"][1]["I"] == 0.0 and ro["runs"][2]["I"] == 0.0        # :hold after a rest holds I = 0
    assert abs(ro["runs"][1]["SOC"] - 1.0 * 180 / 3600) < 1e-6


@pytest.mark.parametrize("variant", ["lco_thermal", "lco_thermal_tdiff"])
def test_thermal_cc_ct_cv_notebook(O, variant):
    """reference examples/fast_charging_CC-CT-CV.ipynb: temperature=true, SOC0=0, T_max=40 C, V_max=4.1, I_max=4, I_min=1/20;
    simulate(p, I=4) -> simulate!(dT=:hold) -> simulate!(V=:hold).  Exercises the T rows, the heat sources, the dT control row and its
    algebraic twin (scalar_residual.jl:347-372).  Run through BOTH thermal oracle variants: `lco_thermal` restates the reference's matrix-form heat conduction
    A_T * T (residuals.jl:299-489); `lco_thermal_tdiff` -- the variant the tight-tolerance GPU suite compares the device with -- evaluates the same stencil on
    temperature differences.  The notebook's printed results pin both."""
    th = O.theta_vector(variant)
    m = O.meta(variant)
    assert m["N"] == 351 and m["nnz"] + 1 == 2883          # SURVEY.md App. D (C3, CC mode)
    b = O.default_bounds(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
    runs = [dict(mode=O.MODE_I, value=4.0, bounds=b), dict(mode=O.MODE_DT, value_kind=O.VAL_HOLD, bounds=b),
            dict(mode=O.MODE_V, value_kind=O.VAL_HOLD, bounds=b)]
    ro = O.simulate(variant, th, 0.0, runs)
    assert ro["rc"] == 0
    for key, r in zip(("thermal_4C", "thermal_dT_hold", "thermal_V_hold"), ro["runs"]):
        k = G["runs"][key]
        assert r["flag"] == k["flag"], (key, r)
        assert abs(r["t_end"] - k["t_end"]) <= k["tol"]["t_end_rel"] * k["t_end"], (key, r["t_end"])
        if "V_abs" in k["tol"]:
            assert abs(r["V"] - k["V_end"]) < k["tol"]["V_abs"]
        if "I_rel" in k["tol"]:
            assert abs(r["I"] - k["I_end"]) <= k["tol"]["I_rel"] * k["I_end"], (key, r["I"])
        assert abs(r["SOC"] - k["SOC_end"]) < 2e-3
    assert abs(ro["runs"][0]["T_avg"] - 313.15) < 1e-6 and abs(ro["runs"][1]["T_avg"] - 313.15) < 1e-4     # CT leg holds 40 C
    assert abs(ro["runs"][2]["T_avg"] - (25.6963 + 273.15)) < 0.05


def test_thermal_tdiff_variant_is_the_same_model(O):
    """`lco_thermal_tdiff` (Model(t_conduction="difference")) against `lco_thermal` (the reference's matrix form) row by row on realistic states with a
    non-trivial T(x): identical sizes / parameter keys / sparsity patterns in every mode; every residual row within 1e-12 of the magnitude of its terms (the two
    forms differ by rounding only: aL T_l + aD T + aU T_r vs aL (T_l - T) + aU (T_r - T) with aD = -(aL + aU) [- h_cell / (h rho Cp) at the two ends]); every Jacobian
    value within 1e-9; the consistent initialisation (incl. the dT twin) within 1e-10; and the default-tolerance CC-CT-CV trajectory with the same decisions."""
    import parity
    A, B = "lco_thermal", "lco_thermal_tdiff"
    ma, mb = O.meta(A), O.meta(B)
    for k in ("N", "N_diff", "nnz", "nnz_alg", "theta_keys", "theta_default", "alg_colptr", "alg_rowval"):
        assert ma[k] == mb[k], k
    th = O.theta_vector(A)
    N = ma["N"]
    Y, YP = parity.realistic_states(O, th, 4, seed=11, variant=A)
    for mode, val in ((O.MODE_I, 3.0), (O.MODE_V, 3.9), (O.MODE_DT, 0.01), (O.MODE_P, 80.0)):
        for i in range(len(Y)):
            Fa, Fb = O.residual(A, th, Y[i], YP[i], mode, val), O.residual(B, th, Y[i], YP[i], mode, val)
            cpa, ria, nza = O.jacobian(A, th, Y[i], YP[i], 0.0, mode, val)
            cpb, rib, nzb = O.jacobian(B, th, Y[i], YP[i], 0.0, mode, val)
            assert np.array_equal(cpa, cpb) and np.array_equal(ria, rib)
            _, _, nz1 = O.jacobian(A, th, Y[i], YP[i], 1.0, mode, val)
            term = np.zeros(N); term[-1] = abs(val)
            for c in range(N):
                sl = slice(cpa[c], cpa[c + 1])
                np.add.at(term, ria[sl], np.abs(nza[sl] * Y[i, c]) + np.abs((nz1[sl] - nza[sl]) * YP[i, c]))
            bad = np.abs(Fa - Fb) > 1e-12 * term + 1e-300
            assert not bad.any(), (mode, i, np.nonzero(bad)[0][:5], np.abs(Fa - Fb)[bad][:5], term[bad][:5])
            rel = np.abs(nza - nzb) / (np.abs(nza) + 1e-300)
            assert rel.max() < 1e-9, (mode, i, rel.max())
    # (dT = 0 from a uniform-temperature rest state would be I = 0 +- sqrt(rounding) -- the heat is quadratic in I --, so the twin is initialised at 0.02 K/s.  Its row SUMS the
    #  fifty T rows: the matrix form's 1e-9 K/s of rounding per row is 1e-6 relative in the current it determines -- DESIGN.md 5 "tight tolerances" -- hence 3e-6 there)
    for mode, val, tol in ((O.MODE_I, 4.0, 1e-10), (O.MODE_DT, 0.02, 3e-6)):
        Yg = O.initial_guess(A, th, 0.3)
        assert np.array_equal(Yg, O.initial_guess(B, th, 0.3))
        (rca, Ya, YPa, ita), (rcb, Yb, YPb, itb) = O.init_consistent(A, th, Yg, mode, val), O.init_consistent(B, th, Yg, mode, val)
        assert rca == rcb == 0 and ita == itb and parity.state_rel_err(Ya, Yb) < tol, (mode, parity.state_rel_err(Ya, Yb))
    b = O.default_bounds(T_max=313.15, V_max=4.1, I_max=4.0, I_min=1 / 20)
    runs = [dict(mode=O.MODE_I, value=4.0, bounds=b), dict(mode=O.MODE_DT, value_kind=O.VAL_HOLD, bounds=b), dict(mode=O.MODE_V, value_kind=O.VAL_HOLD, bounds=b)]
    ra, rb = O.simulate(A, th, 0.0, runs), O.simulate(B, th, 0.0, runs)
    assert [r["flag"] for r in ra["runs"]] == [r["flag"] for r in rb["runs"]]
    # CC and CT legs: identical decisions (the V-hold leg is the "+-1 step" leg of DESIGN.md 5 for any two evaluations that differ in the last bit)
    for k in range(2):
        assert ra["runs"][k]["iterations"] == rb["runs"][k]["iterations"], (k, ra["runs"][k], rb["runs"][k])
        # (the h0 floor of DESIGN.md 5 for two last-bit-different evaluations: 1e-6 on the CC leg, amplified by the restart from a held set point on the CT leg)
        assert abs(ra["runs"][k]["t_end"] - rb["runs"][k]["t_end"]) <= (5e-6, 5e-5)[k] * ra["runs"][k]["t_end"]
    assert abs(ra["runs"][2]["t_end"] - rb["runs"][2]["t_end"]) <= 2e-3 * ra["runs"][2]["t_end"]


@pytest.mark.parametrize("A,B", [("lco_iso", "lco_iso_quiet"), ("lco_thermal", "lco_thermal_quiet"), ("nmc_iso_sei", "nmc_iso_sei_quiet")])
def test_quiet_variants_are_the_same_model(O, A, B):
    """r05.  `<variant>_quiet` evaluates every stencil that cancels large terms on DIFFERENCES (Model.phi_s_form / t_conduction = "difference"): the [1, -2, 1] Laplacian of
    the Phi_s rows as (Phi[i+1] - Phi[i]) - (Phi[i] - Phi[i-1]) before the source term joins, where the generated code of the plain variants adds the ~1e-6 V source to a ~4 V
    potential first and returns the row quantised at ulp(Phi_s) = 8.9e-16 V.  Same model: sizes / keys / patterns identical, every residual row within 1e-12 of the magnitude
    of its terms, every Jacobian value within 1e-9, the consistent initialisation within 1e-10."""
    import parity
    ma, mb = O.meta(A), O.meta(B)
    for k in ("N", "N_diff", "nnz", "nnz_alg", "theta_keys", "theta_default", "alg_colptr", "alg_rowval", "colptr", "rowval"):
        assert ma[k] == mb[k], k
    th = O.theta_vector(A)
    N = ma["N"]
    Y, YP = parity.realistic_states(O, th, 3, seed=5, variant=A)
    modes = ((O.MODE_I, 2.0), (O.MODE_V, 3.9), (O.MODE_P, 80.0)) + (((O.MODE_DT, 0.01),) if "thermal" in A else ())
    for mode, val in modes:
        for i in range(len(Y)):
            Fa, Fb = O.residual(A, th, Y[i], YP[i], mode, val), O.residual(B, th, Y[i], YP[i], mode, val)
            cpa, ria, nza = O.jacobian(A, th, Y[i], YP[i], 0.0, mode, val)
            cpb, rib, nzb = O.jacobian(B, th, Y[i], YP[i], 0.0, mode, val)
            assert np.array_equal(cpa, cpb) and np.array_equal(ria, rib)
            _, _, nz1 = O.jacobian(A, th, Y[i], YP[i], 1.0, mode, val)
            term = np.zeros(N); term[-1] = abs(val)
            for c in range(N):
                sl = slice(cpa[c], cpa[c + 1])
                np.add.at(term, ria[sl], np.abs(nza[sl] * Y[i, c]) + np.abs((nz1[sl] - nza[sl]) * YP[i, c]))
            bad = np.abs(Fa - Fb) > 1e-12 * term + 1e-300
            assert not bad.any(), (mode, i, np.nonzero(bad)[0][:5], np.abs(Fa - Fb)[bad][:5], term[bad][:5])
            rel = np.abs(nza - nzb) / (np.abs(nza) + 1e-300)
            assert rel.max() < 1e-9, (mode, i, rel.max())
    Yg = O.initial_guess(A, th, 0.3)
    assert np.array_equal(Yg, O.initial_guess(B, th, 0.3))
    (rca, Ya, YPa, ita), (rcb, Yb, YPb, itb) = O.init_consistent(A, th, Yg, O.MODE_I, 2.0), O.init_consistent(B, th, Yg, O.MODE_I, 2.0)
    assert rca == rcb == 0 and ita == itb and parity.state_rel_err(Ya, Yb) < 1e-9


def test_the_generated_phi_s_rows_are_quantised_and_the_notebook_shows_it(O):
    """r05: what decides the step sequence of a V = :hold leg -- in the oracle AND in the reference.  (1) The plain variant's Phi_s rows (generated code: `-j x + Phi[i-1] -
    2 Phi[i] + Phi[i+1]`, summed left to right) come out as multiples of ulp(Phi_s); the quiet variant's do not.  (2) That rounding, amplified by J^-1, is ABOVE the local
    error of the first steps of a :hold leg, and IDA's start-up order selection reads it: on the reference notebook's own protocol (model_inputs_and_outputs.ipynb: 2C charge
    to 4.1 V, then V = :hold) the plain variant reproduces the printed 37-point hold leg ending at 2440.61 s with I = 0.1955 C, the quiet variant takes 38 points and ends at
    2441.33 s -- the reference's generated code has this rounding too.  The device's default build evaluates like the quiet variant (and agrees with it to 1e-11 with identical
    decisions: tests/test_device_source_emu.py), its PLH_PREC_F64_REFORDER variants like the plain one."""
    th = O.theta_vector("lco_iso")
    runs = [dict(mode=O.MODE_I, value=2.0, tf=1e6, bounds=O.default_bounds(V_max=4.1)), dict(mode=O.MODE_V, value_kind=O.VAL_HOLD, tf=1e6, bounds=O.default_bounds(V_max=4.1))]
    res = {}
    for v in ("lco_iso", "lco_iso_quiet"):
        r = O.simulate(v, th, 0.0, runs, opts=O.default_opts(exp_yp_alg_zero=1))
        res[v] = r
        assert [q["flag"] for q in r["runs"]] == [2, 4] and r["runs"][0]["iterations"] == 84          # the CC leg: the notebook's 84 points either way
        assert abs(r["runs"][0]["t_end"] - 1388.68) < 5e-3
    kv = G["runs"]["cv_hold_after_2C"]
    assert res["lco_iso"]["runs"][1]["iterations"] == 37 and abs(res["lco_iso"]["runs"][1]["t_end"] - kv["t_end"]) < 5e-3 and abs(res["lco_iso"]["runs"][1]["I"] - kv["I_end"]) < 5e-5
    assert res["lco_iso_quiet"]["runs"][1]["iterations"] == 38 and abs(res["lco_iso_quiet"]["runs"][1]["t_end"] - 2441.33) < 2e-2        # same solution within the tolerance, another step sequence
    assert abs(res["lco_iso_quiet"]["runs"][1]["t_end"] - kv["t_end"]) < 1e-3 * kv["t_end"]
    # (1): quantisation of the cathode's interior Phi_s rows at a converged state
    Y = res["lco_iso"]["Y"]; YP = np.zeros_like(Y)
    m = O.meta("lco_iso")
    o_ps = 280
    Fa = O.residual("lco_iso", th, Y, YP, O.MODE_I, Y[-1]); Fb = O.residual("lco_iso_quiet", th, Y, YP, O.MODE_I, Y[-1])
    ulp = np.spacing(Y[o_ps])
    qa = np.abs(Fa[o_ps + 1:o_ps + 9] / ulp - np.round(Fa[o_ps + 1:o_ps + 9] / ulp))
    assert qa.max() < 1e-6, qa                                                                      # multiples of ulp(Phi_s) (8.9e-16 V)
    assert np.abs(Fa[o_ps:o_ps + 10] - Fb[o_ps:o_ps + 10]).max() <= 1.01 * ulp                      # ... within one ulp of the quiet evaluation


def test_quiet_variants_and_the_notebook_kats(O):
    """the known-answer tests of the reference's notebooks, run through the QUIET oracle variants (the ones the device's default build is compared with
    cell by cell), with the one they fail named.

    pass (same thresholds as the plain, notebook-pinned variants):
      * V(t = 0) of the 2C charge, 4 Newton iterations                      (model_inputs_and_outputs.ipynb; test_V0_known_answer)
      * 1C discharge: 3600.0 s, flag 3, V_end; 2C charge to 4.1 V: t_end, SOC, P; V = :hold to I_min: t_end, I_end       (test_notebook_runs)
      * the printed step history of the 2C charge: 84 saved points, sol.V[1:13] to 1e-8, the step times, five c_e profiles, the last twelve voltages, 1388.68 s
      * the four function-input results                                       (variable_input_functions.ipynb; test_function_inputs_notebook)
      * CC-CT-CV: 357.56 s / 686.41 s / flags / T_avg = 40 C                  (fast_charging_CC-CT-CV.ipynb; lco_thermal_quiet)
    FAIL -- exactly one:
      * the number of saved points of the `simulate!(sol, p, V = :hold)` leg of model_inputs_and_outputs.ipynb: the notebook has 37 (121 in total), the quiet variant takes
        38 and ends at 2441.33 s / 0.1944 C instead of 2440.61 s / 0.1955 C -- the same solution within the integration tolerance on another step sequence.  That is the r05
        finding itself (test_the_generated_phi_s_rows_are_quantised_and_the_notebook_shows_it: the reference's generated Phi_s rows are quantised at ulp(Phi_s), the noise steers
        the start-up order selection of a hold leg); a device build that reproduces it exists (precision = "f64_reforder")."""
    test_V0_known_answer(O, "lco_iso_quiet")
    test_notebook_runs(O, "lco_iso_quiet")
    test_function_inputs_notebook(O, "lco_iso_quiet")
    test_step_history_of_the_2C_charge_notebook(O, "lco_iso_quiet", exact_hold_leg=False)           # everything but the hold leg's point count / end time to two decimals
    with pytest.raises(AssertionError):                                                             # ... which is the one KAT it fails
        test_step_history_of_the_2C_charge_notebook(O, "lco_iso_quiet", exact_hold_leg=True)
    test_thermal_cc_ct_cv_notebook(O, "lco_thermal_quiet")


def test_thermal_jacobian_vs_complex_step(O):
    m = dm.Model("LCO", temperature=True)
    th = O.theta_vector("lco_thermal")
    thd = dict(m.theta)
    ro = O.simulate("lco_thermal", th, 0.2, [dict(mode=O.MODE_I, value=3.0, tf=120.0, bounds=O.default_bounds(T_max=400.0))])
    Y, YP, cj = ro["Y"], ro["YP"], 0.5
    N = 351
    for mode, val in ((O.MODE_I, 3.0), (O.MODE_DT, 0.0)):
        cp, ri, nz = O.jacobian("lco_thermal", th, Y, YP, cj, mode, val)
        J = np.zeros((N, N))
        for c in range(N):
            J[ri[cp[c]:cp[c + 1]], c] = nz[cp[c]:cp[c + 1]]
        ops, h = dm.FloatOps(), 1e-30
        for c in list(range(0, N, 11)) + [N - 1]:
            Yc = [complex(v) for v in Y]; Yc[c] += 1j * h
            dY = np.imag(np.array(dm.residual(m, ops, Yc, [complex(v) for v in YP], thd, mode, val))) / h
            YPc = [complex(v) for v in YP]; YPc[c] += 1j * h
            dYP = np.imag(np.array(dm.residual(m, ops, [complex(v) for v in Y], YPc, thd, mode, val))) / h
            col = dY + cj * dYP
            assert np.abs(J[:, c] - col).max() <= 1e-9 * (np.abs(col).max() + 1e-300), (mode, c)


def test_nmc_sei_variant(O):
    """config C5's model (NMC + aging=:SEI; SEI parameters borrowed from LiC6): sizes, I1C, a GITT pulse."""
    m = O.meta("nmc_iso_sei")
    assert m["N"] == 322 and m["N_diff"] == 241 and m["nnz"] + 1 == 2269 and m["nnz_alg"] + 1 == 314
    assert abs(dm.calc_I1C(dm.theta_NMC()) - 19.966706096404245) < 1e-12                                 # survey-computed, not a reference KAT
    th = O.theta_vector("nmc_iso_sei")
    b = O.default_bounds("NMC")
    runs = [dict(mode=O.MODE_I, value=1.0, tf=180.0, bounds=b), dict(mode=O.MODE_I, value_kind=O.VAL_REST, tf=7200.0, bounds=b)]
    ro = O.simulate("nmc_iso_sei", th, 0.0, runs)
    assert [r["flag"] for r in ro["runs"]] == [0, 0]
    SOH = ro["Y"][30 + 200 + 10]        # layout: c_e 30 | c_s 200 | film 10 | SOH
    film = ro["Y"][230:240]
    assert 0 < 1 - SOH < 1e-5 and (film > 0).all() and film.max() < 1e-9     # SOH after the first pulse ~ 1 - 1.3e-6
    assert abs(ro["runs"][1]["SOC"] - 0.05) < 1e-6


def test_function_inputs_notebook(O, variant="lco_iso"):
    """reference examples/variable_input_functions.ipynb: time-dependent current (a step with and without tdiscon, two ramps) given here as
    piecewise-linear tables (run_function: scalar_residual.jl:169-170, tstops of tdiscon model_evaluation.jl:295-297, checks.jl:251-269,341-364)"""
    th = O.theta_vector(variant)
    I1C = 29.23                                       # printed to 2 decimals in the notebooks; the exact value is pinned in test_I1C
    for key in ("func_step_no_tdiscon", "func_step_tdiscon", "func_ramp_100", "func_ramp_10"):
        k = G["runs"][key]
        ro = O.simulate(variant, th, k["SOC0"], [dict(mode=O.MODE_I, table=k["table"], tf=k["tf"])], opts=O.default_opts(tdiscon=k["tdiscon"]))
        r = ro["runs"][0]
        assert r["flag"] == k["flag"] and abs(r["t_end"] - k["t_end"]) < 1e-9
        assert abs(r["I"] - k["I_end"]) < 1e-12 and abs(r["V"] - k["V_end"]) < k["tol"]["V_abs"], (key, r["V"])
        assert abs(r["SOC"] - k["SOC_end"]) < 1e-4
        P = r["I"] * G["I1C_LCO"]["value"] * r["V"]     # calc_P = I * I1C * V (scalar_residual.jl:87)
        assert abs(P - k["P_end"]) <= k["tol"]["P_rel"] * k["P_end"], (key, P)


def test_function_inputs_notebook_as_closures(O):
    """the same notebook cases with the input given as the closure itself -- `I_fun1(t) = t < 100 ? 1 : 0.5`, `I_ramp(t, p) = p.θ[:ramp_val] * t` -- in the postfix-program form of
    the C ABI (ORC_VAL_EXPR): the oracle's closure path reproduces the reference's printed results like its table path does"""
    th = O.theta_vector("lco_iso")
    OP = dict(CONST=0, T=1, MUL=7, LT=19, SELECT=23)
    step = ([OP["T"], OP["CONST"], OP["LT"], OP["CONST"], OP["CONST"], OP["SELECT"]], [0, 100.0, 0, 1.0, 0.5, 0])            # ifelse(t < 100, 1, 0.5)
    ramp = lambda slope: ([OP["CONST"], OP["T"], OP["MUL"]], [slope, 0, 0])                                                    # ramp_val * t
    for key, prog in (("func_step_no_tdiscon", step), ("func_step_tdiscon", step), ("func_ramp_100", ramp(1 / 100)), ("func_ramp_10", ramp(1 / 10))):
        k = G["runs"][key]
        ro = O.simulate("lco_iso", th, k["SOC0"], [dict(mode=O.MODE_I, expr=prog, tf=k["tf"])], opts=O.default_opts(tdiscon=k["tdiscon"]))
        r = ro["runs"][0]
        assert r["flag"] == k["flag"] and abs(r["t_end"] - k["t_end"]) < 1e-9
        assert abs(r["I"] - k["I_end"]) < 1e-12 and abs(r["V"] - k["V_end"]) < k["tol"]["V_abs"], (key, r["V"])
        assert abs(r["SOC"] - k["SOC_end"]) < 1e-4
        P = r["I"] * G["I1C_LCO"]["value"] * r["V"]
        assert abs(P - k["P_end"]) <= k["tol"]["P_rel"] * k["P_end"], (key, P)

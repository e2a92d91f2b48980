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


def test_V0_known_answer(O):
    th = O.theta_vector("lco_iso")
    Y0 = O.initial_guess("lco_iso", th, 0.0); Y0[-1] = 2.0
    rc, Y, YP, it = O.init_consistent("lco_iso", th, Y0, O.MODE_I, 2.0)
    assert rc == 0 and it == 4
    assert abs((Y[280] - Y[299]) - G["V0_2C_charge"]["value"]) < G["V0_2C_charge"]["tol_abs"]


def test_notebook_runs(O):
    th = O.theta_vector("lco_iso")
    k = G["runs"]["discharge_1C"]
    r = O.simulate("lco_iso", th, 1.0, [dict(mode=O.MODE_I, value=-1.0)])["runs"][0]
    assert r["flag"] == k["flag"] and abs(r["t_end"] - k["t_end"]) <= k["tol"]["t_end_rel"] * k["t_end"]
    assert abs(r["V"] - k["V_end"]) < k["tol"]["V_abs"] and abs(r["SOC"]) < 1e-12
    b = O.default_bounds(V_max=4.1)
    ro = O.simulate("lco_iso", th, 0.0, [dict(mode=O.MODE_I, value=2.0, tf=1800.0, bounds=b),
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

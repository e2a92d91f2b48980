"""The host mirror of simulate()/simulate!() (petlion.jl_amd/api.py) against the reference's notebook outputs, run through the
emulator build of the device source on CPU."""
import json
import os

import numpy as np
import pytest

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "notebook_kats.json"), encoding="utf-8"))


def test_getting_started_discharge(emu_model, pkg):
    p = emu_model
    assert abs(p.θ["I1C"] - G["I1C_LCO"]["value"]) == 0.0
    sol = pkg.simulate(p, I=-1, SOC=1)
    k = G["runs"]["discharge_1C"]
    assert pkg.final_exit_reason(sol) == "Below min. SOC" and sol.results[-1].flag == k["flag"]
    assert abs(sol.t[-1] - k["t_end"]) < 1e-6 * k["t_end"] and abs(sol.V[-1] - k["V_end"]) < k["tol"]["V_abs"]
    assert abs(sol.P[-1] - k["P_end"]) < 5e-3 * abs(k["P_end"])
    assert np.all(np.diff(sol.t) > 0) and np.all(np.diff(sol.SOC) < 0)


def test_cc_cv_chain_with_simulate_bang(emu_model, pkg):
    p = emu_model
    sol = pkg.simulate(p, 1800, I=2, SOC=0, V_max=4.1)
    k = G["runs"]["charge_2C_to_4p1"]
    assert sol.results[0].exit_reason == "Above max. voltage"
    assert abs(sol.t[-1] - k["t_end"]) < k["tol"]["t_end_rel"] * k["t_end"] and abs(sol.SOC[-1] - k["SOC_end"]) < k["tol"]["SOC_abs"]
    assert abs(sol.V[0] - G["V0_2C_charge"]["value"]) < 1e-10
    n1 = len(sol)
    pkg.simulate_b(sol, p, V="hold", V_max=4.1, I_min=1 / 20)
    k = G["runs"]["cv_hold_after_2C"]
    assert pkg.exit_reasons(sol) == ["Above max. voltage", "Above max. SOC"]
    assert abs(sol.t[-1] - k["t_end"]) < k["tol"]["t_end_rel"] * k["t_end"] and abs(sol.I[-1] - k["I_end"]) < k["tol"]["I_rel"] * k["I_end"]
    assert np.allclose(sol.V[n1:], 4.1, atol=1e-9)
    assert sol.t[n1] > sol.t[n1 - 1]                      # t0 = nextfloat(t_end), model_evaluation.jl:112
    assert abs(len(sol) - G["V_first13_2C_charge"]["n_points_total"]) <= 8


def test_updating_parameters(emu_model, pkg):
    """p.θ is re-read at every simulate (reference examples/updating_parameters.ipynb cell 4)."""
    p = emu_model
    old = p.θ["ϵ_p"]
    try:
        ends = []
        for eps in (0.385, 0.485):
            p.θ["ϵ_p"] = eps
            ends.append(pkg.simulate(p, I=-1, SOC=1).V[-1])
        assert abs(ends[0] - ends[1]) > 1e-3
    finally:
        p.θ["ϵ_p"] = old


def test_input_validation(emu_model, pkg):
    import pytest
    with pytest.raises(ValueError):
        pkg.simulate(emu_model, I=1, V=4.0)
    with pytest.raises(ValueError):
        pkg.simulate(emu_model, V="rest")
    with pytest.raises(TypeError):
        pkg.simulate(emu_model, I=1, not_an_option=3)


def test_solution_call_interpolates_like_the_reference(emu_model, pkg):
    """sol(t) (reference src/save_outputs.jl:74-133): per-run interpolating cubic splines (FITPACK, as Dierckx.jl), nearest / extrapolate ends"""
    p = emu_model
    sol = pkg.simulate(p, 1200.0, I=2.0, SOC=0.0, V_max=4.1)
    sol = pkg.simulate_b(sol, p, 600.0, V="hold")
    k1 = sol.results[0].iterations
    # the spline interpolates the saved points of both runs exactly
    s = sol(sol.t)
    assert np.abs(s.V - sol.V).max() < 1e-10 and np.abs(s.I - sol.I).max() < 1e-10 and np.abs(s.SOC - sol.SOC).max() < 1e-12
    # between two saved points of the CC leg the interpolant stays within the bracket of a smooth, monotone V(t)
    tm = 0.5 * (sol.t[10] + sol.t[11])
    vm = sol(tm).V[0]
    assert min(sol.V[10], sol.V[11]) - 1e-4 < vm < max(sol.V[10], sol.V[11]) + 1e-4
    # each time is evaluated on the spline of its own run: the current is 2C just before the switch and below 2C in the CV leg
    t_sw = sol.results[0].tspan[1]
    assert abs(sol(t_sw - 1.0).I[0] - 2.0) < 1e-9 and sol(t_sw + 50.0).I[0] < 2.0
    assert len(sol(np.array([t_sw - 1.0, t_sw + 50.0])).results) == 2 and k1 > 3
    # interp_bc: beyond the end "interpolate" holds the last value, "extrapolate" continues the cubic
    t_end = sol.t[-1]
    assert abs(sol(t_end + 100.0).V[0] - sol.V[-1]) < 1e-12
    assert abs(sol(t_end + 100.0, interp_bc="extrapolate").I[0] - sol.I[-1]) > 1e-6
    with pytest.raises(ValueError):
        sol(1.0, interp_bc="bogus")


def test_simulate_vector_tf_and_failed_run_leave_sol_untouched(emu_model, pkg):
    """simulate(p, tf::Vector) runs to tf[end] and post-interpolates onto tf (model_evaluation.jl:79-80); a failed run raises before `sol` is modified"""
    import numpy as np
    p = emu_model
    sol = pkg.simulate(p, [0.0, 100.0, 250.0, 600.0], I=-1, SOC=1)
    assert np.array_equal(sol.t, [0.0, 100.0, 250.0, 600.0]) and len(sol.V) == 4 and sol.V[0] > sol.V[-1]
    full = pkg.simulate(p, 600.0, I=-1, SOC=1)
    assert abs(sol.V[-1] - full.V[-1]) < 1e-9 and abs(sol.V[1] - np.interp(100.0, full.t, full.V)) < 2e-3
    n0 = len(full.t)
    try:
        pkg.simulate_b(full, p, 100.0, I=-1, maxiters=3)              # "Reached max iterations" -> RuntimeError
        assert False, "expected a RuntimeError"
    except RuntimeError:
        pass
    assert len(full.t) == n0 and len(full.results) == 1
    s2 = pkg.simulate(p, 100.0, I=-1, tstops=[50.0])                  # opts.tstops (model_evaluation.jl:292-294): a saved point lands exactly on the stop
    assert (s2.t == 50.0).sum() == 1 and s2.t[-1] == 100.0


def test_closure_tracer_programs(pkg, emu_model):
    """petlion.jl_amd/closures.py: a closure traced into the C ABI's postfix program evaluates to the closure's own value (host-side interpreter), respects the 16-slot stack,
    and refuses what cannot be traced"""
    import math
    cl = pkg.closures
    p = emu_model
    th = p.theta_vector()
    rng = np.random.default_rng(3)
    Y = rng.random(p.N.tot) + 0.5; YP = rng.standard_normal(p.N.tot)
    ps = p.ind["Φ_s"]
    cases = [(lambda t: 1.5 * cl.sin(t) + cl.cos(2 * t) ** 2, lambda t: 1.5 * math.sin(t) + math.cos(2 * t) ** 2),
             (lambda t, q: q.θ["t₊"] * t / (1 + cl.exp(-t)), lambda t: th[p.θ_keys.index("t₊")] * t / (1 + math.exp(-t))),
             (lambda t: cl.where(t < 3, 1.0, cl.where(t >= 5, -2.0, cl.sqrt(t))), lambda t: 1.0 if t < 3 else (-2.0 if t >= 5 else math.sqrt(t))),
             (lambda t, Y_, q: -cl.minimum(1.0, cl.maximum(0.05, (cl.calc_V(Y_, q) - 0.1) * 2.0)), lambda t: -min(1.0, max(0.05, (Y[ps.start] - Y[ps.stop - 1] - 0.1) * 2.0))),
             (lambda t, Y_, YP_, q: abs(YP_[3]) * cl.tanh(Y_[-1]) - cl.log(Y_[0] + t), lambda t: abs(YP[3]) * math.tanh(Y[-1]) - math.log(Y[0] + t)),
             # == / != on traced values (they used to fall back to Python identity and trace a constant branch)
             (lambda t: cl.where(t == 4.0, 10.0, 1.0) + cl.where(t != 2.5, 0.5, 0.25), lambda t: (10.0 if t == 4.0 else 1.0) + (0.5 if t != 2.5 else 0.25)),
             (lambda t, Y_, q: cl.where(Y_[0] != 0, t, -t) * (t == t), lambda t: t)]
    for f, ref in cases:
        prog = cl.trace(f, p)
        for t in (0.0, 2.5, 4.0, 7.0):
            assert abs(cl.evaluate(prog, t, Y, YP, th) - ref(t)) <= 1e-14 * max(1.0, abs(ref(t)))
    deep = lambda t: t + (t + (t + (t + (t + (t + (t + (t + (t + (t + (t + (t + (t + (t + (t + (t + (t + t))))))))))))))))     # right-nested: 18 operands on the stack
    with pytest.raises(cl.TraceError, match="stack"):
        cl.trace(deep, p)
    assert cl.trace(lambda t: ((((t + t) + t) + t) + t), p)[0].size == 9                                                   # left-nested sums need two slots
    with pytest.raises(cl.TraceError):
        cl.trace(lambda t: 1.0 if t < 1 else 2.0, p)
    with pytest.raises(cl.TraceError):
        cl.trace(lambda t, q: q.θ["I1C"] * t, p)
    with pytest.raises(cl.TraceError):
        cl.trace(lambda a, b, c, d, e: a, p)


def test_closure_symbolic_derivatives(pkg, emu_model):
    """closures.row_derivatives: the programs for d f / d Y[c] (what the reference's Symbolics differentiation of the closure supplies to the Newton matrix,
    scalar_residual.jl:289-303) against central differences of the closure's own program, every rule of the table; closures of YP or of t only get none"""
    cl = pkg.closures
    p = emu_model
    th = p.theta_vector()
    rng = np.random.default_rng(5)
    Y = rng.random(p.N.tot) + 0.5; YP = rng.standard_normal(p.N.tot)
    fs = [lambda t, Y_, q: Y_[0] * Y_[1] / (Y_[2] + t) - 3.0 * Y_[0] ** 2.5 + 2.0 ** Y_[3] + Y_[4] ** Y_[5],
          lambda t, Y_, q: cl.sin(Y_[0] * t) * cl.cos(Y_[1]) + cl.exp(-Y_[2]) * cl.log(Y_[3] + 1.0) + cl.sqrt(Y_[4]) * cl.tanh(Y_[5] - 1.0),
          lambda t, Y_, q: -cl.minimum(1.0, cl.maximum(0.05, (cl.calc_V(Y_, q) + 0.2) * 2.0)) + abs(Y_[7] - 1.0) * q.θ["t₊"],
          lambda t, Y_, q: cl.where(Y_[0] > Y_[1], Y_[0] * Y_[2], Y_[1] / Y_[2]) - (-Y_[3]) + cl.minimum(Y_[4], Y_[5] * Y_[5])]
    for f in fs:
        prog, tree = cl.trace(f, p, with_tree=True)
        cols, progs = cl.row_derivatives(tree)
        assert cols == sorted(cols) and len(cols) >= 3
        for c, dp in zip(cols, progs):
            h = 1e-6
            Yp, Ym = Y.copy(), Y.copy(); Yp[c] += h; Ym[c] -= h
            fd = (cl.evaluate(prog, 1.3, Yp, YP, th) - cl.evaluate(prog, 1.3, Ym, YP, th)) / (2 * h)
            an = cl.evaluate(dp, 1.3, Y, YP, th)
            assert abs(an - fd) <= 1e-7 * max(1.0, abs(fd)), (c, an, fd)
    assert cl.row_derivatives(cl.trace(lambda t: cl.sin(t), p, with_tree=True)[1]) is None
    # a closure of YP: columns N + i for differential states i (the device chains them through the differential equations in the consistent initialisation); YP of an
    # algebraic state has no such equation: the reference's fallback path
    N, Nd = p.N.tot, p.N.diff
    tree = cl.trace(lambda t, Y_, YP_, q: Y_[0] * YP_[1] + cl.sin(YP_[5]), p, with_tree=True)[1]
    cols, progs = cl.row_derivatives(tree, N, Nd)
    assert cols == [0, N + 1, N + 5] and abs(cl.evaluate(progs[1], 0.0, Y, YP, th) - Y[0]) < 1e-15 and abs(cl.evaluate(progs[2], 0.0, Y, YP, th) - np.cos(YP[5])) < 1e-15
    assert cl.row_derivatives(cl.trace(lambda t, Y_, YP_, q: Y_[0] + YP_[Nd + 2], p, with_tree=True)[1], N, Nd) is None
    # the run descriptor carries them: columns ascending, programs behind the main one
    (run,), _ = pkg.make_protocol(p, [{"I": lambda t, Y_, q: -0.5 * cl.calc_V(Y_, q), "tf": 10.0}])
    ps = p.ind["Φ_s"]
    assert run.n_dcol == 2 and [run.dcol[0], run.dcol[1]] == [ps.start, ps.stop - 1] and run.dofs[0] == run.n_tab


def test_kernel_selftest(pkg, emu_model, emu_model_thermal):
    """pkg.selftest: every kernel instantiation of a variant against its plain kernel (run automatically, once, for a grid library compiled at first use on a machine with a GPU)"""
    pkg.selftest(emu_model)
    pkg.selftest(emu_model_thermal, n_cells=1, tf=30.0)
    class Broken:                                                       # a model whose runs come back with another SOC: the check must say which instantiation
        pass
    import petlion_jl_amd.api as api
    real = api.simulate_ensemble
    def fake(p, Th, proto, **kw):
        e = real(p, Th, proto, **kw)
        if isinstance(proto[0]["I"], tuple):
            e.run_info["SOC"][:] += 1e-6
        return e
    api.simulate_ensemble = fake
    try:
        with pytest.raises(RuntimeError, match="table input"):
            pkg.selftest(emu_model)
    finally:
        api.simulate_ensemble = real


def test_register_grid_library_refusals(pkg, emu_model):
    """plh_register_grid_library: a missing file and a library that is not a grid library are refused with a message, registering the same grid library twice is a no-op"""
    import build_emu
    lib = emu_model._lib
    assert lib.plh_register_grid_library(b"/nonexistent/libplh_g1.so") != 0 and b"dlopen" in lib.plh_last_error()
    assert lib.plh_register_grid_library(os.fsencode(lib._name)) != 0 and b"not a grid library" in lib.plh_last_error()
    gl = os.fsencode(build_emu.build_grid((12, 7, 9, 11, 10, 10), [0]))
    assert lib.plh_register_grid_library(gl) == 0 and lib.plh_register_grid_library(gl) == 0
    p = pkg.petlion(pkg.LCO, N_p=12, N_s=7, N_n=9, N_r_p=11, N_r_n=11, _lib_path=lib._name, _grid_lib=False)      # already registered: found without registering again
    assert p.N.tot == 330
    with pytest.raises(pkg._capi.PetlionHipError, match="not instantiated|discretisation"):                          # a variant the grid library does not hold
        pkg.petlion(pkg.NMC, N_p=12, N_s=7, N_n=9, N_r_p=11, N_r_n=11, _lib_path=lib._name, _grid_lib=False)

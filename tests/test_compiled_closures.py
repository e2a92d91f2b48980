"""Compiled input closures (petlion.jl_amd/closure_lib.py, plh_model_attach_closure_library): the PLH_VAL_EXPR programs of a protocol as straight-line device code.

The reference compiles the user's closure and its symbolic derivatives into its control-row functions (scalar_residual.jl:231-416); the assertion is the one its own test suite
makes for closures (test/runtests.jl: `.===` between a closure and the constant it returns): compiled and interpreted evaluation give BIT-IDENTICAL trajectories -- same
operations in the same order -- and a protocol the library was not built for falls back to the interpreter."""
import numpy as np
import pytest


def _protocols(p):
    ps = p.ind["Φ_s"]
    a = [{"I": (lambda t, Y, P_: -1.0 + 0.05 * np.sin(0.01 * t) + 1e-3 * (Y[ps.start] - Y[ps.stop - 1])), "tf": 300.0}, {"I": (lambda t: -0.5 - 0.001 * t), "tf": 100.0}]
    b = [{"I": (lambda t: -1.0 - 0.002 * t), "tf": 100.0}]
    return a, b


def check(pkg, p, n, emu_include=None):
    a, b = _protocols(p)
    Th = pkg.theta_matrix(p, n, {"D_sp": p.θ["D_sp"] * np.linspace(0.7, 1.4, n)})
    lib = p._lib
    e0 = pkg.simulate_ensemble(p, Th, a, SOC=1.0)
    assert lib.plh_last_integrate_compiled(p._h) == 0
    path = p.compile_closures(a, _emu_include=emu_include)
    assert path and p.compile_closures(a, _emu_include=emu_include) == path                     # cached
    e1 = pkg.simulate_ensemble(p, Th, a, SOC=1.0)
    assert lib.plh_last_integrate_compiled(p._h) == 1
    n0 = int(e0.n_pts[0])
    assert np.array_equal(e0.Y, e1.Y) and np.array_equal(e0.n_pts, e1.n_pts) and np.array_equal(e0.V[0, :n0], e1.V[0, :n0]) and np.array_equal(e0.I[0, :n0], e1.I[0, :n0])
    for f in ("n_steps", "n_res", "n_jac", "n_newton"):
        assert np.array_equal(e0.counters[f], e1.counters[f])
    e2 = pkg.simulate_ensemble(p, Th, b, SOC=1.0)                                              # another closure: interpreted
    assert lib.plh_last_integrate_compiled(p._h) == 0 and (e2.run_info["flag"] == 0).all()
    o = pkg.Opts(); o.refine = 1
    e3 = pkg.simulate_ensemble(p, Th, a, SOC=1.0, opts=o)                                      # refinement: the interpreter's instantiation
    assert lib.plh_last_integrate_compiled(p._h) == 0 and (e3.run_info["flag"] == 0).all()
    return e0, e1


def test_compiled_closures_emu(pkg):
    import build_emu
    p = pkg.petlion(pkg.LCO, _lib_path=build_emu.build())          # (its own handle: the attachment is per handle)
    check(pkg, p, 2, emu_include=build_emu.HERE)


@pytest.mark.gpu
def test_compiled_closures_gpu(hip_model, hip_model_thermal, pkg):
    for mk in (lambda: pkg.petlion(pkg.LCO), lambda: pkg.petlion(pkg.LCO, temperature=True)):
        p = mk()
        e0, e1 = check(pkg, p, 64)
        print("%s: compiled closures bit-identical to interpreted on 64 cells; kernel %.3f ms interpreted, %.3f ms compiled" % (p.variant, e0.kernel_ms, e1.kernel_ms))

"""The assertions the reference itself intends (test/runtests.jl:9-51 -- one commented-out block, "the tests work properly if they are run in a terminal"), restated through the host
mirror of simulate / simulate! on the product path: the device source on the CPU wave emulator here, the HIP kernels under -m gpu (test_reference_runtests_on_gpu).  The first group of
the reference ("AD matches symbolic": two Jacobian sources of ONE integrator) becomes device against oracle -- two implementations of the integrator -- on the same three runs, with the
norm-wise `isapprox` the reference uses but at the tolerance the integrators ran at."""
import numpy as np
import pytest

import parity


def isapprox(a, b, rtol):
    a, b = np.asarray(a), np.asarray(b)
    return np.linalg.norm(a - b) <= rtol * max(np.linalg.norm(a), np.linalg.norm(b))


def check_runtests(p, O, pkg):
    cl = pkg.closures
    th = p.theta_vector()
    # runtests.jl:23-26: simulate(model, 0:100:3600, I=-1, SOC=1).V ; simulate(model, 0:100, P=-10, SOC=1).V ; simulate(model, 0:10, V=3.5, SOC=1).V
    for tf, kw, mode in ((np.arange(0.0, 3601.0, 100.0), {"I": -1.0}, O.MODE_I), (np.arange(0.0, 101.0), {"P": -10.0}, O.MODE_P), (np.arange(0.0, 11.0), {"V": 3.5}, O.MODE_V)):
        sol = pkg.simulate(p, tf, SOC=1.0, **kw)
        full = pkg.simulate(p, float(tf[-1]), SOC=1.0, **kw)
        ro = O.simulate(p.variant, th, 1.0, parity.runs_to_oracle(O, p, pkg, [dict(kw, tf=float(tf[-1]))]))
        n = len(ro["t"])
        assert len(sol.V) == len(tf) and len(full.t) == n and isapprox(full.V, ro["V"][:n], 1e-5)
        inside = tf <= full.t[-1]                                       # (the 1C discharge stops on its bound before 3600 s: the reference's sol(t) holds the end value there)
        assert isapprox(sol.V[inside], np.interp(tf[inside], ro["t"][:n], ro["V"][:n]), 2e-3)       # spline (sol) against linear interpolation (here) of the same saved points
    # :29  all outputs work
    assert pkg.simulate(p, 50.0, I=-1.0, SOC=1.0, outputs="all").Y_all.shape[1] == p.N.tot
    # :32-35  functions are working
    assert pkg.simulate(p, 1.0, I=1.0, SOC=0.0).V[-1] != pkg.simulate(p, 1.0, I=lambda t: cl.cos(t), SOC=0.0).V[-1]
    assert pkg.simulate(p, 1.0, P=100.0, SOC=0.0).V[-1] != pkg.simulate(p, 1.0, P=lambda t: 100.0 * cl.cos(t), SOC=0.0).V[-1]
    # :38-39  function matches CC: all(simulate(p, 0:1000, I=1).V .=== simulate(p, 0:1000, I=(t)->1).V) -- bit for bit, through two different kernel instantiations here
    tt = np.arange(0.0, 1001.0)
    a, b = pkg.simulate(p, tt, I=1.0, SOC=0.0), pkg.simulate(p, tt, I=lambda t: 1.0 + 0.0 * t, SOC=0.0)
    assert np.array_equal(a.V, b.V)
    # :41-47  :hold and the I_max stop condition
    sol = pkg.simulate(p, 100.0, I=-0.1, SOC=1.0)
    pkg.simulate(p, 100.0, sol=sol, I=-0.1)
    pkg.simulate(p, 100.0, sol=sol, V="hold")
    pkg.simulate(p, 100.0, sol=sol, V="hold", I_max=-0.05)
    assert abs(sol.I[-1] - (-0.05)) <= 1e-8 * 0.05 + 1e-12                # `sol.I[end] ≈ -0.05` (isapprox: rtol = sqrt(eps))
    pkg.simulate(p, sol=sol, P="hold")
    k = sol.results[-1].iterations
    P_prev_end, P_last_first = sol.P[-k - 1], sol.P[-k]
    assert abs(P_prev_end - P_last_first) <= 1.5e-8 * abs(P_prev_end)     # `sol[end-1].P[end] ≈ sol[end].P[1]`


def test_reference_runtests_on_the_emulator(emu_model, O, pkg):
    check_runtests(emu_model, O, pkg)


@pytest.mark.gpu
def test_reference_runtests_on_gpu(hip_model, O, pkg):
    check_runtests(hip_model, O, pkg)

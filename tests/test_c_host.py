"""A host that is neither Python nor torch on the C ABI: bindings/c/plh_host_demo.c -- plain C99, gcc, include/petlion_hip.h, PLH_HOST pointers --
creates a model, runs config C2 and a two-call CC -> CV chain with Y_init / t_init continuation, and prints every per-cell result in hexadecimal floating point.  The same
calls through the ctypes mirror must give the same bits.  `-m gpu`: against libpetlion_hip.so, 1024 cells (config C2's size).  Here (no GPU): the identical C source linked
against the test-only wave-emulator build of the same library, 3 cells -- it exercises the header, the struct layouts as a C compiler lays them out, and the call sequence."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(text):
    rows = {}
    for ln in text.splitlines():
        w = ln.split()
        if len(w) > 3 and w[1] == "cell":
            rows[(w[0], int(w[2]))] = dict(zip(w[3::2], w[4::2]))
    return rows


def expected(pkg, p, n):
    """the demo's calls through ctypes (petlion.jl_amd/api.py::_integrate is the one place that fills plh_outputs)"""
    api = sys.modules[pkg.__name__ + ".api"]
    Th = pkg.theta_matrix(p, n)
    Th[:, p.θ_keys.index("D_sp")] *= 1.0 + 0.125 * (np.arange(n) % 5)
    o = pkg.Opts()
    out = {}

    def leg(name, proto, soc, Y_init=None, t_init=None):
        runs, _ = pkg.make_protocol(p, proto, n)
        b = api._integrate(p, Th, soc, runs, o, Y_init=Y_init, t_init=t_init, max_points=512)
        for c in range(n):
            ri = b["run_info"][c, 0]
            s = 0.0
            for v in b["Y"][c]:
                s += float(v)
            out[(name, c)] = dict(flag=str(int(ri["flag"])), iterations=str(int(ri["iterations"])), n_pts=str(int(b["n_pts"][c])), t_end=float(ri["t_end"]).hex(),
                                  V=float(ri["V"]).hex(), I=float(ri["I"]).hex(), SOC=float(ri["SOC"]).hex(), Ysum=s.hex())
        return b
    leg("c2", [{"I": -1.0}], np.ones(n))
    b = leg("cc", [{"I": 2.0, "V_max": 4.1}], np.zeros(n))
    leg("cv", [{"V": "hold", "V_max": 4.1}], b["run_info"][:, 0]["SOC"].copy(), Y_init=b["Y"].copy(), t_init=b["run_info"][:, 0]["t_end"].copy())
    return out


def same(a, b):
    return a == b or (a[0] != "-" and float.fromhex(a) == float.fromhex(b))       # (%a and float.hex() normalise the mantissa differently: compare the values, exactly)


def check(exe, pkg, p, n):
    r = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, (r.returncode, r.stdout[-500:], r.stderr[-500:])
    got, want = parse(r.stdout), expected(pkg, p, n)
    assert set(got) == set(want) and len(got) == 3 * n
    for key in sorted(want):
        for f, v in want[key].items():
            g = got[key][f]
            assert (g == v) if f in ("flag", "iterations", "n_pts") else float.fromhex(g) == float.fromhex(v), (key, f, g, v)
    assert all(got[("c2", c)]["flag"] == "3" and abs(float.fromhex(got[("c2", c)]["t_end"]) - 3600.0) < 1e-8 for c in range(n))      # 1C discharge: SOC_min at 3600 s
    assert all(got[("cc", c)]["flag"] == "2" and got[("cv", c)]["flag"] == "4" for c in range(n))                       # V_max, then SOC_max in the hold
    return r.stdout


def test_c_host_on_the_emulator(pkg, emu_model):
    import __graft_entry__ as g
    check(g.build_c_host(emu=True), pkg, emu_model, 3)


@pytest.mark.gpu
def test_c_host_on_gpu(pkg, hip_model):
    import __graft_entry__ as g
    out = check(g.build_c_host(), pkg, hip_model, 1024)
    assert "kernel_ms_positive 1" in out

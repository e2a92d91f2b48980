"""
ORACLE (test infrastructure, NOT product code) -- ctypes front-end of oracle/liboracle.so (ida_oracle.c + generated
model functions).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import json
import math
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
MODE_I, MODE_V, MODE_DT, MODE_P, MODE_ETAP = 0, 1, 2, 3, 4
VAL_CONST, VAL_HOLD, VAL_REST, VAL_TABLE, VAL_EXPR = 0, 1, 2, 3, 4
NAN = math.nan

BOUND_FIELDS = ["V_max", "V_min", "SOC_max", "SOC_min", "T_max", "c_s_n_max", "I_max", "I_min", "eta_plating_min",
                "c_e_min", "dfilm_max"]


class Bounds(C.Structure):
    _fields_ = [(f, C.c_double) for f in BOUND_FIELDS]


class Run(C.Structure):
    _fields_ = [("mode", C.c_int), ("value_kind", C.c_int), ("value", C.c_double), ("tf", C.c_double), ("bounds", Bounds),
                ("n_tab", C.c_int), ("tab_t", C.POINTER(C.c_double)), ("tab_v", C.POINTER(C.c_double)),
                ("n_dcol", C.c_int), ("dcol", C.POINTER(C.c_int)), ("dofs", C.POINTER(C.c_int)), ("dstate", C.c_int)]


class Opts(C.Structure):
    _fields_ = [("abstol", C.c_double), ("reltol", C.c_double), ("abstol_init", C.c_double), ("reltol_init", C.c_double),
                ("maxiters", C.c_int), ("check_bounds", C.c_int), ("interp_final", C.c_int), ("max_order", C.c_int),
                ("jac_every_step", C.c_int), ("init_step", C.c_double), ("n_tdiscon", C.c_int), ("tdiscon", C.POINTER(C.c_double)),
                ("refine", C.c_int), ("fd_perturb", C.c_double), ("perturb_seed", C.c_int), ("n_tstops", C.c_int), ("tstops", C.POINTER(C.c_double)), ("exp_yp_alg_zero", C.c_int), ("res_perturb", C.c_double),
                ("n_stop", C.c_int), ("stop_ops", C.POINTER(C.c_double)), ("stop_args", C.POINTER(C.c_double))]


class RunInfo(C.Structure):
    _fields_ = [("flag", C.c_int), ("iterations", C.c_int), ("t_end", C.c_double), ("V", C.c_double), ("I", C.c_double),
                ("SOC", C.c_double), ("T_avg", C.c_double)]


class Counters(C.Structure):
    _fields_ = [(f, C.c_long) for f in ["n_steps", "n_res", "n_jac", "n_fact", "n_solve", "n_newton", "n_errfail",
                                       "n_convfail", "sum_kp2", "n_init_iters"]]


def build(force=False):
    so = os.path.join(HERE, "liboracle.so")
    if force or not os.path.exists(so):
        subprocess.check_call(["make", "-C", HERE], stdout=subprocess.DEVNULL)
    return so


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.orc_theta_key.restype = C.c_char_p
    return _lib


def meta(variant):
    with open(os.path.join(HERE, "gen", variant + ".json"), encoding="utf-8") as f:
        return json.load(f)


def default_bounds(cathode="LCO", **over):
    from .dfn_model import BOUNDS_DEFAULT
    d = dict(BOUNDS_DEFAULT[cathode])
    d["eta_plating_min"] = d.pop("η_plating_min")
    d.update(over)
    return Bounds(**d)


def default_opts(**over):
    d = dict(abstol=1e-6, reltol=1e-3, maxiters=10000, check_bounds=1, interp_final=1, max_order=5, jac_every_step=0, init_step=0.0, refine=0, fd_perturb=0.0, perturb_seed=0, exp_yp_alg_zero=0, res_perturb=0.0)
    tdiscon = np.ascontiguousarray(list(over.pop("tdiscon", [])), dtype=np.float64)
    tstops = np.ascontiguousarray(list(over.pop("tstops", [])), dtype=np.float64)
    stop = over.pop("stop_program", None)               # (opcodes, operands): opts.stop_function as the postfix program the product's C ABI takes (plh_opts.stop_ops)
    d.update(over)
    d.setdefault("abstol_init", d["abstol"])
    d.setdefault("reltol_init", d["reltol"])
    o = Opts(**d)
    o.n_tdiscon = tdiscon.size
    o.tdiscon = _dp(tdiscon) if tdiscon.size else None
    o.n_tstops = tstops.size
    o.tstops = _dp(tstops) if tstops.size else None
    o._keep = (tdiscon, tstops)
    o.n_stop, o.stop_ops, o.stop_args = 0, None, None
    if stop is not None:
        so, sa = np.ascontiguousarray(stop[0], dtype=np.float64), np.ascontiguousarray(stop[1], dtype=np.float64)
        o.n_stop, o.stop_ops, o.stop_args = so.size, _dp(so), _dp(sa)
        o._keep = o._keep + (so, sa)
    return o


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def theta_vector(variant, overrides=None):
    m = meta(variant)
    th = np.array(m["theta_default"], dtype=np.float64)
    if overrides:
        for k, v in overrides.items():
            th[m["theta_keys"].index(k)] = v
    return th


def simulate(variant, theta, SOC0, runs, opts=None, max_out=20000, Y_init=None, keep_Y=False):
    """runs: list of dicts(mode, value, value_kind, tf, bounds=Bounds).  Returns dict with per-point outputs, final
    state, per-run info and counters."""
    L = lib()
    m = meta(variant)
    N = m["N"]
    opts = opts or default_opts()
    arr = (Run * len(runs))()
    keep = []
    for k, r in enumerate(runs):
        arr[k].mode = r.get("mode", MODE_I)
        arr[k].value_kind = r.get("value_kind", VAL_CONST)
        arr[k].value = r.get("value", 0.0)
        arr[k].tf = r.get("tf", 1e6)
        arr[k].bounds = r.get("bounds") or default_bounds()
        arr[k].dstate = r.get("dstate", 0)
        if r.get("table") is not None:        # (t, v) arrays: piecewise-linear input in run-local time
            tt = np.ascontiguousarray(r["table"][0], dtype=np.float64); vv = np.ascontiguousarray(r["table"][1], dtype=np.float64)
            keep.append((tt, vv))
            arr[k].value_kind = VAL_TABLE; arr[k].n_tab = len(tt); arr[k].tab_t = _dp(tt); arr[k].tab_v = _dp(vv)
        if r.get("expr") is not None:         # (opcodes, operands): closure input as a postfix program (ORC_VAL_EXPR)
            oo = np.ascontiguousarray(r["expr"][0], dtype=np.float64); aa = np.ascontiguousarray(r["expr"][1], dtype=np.float64)
            keep.append((oo, aa))
            arr[k].value_kind = VAL_EXPR; arr[k].n_tab = len(oo) if r.get("n_main") is None else int(r["n_main"]); arr[k].tab_t = _dp(oo); arr[k].tab_v = _dp(aa)
            if r.get("dcol") is not None:     # derivative programs of the control row: columns + instruction offsets into the same arrays (n_main = length of the main program)
                dc = np.ascontiguousarray(r["dcol"], dtype=np.int32); do = np.ascontiguousarray(r["dofs"], dtype=np.int32)
                keep.append((dc, do))
                arr[k].n_dcol = len(dc); arr[k].dcol = dc.ctypes.data_as(C.POINTER(C.c_int)); arr[k].dofs = do.ctypes.data_as(C.POINTER(C.c_int))
    theta = np.ascontiguousarray(theta, dtype=np.float64)
    out = {k: np.zeros(max_out) for k in ("t", "V", "I", "SOC", "T")}
    n_out = C.c_int(0)
    Yf = np.zeros(N)
    YPf = np.zeros(N)
    info = (RunInfo * len(runs))()
    cnt = Counters()
    yi = None if Y_init is None else _dp(np.ascontiguousarray(Y_init, dtype=np.float64))
    Yall = np.zeros((max_out, N)) if keep_Y else None
    rc = L.orc_simulate_all(variant.encode(), _dp(theta), C.c_double(SOC0), len(runs), arr, C.byref(opts), max_out,
                            _dp(out["t"]), _dp(out["V"]), _dp(out["I"]), _dp(out["SOC"]), _dp(out["T"]), None if Yall is None else _dp(Yall),
                            C.byref(n_out), _dp(Yf), _dp(YPf), info, C.byref(cnt), yi)
    n = min(n_out.value, max_out)
    res = {k: v[:n].copy() for k, v in out.items()}
    if keep_Y:
        res["Y_all"] = Yall[:n].copy()
    res.update(rc=rc, Y=Yf, YP=YPf,
               runs=[dict(flag=i.flag, iterations=i.iterations, t_end=i.t_end, V=i.V, I=i.I, SOC=i.SOC, T_avg=i.T_avg) for i in info],
               counters={f: getattr(cnt, f) for f, _ in Counters._fields_})
    return res


def residual(variant, theta, Y, YP, mode=MODE_I, value=0.0):
    L = lib()
    N = meta(variant)["N"]
    out = np.zeros(N)
    rc = L.orc_residual(variant.encode(), _dp(np.ascontiguousarray(theta)), mode, C.c_double(value),
                        _dp(np.ascontiguousarray(Y, dtype=np.float64)), _dp(np.ascontiguousarray(YP, dtype=np.float64)), _dp(out))
    assert rc == 0, rc
    return out


def jacobian(variant, theta, Y, YP, cj, mode=MODE_I, value=0.0):
    """full N x N Jacobian dF/dY + cj dF/dYP (base rows + control row) as (colptr, rowval, nzval)."""
    L = lib()
    N = meta(variant)["N"]
    nnz = C.c_int(0)
    th = np.ascontiguousarray(theta)
    Y = np.ascontiguousarray(Y, dtype=np.float64)
    YP = np.ascontiguousarray(YP, dtype=np.float64)
    L.orc_jacobian(variant.encode(), _dp(th), mode, C.c_double(value), _dp(Y), _dp(YP), C.c_double(cj), C.byref(nnz), None, None, None)
    cp = np.zeros(N + 1, dtype=np.int32)
    ri = np.zeros(nnz.value, dtype=np.int32)
    nz = np.zeros(nnz.value)
    rc = L.orc_jacobian(variant.encode(), _dp(th), mode, C.c_double(value), _dp(Y), _dp(YP), C.c_double(cj), C.byref(nnz),
                        cp.ctypes.data_as(C.POINTER(C.c_int)), ri.ctypes.data_as(C.POINTER(C.c_int)), _dp(nz))
    assert rc == 0, rc
    return cp, ri, nz


def initial_guess(variant, theta, SOC):
    L = lib()
    N = meta(variant)["N"]
    Y = np.zeros(N)
    L.orc_initial_guess(variant.encode(), _dp(np.ascontiguousarray(theta)), C.c_double(SOC), _dp(Y))
    return Y


def init_consistent(variant, theta, Y, mode=MODE_I, value=0.0, reltol_init=1e-3):
    L = lib()
    Y = np.array(Y, dtype=np.float64)
    YP = np.zeros_like(Y)
    it = C.c_int(0)
    rc = L.orc_init_consistent(variant.encode(), _dp(np.ascontiguousarray(theta)), mode, C.c_double(value),
                               C.c_double(reltol_init), _dp(Y), _dp(YP), C.byref(it))
    return rc, Y, YP, it.value


def linear_solve(variant, theta, Y, YP, cj, b, mode=MODE_I, value=0.0, refine=0):
    L = lib()
    b = np.array(b, dtype=np.float64)
    rc = L.orc_linear_solve_refined(variant.encode(), _dp(np.ascontiguousarray(theta)), mode, C.c_double(value),
                                    _dp(np.ascontiguousarray(Y, dtype=np.float64)), _dp(np.ascontiguousarray(YP, dtype=np.float64)),
                                    C.c_double(cj), _dp(b), int(refine))
    assert rc == 0, rc
    return b


def run_batch(variant, thetas, SOC0, runs, n_traj, opts=None):
    """time-able CPU batch (bench.py cpu_baseline): returns (n_ok, t_end_sum, counters dict)"""
    L = lib()
    opts = opts or default_opts()
    arr = (Run * len(runs))()
    for k, r in enumerate(runs):
        arr[k].mode = r.get("mode", MODE_I); arr[k].value_kind = r.get("value_kind", VAL_CONST)
        arr[k].value = r.get("value", 0.0); arr[k].tf = r.get("tf", 1e6); arr[k].bounds = r.get("bounds") or default_bounds()
    thetas = np.ascontiguousarray(np.atleast_2d(thetas), dtype=np.float64)
    s = C.c_double(0.0)
    cnt = Counters()
    ok = L.orc_run_batch(variant.encode(), thetas.shape[0], _dp(thetas), C.c_double(SOC0), len(runs), arr, C.byref(opts), int(n_traj),
                         C.byref(s), C.byref(cnt))
    return ok, s.value, {f: getattr(cnt, f) for f, _ in Counters._fields_}

"""
ORACLE (test infrastructure, NOT product code) -- symbolic code generation for the CPU oracle.

Mirrors the *shape* of the reference pipeline (src/generate_functions.jl:102-164): trace the generic
residual (oracle/dfn_model.py) on symbols, take the sparse Jacobian  J = dF/dY + cj*dF/dYP
(generate_functions.jl:289-307), and emit straight-line C for the five generated functions
  f_diff!, f_alg!, J_y!, J_y_alg!, initial_guess!         (generate_functions.jl:124,135,303,313-314,322)
plus the CSC pattern `J_y_sp` ((N-1) x N, generate_functions.jl:279-280).  The control row is NOT generated
(the reference also adds it separately, scalar_residual.jl:167-229); for thermal models the dT control
row and its "algebraic twin" (scalar_residual.jl:347-372) are generated as extra functions.

Run:  python -m oracle.codegen [lco_iso|lco_thermal|nmc_iso_sei|lco_iso_sei|all]
Output: oracle/gen/<name>.c and oracle/gen/<name>.json (pattern, theta keys) -- generated from THIS repo's
model file only; nothing is read from /root/reference.
"""
from __future__ import annotations

import json
import os
import sys
import time

import sympy as sp
from sympy.printing.c import C99CodePrinter

from . import dfn_model as dm

HERE = os.path.dirname(os.path.abspath(__file__))


class _Printer(C99CodePrinter):
    def _print_Pow(self, expr):
        b, e = expr.as_base_exp()
        if e.is_Integer and 2 <= int(e) <= 4:
            s = self._print(b)
            if not (b.is_Symbol or b.is_Number):
                s = "(" + s + ")"
            return "(" + "*".join([s] * int(e)) + ")"
        if e == -1:
            return "(1.0/(%s))" % self._print(b)
        return super()._print_Pow(expr)


VARIANTS = {
    "lco_iso": dict(cathode="LCO", temperature=False, aging=False),
    "lco_thermal": dict(cathode="LCO", temperature=True, aging=False),
    "lco_iso_sei": dict(cathode="LCO", temperature=False, aging=True),
    "nmc_iso_sei": dict(cathode="NMC", temperature=False, aging=True),
    "nmc_iso": dict(cathode="NMC", temperature=False, aging=False),
    # SURVEY 8(f).4 model variants (one option each on the LCO isothermal model)
    "lco_iso_quad": dict(cathode="LCO", solid_diffusion="quadratic"),
    "lco_iso_poly": dict(cathode="LCO", solid_diffusion="polynomial"),
    "lco_iso_nu": dict(cathode="LCO", thermodynamic_factor="nonlinear"),
    "lco_iso_mhc": dict(cathode="LCO", rxn="MHC"),
    "lgm50_iso": dict(cathode="LGM50"),                  # NMC_LGM50 + LiC6_LGM50 (Chen et al. 2020), reference src/params.jl:514-849
    # the LCO thermal model with the heat-conduction stencil evaluated on temperature differences (dfn_model.Model.t_conduction): same equations, 1e4 x less rounding in the
    # sum of the T rows that the dT control row and its twin form -- the tight-tolerance counterpart of the device's evaluation
    "lco_thermal_tdiff": dict(cathode="LCO", temperature=True, t_conduction="difference"),
    # r05: the same two models with EVERY stencil that cancels large terms evaluated on differences -- T rows as in lco_thermal_tdiff and the [1, -2, 1] Laplacian of the Φ_s rows
    # (Model.phi_s_form): the generated "matrix" rows add the 1e-6 V source to a 4 V potential before the Laplacian cancels and come out quantised at 8.9e-16 V, which J^-1
    # amplifies above the local error of the first steps of a :hold leg.  Same equations; the evaluation order the device has.
    "lco_iso_quiet": dict(cathode="LCO", phi_s_form="difference"),
    "lco_thermal_quiet": dict(cathode="LCO", temperature=True, t_conduction="difference", phi_s_form="difference"),
    "nmc_iso_sei_quiet": dict(cathode="NMC", temperature=False, aging=True, phi_s_form="difference"),      # config C5's model
    "lgm50_thermal": dict(cathode="LGM50", temperature=True),      # ... with temperature = true, the reference default of that chemistry (params.jl:695)
    # other discretisations (reference src/params.jl:119-136); the name suffix is _g<N_p>_<N_s>_<N_n>_<N_r>
    "lco_iso_g12_7_9_11": dict(cathode="LCO", Np=12, Ns=7, Nn=9, Nrp=11, Nrn=11),
    "nmc_iso_sei_g6_5_8_13": dict(cathode="NMC", aging=True, Np=6, Ns=5, Nn=8, Nrp=13, Nrn=13),
    # temperature = true on another grid: _g<N_p>_<N_s>_<N_n>_<N_r>_<N_a>_<N_z>
    "lco_thermal_g8_6_7_11_5_7": dict(cathode="LCO", temperature=True, Np=8, Ns=6, Nn=7, Nrp=11, Nrn=11, Na=5, Nz=7),
    # N_r_p != N_r_n (params.jl:124-136: independent options): _rn<N_r_n> appended
    "lco_iso_g7_6_8_12_rn10": dict(cathode="LCO", Np=7, Ns=6, Nn=8, Nrp=12, Nrn=10),
    "lco_thermal_g8_6_7_11_5_7_rn13": dict(cathode="LCO", temperature=True, Np=8, Ns=6, Nn=7, Nrp=11, Nrn=13, Na=5, Nz=7),
}


def theta_keys_for(model):
    """theta entries read by the residual or the initial guess, sorted by code point (== Julia Symbol sort,
    generate_functions.jl:387)."""
    class Rec(dict):
        def __init__(self, d): super().__init__(d); self.used = set()
        def __getitem__(self, k): self.used.add(k); return super().__getitem__(k)
    th = Rec(model.theta)
    dm.initial_guess(model, 0.5, th)
    keys = set(th.used) | set(dm.used_theta_keys(model))
    return sorted(keys)


def _emit_block(name, args, exprs, out_name, printer, lines, pre_defs=()):
    """CSE + straight-line C for a list of expressions written to out_name[k]; pre_defs = named intermediates that stay named (emitted first, as written)"""
    repl, red = sp.cse(exprs, symbols=sp.numbered_symbols("x"), optimizations=None, order="none")
    lines.append("void %s(%s)\n{" % (name, args))
    for s, e in pre_defs:
        lines.append("  const double %s = %s;" % (s, printer.doprint(e)))
    for s, e in repl:
        lines.append("  const double %s = %s;" % (s, printer.doprint(e)))
    for k, e in enumerate(red):
        lines.append("  %s[%d] = %s;" % (out_name, k, printer.doprint(e)))
    lines.append("}\n")


def generate(name, verbose=True):
    t0 = time.time()
    model = dm.Model(**VARIANTS[name])
    lay = model.lay
    N, Nd = lay.N, lay.N_diff
    keys = theta_keys_for(model)
    P = len(keys)
    Y = [sp.Symbol("Y[%d]" % i, real=True) for i in range(N)]
    YP = [sp.Symbol("YP[%d]" % i, real=True) for i in range(N)]
    TH = [sp.Symbol("th[%d]" % i, real=True) for i in range(P)]
    cj = sp.Symbol("cj", real=True)
    soc = sp.Symbol("SOC", real=True)
    th = dict(model.theta)
    for k, s in zip(keys, TH):
        th[k] = s
    ops = dm.SymOps()
    res = dm.residual(model, ops, Y, YP, th, with_control=False)[: N - 1]
    res = [sp.sympify(r) for r in res]
    if verbose:
        print("[%s] traced residual, N=%d P=%d  (%.1fs)" % (name, N, P, time.time() - t0))

    ysym = {s: i for i, s in enumerate(Y)}
    ypsym = {s: i for i, s in enumerate(YP)}
    aux = getattr(ops, "aux_defs", {})                    # named intermediates (dfn_model.SymOps.aux): chain rule below, substituted back before printing
    daux = {a: {t: sp.diff(e, t) for t in e.free_symbols if t in ysym} for a, e in aux.items()}
    # sparse Jacobian, column-major (CSC) like the reference's SparseMatrixCSC
    entries = {}   # (row, col) -> expr
    for r, e in enumerate(res):
        fs = e.free_symbols
        for s in fs:
            if s in ysym:
                d = sp.diff(e, s)
                if d != 0:
                    entries[(r, ysym[s])] = entries.get((r, ysym[s]), 0) + d
            elif s in ypsym:
                d = sp.diff(e, s)
                if d != 0:
                    entries[(r, ypsym[s])] = entries.get((r, ypsym[s]), 0) + cj * d
            elif s in aux:
                d = sp.diff(e, s)
                if d != 0:
                    for t, dt in daux[s].items():
                        entries[(r, ysym[t])] = entries.get((r, ysym[t]), 0) + d * dt
    # intermediates that stay named in the RESIDUAL code (dT_k: the temperature differences of Model.t_conduction = "difference" -- substituting them back would let sympy
    # merge (T[k+1] - T[k]) - (T[k] - T[k-1]) into T[k-1] - 2 T[k] + T[k+1], the very cancellation they avoid); the Jacobian entries take all of them back
    kept = {a: e for a, e in aux.items() if str(a).startswith(("dT_", "dPs_"))}          # (dPs_k: the potential differences of Model.phi_s_form = "difference", for the same reason)
    kept_defs = sorted(kept.items(), key=lambda ae: (str(ae[0]).split("_")[0], int(str(ae[0]).split("_")[1])))
    if aux:
        sub = {a: e for a, e in aux.items() if a not in kept}
        res = [e.xreplace(sub) for e in res]
        entries = {k: sp.sympify(v).xreplace(aux) for k, v in entries.items()}
    cols = [[] for _ in range(N)]
    for (r, c) in entries:
        cols[c].append(r)
    colptr = [0]
    rowval = []
    nzexpr = []
    for c in range(N):
        for r in sorted(cols[c]):
            rowval.append(r)
            nzexpr.append(entries[(r, c)])
        colptr.append(len(rowval))
    Z = len(rowval)
    # algebraic block: rows Nd..N-2, cols Nd..N-1  (generate_functions.jl:318-325)
    a_colptr = [0]
    a_rowval = []
    a_expr = []
    for c in range(Nd, N):
        for r in sorted(cols[c]):
            if r >= Nd:
                a_rowval.append(r - Nd)
                a_expr.append(entries[(r, c)])
        a_colptr.append(len(a_rowval))
    if verbose:
        print("[%s] jacobian: nnz=%d alg nnz=%d  (%.1fs)" % (name, Z, len(a_rowval), time.time() - t0))

    pr = _Printer()
    L = []
    L.append("/* GENERATED by oracle/codegen.py from oracle/dfn_model.py -- ORACLE (test infrastructure), do not edit.")
    L.append(" * variant %s: N=%d N_diff=%d nnz(J_y)=%d nnz(J_y_alg)=%d P=%d */" % (name, N, Nd, Z, len(a_rowval), P))
    L.append("#include <math.h>\n#ifndef M_PI\n#define M_PI 3.14159265358979323846\n#endif\n")
    pre = "orc_" + name
    L.append("const int %s_N = %d, %s_NDIFF = %d, %s_NNZ = %d, %s_NNZ_ALG = %d, %s_P = %d;" % (pre, N, pre, Nd, pre, Z, pre, len(a_rowval), pre, P))
    L.append("const int %s_colptr[%d] = {%s};" % (pre, N + 1, ",".join(map(str, colptr))))
    L.append("const int %s_rowval[%d] = {%s};" % (pre, Z, ",".join(map(str, rowval))))
    L.append("const int %s_alg_colptr[%d] = {%s};" % (pre, len(a_colptr), ",".join(map(str, a_colptr))))
    L.append("const int %s_alg_rowval[%d] = {%s};" % (pre, max(1, len(a_rowval)), ",".join(map(str, a_rowval)) or "0"))
    L.append("const char* const %s_theta_keys[%d] = {%s};\n" % (pre, P, ",".join('"%s"' % k for k in keys)))
    sig = "double* out, const double* Y, const double* YP, const double* th"
    _emit_block(pre + "_f_diff", sig, res[:Nd], "out", pr, L, kept_defs)
    _emit_block(pre + "_f_alg", sig, res[Nd:], "out", pr, L, kept_defs)
    sigj = "double* nz, const double* Y, const double* YP, double cj, const double* th"
    _emit_block(pre + "_jac", sigj, nzexpr, "nz", pr, L)
    _emit_block(pre + "_jac_alg", sigj, a_expr, "nz", pr, L)
    # initial guess (states_definition.jl:80-121): Y0[0..N-2] as a function of SOC and theta
    class SymTh(dict):
        pass
    y0 = dm.initial_guess_generic(model, ops, soc, th)[: N - 1]
    _emit_block(pre + "_initial_guess", "double* out, double SOC, const double* th", [sp.sympify(v) for v in y0], "out", pr, L)

    extra = {}
    if lay.temperature:
        # dT control row  value - sum(w_i YP_T_i)/L  (input_methods.jl:182-189) is linear with constant
        # coefficients; its algebraic twin substitutes YP_T -> rhs_T(Y) (scalar_residual.jl:347-372):
        w, Ltot = dm.temperature_weights(th, lay)
        rhsT = [res[lay.T[0] + i] + YP[lay.T[0] + i] for i in range(lay.T[1] - lay.T[0])]
        twin = -sum(w[i] * rhsT[i] for i in range(len(w))) / Ltot      # + value added by the caller
        tw_cols, tw_expr = [], []
        twin_full = twin.xreplace(kept) if kept else twin
        for c in range(Nd, N):
            d = sp.diff(twin_full, Y[c])
            if d != 0:
                tw_cols.append(c - Nd)
                tw_expr.append(d)
        L.append("const int %s_NNZ_DT_TWIN = %d;" % (pre, len(tw_cols)))
        L.append("const int %s_dT_twin_cols[%d] = {%s};" % (pre, len(tw_cols), ",".join(map(str, tw_cols))))
        _emit_block(pre + "_dT_twin", sig, [twin], "out", pr, L, kept_defs)
        _emit_block(pre + "_dT_twin_jac", sigj, tw_expr, "nz", pr, L)
        L.append("void %s_dT_weights(double* w, const double* th)\n{" % pre)
        for i in range(len(w)):
            L.append("  w[%d] = %s;" % (i, pr.doprint(sp.sympify(w[i]) / Ltot)))
        L.append("}\n")
        extra["dT_twin_cols"] = tw_cols

    with open(os.path.join(HERE, "gen", name + ".c"), "w") as f:
        f.write("\n".join(L))
    meta = dict(name=name, N=N, N_diff=Nd, nnz=Z, nnz_alg=len(a_rowval), P=P, theta_keys=keys, colptr=colptr,
                rowval=rowval, alg_colptr=a_colptr, alg_rowval=a_rowval,
                theta_default=[float(model.theta[k]) for k in keys], **extra)
    with open(os.path.join(HERE, "gen", name + ".json"), "w") as f:
        json.dump(meta, f, ensure_ascii=False)
    if verbose:
        print("[%s] wrote gen/%s.c (%.1fs)" % (name, name, time.time() - t0))
    return meta


if __name__ == "__main__":
    which = sys.argv[1:] or ["lco_iso"]
    if which == ["all"]:
        which = list(VARIANTS)
    for w in which:
        generate(w)

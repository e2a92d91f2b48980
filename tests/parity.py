"""Parity checks of the C-ABI library against the oracle, shared by the emulator (CPU) and HIP (GPU) test modules."""
import ctypes as C

import numpy as np

SECTIONS = [("c_e", 0, 30), ("c_s", 30, 230), ("j", 230, 250), ("Phi_e", 250, 280), ("Phi_s", 280, 300), ("I", 300, 301)]
SECTIONS_SEI = [("c_e", 0, 30), ("c_s", 30, 230), ("film", 230, 240), ("SOH", 240, 241), ("j", 241, 261), ("Phi_e", 261, 291),
                ("Phi_s", 291, 311), ("j_s", 311, 321), ("I", 321, 322)]


SECTIONS_THERMAL = [("c_e", 0, 30), ("c_s", 30, 230), ("T", 230, 280), ("j", 280, 300), ("Phi_e", 300, 330), ("Phi_s", 330, 350), ("I", 350, 351)]


SECTIONS_QUAD = [("c_e", 0, 30), ("c_s", 30, 50), ("j", 50, 70), ("Phi_e", 70, 100), ("Phi_s", 100, 120), ("I", 120, 121)]
SECTIONS_POLY = [("c_e", 0, 30), ("c_s", 30, 50), ("Q", 50, 70), ("j", 70, 90), ("Phi_e", 90, 120), ("Phi_s", 120, 140), ("I", 140, 141)]


def grid_sections(Np, Ns, Nn, Nr, sei=False, thermal=None, Nrn=None):
    """state sections of a Fickian model on another discretisation (reference state layout, src/external.jl:275-365); thermal = (N_a, N_z) with temperature = true"""
    out, o = [], 0
    for name, n in (("c_e", Np + Ns + Nn), ("c_s", Np * Nr + Nn * (Nrn or Nr))) + ((("T", thermal[0] + Np + Ns + Nn + thermal[1]),) if thermal else ()) + ((("film", Nn), ("SOH", 1)) if sei else ()) + (("j", Np + Nn), ("Phi_e", Np + Ns + Nn), ("Phi_s", Np + Nn)) + \
                   ((("j_s", Nn),) if sei else ()) + (("I", 1),):
        out.append((name, o, o + n)); o += n
    return out


# (keyed by the number of states: every model the tests build has its own)
SECTION_TABLES = {301: SECTIONS, 322: SECTIONS_SEI, 351: SECTIONS_THERMAL, 121: SECTIONS_QUAD, 141: SECTIONS_POLY,
                  330: grid_sections(12, 7, 9, 11), 266: grid_sections(6, 5, 8, 13, sei=True), 271: grid_sections(8, 6, 7, 11, thermal=(5, 7)),
                  237: grid_sections(7, 6, 8, 12, Nrn=10), 285: grid_sections(8, 6, 7, 11, thermal=(5, 7), Nrn=13)}


def sections_for(n_states):
    return SECTION_TABLES[n_states]


def realistic_states(O, th, n, seed=0, variant="lco_iso"):
    """states along a 1C discharge + random perturbations (so that every term of the equations is exercised)."""
    rng = np.random.default_rng(seed)
    if "_thermal" in variant:  # a 3C charge heats the cell: non-trivial T(x) and heat sources
        ro = O.simulate(variant, th, 0.1, [dict(mode=O.MODE_I, value=3.0, tf=200.0 * (1 + 3 * rng.random()))])
    elif "_sei" in variant:    # the side reaction is active only while charging (residuals.jl:519-552)
        ro = O.simulate(variant, th, 0.1, [dict(mode=O.MODE_I, value=1.0, tf=600.0 * (1 + 3 * rng.random()))])
    else:
        ro = O.simulate(variant, th, 1.0, [dict(mode=O.MODE_I, value=-1.0, tf=600.0 * (1 + 4 * rng.random()))])
    Ys, YPs = [], []
    for _ in range(n):
        Ys.append(ro["Y"] * (1 + 1e-3 * rng.standard_normal(ro["Y"].size)))
        YPs.append(ro["YP"] * (1 + 1e-2 * rng.standard_normal(ro["Y"].size)))
    return np.array(Ys), np.array(YPs)


def check_keys_and_pattern(p, O):
    VARIANT = p.variant
    meta = O.meta(VARIANT)
    assert p.θ_keys == meta["theta_keys"]
    assert np.array_equal(p.theta_vector(), np.array(meta["theta_default"]))
    th = p.theta_vector()
    N = p.N.tot
    Z = meta["nnz"] + 1                                          # Z = 2139 in CC mode (+130 with SEI; 2883 thermal); 519 / 579 quadratic / polynomial
    if p.variant in ("lco_iso", "nmc_iso", "lco_iso_nu", "lco_iso_mhc"):
        assert Z == 2139
    if p.variant in ("lco_iso_sei", "nmc_iso_sei"):
        assert Z == 2269
    if p.variant == "lco_thermal":
        assert Z == 2883
    nT = (p.ind["T"].stop - p.ind["T"].start) if p.temperature else 0      # the dT control row has one entry per temperature node (2883 - 1 + 50 = 2932 on the default grid)
    for mode, nnz_expect in ((0, Z), (1, Z + 1), (3, Z + 2), (4, Z + 1)) + (((2, Z - 1 + nT),) if p.temperature else ()):
        cp, ri = p.jac_pattern(mode)
        ocp, ori, _ = O.jacobian(VARIANT, th, np.ones(N), np.zeros(N), 1.0, mode, 0.0)
        assert len(ri) == nnz_expect
        assert np.array_equal(cp, ocp) and np.array_equal(ri, ori)
    check_alg_block_pattern(p, O)


def check_alg_block_pattern(p, O):
    """seam 1's J_y_alg! (generate_functions.jl:318-325) is the block J[N_diff:N-1, N_diff:N] of the full Jacobian at gamma = 0: its pattern,
    extracted from plh_jac_pattern, must equal the one the oracle's symbolic pipeline produced for J_y_alg (245 nz for C1 with the CC row)."""
    meta = O.meta(p.variant)
    Nd, N = p.N.diff, p.N.tot
    cp, ri = p.jac_pattern(0)
    acp, ari = [0], []
    for c in range(Nd, N):
        rows = [int(r) - Nd for r in ri[cp[c]:cp[c + 1]] if r >= Nd and r < N - 1]     # generated rows only (the control row is separate)
        ari += rows; acp.append(len(ari))
    assert acp == list(meta["alg_colptr"]) and ari == list(meta["alg_rowval"])
    assert len(ari) == meta["nnz_alg"]


def check_evaluators(p, O, n_cells=3, solve_tol=1e-8):
    VARIANT = p.variant
    lib, h = p._lib, p._h
    th = p.theta_vector()
    N = p.N.tot
    Y, YP = realistic_states(O, th, n_cells, variant=VARIANT)
    Th = np.tile(th, (n_cells, 1))
    Th[:, p.θ_keys.index("D_sp")] *= np.linspace(0.5, 2.0, n_cells)
    Th[:, p.θ_keys.index("k_n")] *= np.linspace(2.0, 0.5, n_cells)
    Th[1:, p.θ_keys.index("T₀")] = 310.0          # exercises the Arrhenius / dU/dT branches (T0 != T_ref)
    Th = np.ascontiguousarray(Th)
    # initial guess
    soc = np.linspace(0.1, 0.9, n_cells)
    Yg = np.zeros((n_cells, N))
    assert lib.plh_initial_guess(h, n_cells, Th.ctypes.data, soc.ctypes.data, Yg.ctypes.data, 0, None) == 0
    for i in range(n_cells):
        assert np.allclose(Yg[i], O.initial_guess(VARIANT, Th[i], soc[i]), rtol=1e-12, atol=0)
    for mode, val in ((0, -1.0), (1, 3.9), (3, -80.0), (4, 0.05)) + (((2, 0.01),) if p.temperature else ()):   # I, V, P, eta_p (, dT)
        F = np.zeros((n_cells, N))
        assert lib.plh_residual(h, n_cells, Th.ctypes.data, Y.ctypes.data, YP.ctypes.data, mode, val, F.ctypes.data, 0, None) == 0
        nnz = len(p.jac_pattern(mode)[1])
        cj = 0.37
        nz = np.zeros((n_cells, nnz))
        assert lib.plh_jacobian(h, n_cells, Th.ctypes.data, Y.ctypes.data, YP.ctypes.data, cj, mode, nz.ctypes.data, 0, None) == 0
        rng = np.random.default_rng(1)
        b = rng.standard_normal((n_cells, N))
        x = b.copy()
        assert lib.plh_linear_solve(h, n_cells, Th.ctypes.data, Y.ctypes.data, YP.ctypes.data, cj, mode, x.ctypes.data, 0, None) == 0
        for i in range(n_cells):
            Fo = O.residual(VARIANT, Th[i], Y[i], YP[i], mode, val)
            # rounding-level criterion: |dF_i| <= 1e-12 * (magnitude of the terms entering row i) = sum_k |J_ik Y_k| + |YP_i|
            ocp, ori_, onz_ = O.jacobian(VARIANT, Th[i], Y[i], YP[i], 0.0, mode, val)
            _, _, onz1 = O.jacobian(VARIANT, Th[i], Y[i], YP[i], 1.0, mode, val)     # J(cj=1) - J(cj=0) = dF/dYP
            term = np.zeros(N)
            term[-1] = abs(val)
            for c in range(N):
                sl = slice(ocp[c], ocp[c + 1])
                np.add.at(term, ori_[sl], np.abs(onz_[sl] * Y[i, c]) + np.abs((onz1[sl] - onz_[sl]) * YP[i, c]))
            bad = np.abs(F[i] - Fo) > 1e-12 * term + 1e-300
            assert not bad.any(), (mode, i, np.nonzero(bad)[0][:5], np.abs(F[i] - Fo)[bad][:5], term[bad][:5])
            _, ori, onz = O.jacobian(VARIANT, Th[i], Y[i], YP[i], cj, mode, val)
            rel = np.abs(nz[i] - onz) / (np.abs(onz) + 1e-300)
            assert rel.max() < 1e-9, (mode, i, rel.max(), int(ori[rel.argmax()]))
            xo = O.linear_solve(VARIANT, Th[i], Y[i], YP[i], cj, b[i], mode, val)
            for name, a, e in sections_for(N):
                # measured against an extended-precision solve (check_solver_accuracy): both solvers are within 3e-9 of the truth (isothermal: the
                # structured solve 1e-11, the oracle's LU 3e-9); thermal 2e-9 each = the conditioning floor of the fp64 Jacobian entries themselves
                assert np.abs(x[i, a:e] - xo[a:e]).max() <= solve_tol * (np.abs(xo[a:e]).max() + 1e-300), (mode, i, name)


def check_init(p, O, V0_expected=2.863495104606893):
    VARIANT = p.variant
    lib, h = p._lib, p._h
    th = p.theta_vector()
    N = p.N.tot
    Y0 = O.initial_guess(VARIANT, th, 0.0)
    Y0[-1] = 2.0
    rc, Yo, YPo, ito = O.init_consistent(VARIANT, th, Y0, O.MODE_I, 2.0)
    Y, YP = Y0.copy(), np.zeros(N)
    st, it = np.zeros(1, np.int32), np.zeros(1, np.int32)
    assert lib.plh_init_consistent(h, 1, th.ctypes.data, 0, 2.0, 1e-3, Y.ctypes.data, YP.ctypes.data, st.ctypes.data, it.ctypes.data, 0, None) == 0
    assert st[0] == 0 and it[0] == ito
    assert np.abs(Y - Yo).max() <= 1e-12 * np.abs(Yo).max()
    # the finite-difference estimate of YP_alg is intrinsically noisy (difference quotient of a Newton update): 1e-6 of scale
    assert np.abs(YP - YPo).max() <= 1e-6 * np.abs(YPo).max()
    ps, pe = dict((n, (a, e)) for n, a, e in sections_for(N))["Phi_s"]
    V0 = Y[ps] - Y[pe - 1]
    if V0_expected is not None:
        assert abs(V0 - V0_expected) < 1e-10       # reference examples/model_inputs_and_outputs.ipynb:152
    return V0


def runs_to_oracle(O, p, pkg, protocol, cell=None, n_cells=None):
    """the protocol as oracle run dicts; with per-cell inputs (arrays of length n_cells) the values of cell `cell`"""
    runs, _ = pkg.make_protocol(p, protocol, n_cells)
    out = []
    for r in runs:
        b = O.Bounds(**{f: getattr(r.bounds, f) for f in O.BOUND_FIELDS})
        d = dict(mode=r.mode, value_kind=r.value_kind, value=r.value_cell[cell] if r.value_cell else r.value,
                 tf=r.tf_cell[cell] if r.tf_cell else r.tf, bounds=b, dstate=r.dstate)
        if r.value_kind == 3:       # PLH_VAL_TABLE
            d["table"] = (np.array(r._keep[0]), np.array(r._keep[1]))
        if r.value_kind == 4:       # PLH_VAL_EXPR
            d["expr"] = (np.array(r._keep[0]), np.array(r._keep[1]))
            if r.n_dcol > 0:              # derivative programs of a closure of the state (behind the main program in the same arrays)
                d["n_main"] = int(r.n_tab); d["dcol"] = np.array(r._keep[2]); d["dofs"] = np.array(r._keep[3])
        out.append(d)
    return out


def state_rel_err(Y, Yo):
    """the parity metric for state vectors: max over the state sections of  max|dY| / max|Y_oracle|  (each physical field is
    compared relative to its own scale; a component-wise ratio would blow up on near-zero entries such as j in the separator-side
    nodes or Phi_e next to the reference node)."""
    worst = 0.0
    for name, a, e in sections_for(len(Yo)):
        # a field that is identically zero in exact arithmetic (j_s while not charging, I at rest) holds only solver round-off
        # (~1e-23 in the oracle's sparse LU, exactly 0 on the device): floor its scale well below any physical magnitude
        floor = {"j_s": 1e-13, "j": 1e-12, "I": 1e-9, "film": 1e-14}.get(name, 1e-300)
        worst = max(worst, np.abs(Y[a:e] - Yo[a:e]).max() / max(np.abs(Yo[a:e]).max(), floor))
    return worst


def compare_trajectory(ens, i, ro, rtol_state=1e-6, same_decisions=True):
    info = ens.run_info[i]
    for k, rr in enumerate(ro["runs"]):
        assert info[k]["flag"] == rr["flag"], (i, k, info[k], rr)
        if same_decisions:
            assert info[k]["iterations"] == rr["iterations"], (i, k, info[k], rr)
        assert abs(info[k]["t_end"] - rr["t_end"]) <= rtol_state * max(1.0, rr["t_end"]), (i, k, info[k]["t_end"], rr["t_end"])
    n = int(ens.n_pts[i])
    if same_decisions:
        assert n == len(ro["t"])
        assert np.abs(ens.t[i, :n] - ro["t"]).max() <= 10 * rtol_state * max(1.0, ro["t"][-1])
        # V is compared at equal step index; the step times themselves agree only to ~rtol_state * t (the step-size controller is a
        # continuous function of rounding-level differences), so near a voltage knee the comparison allows the first-order
        # effect of that time offset: |dV| <= 10 rtol * 4 V + 2 |dV/dt| |dt|
        dt = np.abs(ens.t[i, :n] - ro["t"])
        slope = np.zeros(n)
        if n > 2:
            dtt = np.diff(ro["t"])
            sl = np.abs(np.diff(ro["V"])) / np.where(dtt > 0, dtt, np.inf)     # run boundaries repeat t: no slope there
            slope[1:] = sl; slope[:-1] = np.maximum(slope[:-1], sl)
        assert (np.abs(ens.V[i, :n] - ro["V"]) <= 10 * rtol_state * 4.0 + 2 * slope * dt).all()
        for f in ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail"):
            assert ens.counters[i][f] == ro["counters"][f], f
    assert state_rel_err(ens.Y[i], ro["Y"]) <= rtol_state, (i, state_rel_err(ens.Y[i], ro["Y"]))


# ---- which solver is closer to the truth?   ----
def ld_solve(A, b):
    """dense LU with partial pivoting in 80-bit extended precision (numpy longdouble, eps 1.1e-19) + two refinement steps: the reference solution
    of J x = b for the fp64 Jacobian entries in A"""
    A = A.astype(np.longdouble); b = b.astype(np.longdouble)
    n = A.shape[0]
    LU = A.copy(); piv = np.arange(n)
    for k in range(n):
        q = k + int(np.argmax(np.abs(LU[k:, k])))
        if q != k:
            LU[[k, q]] = LU[[q, k]]; piv[[k, q]] = piv[[q, k]]
        LU[k + 1:, k] /= LU[k, k]
        LU[k + 1:, k + 1:] -= np.outer(LU[k + 1:, k], LU[k, k + 1:])

    def sub(r):
        y = r[piv].copy()
        for k in range(n):
            y[k + 1:] -= LU[k + 1:, k] * y[k]
        for k in range(n - 1, -1, -1):
            y[k] /= LU[k, k]
            y[:k] -= LU[:k, k] * y[k]
        return y
    x = sub(b)
    for _ in range(2):
        x = x + sub(b - A @ x)
    return x.astype(np.float64)


def dense_from_csc(N, cp, ri, nz):
    A = np.zeros((N, N), dtype=np.longdouble)
    for c in range(N):
        A[ri[cp[c]:cp[c + 1]], c] = nz[cp[c]:cp[c + 1]]
    return A


def section_err(x, xt):
    return max(np.abs(x[a:e] - xt[a:e]).max() / (np.abs(xt[a:e]).max() + 1e-300) for _, a, e in sections_for(len(xt)))


def solver_accuracy_rows(p, O, n_states=3, refine=0, modes=((0, -1.0), (1, 3.9)), cjs=(0.37, 25.0)):
    """rows (mode, cj, state, device-vs-truth, oracle-vs-truth, device-vs-oracle): plh_linear_solve[_refined] and the oracle's sparse LU against ld_solve"""
    th = p.theta_vector(); N = p.N.tot
    Y, YP = realistic_states(O, th, n_states, variant=p.variant)
    rng = np.random.default_rng(1)
    rows = []
    for mode, val in modes:
        for cj in cjs:
            for i in range(n_states):
                b = rng.standard_normal(N)
                cp, ri, nz = O.jacobian(p.variant, th, Y[i], YP[i], cj, mode, val)
                xt = ld_solve(dense_from_csc(N, cp, ri, nz), b)
                xo = O.linear_solve(p.variant, th, Y[i], YP[i], cj, b, mode, val, refine=refine)
                xd = b[None, :].copy()
                Thm, Yi, YPi = np.ascontiguousarray(th[None, :]), np.ascontiguousarray(Y[i][None, :]), np.ascontiguousarray(YP[i][None, :])
                assert p._lib.plh_linear_solve_refined(p._h, 1, Thm.ctypes.data, Yi.ctypes.data, YPi.ctypes.data, cj, mode, xd.ctypes.data, refine, 0, None) == 0
                rows.append((mode, cj, i, section_err(xd[0], xt), section_err(xo, xt), section_err(xd[0], xo)))
    return rows


def check_solver_accuracy(p, O):
    """the structured device solve is at least as close to the exact solution as the sparse LU it is compared with; one refinement step brings both to
    the rounding level of the fp64 residual (isothermal models), where they agree with each other to 1e-11"""
    r0 = np.array([r[3:] for r in solver_accuracy_rows(p, O, refine=0)])
    r1 = np.array([r[3:] for r in solver_accuracy_rows(p, O, refine=1)])
    floor = 1e-8 if p.temperature else 2e-11            # thermal: the T rows (conduction coefficients ~1e8 1/s) put the conditioning floor of the entries at ~2e-9
    assert r0[:, 0].max() <= 1e-8 and r0[:, 1].max() <= 1e-8, r0.max(axis=0)
    assert np.median(r0[:, 0]) <= 2.0 * np.median(r0[:, 1]) + 1e-12, (np.median(r0[:, 0]), np.median(r0[:, 1]))     # the device is not the less accurate of the two
    assert r1[:, 0].max() <= floor and r1[:, 1].max() <= floor and r1[:, 2].max() <= floor, r1.max(axis=0)
    return r0, r1


def oracle_noise_band(O, variant, th, soc, runs, opts_kw=None, seeds=6, eps=2.2e-16):
    """reproducibility floor of the reference algorithm for one cell: the oracle re-run with last-bit perturbations of the shifted state of the
    finite-difference estimate of YP_alg (orc_opts.fd_perturb): max state deviation from the unperturbed run over `seeds` perturbations"""
    kw = dict(opts_kw or {})
    r0 = O.simulate(variant, th, soc, runs, opts=O.default_opts(**kw))
    band = 0.0
    for seed in range(1, seeds + 1):
        rk = O.simulate(variant, th, soc, runs, opts=O.default_opts(fd_perturb=eps, perturb_seed=seed, **kw))
        band = max(band, state_rel_err(rk["Y"], r0["Y"]))
    return r0, band


# ---- tight-tolerance parity: the regime where the reproducibility floor of the default tolerances vanishes (VERDICT r02 "next" 1) ----
TIGHT = dict(reltol=1e-8, abstol=1e-10)


class RunFails(AssertionError):
    """device or oracle did not complete the protocol at the requested tolerances (`who` = "oracle" / "device")"""
    def __init__(self, who, detail):
        super().__init__("%s fails on this protocol at these tolerances: %r" % (who, detail))
        self.who = who


def _tight_tstops(ro, runs, sample_dt, fine_dt=0.05, fine_span=1.0):
    """run-local stop times (opts.tstops, model_evaluation.jl:292-294) of the second pass: a coarse grid over the longest leg -- the equal times at which the two
    trajectories are compared -- and a fine grid around the run-local end time of every leg that ended on a BOUND in the first pass.  The reference replaces the
    last point of such a run by a LINEAR interpolation between the last two accepted points (interp_final_points!, model_evaluation.jl:369-382): its end state
    depends on where those two points fall (h^2 y''/8: 1e-5 of Phi_s at a voltage knee with the ~5 s steps of reltol 1e-8, and the step grids of two
    implementations do differ at tight tolerances); the fine grid makes both bracket the crossing within the same 0.05 s."""
    ts, t0, longest = [], 0.0, 0.0
    for rr in ro["runs"]:
        loc = rr["t_end"] - t0
        longest = max(longest, loc)
        if rr["flag"] > 0:
            ts.append(loc + np.arange(-fine_span, fine_span + fine_dt / 2, fine_dt))
        else:
            # a run that reaches tf: bounds are tested at accepted steps only, and not at t = tf itself (check_simulation_stop!, checks.jl:1-10) -- whether a bound that is
            # crossed within the last step before tf fires depends on where that step starts.  The same fine grid makes it the same 0.05 s for both.
            ts.append(loc + np.arange(-fine_span, fine_dt / 2, fine_dt))
        t0 = rr["t_end"]
    ts.append(np.arange(sample_dt, longest + fine_span, sample_dt))
    ts = np.unique(np.round(np.concatenate(ts), 9))
    return ts[ts > 1.5]                  # (a continuation run has its own tstop at 1 s)


def _quad_interp(tk, Yk, t):
    """Lagrange interpolation through three points (rows of Yk at times tk) at time t"""
    (a, b, c) = tk
    return Yk[0] * ((t - b) * (t - c) / ((a - b) * (a - c))) + Yk[1] * ((t - a) * (t - c) / ((b - a) * (b - c))) + Yk[2] * ((t - a) * (t - b) / ((c - a) * (c - b)))


def tight_compare(pkg, p, O, th, soc, protocol, sample_dt=50.0, tol=None, max_points=40000, extra_opts=None, variant=None):
    """ONE cell, device and oracle BOTH at reltol 1e-8 / abstol 1e-10 (`tol`).  Pass 1 (oracle alone) locates the leg ends; pass 2 runs both with the same opts.tstops
    (_tight_tstops) and outputs = :all.  Every deviation is relative to the scale of its field over the whole trajectory (max |field| over the oracle's saved points).
    Returns
      traj     max over the common stop times and the state sections of |Y_dev(t) - Y_orc(t)| / scale        (state trajectories at equal times)
      V        max over the common stop times of |V_dev - V_orc| / |V_orc|
      legs     per run: (flag_dev, flag_orc, t_end_dev, t_end_orc, end-state deviation), the last one measured against the oracle's trajectory at the DEVICE's own end
               time (quadratic interpolation of the oracle's saved states on the fine stop grid): the end state of a run cannot be compared at unequal times
      n_times  number of common stop times;   worst = (state section, run index, run-local time) of `traj`;   by_field[section] = [its part of `traj`, its scale]"""
    tol = dict(tol or TIGHT)
    runs = runs_to_oracle(O, p, pkg, protocol)
    okw = dict(maxiters=120000, **tol, **(extra_opts or {}))
    variant = variant or p.variant                       # (an oracle variant of the same model, e.g. lco_thermal_tdiff)
    r1 = O.simulate(variant, th, soc, runs, opts=O.default_opts(**okw), max_out=max_points)
    if min(r["flag"] for r in r1["runs"]) < 0:
        raise RunFails("oracle", [(r["flag"], r["iterations"], r["t_end"]) for r in r1["runs"]])
    ts = _tight_tstops(r1, runs, sample_dt)
    ro = O.simulate(variant, th, soc, runs, opts=O.default_opts(tstops=ts, **okw), max_out=max_points, keep_Y=True)
    if min(r["flag"] for r in ro["runs"]) < 0:
        raise RunFails("oracle", [(r["flag"], r["iterations"], r["t_end"]) for r in ro["runs"]])
    o = pkg.Opts(); o.reltol = tol["reltol"]; o.abstol = tol["abstol"]; o.maxiters = 120000; o.tstops = list(ts)
    for k, v in (extra_opts or {}).items():
        setattr(o, k, v)
    ens = pkg.simulate_ensemble(p, np.ascontiguousarray(th[None, :]), protocol, SOC=soc, opts=o, max_points=max_points, outputs="all")
    info = ens.run_info[0]
    n = int(ens.n_pts[0])
    if min(int(f) for f in info["flag"]) < 0:
        raise RunFails("device", [(int(r["flag"]), int(r["iterations"]), float(r["t_end"])) for r in info])
    assert n < max_points and len(ro["t"]) < max_points, "output buffers too small for the tight-tolerance run"
    td, Vd, Yd = np.asarray(ens.t[0, :n]), np.asarray(ens.V[0, :n]), np.asarray(ens.Y_all[0, :n])
    to, Vo, Yo = ro["t"], ro["V"], ro["Y_all"]
    secs = sections_for(Yo.shape[1])
    scale = {name: max(np.abs(Yo[:, a:e]).max(), 1e-300) for name, a, e in secs}
    # saved points of run k: [start[k], start[k+1])   (run.info.iterations = points of the run)
    sd = np.concatenate([[0], np.cumsum([int(x) for x in info["iterations"]])]); so = np.concatenate([[0], np.cumsum([r["iterations"] for r in ro["runs"]])])
    traj, dV, ntimes, legs, worst = 0.0, 0.0, 0, [], None
    by_field = {name: [0.0, float(scale[name])] for name, a, e in secs}
    t0d = t0o = 0.0
    for k, rr in enumerate(ro["runs"]):
        a_d, e_d, a_o, e_o = int(sd[k]), int(sd[k + 1]), int(so[k]), int(so[k + 1])
        # run-local times of the accepted points (the last point of a run that ended on a bound is the back-interpolated one: not on the grid)
        ld, lo = td[a_d:e_d] - td[a_d], to[a_o:e_o] - to[a_o]
        common, ia, ib = np.intersect1d(np.round(ld[:-1], 6), np.round(lo[:-1], 6), return_indices=True)
        on_grid = np.isin(common, np.round(ts, 6))
        ia, ib = ia[on_grid] + a_d, ib[on_grid] + a_o
        for name, a, e in secs:
            if len(ia):
                dev = np.abs(Yd[ia, a:e] - Yo[ib, a:e]).max(axis=1) / scale[name]
                by_field[name][0] = max(by_field[name][0], float(dev.max()))
                if float(dev.max()) > traj:
                    traj, worst = float(dev.max()), (name, k, float(ld[ia[int(dev.argmax())] - a_d]))      # (section, run, run-local time of the largest deviation)
        if len(ia):
            dV = max(dV, float((np.abs(Vd[ia] - Vo[ib]) / np.abs(Vo[ib])).max()))
        ntimes += len(ia)
        # end state of the run against the oracle's trajectory at the device's own end time
        te_d, te_o = float(info[k]["t_end"]), rr["t_end"]
        Yend = Yd[e_d - 1]
        if int(info[k]["flag"]) == 0 or e_o - a_o < 5:
            Yref = Yo[e_o - 1]
        else:
            loc = te_d - td[a_d]                                       # device's end in run-local time
            grid = lo[:-1]
            j = int(np.clip(np.searchsorted(grid, loc), 2, len(grid) - 1))
            idx = [j - 2, j - 1, j] if j == len(grid) - 1 or abs(grid[j - 1] - loc) < abs(grid[j] - loc) or j + 1 >= len(grid) else [j - 1, j, j + 1]
            Yref = _quad_interp(grid[idx], Yo[a_o:e_o][idx], loc)
        end_err = max(float(np.abs(Yend[a:e] - Yref[a:e]).max() / scale[name]) for name, a, e in secs)
        legs.append((int(info[k]["flag"]), rr["flag"], te_d, te_o, end_err))
    return dict(tol=tol, traj=traj, worst=worst, by_field=by_field, V=dV, legs=legs, n_times=ntimes, steps=(int(ens.counters[0]["n_steps"]), ro["counters"]["n_steps"]), kernel_ms=ens.kernel_ms)


# ---- forward sensitivities: the oracle differenced (there is no reference vector for derivatives) ----
def oracle_fd_sens(O, variant, th, soc, runs, col, ts, rel_h=0.05, tol=None, max_out=40000):
    """d Y(t_end) / d theta[col] and d V(ts) / d theta[col] by SIXTH-order central differences of the oracle at tight tolerance, step rel_h * theta[col], all runs with the same
    stop times `ts` (so that V is compared at equal times and the adaptive step grids stay correlated).  What limits it is the integration error of the differenced runs
    (~1e-7 of the states at 1e-8) divided by the step: ~2e-6 of theta * |dY/dtheta| relative to the state -- sections that barely depend on the parameter are noise."""
    tol = dict(tol or TIGHT)
    h = rel_h * th[col]
    kw = dict(maxiters=400000, tstops=list(ts), **tol)

    def run(f):
        t2 = th.copy(); t2[col] += f * h
        r = O.simulate(variant, t2, soc, runs, opts=O.default_opts(**kw), max_out=max_out)
        if min(x["flag"] for x in r["runs"]) < 0:
            raise RunFails("oracle", [(x["flag"], x["t_end"]) for x in r["runs"]])
        return r
    rs = {f: run(f) for f in (3, 2, 1, -1, -2, -3)}
    d6 = lambda g: (g(rs[3]) / 60 - 3 * g(rs[2]) / 20 + 3 * g(rs[1]) / 4 - 3 * g(rs[-1]) / 4 + 3 * g(rs[-2]) / 20 - g(rs[-3]) / 60) / h

    def V_at(r):
        keep = np.concatenate([[True], np.diff(r["t"]) > 0])           # (a run boundary repeats its time: keep the first of the pair)
        return np.interp(ts, r["t"][keep], r["V"][keep])
    return d6(lambda r: r["Y"]), d6(V_at), [tuple(x["flag"] for x in r["runs"]) for r in rs.values()]


def sens_compare(O, p, pkg, ens, i, th, soc, protocol, keys, ts, variant=None, rel_h=0.05):
    """device sensitivities of cell i (ens from simulate_ensemble(..., sens=keys) with opts.tstops = ts) against oracle_fd_sens.  Returns per key: (dV error relative to
    max |dV/dtheta| over the stop times, {section: (dY error relative to the section's max |dY/dtheta|, theta |dY/dtheta| / |Y| of the section)})"""
    runs = runs_to_oracle(O, p, pkg, protocol)
    n = int(ens.n_pts[i]); td = np.asarray(ens.t[i, :n])
    idx = [int(np.argmin(np.abs(td - t))) for t in ts]
    assert np.abs(td[idx] - np.asarray(ts)).max() < 1e-6, "a stop time is not among the saved points"
    out = {}
    for k, key in enumerate(keys):
        col = p.θ_keys.index(key)
        dY, dV, _ = oracle_fd_sens(O, variant or p.variant, th, soc, runs, col, ts, rel_h=rel_h)
        dVd = np.asarray(ens.dV_dtheta[i, k])[idx]
        eV = (float(np.abs(dVd - dV).max() / np.abs(dV).max()), float(abs(th[col]) * np.abs(dV).max()))      # (error relative to max |dV/dtheta|, theta max |dV/dtheta| in volts)
        sec = {}
        Yd = np.asarray(ens.Y[i])
        for name, a, e in sections_for(len(dY)):
            sc = np.abs(dY[a:e]).max()
            if sc > 0:
                # (relative sensitivity against the OPERATING scale of the section: j, j_s, Phi_e and I relax to ~0 at rest, where a ratio to the state itself is noise over noise)
                floor = {"j": 1e-6, "j_s": 1e-9, "Phi_e": 1e-3, "I": 1e-3, "film": 1e-14}.get(name, 1e-300)
                sec[name] = (float(np.abs(np.asarray(ens.dY_dtheta[i, k, a:e]) - dY[a:e]).max() / sc), float(abs(th[col]) * sc / max(np.abs(Yd[a:e]).max(), floor)))
        out[key] = (eV, sec)
    return out


# ---- r05: the quiet oracle, the reference-order device variants, the stop function ----
def check_quiet_oracle_parity(p, O, pkg, n_cells=6, tol=1e-6, thermal_proto=False, min_same=0.98):
    """The device's default build against `<variant>_quiet` -- the oracle with the cancelling stencils evaluated on differences (oracle/codegen.py) -- through :hold legs, at the
    DEFAULT tolerances, per cell, no floor: exit flags equal in every cell, identical integrator decisions in (at least `min_same` of) the cells -- on the GPU 255 of 256 C3 cells,
    every isothermal one -- and for every such cell end states / run-end times within `tol` (measured: median 1e-12 on the emulator, whose sums round like the oracle's, 1e-8
    on the GPU, where fused multiply-adds round differently; a cell that takes another decision somewhere is within the integration tolerance).  (Against the plain oracle
    variants the same cells differ by 1e-6 ... 1e-2: their generated Phi_s rows are quantised at ulp(Phi_s) and IDA's start-up order selection in a :hold leg reads that
    rounding -- tests/test_oracle_golden.py::test_the_generated_phi_s_rows_are_quantised_and_the_notebook_shows_it)"""
    q = p.variant + "_quiet"
    if thermal_proto:
        cfg = pkg.configs.c3(p, 4096)
        Th = np.ascontiguousarray(cfg["theta"][:: 4096 // n_cells][:n_cells]); protos = [(cfg["protocol"], 0.0)]
        kw = dict(T_max=400.0, V_max=5.0, I_max=10.0, I_min=0.0, SOC_max=2.0)
        protos.append(([dict(I=4.0, tf=300.0, **kw), dict(dT="hold", tf=200.0, **kw), dict(V="hold", tf=300.0, **kw)], 0.0))
    else:
        Th = np.ascontiguousarray(pkg.configs.sweep_theta(p, np.arange(n_cells), 4))
        protos = [([dict(I=2.0, tf=900.0, V_max=5.0), dict(V="hold", tf=600.0, V_max=5.0, I_min=0.0), dict(P="hold", tf=100.0, V_max=5.0), dict(I="rest", tf=300.0), dict(I=-1.0, tf=600.0)], 0.0),
                  ([{"I": 2.0, "tf": 1800.0, "V_max": 4.1}, {"V": "hold", "V_max": 4.1, "I_min": 1 / 20}], 0.0), ([{"I": -1.0}], 1.0)]
    worst, n_same, n_all = 0.0, 0, 0
    for proto, soc in protos:
        ens = pkg.simulate_ensemble(p, Th, proto, SOC=soc)
        runs = runs_to_oracle(O, p, pkg, proto)
        for i in range(len(Th)):
            ro = O.simulate(q, Th[i], soc, runs)
            assert [int(f) for f in ens.run_info[i]["flag"]] == [r["flag"] for r in ro["runs"]], (i, ens.run_info[i], ro["runs"])
            same = all(int(ens.counters[i][f]) == ro["counters"][f] for f in ("n_steps", "n_res", "n_jac", "n_newton", "n_errfail", "n_convfail"))
            e = state_rel_err(ens.Y[i], ro["Y"])
            te = max(abs(float(ens.run_info[i, k]["t_end"]) - r["t_end"]) / max(1.0, r["t_end"]) for k, r in enumerate(ro["runs"]))
            n_all += 1; n_same += same
            if same:
                assert e <= tol and te <= tol, (i, e, te)
                worst = max(worst, e, te)
            else:
                assert e <= 2e-2, (i, e)
    assert n_same >= min_same * n_all, (n_same, n_all)
    return worst


def check_reforder_variant(pr, O, pkg, variant):
    """PLH_PREC_F64_REFORDER (precision = "f64_reforder"): same keys / patterns / evaluators as the plain variant's oracle (1e-12 of the terms: the rows differ by their own
    rounding only), consistent initialisation to 1e-12, and a CC discharge -- whose decisions do not hang on rounding -- with the oracle's decisions"""
    assert pr.precision == "f64_reforder" and pr.variant == variant
    check_keys_and_pattern(pr, O)
    check_evaluators(pr, O)
    th = pr.theta_vector()
    soc, proto = (0.0, [{"I": 2.0, "tf": 600.0}]) if pr.temperature else (1.0, [{"I": -1.0, "tf": 1200.0}])
    ens = pkg.simulate_ensemble(pr, th[None, :], proto, SOC=soc)
    ro = O.simulate(variant, th, soc, runs_to_oracle(O, pr, pkg, proto))
    compare_trajectory(ens, 0, ro, rtol_state=1e-5)


def check_stop_function(p, O, pkg, variant=None):
    """opts.stop_function (reference src/checks.jl:26, src/structures.jl:283) as a traced closure g(t, Y, YP, p): the run ends when g > 0, exit flag 12, back-interpolated like a
    built-in bound -- same step, same end point as the oracle; a hook that never fires changes nothing; a built-in bound that fires earlier wins."""
    variant = variant or p.variant
    ce = p.ind["c_e"]
    Th = np.ascontiguousarray(pkg.configs.sweep_theta(p, np.arange(3), 4)) if not p.temperature and not p.aging else np.tile(p.theta_vector(), (2, 1))
    cases = [("c_e gradient", lambda t, Y, YP, P: (Y[ce.stop - 1] - Y[ce.start]) - 400.0, [{"I": -2.0}], 1.0, 12)]
    if p.temperature:
        T = p.ind["T"]
        # T_max per NODE (the reference's built-in T bound is on the average): the hottest of three nodes, written with max
        cases.append(("node temperature", lambda t, Y, YP, P: pkg.closures.maximum(pkg.closures.maximum(Y[T.start + 12], Y[T.start + 25]), Y[T.start + 38]) - 303.0, [{"I": 4.0, "T_max": 400.0}], 0.0, 12))
    cases.append(("never fires", lambda t, Y, YP, P: Y[ce.start] - 1e9, [{"I": -1.0, "tf": 600.0}], 1.0, 0))
    cases.append(("a built-in bound wins", lambda t, Y, YP, P: t - 5000.0, [{"I": -2.0, "V_min": 3.6}], 1.0, 1))
    cases.append(("time, second run of a chain", lambda t, Y, YP, P: t - 100.0, [{"I": -1.0, "tf": 50.0}, {"I": -0.5, "tf": 600.0}], 1.0, 12))
    for name, g, proto, soc, flag in cases:
        o = pkg.Opts(); o.stop_function = g
        ens = pkg.simulate_ensemble(p, Th, proto, SOC=soc, opts=o)
        prog = pkg.closures.trace(g, p)
        runs = runs_to_oracle(O, p, pkg, proto)
        for i in range(len(Th)):
            ro = O.simulate(variant, Th[i], soc, runs, opts=O.default_opts(stop_program=prog))
            assert int(ens.run_info[i, -1]["flag"]) == ro["runs"][-1]["flag"] == flag, (name, i, ens.run_info[i], ro["runs"])
            compare_trajectory(ens, i, ro, rtol_state=1e-5)
            if flag == 12 and "time" not in name:
                assert abs(pkg.closures.evaluate(prog, 0.0, ens.Y[i], ens.YP[i], Th[i])) < 1e-6 * 400.0, name          # linear in Y: the back-interpolated end point sits on g = 0
        if name == "never fires":
            base = pkg.simulate_ensemble(p, Th, proto, SOC=soc)
            assert np.array_equal(base.run_info["t_end"], ens.run_info["t_end"]) and np.abs(base.Y - ens.Y).max() <= 1e-9 * np.abs(base.Y).max()


def check_initial_states(p, O, pkg):
    """simulate(p, ...; initial_states = Y) (reference src/model_evaluation.jl:15, 102-110, 193-199): a NEW solution (t0 = 0, no tstop at 1 s) from a caller-supplied state vector
    instead of initial_guess!, its SOC estimated from the anode's mean concentration (calc_SOC) -- same run as the oracle's from the same states."""
    Th = np.ascontiguousarray(pkg.configs.sweep_theta(p, np.arange(2), 4)) if not p.temperature and not p.aging else np.tile(p.theta_vector(), (2, 1))
    first = pkg.simulate_ensemble(p, Th, [{"I": -1.0, "tf": 900.0}], SOC=1.0)
    Y0 = np.ascontiguousarray(first.Y)
    soc = pkg.api.calc_SOC(p, Y0)
    assert np.abs(soc - first.run_info["SOC"][:, 0]).max() < 2e-2           # (the estimate of the trapezoid SOC: the particles' mean, not their volume average)
    proto = [{"I": 1.0, "tf": 400.0}, {"V": "hold", "tf": 100.0}]
    ens = pkg.simulate_ensemble(p, Th, proto, initial_states=Y0)
    runs = runs_to_oracle(O, p, pkg, proto)
    for i in range(len(Th)):
        ro = O.simulate(p.variant + ("_quiet" if p.variant in ("lco_iso", "lco_thermal", "nmc_iso_sei") else ""), Th[i], float(soc[i]), runs, Y_init=Y0[i])
        compare_trajectory(ens, i, ro, rtol_state=1e-6)
        assert abs(float(ens.t[i, 0])) == 0.0 and abs(float(ens.SOC[i, 0]) - soc[i]) < 1e-15
    # the single-cell front end, and its refusal to combine initial_states with a continued solution (model_evaluation.jl:105-108)
    th0 = p.theta_vector()
    sol = pkg.simulate(p, 400.0, I=1.0, initial_states=Y0[0]) if np.array_equal(Th[0], th0) else None
    if sol is not None:
        assert abs(sol.t[0]) == 0.0 and abs(sol.t[-1] - 400.0) < 1e-9
        with pytest_raises(ValueError):
            pkg.simulate(p, 100.0, I=1.0, sol=sol, initial_states=Y0[0])


def check_save_start(p, pkg):
    """opts.save_start (reference src/model_evaluation.jl:384-411): the first simulate() with a key fills p.save_start_dict with the initialised algebraic states and is otherwise
    the plain call; the second starts its consistent initialisation from them -- fewer Newton iterations, the same trajectory to the initialisation tolerance."""
    p.save_start_dict.clear()
    plain = pkg.simulate(p, 600.0, I=-1.0, SOC=0.8)
    a = pkg.simulate(p, 600.0, I=-1.0, SOC=0.8, save_start=True)
    assert list(p.save_start_dict) == [("I", 0.8, -1.0)] and np.array_equal(a.V, plain.V) and np.array_equal(a.t, plain.t)
    b = pkg.simulate(p, 600.0, I=-1.0, SOC=0.8, save_start=True)
    assert len(p.save_start_dict) == 1 and abs(len(b.t) - len(a.t)) <= 2
    # (another first guess, another last bit of the initialised state, another h0: the step grids differ by 1e-3 of h -- the trajectories are compared at equal times)
    tt = a.t[1:-1]
    dV = np.abs(np.asarray(b(tt).V) - a.V[1:-1]).max()
    assert abs(b.V[0] - a.V[0]) < 1e-6 and dV < 1e-4 and abs(b.t[-1] - a.t[-1]) < 1e-9, (dV, b.V[0] - a.V[0])
    ia, ib = int(a.counters["n_init_iters"]), int(b.counters["n_init_iters"])
    assert ib < ia, (ia, ib)
    c = pkg.simulate(p, 600.0, I=-2.0, SOC=0.8, save_start=True)                 # another key
    assert len(p.save_start_dict) == 2 and c.V[-1] < a.V[-1]
    print("save_start: initialisation Newton iterations %d -> %d with the cached algebraic states; |dV| at equal times %.1e" % (ia, ib, dV))
    p.save_start_dict.clear()


def pytest_raises(exc):
    import pytest
    return pytest.raises(exc)

// dfn_sens.h -- forward parameter sensitivities s_k(t) = dY(t)/d theta_k, integrated alongside the states ("parameter-sensitivity (forward) outputs for
// estimation workflows").  The reference has no such output (its users difference whole simulate() calls); what is restated here is the textbook staggered-direct method of
// DAE sensitivity analysis (Maly & Petzold 1996; the method SUNDIALS IDAS documents as IDA_STAGGERED with difference-quotient sensitivity residuals) on top of the
// fixed-leading-coefficient BDF of dfn_integrate.h:
//
//   F(t, y, y', theta) = 0   =>   F_y s + F_y' s' + F_theta = 0        (one linear DAE per parameter, same step sizes and orders as y)
//
//   * after every ACCEPTED step (y_n, y'_n converged, the step's coefficients final) each s_k gets the BDF treatment of y: predictor from its own history of modified divided
//     differences (same beta / gamma), s' = s'_pred + cj (s - s_pred), corrector  s <- s - c J~^-1 r(s)  with r(s) = F_y s + F_y' s' + F_theta and the integrator's CURRENT
//     (possibly stale) factorisation J~ -- the matrix the Newton iteration of y has just converged with; c = 2 / (1 + cj/cj_factor) as in IDANls.  r is linear in s, so the
//     iteration converges at the rate of that Newton iteration; it runs until a correction times the parameter is below max(1e-7, reltol / 100) of the scale of every state.
//   * the residual of the sensitivity system is formed by directional difference quotients of the model's own residual -- no second set of equations to keep in step:
//     F_y s + F_y' s' = (F(y + e s, y' + e s') - F(y, y')) / e ,  F_theta = (F(.; theta + d e_k) - F(.; theta)) / d  (the cell constants are recomputed from the perturbed
//     theta row for that one evaluation: cell_setup<M, false>).
//   * the states are NOT affected: no error control on s (IDAS' errconS = false), the factorisation is only used, never refreshed, the solution point of the step is saved and
//     restored around the phase -- a run with sensitivities takes bit for bit the steps of the run without.
//   * start of a run: s_diff from the initial guess (difference quotient of initial_guess! in theta) or carried from the previous run; s_alg from the algebraic equations with
//     the consistent initialisation's own factorisation; s'_diff = d rhs / d theta, s'_alg = 0 (the first step is BDF1: s' only seeds the predictor).
//   * end of a run on a bound: the reference replaces the last point by a linear interpolation between the last two accepted points (interp_final_points!,
//     model_evaluation.jl:369-382); s gets the same interpolation AND the shift of the crossing with theta (sens_finish): what is reported, and what the next run continues
//     from, is the derivative of the end state as simulate() returns it.
// The history of each s_k (MAXORD + 1 vectors) lives in HBM, lane-strided like every state vector; everything else is registers and the three LDS work vectors of the
// integrator, which are dead between the output of a step and the next predictor.
#pragma once
// (included from dfn_integrate.h, inside namespace pl, after IdaScalars / PL_VEC / EWT)
// Every vector statement of this file is lane-masked (PL_VECG): the sensitivity histories live in GLOBAL memory at the unpadded stride (M::NPADG), and the padding of the LDS
// vectors (M::VPAD) must stay zero -- the masked form never touches it.
#pragma push_macro("PL_VEC")
#undef PL_VEC
#define PL_VEC(n) PL_VECG(n)

constexpr int SENS_MAXIT = 64;
// weights of the sensitivity norms: 1 / (|y_n| + abstol / reltol) = the integrator's error weights with the tolerance divided out -- the corrector stops when a correction,
// times the parameter, is below SENS_TOL of the scale of each state (the difference quotients carry ~1e-9 of rounding: a criterion that tightened with reltol could not be met)
// (and not below 1 % of the integration's own relative tolerance: at reltol 1e-3 a corrector driven to 1e-7 spends its iterations on digits the step does not have)
constexpr double SENS_TOL = 1e-7, SENS_FD = 1e-7;
__device__ __forceinline__ double sens_tol(double rtol) { const double t = 0.01 * rtol; return t > SENS_TOL ? t : SENS_TOL; }

__device__ __forceinline__ double wave_max(double v) {
  for (int o = WAVE / 2; o >= 1; o >>= 1) { const double r = __shfl_xor(v, o); v = v > r ? v : r; }
  return v;
}

template <class M> struct SensCell {
  SensArgs a; const double* th0; int cell, P, max_pts;
  bool first;                   // the next accepted step is the first of this integrator instance: hist[1] holds s'(t0), not h s'
  int n_it, n_fail, n_refresh;   // corrector iterations; solves that did not converge (after the refresh); steps / initialisations that factored their own matrix
  __device__ __forceinline__ double* hist(int k, int j) const { return a.hist + (((size_t)cell * a.n_sens + k) * (MAXORD + 1) + j) * M::NPADG; }
  __device__ __forceinline__ const double* thp(int k) const { return a.theta_pert + ((size_t)cell * a.n_sens + k) * P; }
  __device__ __forceinline__ double* aux(int k) const { return a.aux + ((size_t)cell * a.n_sens + k) * 4; }
};

// the quantity a :hold run holds, as a linear functional of a state vector (initial_current!, input_methods.jl:11-74: V = Phi_s[1] - Phi_s[end], I, eta_plating); power is
// bilinear (I I1C V) and handled where it is used
template <class M>
__device__ __forceinline__ double sens_hold_g(int mode, const double* v) {
  PL_MODEL(M);
  if (mode == PLH_MODE_V) return v[O_PS] - v[O_PS + NJ - 1];
  if (mode == PLH_MODE_I) return v[O_I];
  if (mode == PLH_MODE_ETA_P) return v[O_PS + NP] - v[O_PE + NP + NS];
  return 0.0;
}

// The cell's theta-derived constants (CellConst, the conduction / weighting tables of the thermal model, the SOH quadrature weights of the SEI model) are SAVED once and COPIED
// back after every evaluation with a perturbed theta row.  Recomputing them from the unperturbed row would be the same arithmetic -- but in another inlined copy of cell_setup,
// which the compiler is free to contract differently: measured on the GPU, a last-bit difference in one constant is enough to move a few of 8192 trajectories off the ones of the
// plain kernel.  A copy is exact.
template <bool SAVE, class M>
PL_DEV void sens_consts(CellLDS<M>& S, const SensCell<M>& X) {
  const int lane = lane_id();
  double* b = X.a.cbak + (size_t)X.cell * SENS_CBAK;
  constexpr int NC = (int)(sizeof(CellConst) / sizeof(double));
  static_assert(sizeof(CellConst) % sizeof(double) == 0, "CellConst is copied as doubles");
  double* c = reinterpret_cast<double*>(&S.cc);
  int o = 0;
  for (int k = lane; k < NC; k += WAVE) { if (SAVE) b[o + k] = c[k]; else c[k] = b[o + k]; }
  o += NC;
  if constexpr (M::THERMAL) {
    double* t0 = &S.th.aL[0];
    const int nt = (int)(&S.th.qI[0] + 2 - t0);             // aL, aU, wT5, aC2, rc5, qI: consecutive members of ThermalPool
    static_assert(NC + 2 * NT + 16 + NN <= SENS_CBAK, "SENS_CBAK");
    for (int k = lane; k < nt; k += WAVE) { if (SAVE) b[o + k] = t0[k]; else t0[k] = b[o + k]; }
    o += nt;
  }
  if constexpr (M::SEI) { for (int k = lane; k < NN; k += WAVE) { if (SAVE) b[o + k] = S.sei.sohw[k]; else S.sei.sohw[k] = b[o + k]; } o += NN; }
  if constexpr (!M::THERMAL) {                              // the node-pass tables (NodeTab): copies of CellConst entries
    constexpr int NTB = (int)(sizeof(S.nt) / sizeof(double));
    static_assert(NC + NN + NTB <= SENS_CBAK, "SENS_CBAK");
    double* t0 = reinterpret_cast<double*>(&S.nt);
    for (int k = lane; k < NTB; k += WAVE) { if (SAVE) b[o + k] = t0[k]; else t0[k] = b[o + k]; }
  }
  PL_XSYNC();
}

// F(phi[0] + e s, ypn + e sp) -> S.delta   (S.yy / S.yp are the work vectors)
template <class M>
PL_DEV void sens_eval(CellLDS<M>& S, LaneRegs& R, const double (&s)[M::NTRIP], const double (&sp)[M::NTRIP], const double (&ypn)[M::NTRIP], double e, int mode, double value) {
  PL_MODEL(M);
  const int lane = lane_id();
  PL_VEC(n) { S.yy[n] = S.phi[0][n] + e * s[k__]; S.yp[n] = ypn[k__] + e * sp[k__]; }
  PL_XSYNC();
  cell_residual(S, R, S.yy, S.yp, S.delta, mode, value);
  PL_XSYNC();
}

// r05: the factorisation as DATA -- everything cell_res_jac / cell_factor write: the structured Jacobian pool, the eliminated system, the resolvents and the SEI / thermal pools
// (consecutive members of CellLDS from ceL up to the BDF coefficient arrays) and the per-lane registers.  A sensitivity corrector that does not converge with the integrator's
// stale matrix (rate >= 0.97, or the iteration cap) saves the factorisation to HBM, factors the matrix of the step's OWN solution point and coefficient -- with which the linear
// sensitivity system converges in one or two iterations -- and copies the integrator's factorisation back afterwards.  A copy, not a recomputation: a second inlined copy of the
// Jacobian pass may be contracted differently by the compiler (see sens_consts), and the states must keep taking the steps of the run without sensitivities bit for bit.
template <bool SAVE, class M>
PL_DEV void sens_factor_copy(CellLDS<M>& S, LaneRegs& R, const SensCell<M>& X) {
  const int lane = lane_id();
  double* r0 = &S.ceL[0];
  const int nd = (int)(&S.ida_psi[0] - r0);
  double* b = X.a.fsave + (size_t)X.cell * X.a.fsave_stride;
  PL_XSYNC();
  for (int k = lane; k < nd; k += WAVE) { if (SAVE) b[k] = r0[k]; else r0[k] = b[k]; }
  double* br = b + nd + (size_t)(wave_id() * WAVE + lane) * (2 * LR_PASS);
  for (int q = 0; q < LR_PASS; q++) {
    if (SAVE) { br[q] = R.wreg[q]; br[LR_PASS + q] = R.rcp[q]; } else { R.wreg[q] = br[q]; R.rcp[q] = br[LR_PASS + q]; }
  }
  PL_XSYNC();
}

// the matrix of the point in (S.phi[0], ypn) with coefficient cj, factored in place (alg_only: the algebraic block of the consistent initialisation)
template <class M>
PL_DEV void sens_refactor(CellLDS<M>& S, LaneRegs& R, const double (&ypn)[M::NTRIP], double cj, int mode, double value, bool alg_only) {
  PL_MODEL(M);
  const int lane = lane_id();
  PL_VEC(n) { S.yy[n] = S.phi[0][n]; S.yp[n] = ypn[k__]; }
  PL_XSYNC();
  cell_res_jac(S, R, S.yy, S.yp, S.delta, mode, value);
  cell_factor(S, R, S.tb, cj, mode, alg_only);
  PL_XSYNC();
}

// dV/dtheta_k of the point just saved: S.delta holds s_k
template <class M>
PL_DEV void sens_put_V(CellLDS<M>& S, const SensCell<M>& X, int k, int idx) {
  PL_MODEL(M);
  if (lane_id() == 0 && wave_id() == 0 && X.a.dV && idx >= 0 && idx < X.max_pts)
    X.a.dV[((size_t)X.cell * X.a.n_sens + k) * X.max_pts + idx] = S.delta[O_PS] - S.delta[O_PS + NJ - 1];
}

// start of a run (after the consistent initialisation and ida_reinit: S.phi[0] = y0, S.yp = y'0, the algebraic block factored): s_k(t0), s'_k(t0) -> history
template <class M>
PL_DEV void sens_init(CellLDS<M>& S, SensCell<M>& X, int mode, double value, bool new_solution, double SOC0, double rtol, double atol, int idx,
                      bool hold = false, double prev_V = 0.0, double prev_I = 0.0) {
  PL_MODEL(M);
  const int lane = lane_id();
  LaneRegs Ra;                                            // (the algebraic solves do not touch the particle registers)
  for (int q = 0; q < LR_PASS; q++) { Ra.wreg[q] = 0.0; Ra.rcp[q] = 0.0; }
  sens_consts<true>(S, X);
  bool refreshed = false;
  const double I1C0 = S.cc.I1C;
  const int amode = (M::THERMAL && mode == PLH_MODE_DT) ? PL_MODE_DT_TWIN : mode;      // the algebraic form of the dT row (cell_init_consistent)
  double yn[NTRIP], ypn[NTRIP], f0[NTRIP], w[NTRIP], zero[NTRIP];
  PL_VEC(n) { yn[k__] = S.yy[n]; ypn[k__] = S.yp[n]; zero[k__] = 0.0; w[k__] = 1.0 / (fabs(S.phi[0][n]) + atol / rtol); }
  PL_XSYNC();
  sens_eval(S, Ra, zero, zero, ypn, 0.0, amode, value);
  PL_VEC(n) f0[k__] = S.delta[n];
  for (int k = 0; k < X.a.n_sens; k++) {
    const double* tp = X.thp(k); const int col = X.a.cols[k];
    const double dth = tp[col] - X.th0[col], psc = fabs(X.th0[col]) > 0.0 ? fabs(X.th0[col]) : 1.0;
    double s[NTRIP], fp[NTRIP];
    double* h0 = X.hist(k, 0);
    PL_XSYNC();
    cell_setup<M, false>(S, Ra, S.tb, tp);                // constants of the perturbed theta row
    if (new_solution) {                                   // d(initial guess)/d theta (initial_guess!, states_definition.jl:80-121): the differential states
      cell_initial_guess(S, S.delta, SOC0);
      PL_XSYNC();
      PL_VEC(n) s[k__] = n < NDIFF ? (S.delta[n] - S.phi[0][n]) / dth : 0.0;
      PL_XSYNC();
    } else { PL_VEC(n) s[k__] = h0[n]; }                  // carried from the end of the previous run (the algebraic part: first guess only)
    // A :hold run holds the value its quantity had at the end of the previous run (initial_current!, input_methods.jl:11-74) -- which depends on theta: d value / d theta_k is
    // that quantity's sensitivity at the previous run's end (h0 still holds it), one constant term -d value / d theta_k in F_theta of the control row for the whole run.
    // Power: value = I I1C V with I1C a function of theta (the perturbed constants are current here).  dT = :hold holds 0 K/s: no term.  A new solution holds nothing.
    double dval = 0.0;
    if (hold && !new_solution) {
      if (mode == PLH_MODE_P) {
        const double I1Cp = S.cc.I1C;                      // (of the perturbed row: cell_setup<M, false> above; I1C0: read before the k loop)
        const double sV = h0[O_PS] - h0[O_PS + NJ - 1], sI = h0[O_I];
        dval = I1C0 * (sI * prev_V + prev_I * sV) + prev_I * prev_V * ((I1Cp - I1C0) / dth);
      } else dval = sens_hold_g<M>(mode, h0);
    }
    if (lane == 0 && wave_id() == 0) { double* ax = X.aux(k); ax[0] = dval; if (new_solution) { ax[1] = 0.0; ax[3] = 0.0; } }
    sens_eval(S, Ra, zero, zero, ypn, 0.0, amode, value);
    PL_VEC(n) { fp[k__] = (S.delta[n] - f0[k__]) / dth; if (n == O_I) fp[k__] -= dval; }
    sens_consts<false>(S, X);
    PL_XSYNC();
    // algebraic part: G_ya s_a = -(G_yd s_d + G_theta), Newton-like with the factorisation of the last initialisation iterate
    bool conv = false;
    for (int attempt = 0; attempt < 2 && !conv; attempt++) {
    // (not converged with the factorisation the initialisation left behind: the algebraic block of the initial point itself.  Nothing to put back: the integrator's first
    //  step sets up its own matrix, ida_nls nst == 0)
    if (attempt == 1) { if (refreshed) break; sens_refactor(S, Ra, ypn, 0.0, amode, value, true); refreshed = true; X.n_refresh++; }
#ifdef PL_TEST_SENS_FORCE_REFRESH
    if (attempt == 0 && k == 0) continue;
#endif
    double nr_old = 0.0;
    for (int it = 0; it < SENS_MAXIT; it++) {
      double m = 0.0;
      PL_VEC(n) { const double q = fabs(s[k__]) * w[k__]; m = q > m ? q : m; }
      m = wave_max(m);
      const double e = m > 0.0 ? SENS_FD / m : 1.0, re = 1.0 / e;
      sens_eval(S, Ra, s, zero, ypn, e, amode, value);
#ifdef PL_WAVE_EMU
      if (getenv("PL_EMU_TRACE_SENS") && lane == 0 && (attempt == 1 || it >= atoi(getenv("PL_EMU_TRACE_SENS")))) {
        double rm = 0.0; int ir = 0;
        fprintf(stderr, "   sI %.9e ctrl row F %.6e ", (S.yy[O_I] - S.phi[0][O_I]) * re, S.delta[O_I]);
      }
#endif
      PL_VEC(n) S.delta[n] = (S.delta[n] - f0[k__]) * re + fp[k__];
      PL_XSYNC();
#ifdef PL_WAVE_EMU
      if (getenv("PL_EMU_TRACE_SENS") && lane == 0 && (attempt == 1 || it >= atoi(getenv("PL_EMU_TRACE_SENS")))) {
        double rm = 0.0; int ir = NDIFF;
        for (int n = NDIFF; n < NST; n++) if (fabs(S.delta[n]) > rm) { rm = fabs(S.delta[n]); ir = n; }
        fprintf(stderr, "r[I] %.6e largest alg r row %d %.6e\n", S.delta[O_I], ir, S.delta[ir]);
      }
#endif
      cell_solve(S, Ra, S.delta, amode, true);
      PL_XSYNC();
      double nr = 0.0;
      PL_VEC(n) if (n >= NDIFF) { const double d = S.delta[n]; s[k__] -= d; const double q = d * w[k__] * psc; nr += q * q; }
      nr = sqrt(wave_sum(nr) * (1.0 / NST));
      X.n_it++;
      PL_XSYNC();
#ifdef PL_WAVE_EMU
      if (getenv("PL_EMU_TRACE_SENS") && lane == 0 && (attempt == 1 || it >= atoi(getenv("PL_EMU_TRACE_SENS")))) {
        int im = NDIFF; double vm = 0.0;
        for (int n = NDIFF; n < NST; n++) { const double q = fabs(S.delta[n]) / (fabs(S.phi[0][n]) + atol / rtol); if (q > vm) { vm = q; im = n; } }
        fprintf(stderr, "sens init cell %d mode %d k %d attempt %d it %d nr %.3e m %.3e e %.3e worst row %d d %.3e y %.3e\n", X.cell, mode, k, attempt, it, nr, m, e, im, S.delta[im], S.phi[0][im]);
      }
#endif
      if (nr <= sens_tol(rtol)) { conv = true; break; }
      if (!(nr == nr)) break;
      // the rate-based test of sens_step (IDAS' criterion) here too: at the start of a dT = :hold run the heat sources make the twin control row's difference quotient noisy
      // (1e-5 ... 2e-4 in this norm, measured on C3 cells: the iteration contracts to that floor and is kicked off it again, for ever); 0.33 reltol is above that floor
      // at the default tolerances and below sens_tol at tight ones, where nothing changes
      if (it > 0) {
        const double q = nr / nr_old;
        if (q < 0.97 && q / (1.0 - q) * nr <= 0.33 * rtol) { conv = true; break; }
        if (attempt == 1 && it >= 2 && q >= 0.97 && nr <= 0.33 * rtol) { conv = true; break; }     // (the rounding floor of the difference quotients, as in sens_step)
      }
      nr_old = nr;
    }
    }
    if (!conv) X.n_fail++;
    // s'_diff = d rhs_diff / dy s + d rhs_diff / d theta (F_diff = rhs - y'); s'_alg = 0
    {
      double m = 0.0;
      PL_VEC(n) { const double q = fabs(s[k__]) * w[k__]; m = q > m ? q : m; }
      m = wave_max(m);
      const double e = m > 0.0 ? SENS_FD / m : 1.0, re = 1.0 / e;
      sens_eval(S, Ra, s, zero, ypn, e, amode, value);
      double* h1 = X.hist(k, 1);
      PL_VEC(n) { h0[n] = s[k__]; h1[n] = n < NDIFF ? (S.delta[n] - f0[k__]) * re + fp[k__] : 0.0; }
      PL_XSYNC();
      PL_VEC(n) S.delta[n] = s[k__];
      PL_XSYNC();
      sens_put_V(S, X, k, idx);
      if (lane == 0 && wave_id() == 0) X.aux(k)[2] = S.delta[O_I];      // dI/dtheta at the first saved point of the run (the SOC trapezoid of the first step)
      PL_XSYNC();
    }
  }
  X.first = true;
  PL_VEC(n) { S.yy[n] = yn[k__]; S.yp[n] = ypn[k__]; }
  PL_XSYNC();
}

// after an accepted step: advance every s_k over the same step (S.phi = history after IDACompleteStep, S.yy / S.yp = y(tn), y'(tn), I = the step's coefficients)
template <class M>
PL_DEV void sens_step(CellLDS<M>& S, LaneRegs& R, const IdaScalars& I, SensCell<M>& X, int mode, double value, int idx, double dt_saved = 0.0) {
  PL_MODEL(M);
  const int lane = lane_id();
  const int ku = I.kused; const double cj = I.cj;
  double sc = (I.cjratio != 1.0) ? 2.0 / (1.0 + I.cjratio) : 1.0;
  bool refreshed = false;
  double yn[NTRIP], ypn[NTRIP], f0[NTRIP], zero[NTRIP];
  PL_VEC(n) { yn[k__] = S.yy[n]; ypn[k__] = S.yp[n]; zero[k__] = 0.0; }
  PL_XSYNC();
  sens_eval(S, R, zero, zero, ypn, 0.0, mode, value);      // F(y_n, y'_n): the base of every difference quotient of this step
  PL_VEC(n) f0[k__] = S.delta[n];
  for (int k = 0; k < X.a.n_sens; k++) {
    const double* tp = X.thp(k); const int col = X.a.cols[k];
    const double dth = tp[col] - X.th0[col], psc = fabs(X.th0[col]) > 0.0 ? fabs(X.th0[col]) : 1.0;
    double fp[NTRIP], a_[NTRIP], b_[NTRIP], s[NTRIP];
    PL_XSYNC();
    cell_setup<M, false>(S, R, S.tb, tp);
    sens_eval(S, R, zero, zero, ypn, 0.0, mode, value);
    const double dval = X.aux(k)[0];                       // d(held value)/d theta_k of a :hold run (sens_init), 0 otherwise: the control row is g(Y) - value
    PL_VEC(n) { fp[k__] = (S.delta[n] - f0[k__]) / dth; if (n == O_I) fp[k__] -= dval; }
    sens_consts<false>(S, X);
    PL_XSYNC();
    // predictor from the history (IDASetCoeffs' rescaling phi*_j = beta_j phi_j for j >= ns is applied in place, as form_iterate does for y)
    double* h0 = X.hist(k, 0);
    PL_VEC(n) { a_[k__] = h0[n]; b_[k__] = 0.0; }
    for (int j = 1; j <= MAXORD; j++) if (j <= ku) {
      double* hj = X.hist(k, j);
      const double g = S.ida_gamma[j];
      double bt = j >= I.ns ? S.ida_beta[j] : 1.0;
      if (X.first && j == 1) bt = I.hused;               // first step of the integrator: hist[1] = s'(t0) -> h s'(t0) (beta_1 = 1 there)
      PL_VEC(n) { const double p = hj[n] * bt; hj[n] = p; a_[k__] += p; b_[k__] += g * p; }
    }
    PL_VEC(n) s[k__] = a_[k__];
    bool conv = false;
    const bool fresh0 = refreshed;                          // (an earlier parameter of this step already factored the step's matrix)
    for (int attempt = 0; attempt < 2 && !conv; attempt++) {
    if (attempt == 1) {
      // not converged with the integrator's matrix: the step's own matrix (this and the remaining parameters of the step use it; the integrator gets its own back below)
      if (!X.a.fsave || fresh0) break;
      sens_factor_copy<true>(S, R, X); sens_refactor(S, R, ypn, cj, mode, value, false); refreshed = true; sc = 1.0; X.n_refresh++;
      PL_VEC(n) s[k__] = a_[k__];
    }
#ifdef PL_TEST_SENS_FORCE_REFRESH
    if (attempt == 0 && k == 0 && (X.n_it % 3) == 0) continue;      // (test build: every few steps the first parameter is sent through the refresh path)
#endif
    double nr_old = 0.0;
    for (int it = 0; it < SENS_MAXIT; it++) {
      double sp[NTRIP], m = 0.0;
      PL_VEC(n) { sp[k__] = b_[k__] + cj * (s[k__] - a_[k__]); const double q = fabs(s[k__]) * EWT(n); m = q > m ? q : m; }
      m = wave_max(m) * I.rtol;
      const double e = m > 0.0 ? SENS_FD / m : 1.0, re = 1.0 / e;
      sens_eval(S, R, s, sp, ypn, e, mode, value);
      PL_VEC(n) S.delta[n] = (S.delta[n] - f0[k__]) * re + fp[k__];
      PL_XSYNC();
      cell_solve(S, R, S.delta, mode, false);
      PL_XSYNC();
      double nr = 0.0;
      PL_VEC(n) { const double d = S.delta[n] * sc; s[k__] -= d; const double q = d * EWT(n) * (psc * I.rtol); nr += q * q; }
      nr = sqrt(wave_sum(nr) * (1.0 / NST));
      X.n_it++;
      PL_XSYNC();
#ifdef PL_WAVE_EMU
      if (getenv("PL_EMU_TRACE_SENS") && lane == 0 && (attempt == 1 || it >= atoi(getenv("PL_EMU_TRACE_SENS")))) {
        int im = 0; double vm = 0.0;
        for (int n = 0; n < NST; n++) { const double q = fabs(S.delta[n]) / (fabs(S.phi[0][n]) + 1e-3); if (q > vm) { vm = q; im = n; } }
        fprintf(stderr, "sens step cell %d nst %d t %.6f h %.3e k %d attempt %d it %d nr %.3e q %.3f cjratio %.3f worst %d d %.3e y %.3e\n", X.cell, I.nst, I.tn, I.hused, k, attempt, it, nr, it ? nr / nr_old : 0.0, I.cjratio, im, S.delta[im], S.phi[0][im]);
      }
#endif
      if (nr <= sens_tol(I.rtol)) { conv = true; break; }
      if (!(nr == nr)) break;
      // r05: IDANls' own convergence test beside the fixed one -- the corrector is linear, its rate is that of the integrator's (possibly stale) matrix: with the estimated
      // rate q the remaining error is q / (1 - q) x the last correction, and a sensitivity is converged for a step integrated at reltol once that is below 0.33 reltol of the
      // state scale per unit relative parameter change (IDAS' criterion).  At tight tolerances 0.33 reltol is below sens_tol and nothing changes; at the default ones the
      // steps near a voltage knee, where the stale matrix converges at 0.8 ... 0.9, no longer run into the iteration cap (r04: 4263 of 4.7 M solves of the C4 shard did)
      if (it > 0) {
        const double q = nr / nr_old;
        if (q < 0.97 && q / (1.0 - q) * nr <= 0.33 * I.rtol) { conv = true; break; }
        if (attempt == 0 && it >= 8 && q >= 0.97 && X.a.fsave) break;          // the stale matrix does not contract: no point in the other 55 iterations
        // with the step's OWN matrix a correction that neither contracts nor exceeds 0.33 reltol is the difference quotients' rounding, not an error of the iterate: at the
        // first steps of a hold run (h ~ 1e-6 s) dI/dtheta can exceed every other sensitivity by 1e5 in weighted size, the common perturbation scale then leaves the other
        // states three digits, and the control row's quotient turns into a staircase the iteration hops on (traced on C3 cells: a 2-cycle of 4e-5 in this norm)
        if (attempt == 1 && it >= 2 && q >= 0.97 && nr <= 0.33 * I.rtol) { conv = true; break; }
      }
      nr_old = nr;
    }
    }
    if (!conv) X.n_fail++;
    // history update (IDACompleteStep): phi[ku+1] = e, phi[ku] += e, phi[j] += phi[j+1]
    {
      double acc[NTRIP];
      PL_VEC(n) acc[k__] = s[k__] - a_[k__];
      if (ku < I.maxord) { double* hn = X.hist(k, ku + 1); PL_VEC(n) hn[n] = acc[k__]; }
      for (int j = MAXORD; j >= 0; j--) if (j <= ku) { double* hj = X.hist(k, j); PL_VEC(n) { acc[k__] += hj[n]; hj[n] = acc[k__]; } }
      PL_VEC(n) S.delta[n] = acc[k__];                     // = s_k(t_n)
      PL_XSYNC();
      sens_put_V(S, X, k, idx);
      // dSOC/dtheta: the trapezoid of dI/dtheta over the saved points, like calc_SOC (scalar_residual.jl:103-111)
      if (lane == 0 && wave_id() == 0) { double* ax = X.aux(k); const double sI = S.delta[O_I], inc = 0.5 * dt_saved * (sI + ax[2]) / 3600.0; ax[1] += inc; ax[3] = inc; ax[2] = sI; }
      PL_XSYNC();
    }
  }
  X.first = false;
  if (refreshed) sens_factor_copy<false>(S, R, X);        // the integrator's own factorisation, bit for bit
  PL_VEC(n) { S.yy[n] = yn[k__]; S.yp[n] = ypn[k__]; }
  PL_XSYNC();
}

// the point check_solve's first-step retry repeats (checks.jl:227-237): dV/dtheta of the point it repeats (hist[0] = s at the start of the run)
template <class M>
PL_DEV void sens_repeat_point(CellLDS<M>& S, const SensCell<M>& X, int idx) {
  PL_MODEL(M);
  const int lane = lane_id();
  for (int k = 0; k < X.a.n_sens; k++) {
    const double* h0 = X.hist(k, 0);
    PL_XSYNC();
    if (lane == 0 && wave_id() == 0 && X.a.dV && idx >= 0 && idx < X.max_pts) X.a.dV[((size_t)X.cell * X.a.n_sens + k) * X.max_pts + idx] = h0[O_PS] - h0[O_PS + NJ - 1];
  }
  PL_XSYNC();
}

// the linear functional of the state whose crossing of a bound ended the run (check_stop), applied to an LDS vector; ix: the entry that held the extreme value for the
// two max / min bounds.  Wave-uniform; every lane must call it (the average temperature is a wave reduction).
template <class M>
PL_DEV double sens_event_g(const CellLDS<M>& S, int flag, int ix, const double* v) {
  PL_MODEL(M);
  if constexpr (M::THERMAL) { if (flag == 5) return cellTavg<M>(S, v); }
  if (flag == 1 || flag == 2) return v[O_PS] - v[O_PS + NJ - 1];
  if (flag == 7 || flag == 8) return v[O_I];
  if (flag == 11) return v[O_PS + NP] - v[O_PE + NP + NS];
  if (flag == 6 || flag == 9) return v[ix];
  return 0.0;
}

// end of a run: s_k at the run's last point -> hist[0] (what the next run continues from), dY, dV.  A run that ended on a bound is back-interpolated by the reference between
// its last two accepted points, y_e = y_(n-1) + fr (y_n - y_(n-1)) with fr = (g_(n-1) - b) / (g_(n-1) - g_n) from the bounded quantity g (interp_final_points!,
// model_evaluation.jl:369-382; check_stop_*, checks.jl:31-224).  Its derivative has two parts: the same interpolation of s, and the shift of the crossing itself,
// (y_n - y_(n-1)) d fr / d theta with d g / d theta = g(s) (g is linear in the state: V, I, the average temperature, eta_plating, one surface / electrolyte concentration).
// Reported is the sum -- the derivative of the end state AS simulate() RETURNS IT; in particular dV/dtheta of a run that ends on a voltage bound is 0.  An SOC bound under a
// constant current is crossed at a time that does not depend on theta (fr fixed); an SOC bound in another mode, and the dfilm bound (a bound on YP), give NaN.
template <class M>
PL_DEV void sens_finish(CellLDS<M>& S, SensCell<M>& X, bool interp, double fr, bool failed, int idx, int flag, const plh_bounds& bd, int mode,
                        double soc_n = 0.0, double soc_nm1 = 0.0, double dt_step = 0.0, double I_end = 0.0) {
  // (soc_n, soc_nm1: the trapezoid SOC at the last two accepted points, before the back-interpolation; dt_step = t_n - t_(n-1); I_end: the current of the interpolated end point)
  PL_MODEL(M);
  const int lane = lane_id();
  const double nan = __builtin_nan("");
  interp = interp && !X.first;                             // (hist[1] = s_n - s_(n-1) once a step has been completed)
  const bool soc_ev = flag == 3 || flag == 4;              // an SOC bound: g = the trapezoid SOC, d g / d theta = the trapezoid of dI/dtheta (aux[1]); constant current: 0, fr fixed
  bool shift = interp, unknown = interp && flag == 10;     // (the dfilm bound is a bound on YP)
  double gn = 0.0, gp = 0.0, bnd = 0.0; int ix = 0;
  if (shift && !unknown) {
    const double* Y = S.phi[0];
    if (flag == 6) { double cm = -1e300; for (int i = 0; i < NN; i++) { const int q = M::SD == 0 ? O_CS + cs_surf(NP + i) : O_CS + NP + i; if (Y[q] > cm) { cm = Y[q]; ix = q; } } }
    if (flag == 9) { double cm = 1e300; for (int i = 0; i < NE; i++) { if (Y[O_CE + i] < cm) { cm = Y[O_CE + i]; ix = O_CE + i; } } }
    bnd = flag == 1 ? bd.V_min : flag == 2 ? bd.V_max : flag == 5 ? bd.T_max : flag == 6 ? bd.c_s_n_max * S.cc.cmaxn : flag == 7 ? bd.I_max : flag == 8 ? bd.I_min
        : flag == 9 ? bd.c_e_min : bd.eta_plating_min;
    if (soc_ev) { gn = soc_n; gp = soc_nm1; bnd = flag == 3 ? bd.SOC_min : bd.SOC_max; }
    else {
    gn = sens_event_g(S, flag, ix, S.phi[0]);
    gp = gn - sens_event_g(S, flag, ix, S.phi[1]);         // g(y_(n-1)) = g(phi0) - g(phi1)
    }
  }
  const int col_cmaxn = S.tb->thidx[K_c_max_n];
  for (int k = 0; k < X.a.n_sens; k++) {
    double* h0 = X.hist(k, 0); const double* h1 = X.hist(k, 1);
    double dfr = 0.0;
    if (shift && !unknown) {
      PL_XSYNC();
      PL_VEC(n) S.delta[n] = h0[n];
      PL_XSYNC();
      const double gsn = sens_event_g(S, flag, ix, S.delta);
      PL_XSYNC();
      PL_VEC(n) S.delta[n] = h1[n];
      PL_XSYNC();
      double gsp = gsn - sens_event_g(S, flag, ix, S.delta);
      double gsn_ = gsn;
      if (soc_ev) { const double* ax = X.aux(k); gsn_ = ax[1]; gsp = ax[1] - ax[3]; }
      const double den = gp - gn;
      dfr = (gsp * den - (gp - bnd) * (gsp - gsn_)) / (den * den);
    }
    // the c_s_n_max bound is c_s_n_max * c_max_n: with c_max_n among the parameters the bound itself moves -- not differentiated: NaN for that column (as every unknown case)
    const bool unk_k = unknown || (interp && flag == 6 && X.a.cols[k] == col_cmaxn);
    PL_XSYNC();
    PL_VEC(n) {
      double v = h0[n];
      if (interp) { const double d1 = h1[n]; v = (v - d1) + fr * d1 + S.phi[1][n] * dfr; }
      if (failed || unk_k) v = nan;
      h0[n] = v; S.delta[n] = v;
      if (X.a.dY) X.a.dY[((size_t)X.cell * X.a.n_sens + k) * NST + n] = v;
    }
    PL_XSYNC();
    if (interp || failed) sens_put_V(S, X, k, idx);
    // SOC of the interpolated end point: SOC_n + (fr - 1) dt I_e / 3600 (cell_simulate) -- its derivative, for the runs that follow
    if (interp && lane == 0 && wave_id() == 0) { double* ax = X.aux(k); ax[1] += (dfr * dt_step * I_end + (fr - 1.0) * dt_step * S.delta[O_I]) / 3600.0; if (failed || unk_k) ax[1] = nan; }
    PL_XSYNC();
  }
  PL_XSYNC();
}
#pragma pop_macro("PL_VEC")

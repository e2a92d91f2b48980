// dfn_thermal.h -- the temperature = true model (config C3): lumped-in-y 1D heat equation over a | p | s | n | z coupled to the DFN rows.
//
// Replaces, for ensembles, what the reference generates from residuals_T! (src/physics_equations/residuals.jl:299-489),
// build_heat_generation_rates! (auxiliary_states_and_coefficients.jl:344-518), the T-dependent closures
// (custom_functions.jl:1-57,96,123-152) and the dT control row (input_methods.jl:182-189, scalar_residual.jl:167-172,347-372).
//
// Structure of the Newton matrix and how it is solved (one wavefront per cell, included at the end of dfn_cell.h):
//   * particles: (kappa_i M - cj I) with kappa_i = D_s(T_i)/Rp^2 differing per node -> spectral resolvent
//     V diag(1/(kappa_i lam - cj)) W applied as two 10x10 mat-vecs; the T_i column of the particle rows folds into the node block
//   * j: node-local pivot as in the isothermal model (with the extra T column)
//   * current collectors: two scalar tridiagonal chains (10 nodes each), eliminated onto T of node 0 / node 29 and the column of I
//   * 30 cell-sandwich nodes: block-tridiagonal, 4x4 blocks over (c_e, Phi_e, Phi_s, T), systolic DPP sweeps
//   * the one-sided 3-point gradient stencils of the ohmic heat (nodes 0, 9, 20, 29) reach a second neighbour: rank-4 Woodbury
//   * control row (I, V, dT, and the algebraic twin of dT used by the consistent initialisation): bordered solve
#pragma once

namespace pl {

// K_eff(c_e, T) with both partials, custom_functions.jl:96
__device__ __forceinline__ void keff_T(double c, double T, double& K, double& dKc, double& dKT) {
  const double A = -10.5 + 0.668 * 1e-3 * c + 0.494 * 1e-6 * c * c;
  const double B = 0.074 - 1.78 * 1e-5 * c - 8.86 * 1e-10 * c * c;
  const double C = -6.96 * 1e-5 + 2.8 * 1e-8 * c;
  const double P = A + B * T + C * T * T;
  const double dPc = (0.668 * 1e-3 + 2 * 0.494 * 1e-6 * c) + (-1.78 * 1e-5 - 2 * 8.86 * 1e-10 * c) * T + 2.8 * 1e-8 * T * T;
  const double dPT = B + 2.0 * C * T;
  K = 1e-4 * c * P * P;
  dKc = 1e-4 * (P * P + 2.0 * c * P * dPc);
  dKT = 2e-4 * c * P * dPT;
}

// OCV_LCO with the entropic term always on (temperature = true), custom_functions.jl:123-136: U, dU/dx, dU/dT, d2U/dxdT
__device__ __forceinline__ void ocv_lco_T(double x, double T, double& U, double& dUdx, double& dUdT, double& ddUdT) {
  const double x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x6 = x4 * x2, x8 = x4 * x4, x10 = x8 * x2;
  const double P = -4.656 + 88.669 * x2 - 401.119 * x4 + 342.909 * x6 - 462.471 * x8 + 433.434 * x10;
  const double Q = -1 + 18.933 * x2 - 79.532 * x4 + 37.311 * x6 - 73.083 * x8 + 95.96 * x10;
  const double dP = x * (2 * 88.669 - 4 * 401.119 * x2 + 6 * 342.909 * x4 - 8 * 462.471 * x6 + 10 * 433.434 * x8);
  const double dQ = x * (2 * 18.933 - 4 * 79.532 * x2 + 6 * 37.311 * x4 - 8 * 73.083 * x6 + 10 * 95.96 * x8);
  const double n = 0.199521039 - 0.928373822 * x + 1.364550689000003 * x2 - 0.6115448939999998 * x3;
  const double d = 1 - 5.661479886999997 * x + 11.47636191 * x2 - 9.82431213599998 * x3 + 3.048755063 * x4;
  const double dn = -0.928373822 + 2 * 1.364550689000003 * x - 3 * 0.6115448939999998 * x2;
  const double dd = -5.661479886999997 + 2 * 11.47636191 * x - 3 * 9.82431213599998 * x2 + 4 * 3.048755063 * x3;
  const double rd = pl_rcp(d), rQ = pl_rcp(Q), nd = n * rd, PQ = P * rQ;      // (one reciprocal per rational function serves its value and its derivative: dfn_cell.h ocv_lco)
  dUdT = -0.001 * nd;
  ddUdT = -0.001 * (dn - nd * dd) * rd;
  U = PQ + dUdT * (T - TREF);
  dUdx = (dP - PQ * dQ) * rQ + ddUdT * (T - TREF);
}

// OCV_LiC6, custom_functions.jl:139-152
__device__ __forceinline__ void ocv_lic6_T(double x, double T, double& U, double& dUdx, double& dUdT, double& ddUdT) {
  // (1 / x and 1 / sqrt(max(x, 1e-4)) once, as in dfn_cell.h ocv_lic6)
  const bool big = x > 1e-4;
  const double xm = big ? x : 1e-4;
  double s1, rs1; pl_sqrt_rsqrt(xm, s1, rs1);
  double s0 = s1, rs0 = rs1;
  if (!big) { s0 = sqrt(x > 0.0 ? x : 0.0); rs0 = 1.0 / s0; }
  const double rx = pl_rcp(x), rx2 = rx * rx;
  const double e1 = pl_exp(0.9 - 15 * x), e2 = pl_exp(0.4465 * x - 0.4108);
  U = 0.7222 + 0.1387 * x + 0.029 * s0 - 0.0172 * rx + 0.0019 * (rs1 * rx) + 0.2808 * e1 - 0.7984 * e2;
  double dv = 0.1387 + 0.0172 * rx2 - 0.2808 * 15 * e1 - 0.7984 * 0.4465 * e2;
  if (x > 0.0) dv += 0.029 * 0.5 * rs0;
  dv += (big ? 0.0019 * (-1.5) : -0.0019) * (rx2 * rs1);
  const double x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x, x6 = x3 * x3, x7 = x6 * x, x8 = x4 * x4;
  const double n = 0.001 * (0.005269056 + 3.299265709 * x - 91.79325798 * x2 + 1004.911008 * x3 - 5812.278127 * x4 + 19329.7549 * x5 - 37147.8947 * x6 + 38379.18127 * x7 - 16515.05308 * x8);
  const double q = 1 - 48.09287227 * x + 1017.234804 * x2 - 10481.80419 * x3 + 59431.3 * x4 - 195881.6488 * x5 + 374577.3152 * x6 - 385821.1607 * x7 + 165705.8597 * x8;
  const double dn = 0.001 * (3.299265709 - 2 * 91.79325798 * x + 3 * 1004.911008 * x2 - 4 * 5812.278127 * x3 + 5 * 19329.7549 * x4 - 6 * 37147.8947 * x5 + 7 * 38379.18127 * x6 - 8 * 16515.05308 * x7);
  const double dq = -48.09287227 + 2 * 1017.234804 * x - 3 * 10481.80419 * x2 + 4 * 59431.3 * x3 - 5 * 195881.6488 * x4 + 6 * 374577.3152 * x5 - 7 * 385821.1607 * x6 + 8 * 165705.8597 * x7;
  const double rq = pl_rcp(q);
  dUdT = n * rq;
  ddUdT = (dn - dUdT * dq) * rq;
  U += dUdT * (T - TREF);
  dUdx = dv + ddUdT * (T - TREF);
}

__device__ __forceinline__ int tsec_of(int it) { return it < NA ? 0 : (it < NA + NP ? 1 : (it < NA + NP + NS ? 2 : (it < NA + NE ? 3 : 4))); }

// ------------------------------------------------------------------------------------------------------------------
// constants of the heat equation (lanes 0..49 own one T node each)
// ------------------------------------------------------------------------------------------------------------------
// diagonal of the scaled heat-conduction stencil, aD = -(aL + aU) - (convective coefficient at the two outer faces): every interior and interface row of residuals_T!
// conserves the conductive flux (residuals.jl:299-489), so the diagonal need not be stored (Jacobian entries only -- the residual uses the difference form)
template <class TP_> __device__ __forceinline__ double thermal_aD(const TP_& TP, int it) {
  const double e = it == 0 ? TP.aC2[0] : (it == NT - 1 ? TP.aC2[1] : 0.0);
  return -(TP.aL[it] + TP.aU[it]) - e;
}

template <class M, bool INIT>
PL_DEV void thermal_setup(CellLDS<M>& S, const Tables* __restrict__ tb, const double* __restrict__ th) {
  static_assert((M::CHEM == PLH_CHEM_LCO_LIC6 || M::CHEM == PLH_CHEM_LGM50) && !M::SEI, "temperature = true is instantiated for LCO/LiC6 and NMC_LGM50/LiC6_LGM50 without aging");
  // Discretisations with temperature = true (reference src/params.jl:119-136 takes any N_p, N_s, N_n, N_a, N_z, N_r).  What the elimination needs: each electrode inside its
  // own half of the twisted sweeps (the T rows of nodes N_p - 1 and N_p + N_s reach back to a second neighbour that must already be final in the same chain), at least five
  // nodes per electrode (that second neighbour must not be the node next to the chain head, whose off-diagonal block the head's own one-sided stencil modifies), one lane per
  // temperature node and per collector row.
  static_assert(!M::THERMAL || (NP >= 5 && NN >= 5 && NP <= TW_MID && NN < NE - TW_MID && NA >= 2 && NZ >= 2 && NA + NZ <= 30 && NT <= WAVE),
                "temperature = true: 5 <= N_p <= (N_p + N_s + N_n) / 2, 5 <= N_n < (N_p + N_s + N_n + 1) / 2, 2 <= N_a, N_z, N_a + N_z <= 30, N_a + N_p + N_s + N_n + N_z <= 64");
  const int lane = lane_id();
  const CellConst& c = S.cc;
  const int* ix = tb->thidx;
  auto& TP = S.th;
  if (lane < NT) {
    const int it = lane, k = tsec_of(it);
    const double la = th[ix[K_l_a]], lz = th[ix[K_l_z]];
    const double hh[5] = {la / NA, c.h[0], c.h[1], c.h[2], lz / NZ};
    const double lam[5] = {th[ix[K_lam_a]], th[ix[K_lam_p]], th[ix[K_lam_s]], th[ix[K_lam_n]], th[ix[K_lam_z]]};
    const double rcp[5] = {th[ix[K_rho_a]] * th[ix[K_Cp_a]], th[ix[K_rho_p]] * th[ix[K_Cp_p]], th[ix[K_rho_s]] * th[ix[K_Cp_s]],
                           th[ix[K_rho_n]] * th[ix[K_Cp_n]], th[ix[K_rho_z]] * th[ix[K_Cp_z]]};
    const int sec_start = k == 0 ? 0 : k == 1 ? NA : k == 2 ? NA + NP : k == 3 ? NA + NP + NS : NA + NE, sec_len = k == 0 ? NA : k == 1 ? NP : k == 2 ? NS : k == 3 ? NN : NZ;
    const int loc = it - sec_start;
    const bool first = loc == 0, last = loc == sec_len - 1;
    const double h = hh[k], lm = lam[k];
    double aL = first ? 0.0 : lm / (h * h), aU = last ? 0.0 : lm / (h * h), aD = -(aL + aU), aC = 0.0;
    if (last && k < 4) {                                  // left CV of an interface (residuals.jl:354-439)
      const double hl = h, hr = hh[k + 1];
      const double beta = (hl / 2) / (hl / 2 + hr / 2);
      const double lif = hmean(beta, lm, lam[k + 1]);
      const double last_l = lm / hl, first_r = lif / (hr / 2 + hl / 2);
      aL = last_l / hl; aD = -(last_l + first_r) / hl; aU = first_r / hl;
    }
    if (first && k > 0) {                                 // right CV of an interface
      const double hl = hh[k - 1], hr = h;
      const double beta = (hl / 2) / (hl / 2 + hr / 2);
      const double lif = hmean(beta, lam[k - 1], lm);
      const double first_r = lif / (hr / 2 + hl / 2), second_r = lm / hr;
      aL = first_r / hr; aD = -(second_r + first_r) / hr; aU = second_r / hr;
    }
    const double hc = th[ix[K_h_cell]], Tamb = th[ix[K_T_amb]];
    if (it == 0) { aD -= hc / h; aC = hc * Tamb / h; }                   // convective BCs at the two outer faces
    if (it == NT - 1) { aD -= hc / h; aC = hc * Tamb / h; }
    const double rc = 1.0 / rcp[k];
    TP.aL[it] = aL * rc; TP.aU[it] = aU * rc;      // (aD: thermal_aD)
    (void)aD;
    if (it == 0) TP.aC2[0] = hc / h * rc;                                 // (the end rows are evaluated as  aL (T_l - T) + aU (T_r - T) + aC2 (T_amb - T): see thermal_node_pass)
    if (it == NT - 1) TP.aC2[1] = hc / h * rc;
    if (loc == 0) TP.rc5[k] = rc;
    if (loc == 0) TP.wT5[k] = h / (la + (c.h[0] * NP + c.h[1] * NS + c.h[2] * NN) + lz);
    if (it == 0) TP.qI[0] = c.I1C * c.I1C / th[ix[K_sig_a]] * rc;
    if (it == NT - 1) TP.qI[1] = c.I1C * c.I1C / th[ix[K_sig_z]] * rc;
  }
  if constexpr (INIT) {
    if (lane < 12) (&TP.TX2[0][0])[lane] = 0.0;
    if (lane == 0) TP.cjf = 0.0;
  }
  PL_SYNC();
}

// ------------------------------------------------------------------------------------------------------------------
// node pass: lanes 0..29 own control volume i / edge i of the cell sandwich; lanes 32..51 own the 20 collector T rows
// ------------------------------------------------------------------------------------------------------------------
template <bool WANT_RES, bool WANT_JAC, class M>
PL_DEV void thermal_node_pass(CellLDS<M>& S, const double* Y, const double* YP, double* Fo, int mode, double value) {
  PL_MODEL(M);
  constexpr int O_T = M::O_T;
  const int lane = lane_id();
  const CellConst& c = S.cc;
  auto& TP = S.th;
  const int i = lane < NE ? lane : NE - 1;
  const bool act = lane < NE;
  const int sc = sec_of(i);
  const bool elec = sc != 1;
  const int jx = sc == 0 ? i : i - NS;
  const int it = NA + i;
  const double ce = Y[O_CE + i], pe = Y[O_PE + i], T = Y[O_T + it], Tl = Y[O_T + it - 1], Tr = Y[O_T + it + 1];
  const double jv_l = Y[O_J + jx], ps_l = Y[O_PS + jx], cs_l = Y[O_CS + cs_surf(jx)], yI = Y[O_I];
  const double ypce = WANT_RES ? YP[O_CE + i] : 0.0, ypT = WANT_RES ? YP[O_T + it] : 0.0;
  const double h0 = c.h[0], h1 = c.h[1], h2 = c.h[2];
  const double h = sc == 0 ? h0 : (sc == 1 ? h1 : h2);
  const double epsc = c.eps[sc], bfc = c.bf[sc];
  const double cKfac = c.Kfac, ctplus = c.tplus, cI1C = c.I1C;
  constexpr bool LGM = M::CHEM == PLH_CHEM_LGM50;                       // K_eff_LGM50(c_e), D_eff_LGM50(c_e), tanh OCVs with dU/dT = 0 (reference src/params.jl:563-672)
  double K, dKc, dKT;
  if constexpr (LGM) { keff_lgm50(ce, K, dKc); dKT = 0.0; } else keff_T(ce, T, K, dKc, dKT);
  K *= bfc; dKc *= bfc; dKT *= bfc;
  double D, dD = 0.0;
  if constexpr (LGM) { deff_lgm50(ce, c.De, D, dD); D *= bfc; dD *= bfc; }
  else D = c.Dc[sc];                                                    // D_eff_linear: no c_e / T dependence (custom_functions.jl:59-69)
  const double ce_n = shift_down1(ce), pe_n = shift_down1(pe), K_n = shift_down1(K), dKc_n = shift_down1(dKc), dKT_n = shift_down1(dKT), T_n = shift_down1(T);
  const double ce_p = shift_up1(ce), pe_p = shift_up1(pe);
  // reciprocals of per-cell constants come from cell_setup (see iso_node_pass)
  const double rh = c.rh[sc], reps = c.reps[sc];
  double beta = 0.5, rdist = rh;
  if (i == NP - 1) { beta = c.beta_ps; rdist = c.rd_ps; }
  if (i == NP + NS - 1) { beta = c.beta_sn; rdist = c.rd_sn; }
  const bool edge = i < NE - 1;
  const double rdenK = pl_rcp(beta * K_n + (1 - beta) * K), Kh = K * K_n * rdenK;
  const double D_n = shift_down1(D);
  const double dD_n = (WANT_JAC && LGM) ? shift_down1(dD) : 0.0;
  double rdenD = 0.0, Dhm;
  if constexpr (LGM) { rdenD = pl_rcp(beta * D_n + (1 - beta) * D); Dhm = D * D_n * rdenD; }     // harmonic edge mean of D_eff(c_e)
  else Dhm = (i == NP - 1) ? c.Dh_ps : ((i == NP + NS - 1) ? c.Dh_sn : D);   // constant edge means of D_eff_linear (cell_setup)
  const double denC = beta * ce_n + (1 - beta) * ce, rcb = denC * pl_rcp(ce * ce_n);
  const double rdenT = pl_rcp(beta * T_n + (1 - beta) * T), Tb = T * T_n * rdenT;
  const double dc = (ce_n - ce) * rdist;
  const double w = Kh * rdist;
  const double g = Kh * Tb * dc * rcb;
  const double E = edge ? w * (pe - pe_n) + cKfac * g : 0.0;
  const double Nf = edge ? Dhm * dc : 0.0;
  const double E_p = shift_up1(E), Nf_p = shift_up1(Nf);
  const double Em = i > 0 ? E_p : 0.0, Nm = i > 0 ? Nf_p : 0.0;
  [[maybe_unused]] double w_m = 0.0, g_m = 0.0, dcoef_m = 0.0;           // PLH_PREC_F64_REFORDER: coefficients of the left edge (cross-lane moves outside divergent control flow)
  if constexpr (M::REFORD) { w_m = shift_up1(w); g_m = shift_up1(edge ? g : 0.0); dcoef_m = shift_up1(Dhm * rdist); }
  // electrode quantities with per-node Arrhenius factors
  const double a = sc == 0 ? c.a_p : c.a_n;
  const double jv = elec ? jv_l : 0.0, ps = elec ? ps_l : 0.0, cs = elec ? cs_l : 1.0;
  const double cmax = sc == 0 ? c.cmaxp : c.cmaxn;
  const double sg = sc == 0 ? c.sig_p : c.sig_n;
  const double rT = pl_rcp(T);
  const double dinv = rT - 1.0 / TREF;
  const double EaK = sc == 0 ? c.EaKp : c.EaKn, EaD = sc == 0 ? c.EaDp : c.EaDn;
  const double kk = (sc == 0 ? c.kp : c.kn) * pl_exp(-EaK * dinv);
  const double kap = (sc == 0 ? c.kap_p : c.kap_n) * pl_exp(-EaD * dinv);
  if (act && elec) { TP.kapP[jx] = kap; if (WANT_JAC) TP.dkapP[jx] = kap * EaD * rT * rT; }
  double U = 0, dU = 0, dUdT = 0, ddUdT = 0;
  const double rcm = sc == 0 ? c.rcm_p : c.rcm_n, rsg = sc == 0 ? c.rsg_p : c.rsg_n;
  if constexpr (LGM) {
    if (sc == 0) ocv_nmc_lgm50(cs * rcm, U, dU);
    else if (sc == 2) ocv_lic6_lgm50(cs * rcm, U, dU);
  } else {
    if (sc == 0) ocv_lco_T(cs * rcm, T, U, dU, dUdT, ddUdT);
    else if (sc == 2) ocv_lic6_T(cs * rcm, T, U, dU, dUdT, ddUdT);
  }
  const double eta = ps - pe - U;
  const double arg = ce * cs * (cmax - cs);
  double sq, inv_sq;                                     // (dfn_cell.h iso_node_pass)
  pl_sqrt_rsqrt(arg > 0.0 ? arg : 1.0, sq, inv_sq);
  if (!(arg > 0.0)) { sq = 0.0; inv_sq = 0.0; }
  const double fRT = 0.5 * FAR / RGAS * rT;
  const double xx = fRT * eta;
  double sh, chh; sinh_cosh(xx, sh, chh);
  const double ps_p = shift_up1(ps), ps_n = shift_down1(ps);
  const bool first = (i == 0) || (i == NP + NS), last = (i == NP - 1) || (i == NE - 1);
  // gradient stencils of the heat sources (build_heat_generation_rates!, aux...jl:344-518): X' = em X[i-1] + e0 X[i] + ep X[i+1] + e2 X[i+-2]
  const bool far_right = (i == 0) || (i == NP + NS);                    // the second neighbour is i+2 (else i-2)
  // (both shifts are issued by every lane: cross-lane operations must not sit in divergent control flow)
  const double ce_nn = shift_down1(ce_n), ce_pp = shift_up1(ce_p), pe_nn = shift_down1(pe_n), pe_pp = shift_up1(pe_p);
  const double ps_nn = shift_down1(ps_n), ps_pp = shift_up1(ps_p);
  const double ce_2 = far_right ? ce_nn : ce_pp, pe_2 = far_right ? pe_nn : pe_pp, ps_2 = far_right ? ps_nn : ps_pp;
  const double r2 = c.r2h[sc];
  double em = -r2, e0 = 0.0, ep = r2, e2 = 0.0;                         // electrolyte quantities (Phi_e, c_e)
  if (i == 0) { em = 0.0; e0 = -3 * r2; ep = 4 * r2; e2 = -r2; }
  else if (i == NP - 1) { em = -c.qps_r; ep = c.qps_r; }
  else if (i == NP) { em = -c.qps_l; ep = c.qps_l; }
  else if (i == NP + NS - 1) { em = -c.qsn_r; ep = c.qsn_r; }
  else if (i == NP + NS) { em = -c.qsn_l; ep = c.qsn_l; }
  else if (i == NE - 1) { ep = 0.0; e0 = 3 * r2; em = -4 * r2; e2 = r2; }
  double sm = -r2, s0 = 0.0, sp = r2, s2 = 0.0;                         // Phi_s within one electrode
  if (first) { sm = 0.0; s0 = -3 * r2; sp = 4 * r2; s2 = -r2; }
  else if (last) { sp = 0.0; s0 = 3 * r2; sm = -4 * r2; s2 = r2; }
  if (!elec) { sm = s0 = sp = s2 = 0.0; }
  const double dPe = em * pe_p + e0 * pe + ep * pe_n + e2 * pe_2;
  const double dce = em * ce_p + e0 * ce + ep * ce_n + e2 * ce_2;
  const double dPs = sm * ps_p + s0 * ps + sp * ps_n + s2 * ps_2;
  const double rc = TP.rc5[tsec_of(it)];
  const double Faj = elec ? FAR * a * jv : 0.0;
  if (WANT_RES) {
    if (act) {
      const double src = elec ? (1 - ctplus) * 1.0 * a * jv : 0.0;
      if constexpr (!M::REFORD) {
      Fo[O_CE + i] = ((Nf - Nm) * rh + src) * reps - ypce;                 // residuals_c_e!, residuals.jl:6-106
      Fo[O_PE + i] = (i < NE - 1) ? (E - Em - (elec ? h * FAR * a * jv : 0.0)) : pe;   // residuals_Φ_e!, residuals.jl:554-654
      } else {                                                             // PLH_PREC_F64_REFORDER: the reference's matrix form (see iso_node_pass)
        const double dcoef = Dhm * rdist, dL = i > 0 ? dcoef_m : 0.0, dU_ = edge ? dcoef : 0.0;
        const double accC = (pl_rounded(dL * ce_p) + pl_rounded(-(dL + dU_) * ce)) + pl_rounded(dU_ * ce_n);
        Fo[O_CE + i] = (accC * rh + src) * reps - ypce;
        const double wL = i > 0 ? w_m : 0.0, wU = edge ? w : 0.0;
        const double accP = (pl_rounded(-wL * pe_p) + pl_rounded((wL + wU) * pe)) + pl_rounded(-wU * pe_n);
        const double fE = pl_rounded(-cKfac * ((edge ? g : 0.0) - (i > 0 ? g_m : 0.0))) + (elec ? h * FAR * a * jv : 0.0);
        Fo[O_PE + i] = (i < NE - 1) ? accP - fE : pe;
      }
      if (elec) {
        Fo[O_J + jx] = 2.0 * kk * sq * sh - jv;                                        // residuals_j!, residuals.jl:491-517
        const double lap = first ? (-ps + ps_n) : (last ? (ps_p - ps) : (ps_p - 2 * ps + ps_n));
        double f = h * h * a * FAR * jv;
        const double Idens = yI * cI1C;
        if (i == 0) f += -Idens * h;
        if (i == NE - 1) f += Idens * h;
        if constexpr (!M::REFORD) Fo[O_PS + jx] = lap - f * rsg;                       // residuals_Φ_s!, residuals.jl:656-703
        else Fo[O_PS + jx] = phi_s_row_reford(i, first, last, ps_p, ps, ps_n, f * rsg);
      }
      // residuals_T!, residuals.jl:299-489 ; heat sources aux...jl:344-518
      const double qrr = Faj * (T * dUdT + eta);
      const double rce = pl_rcp(ce);
      const double qohm = K * dPe * dPe + cKfac * K * T * (dce * rce) * dPe + (elec ? sg * dPs * dPs : 0.0);
      // Conduction in DIFFERENCE form, aL (T_l - T) + aU (T_r - T)  (aD = -(aL + aU) on every interior row): the matrix form aL T_l + aD T + aU T_r of the reference
      // (residuals.jl:299-489) sums three terms of 6e6 K/s that cancel to ~0.1 K/s, i.e. carries 1e-9 K/s of rounding per row -- harmless for the row itself, but the
      // dT control row and its algebraic twin SUM the fifty rows (their conduction parts telescope to zero) and find the current from what is left: 1e-6 relative noise in I,
      // which at reltol <= 1e-6 is what made the dT = :hold leg stall on the device more often than in the oracle.  Differences of neighbouring temperatures
      // are exact to their own last bit, so the noise drops by T / dT ~ 1e4.
      if constexpr (!M::REFORD) Fo[O_T + it] = TP.aL[it] * (Tl - T) + TP.aU[it] * (Tr - T) + (qrr + qohm) * rc - ypT;
      else Fo[O_T + it] = ((pl_rounded(TP.aL[it] * Tl) + pl_rounded(thermal_aD(TP, it) * T)) + pl_rounded(TP.aU[it] * Tr)) + (qrr + qohm) * rc - ypT;   // A_T * T, residuals.jl:299-489
    }
    if (lane >= 32 && lane < 32 + NA + NZ) {                                          // current-collector rows
      const int k = lane - 32, ic = k < NA ? k : NA + NE + (k - NA);
      const double Tc = Y[O_T + ic], Tcl = ic > 0 ? Y[O_T + ic - 1] : 0.0, Tcr = ic < NT - 1 ? Y[O_T + ic + 1] : 0.0;
      const double hcv = ic == 0 ? TP.aC2[0] : (ic == NT - 1 ? TP.aC2[1] : 0.0);      // convective end rows: h_cell (T_amb - T) / (h rho Cp)
      if constexpr (!M::REFORD) Fo[O_T + ic] = TP.aL[ic] * (Tcl - Tc) + TP.aU[ic] * (Tcr - Tc) + hcv * (c.Tamb - Tc) + TP.qI[k < NA ? 0 : 1] * yI * yI - YP[O_T + ic];
      else Fo[O_T + ic] = (((pl_rounded(TP.aL[ic] * Tcl) + pl_rounded(thermal_aD(TP, ic) * Tc)) + pl_rounded(TP.aU[ic] * Tcr)) + pl_rounded(hcv * c.Tamb)) + TP.qI[k < NA ? 0 : 1] * yI * yI - YP[O_T + ic];
    }
    if (mode == PLH_MODE_I || mode == PLH_MODE_V || mode == PLH_MODE_P || mode == PLH_MODE_ETA_P || mode == PLH_MODE_RES) {   // scalar_residual!, scalar_residual.jl:167-172
      const double Vc = Y[O_PS] - Y[O_PS + NJ - 1];
      if (lane == 0) Fo[O_I] = mode == PLH_MODE_I ? yI - value : (mode == PLH_MODE_V ? Vc - value : (mode == PLH_MODE_P ? yI * cI1C * Vc - value
                                                                  : (mode == PLH_MODE_RES ? -value : Y[O_PS + NP] - Y[O_PE + NP + NS] - value)));
    } else if (mode == PLH_MODE_DT) {                                                  // constant_temperature: value - sum w_i YP[T_i] / L
      const double sT = wave_sum(lane < NT ? TP.wT5[tsec_of(lane)] * YP[O_T + lane] : 0.0);
      if (lane == 0) Fo[O_I] = value - sT;
    } else {                                                                           // algebraic twin: YP_T -> rhs_T(Y), scalar_residual.jl:347-372
      PL_SYNC();
      const double sT = wave_sum(lane < NT ? TP.wT5[tsec_of(lane)] * (Fo[O_T + lane] + YP[O_T + lane]) : 0.0);
      if (lane == 0) Fo[O_I] = value - sT;
    }
  }
  if (WANT_JAC) {
    if (lane == 0) { TP.qIJ[0] = 2.0 * TP.qI[0] * yI; TP.qIJ[1] = 2.0 * TP.qI[1] * yI; }   // d(collector row)/dI (Joule heat ~ I^2)
    if (lane == 0) { S.ctrlJ[0] = yI * cI1C; S.ctrlJ[1] = (Y[O_PS] - Y[O_PS + NJ - 1]) * cI1C; }   // scalar_jacobian! of method_P
    const double dKh_a = beta * K_n * K_n * (rdenK * rdenK), dKh_b = (1 - beta) * K * K * (rdenK * rdenK);   // dKh/dK_i, dKh/dK_{i+1}
    const double rdenC = pl_rcp(denC);
    const double dcb_a = beta * ce_n * ce_n * (rdenC * rdenC), dcb_b = (1 - beta) * ce * ce * (rdenC * rdenC);
    const double dTb_a = beta * T_n * T_n * (rdenT * rdenT), dTb_b = (1 - beta) * T * T * (rdenT * rdenT);
    const double Tq = Tb * rdist;
    const double dcn = ce_n - ce;
    const double dg_a = Tq * (dKh_a * dKc * dcn * rcb - Kh * rcb - Kh * dcn * dcb_a * (rcb * rcb));
    const double dg_b = Tq * (dKh_b * dKc_n * dcn * rcb + Kh * rcb - Kh * dcn * dcb_b * (rcb * rcb));
    const double Ea = edge ? (pe - pe_n) * dKh_a * dKc * rdist + cKfac * dg_a : 0.0;
    const double Eb = edge ? (pe - pe_n) * dKh_b * dKc_n * rdist + cKfac * dg_b : 0.0;
    const double gq = dc * rcb;                                                         // g = Kh Tb gq
    const double ETa = edge ? (pe - pe_n) * dKh_a * dKT * rdist + cKfac * gq * (dKh_a * dKT * Tb + Kh * dTb_a) : 0.0;
    const double ETb = edge ? (pe - pe_n) * dKh_b * dKT_n * rdist + cKfac * gq * (dKh_b * dKT_n * Tb + Kh * dTb_b) : 0.0;
    const double we = edge ? w : 0.0;
    const double dDh_a = LGM ? dD * beta * D_n * D_n * (rdenD * rdenD) : 0.0, dDh_b = LGM ? dD_n * (1 - beta) * D * D * (rdenD * rdenD) : 0.0;
    const double Na = edge ? (dDh_a * (ce_n - ce) - Dhm) * rdist : 0.0, Nb = edge ? (dDh_b * (ce_n - ce) + Dhm) * rdist : 0.0;
    const double Ea_p = shift_up1(Ea), Eb_p = shift_up1(Eb), we_p = shift_up1(we), Na_p = shift_up1(Na), Nb_p = shift_up1(Nb);
    const double ETa_p = shift_up1(ETa), ETb_p = shift_up1(ETb);
    if (act) {
      const double rhe = rh * reps;
      S.ceL[i] = i > 0 ? -Na_p * rhe : 0.0;
      S.ceD[i] = (Na - (i > 0 ? Nb_p : 0.0)) * rhe;
      S.ceU[i] = Nb * rhe;
      S.ceJ[i] = elec ? (1 - ctplus) * a * reps : 0.0;
      if (i < NE - 1) {
        S.peL[i] = i > 0 ? -we_p : 0.0; S.peD[i] = (i > 0 ? we_p : 0.0) + we; S.peU[i] = -we;
        S.pcL[i] = i > 0 ? -Ea_p : 0.0; S.pcD[i] = Ea - (i > 0 ? Eb_p : 0.0); S.pcU[i] = Eb;
        TP.ptL[i] = i > 0 ? -ETa_p : 0.0; TP.ptD[i] = ETa - (i > 0 ? ETb_p : 0.0); TP.ptU[i] = ETb;
        S.peJ[i] = elec ? -h * FAR * a : 0.0;
      } else {
        S.peL[i] = 0; S.peD[i] = 1.0; S.peU[i] = 0; S.pcL[i] = 0; S.pcD[i] = 0; S.pcU[i] = 0; S.peJ[i] = 0;
        TP.ptL[i] = 0; TP.ptD[i] = 0; TP.ptU[i] = 0;
      }
      if (elec) {
        S.gce[jx] = kk * sh * cs * (cmax - cs) * inv_sq;
        S.gcs[jx] = 2.0 * kk * (sh * ce * (cmax - 2 * cs) * 0.5 * inv_sq + sq * chh * fRT * (-dU * rcm));
        S.gps[jx] = 2.0 * kk * sq * chh * fRT;
        S.gpe[jx] = -S.gps[jx];
        TP.gT[jx] = 2.0 * kk * sq * (EaK * rT * rT * sh + chh * (-xx * rT - fRT * dUdT));
        S.psJ[jx] = -h * h * a * FAR * rsg;
        TP.TJ[jx] = rc * FAR * a * (T * dUdT + eta);
        TP.Tcs[jx] = rc * Faj * (T * ddUdT - dU) * rcm;
      }
      // T row: couplings to (c_e, Phi_e, Phi_s, T) of nodes i-1, i, i+1 (+ the second neighbour at the four one-sided stencils)
      const double rce = pl_rcp(ce);
      const double qPe = 2.0 * K * dPe + cKfac * K * T * (dce * rce);               // dQ/d(dPe)
      const double qCe = cKfac * K * T * dPe * rce;                                  // dQ/d(dce)
      const double qPs = elec ? 2.0 * sg * dPs : 0.0;                                // dQ/d(dPs)
      const double qce_loc = dKc * dPe * dPe + cKfac * T * dPe * (dKc * dce * rce - K * dce * (rce * rce));
      TP.TcL[i] = rc * qCe * em; TP.TcD[i] = rc * (qCe * e0 + qce_loc); TP.TcU[i] = rc * qCe * ep;
      TP.TeL[i] = rc * qPe * em; TP.TeD[i] = rc * (qPe * e0 - Faj); TP.TeU[i] = rc * qPe * ep;
      TP.TsL[i] = rc * qPs * sm; TP.TsD[i] = rc * (qPs * s0 + Faj); TP.TsU[i] = rc * qPs * sp;
      TP.TtD[i] = rc * (dKT * dPe * dPe + cKfac * (dKT * T + K) * (dce * rce) * dPe);   // (the reversible + reaction heat has no net dT term)
      if (i == 0) { TP.TX2[0][0] = rc * qCe * e2; TP.TX2[0][1] = rc * qPe * e2; TP.TX2[0][2] = rc * qPs * s2; }
      if (i == NP - 1) { TP.TX2[1][0] = 0.0; TP.TX2[1][1] = 0.0; TP.TX2[1][2] = rc * qPs * s2; }
      if (i == NP + NS) { TP.TX2[2][0] = 0.0; TP.TX2[2][1] = 0.0; TP.TX2[2][2] = rc * qPs * s2; }
      if (i == NE - 1) { TP.TX2[3][0] = rc * qCe * e2; TP.TX2[3][1] = rc * qPe * e2; TP.TX2[3][2] = rc * qPs * s2; }
    }
  }
}

// c_s rows with per-particle kappa(T) (residuals_c_s_avg!, residuals.jl:128-180); WANT_JAC also stores W c (needed for the T column)
template <bool WANT_JAC, class M>
PL_DEV void thermal_cs_rows(CellLDS<M>& S, const Tables* __restrict__ tb, const double* Y, const double* YP, double* Fo) {
  PL_MODEL(M);
  const int lane = lane_id();
  const CellConst& c = S.cc;
  auto& TP = S.th;
  if constexpr (PL_THROWB) {
    // r06, ROW layout (dfn_cell.h rowb_fmac): one particle per 16-lane DPP row, lane -> (particle = pass * 4 + lane / 16, row = lane % 16); the particle's vector costs one
    // LDS load per lane and pass, the sums run over row_newbcast operands -- same terms, same order as the LDS form below
    const int q = lane >> 4, rr = lane & 15, rc = rr < NR ? rr : NR - 1;
    double Mr_[NR], Wr_[WANT_JAC ? NR : 1];
    for (int k = 0; k < NR; k++) { Mr_[k] = S.Mr[rc * NR + k]; if constexpr (WANT_JAC) Wr_[k] = S.Mr[S.OFF_WR + rc * NR + k]; }
    [[maybe_unused]] double MrN[NR_EQ ? 1 : NR], WrN[(NR_EQ || !WANT_JAC) ? 1 : NR];
    if constexpr (!NR_EQ) for (int k = 0; k < NR; k++) { MrN[k] = S.Mr[S.mr_el(1) + rc * NR + k]; if constexpr (WANT_JAC) WrN[k] = S.Mr[S.mr_el(1) + S.OFF_WR + rc * NR + k]; }
    int pp[CSD_PASS]; double acc[CSD_PASS], wc[CSD_PASS], cv[CSD_PASS], jv[CSD_PASS], ypv[CSD_PASS], kp[CSD_PASS];
#pragma unroll
    for (int pass = 0; pass < CSD_PASS; pass++) {
      const int p0 = pass * CSD_G + q; pp[pass] = p0 < NJ ? p0 : NJ - 1; acc[pass] = 0.0; wc[pass] = 0.0;
      const int rk = rc < nr_of(pp[pass]) ? rc : nr_of(pp[pass]) - 1;
      cv[pass] = Y[O_CS + cs_off(pp[pass]) + rk]; jv[pass] = Y[O_J + pp[pass]]; ypv[pass] = YP[O_CS + cs_off(pp[pass]) + rk]; kp[pass] = TP.kapP[pp[pass]];
    }
    csd_settle(cv);
    static_for<0, NR>([&](auto kc) {
      constexpr int k = decltype(kc)::value;
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) {
        if constexpr (NR_EQ) { rowb_fmac<k>(acc[pass], cv[pass], Mr_[k]); if constexpr (WANT_JAC) rowb_fmac<k>(wc[pass], cv[pass], Wr_[k]); }
        else { rowb_fmac<k>(acc[pass], cv[pass], pp[pass] < NP ? Mr_[k] : MrN[k]); if constexpr (WANT_JAC) rowb_fmac<k>(wc[pass], cv[pass], pp[pass] < NP ? Wr_[k] : WrN[k]); }
      }
    });
#pragma unroll
    for (int pass = 0; pass < CSD_PASS; pass++) {
      const int p0 = pass * CSD_G + q, p = pp[pass];
      double rhs = kp[pass] * acc[pass];
      if (rr == nr_of(p) - 1) rhs += (p < NP ? c.bj_p : c.bj_n) * jv[pass];
      if (p0 < NJ && rr < NR) { if (rr < nr_of(p)) Fo[O_CS + cs_off(p) + rr] = rhs - ypv[pass]; if (WANT_JAC) TP.AinvQ[p][rr] = wc[pass]; }
    }
    return;
  }
  const int r = lane % NR, g = lane < CS_LANES ? lane / NR : CS_G - 1;
  double Mrow[NR], Wrow[NR];
  // (the radial operator from its LDS copy, as in the isothermal models: r03 fetched it from the table in global memory in every residual and every solve)
  for (int k = 0; k < NR; k++) { Mrow[k] = S.Mr[r * NR + k]; if (WANT_JAC) Wrow[k] = S.Mr[S.OFF_WR + r * NR + k]; }
  // (N_r_p != N_r_n: the anode's rows of M and W, zero-padded to the common stride like the cathode's -- a sum over k < NR is the sum over the particle's own rows, a lane
  //  whose row does not exist in its particle computes 0 for W c and stores no residual)
  [[maybe_unused]] double MrowN[NR_EQ ? 1 : NR], WrowN[NR_EQ ? 1 : NR];
  if constexpr (!NR_EQ) for (int k = 0; k < NR; k++) { MrowN[k] = S.Mr[S.mr_el(1) + r * NR + k]; if (WANT_JAC) WrowN[k] = S.Mr[S.mr_el(1) + S.OFF_WR + r * NR + k]; }
#pragma unroll
  for (int pass = 0; pass < CS_PASS; pass++) {
    const int p0 = pass * CS_G + g, p = p0 < NJ ? p0 : NJ - 1;
    double acc = 0.0, wc = 0.0;
#pragma unroll
    for (int k = 0; k < NR; k++) {
      const double v = Y[O_CS + cs_off(p) + k];
      if constexpr (NR_EQ) { acc += Mrow[k] * v; if (WANT_JAC) wc += Wrow[k] * v; }
      else { acc += (p < NP ? Mrow[k] : MrowN[k]) * v; if (WANT_JAC) wc += (p < NP ? Wrow[k] : WrowN[k]) * v; }
    }
    // (every load unconditional -- p is clamped --, only the stores guarded: a load under `if` is one exec-masked LDS round trip of its own)
    const double jv = Y[O_J + p], ypv = YP[O_CS + cs_off(p) + r];
    double rhs = TP.kapP[p] * acc;
    if (r == nr_of(p) - 1) rhs += (p < NP ? c.bj_p : c.bj_n) * jv;
    if (lane < CS_LANES && p0 < NJ) { if (r < nr_of(p)) Fo[O_CS + cs_off(p) + r] = rhs - ypv; if (WANT_JAC) TP.AinvQ[p][r] = wc; }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// linear algebra
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv4(const double* A, double* B) {
  // cofactor expansion through the 2x2 minors of the top and bottom row pairs
  const double s0 = A[0] * A[5] - A[4] * A[1], s1 = A[0] * A[6] - A[4] * A[2], s2 = A[0] * A[7] - A[4] * A[3];
  const double s3 = A[1] * A[6] - A[5] * A[2], s4 = A[1] * A[7] - A[5] * A[3], s5 = A[2] * A[7] - A[6] * A[3];
  const double c5 = A[10] * A[15] - A[14] * A[11], c4 = A[9] * A[15] - A[13] * A[11], c3 = A[9] * A[14] - A[13] * A[10];
  const double c2 = A[8] * A[15] - A[12] * A[11], c1 = A[8] * A[14] - A[12] * A[10], c0 = A[8] * A[13] - A[12] * A[9];
  const double det = s0 * c5 - s1 * c4 + s2 * c3 + s3 * c2 - s4 * c1 + s5 * c0;
  const double id = pl_rcp(det);
  B[0] = (A[5] * c5 - A[6] * c4 + A[7] * c3) * id;
  B[1] = (-A[1] * c5 + A[2] * c4 - A[3] * c3) * id;
  B[2] = (A[13] * s5 - A[14] * s4 + A[15] * s3) * id;
  B[3] = (-A[9] * s5 + A[10] * s4 - A[11] * s3) * id;
  B[4] = (-A[4] * c5 + A[6] * c2 - A[7] * c1) * id;
  B[5] = (A[0] * c5 - A[2] * c2 + A[3] * c1) * id;
  B[6] = (-A[12] * s5 + A[14] * s2 - A[15] * s1) * id;
  B[7] = (A[8] * s5 - A[10] * s2 + A[11] * s1) * id;
  B[8] = (A[4] * c4 - A[5] * c2 + A[7] * c0) * id;
  B[9] = (-A[0] * c4 + A[1] * c2 - A[3] * c0) * id;
  B[10] = (A[12] * s4 - A[13] * s2 + A[15] * s0) * id;
  B[11] = (-A[8] * s4 + A[9] * s2 - A[11] * s0) * id;
  B[12] = (-A[4] * c3 + A[5] * c1 - A[6] * c0) * id;
  B[13] = (A[0] * c3 - A[1] * c1 + A[2] * c0) * id;
  B[14] = (-A[12] * s3 + A[13] * s1 - A[14] * s0) * id;
  B[15] = (A[8] * s3 - A[9] * s1 + A[10] * s0) * id;
}

// the sparse off-diagonal node blocks.  Lower block of node i (rows of i x unknowns of i-1) / upper block (rows of i x unknowns of i+1):
//   [ ce 0 0 0 ; pc pe 0 pt ; 0 0 s 0 ; Tc Te Ts Tt ]
struct OffBlk { double ce, pc, pe, pt, s, Tc, Te, Ts, Tt; };
// (every LDS operand is loaded unconditionally -- i is a valid node in every lane -- and masked afterwards: `cond ? 0.0 : S.x[i]` compiles to one exec-masked branch with
//  its own LDS round trip PER ELEMENT, which serialised the prologue of every solve: 16 + 16 + 9 dependent round trips in thermal_sweeps)
template <class M> __device__ __forceinline__ OffBlk lower_blk(const CellLDS<M>& S, int i, bool alg_only) {
  OffBlk b; const auto& TP = S.th;
  const int sci = sec_of(i), scp = sec_of(i > 0 ? i - 1 : 0);
  const bool z = i == 0;
  const double ce = S.ceL[i], pc = S.pcL[i], pe = S.peL[i], pt = TP.ptL[i], Tc = TP.TcL[i], Te = TP.TeL[i], Ts = TP.TsL[i], Tt = TP.aL[NA + i];
  const bool zt = alg_only || z;
  b.ce = zt ? 0.0 : ce; b.pc = zt ? 0.0 : pc; b.pe = z ? 0.0 : pe; b.pt = zt ? 0.0 : pt;
  b.s = (!z && sci != 1 && scp == sci) ? 1.0 : 0.0;
  b.Tc = zt ? 0.0 : Tc; b.Te = zt ? 0.0 : Te; b.Ts = zt ? 0.0 : Ts; b.Tt = zt ? 0.0 : Tt;
  return b;
}
template <class M> __device__ __forceinline__ OffBlk upper_blk(const CellLDS<M>& S, int i, bool alg_only) {
  OffBlk b; const auto& TP = S.th;
  const int sci = sec_of(i), scn = sec_of(i < NE - 1 ? i + 1 : NE - 1);
  const bool z = i == NE - 1;
  const double ce = S.ceU[i], pc = S.pcU[i], pe = S.peU[i], pt = TP.ptU[i], Tc = TP.TcU[i], Te = TP.TeU[i], Ts = TP.TsU[i], Tt = TP.aU[NA + i];
  const bool zt = alg_only || z;
  b.ce = zt ? 0.0 : ce; b.pc = zt ? 0.0 : pc; b.pe = z ? 0.0 : pe; b.pt = zt ? 0.0 : pt;
  b.s = (!z && sci != 1 && scn == sci) ? 1.0 : 0.0;
  b.Tc = zt ? 0.0 : Tc; b.Te = zt ? 0.0 : Te; b.Ts = zt ? 0.0 : Ts; b.Tt = zt ? 0.0 : Tt;
  return b;
}
// the block of node i on the side given per lane (top half: upper block, bottom half: lower block, or the other way round): ONE set of loads through a per-lane address
template <class M> __device__ __forceinline__ OffBlk side_blk(const CellLDS<M>& S, int i, bool upper, bool alg_only) {
  OffBlk b; const auto& TP = S.th;
  const int sci = sec_of(i), scx = sec_of(upper ? (i < NE - 1 ? i + 1 : NE - 1) : (i > 0 ? i - 1 : 0));
  const bool z = upper ? i == NE - 1 : i == 0;
  // ceL / ceD / ceU ... are consecutive arrays of NE doubles: the upper block is the lower one 2 NE doubles further on (ceL, ceD, ceU; likewise pe, pc, pt, Tc, Te, Ts); aL -> aU: NT
  const int o = upper ? 2 * NE : 0;
  static_assert(sizeof(S.ceL) == NE * sizeof(double), "layout");
  const double ce = (&S.ceL[0])[o + i], pc = (&S.pcL[0])[o + i], pe = (&S.peL[0])[o + i], pt = (&TP.ptL[0])[o + i], Tc = (&TP.TcL[0])[o + i], Te = (&TP.TeL[0])[o + i],
               Ts = (&TP.TsL[0])[o + i], Tt = (&TP.aL[0])[(upper ? NT : 0) + NA + i];
  const bool zt = alg_only || z;
  b.ce = zt ? 0.0 : ce; b.pc = zt ? 0.0 : pc; b.pe = z ? 0.0 : pe; b.pt = zt ? 0.0 : pt;
  b.s = (!z && sci != 1 && scx == sci) ? 1.0 : 0.0;
  b.Tc = zt ? 0.0 : Tc; b.Te = zt ? 0.0 : Te; b.Ts = zt ? 0.0 : Ts; b.Tt = zt ? 0.0 : Tt;
  return b;
}

// twisted systolic forward/backward substitution for NRHS right-hand sides at once, in the mirrored lane layout of thomas_sweeps
// (dfn_cell.h): the lane of node n (tw_lane) holds r[q][0..3] on entry and the solution on exit
template <int NRHS, class M>
__device__ __forceinline__ void thermal_sweeps(const CellLDS<M>& S, bool alg_only, double (&r)[NRHS][4]) {
  const int lane = lane_id();
  const int nd = tw_node(lane);
  const bool act = nd >= 0, top = lane < TW_MID;
  const int i = act ? nd : 0;
  // register diet: C (forward sweep), then Lm (closing), then G (backward sweep) are formed one after the other, so that at most two of the 4x4 blocks
  // are live at a time next to the integrator's per-lane state (the three at once were half of the register file); PL_SYNC keeps the loads where they are
  double C[16], Di[16];
  // (r06: no masks -- an idle lane reads node 0's factors, the head of the forward chain whose L D'^-1 is zero, and with its right-hand side zeroed below its recurrence
  //  stays at zero: dfn_cell.h thomas_sweeps.  64 selects less per solve)
  for (int k = 0; k < 16; k++) { C[k] = S.LD[k][i]; Di[k] = S.Dinv[k][i]; }
  // ---- the four T rows with a second-neighbour entry (one-sided stencils at nodes 0, 9, 20, 29) inside the twisted elimination ----
  // "far ahead" (node 0 -> node 2, node 29 -> node 27): eliminating x_0 puts -LD_1[:,3] (x) w into U_1, so node 1's (node 28's)
  // back-substitution block is G - (Dinv C[:,3]) (x) w, and x_0 (x_29) gets -Dinv[:,3] (w . x_2) after the sweep.
  // "far behind" (node 9 -> node 7, node 20 -> node 22): the source is already eliminated, its share q . y_src leaves the right-hand side
  // after stage 7 (q and the modified L_9 / U_20 come from the factorisation).
  const auto& TPs = S.th;
  const bool far_ahead_nb = !alg_only && (nd == 1 || nd == NE - 2), far_ahead = !alg_only && (nd == 0 || nd == NE - 1);
  const bool far_behind = !alg_only && (nd == NP - 1 || nd == NP + NS);
  const int fk = (nd == 0 || nd == 1) ? 0 : 3;
  const double w0 = TPs.TX2[fk][0], w1 = TPs.TX2[fk][1], w2 = TPs.TX2[fk][2];
  const double cT[4] = {C[3], C[7], C[11], C[15]};        // T column of LD (outlives C: the far-ahead correction of G)
  double qf[4] = {0.0, 0.0, 0.0, 0.0};
  if (far_behind) for (int k = 0; k < 4; k++) qf[k] = TPs.qfar[nd == NP - 1 ? 0 : 1][k];
  double y[NRHS][4];
  for (int q = 0; q < NRHS; q++) for (int k = 0; k < 4; k++) { if (!act) r[q][k] = 0.0; y[q][k] = r[q][k]; }
  // nodes N_p - 3 (top chain) and N_p + N_s + 2 (bottom chain) are final after FAR_T = N_p - 3 and FAR_B = N_n - 3 stages of their chains (7 and 7 on the default
  // grid: one interruption of the stage loop serves both; two when N_p != N_n)
  constexpr int FAR_T = NP - 3, FAR_B = NN - 3, FAR_LO = FAR_T < FAR_B ? FAR_T : FAR_B, FAR_HI = FAR_T < FAR_B ? FAR_B : FAR_T;
  constexpr int NSEG = FAR_LO == FAR_HI ? 2 : 3;
  // r06b: one level of recursive doubling, as in thomas_sweeps (dfn_cell.h): y_n = (r_n - C_n r_{n-1}) + (C_n C_{n-1}) y_{n-2} -- the odd and the even nodes of a half
  // advance together two lanes apart (row_shr:2: each half sits inside one 16-lane DPP row) in 7 stages instead of 14, for one parallel pre-pass (the 4x4 product P = C C_prev
  // and the shifted right-hand side: 80 FMAs).  The chain of DPP shift + four dependent FMAs per stage is latency at one wave per SIMD: 5.7 k cycles per solve before.  The
  // far-behind share q . y(node N_p - 3) leaves r_{N_p - 1} once that y is final (after stage (N_p - 3) / 2); r_{N_p - 1} sits in TWO of the combined right-hand sides,
  // its own and the next node's (through -C_next r_{N_p - 1}), so the next lane gets + C_next[:, 3] x the same share.  Default-shaped grids only (both electrodes alike,
  // halves inside a DPP row); one right-hand side.
#ifdef PL_NO_STRIDE2T
  constexpr bool STRIDE2T = false;
#else
  constexpr bool STRIDE2T = NRHS == 1 && FAR_T == FAR_B && FAR_T >= 1 && TW_FWD <= 15 && TW_MID <= 15;
#endif
  if constexpr (STRIDE2T) {
    double P[16];
    {
      double Cs[16];
#pragma unroll
      for (int k = 0; k < 16; k++) Cs[k] = shift_up1(C[k]);                 // C of the previous node of the chain
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) P[a * 4 + b] = ((C[a * 4] * Cs[b] + C[a * 4 + 1] * Cs[4 + b]) + C[a * 4 + 2] * Cs[8 + b]) + C[a * 4 + 3] * Cs[12 + b];
      const double s0 = shift_up1(r[0][0]), s1 = shift_up1(r[0][1]), s2 = shift_up1(r[0][2]), s3 = shift_up1(r[0][3]);
#pragma unroll
      for (int rr = 0; rr < 4; rr++) r[0][rr] = PL_NMS4(r[0][rr], C[rr * 4], s0, C[rr * 4 + 1], s1, C[rr * 4 + 2], s2, C[rr * 4 + 3], s3);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) y[0][k] = r[0][k];
    constexpr int S_INT = FAR_T / 2, NST2 = (TW_FWD - 1 + 1) / 2;          // positions <= 2 s + 1 of a chain are final after s stages; the last position is TW_FWD - 1
    auto stages = [&](int lo, int hi) {
#pragma unroll
      for (int it = lo; it < hi; it++) {
        const double p0 = row_up2(y[0][0]), p1 = row_up2(y[0][1]), p2 = row_up2(y[0][2]), p3 = row_up2(y[0][3]);
#pragma unroll
        for (int rr = 0; rr < 4; rr++) y[0][rr] = (((r[0][rr] + P[rr * 4] * p0) + P[rr * 4 + 1] * p1) + P[rr * 4 + 2] * p2) + P[rr * 4 + 3] * p3;
      }
    };
    stages(0, S_INT);
    {
      double d = 0.0;
#pragma unroll
      for (int k = 0; k < 4; k++) { const double a7 = lane_bcast(y[0][k], tw_lane(NP - 3)), a22 = lane_bcast(y[0][k], tw_lane(NP + NS + 2)); d += qf[k] * (nd == NP - 1 ? a7 : a22); }
      const double dm = far_behind ? d : 0.0, dn = shift_up1(dm);          // (qf = 0 outside the two far-behind lanes; dn: the share as the next lane of the chain sees it)
      r[0][3] -= dm;
#pragma unroll
      for (int k = 0; k < 4; k++) r[0][k] += cT[k] * dn;
    }
    stages(S_INT, NST2);
  } else {
#pragma unroll 1
  for (int seg = 0; seg < NSEG; seg++) {
    const int lo = seg == 0 ? 1 : (seg == 1 ? FAR_LO + 1 : FAR_HI + 1), hi = seg == 0 ? FAR_LO + 1 : ((seg == 1 && NSEG == 3) ? FAR_HI + 1 : TW_FWD);
#pragma unroll 2
    for (int itr = lo; itr < hi; itr++) {
#pragma unroll
      for (int q = 0; q < NRHS; q++) {
        const double p0 = shift_up1(y[q][0]), p1 = shift_up1(y[q][1]), p2 = shift_up1(y[q][2]), p3 = shift_up1(y[q][3]);
#pragma unroll
        for (int rr = 0; rr < 4; rr++) y[q][rr] = PL_NMS4(r[q][rr], C[rr * 4], p0, C[rr * 4 + 1], p1, C[rr * 4 + 2], p2, C[rr * 4 + 3], p3);
      }
    }
    if (seg < NSEG - 1) {                                   // right-hand side of node N_p - 1 / N_p + N_s: minus q . y(node N_p - 3 / N_p + N_s + 2), once that y is final
      const int stage = seg == 0 ? FAR_LO : FAR_HI;
      const bool mine = (nd == NP - 1 && stage == FAR_T) || (nd == NP + NS && stage == FAR_B);
#pragma unroll
      for (int q = 0; q < NRHS; q++) {
        double d = 0.0;
#pragma unroll
        for (int k = 0; k < 4; k++) { const double a7 = lane_bcast(y[q][k], tw_lane(NP - 3)), a22 = lane_bcast(y[q][k], tw_lane(NP + NS + 2)); d += qf[k] * (nd == NP - 1 ? a7 : a22); }
        if (mine) r[q][3] -= d;                             // (qf = 0 in every other lane)
      }
    }
  }
  }
  PL_SYNC();
  double Lm[16];
  for (int k = 0; k < 16; k++) Lm[k] = S.LDmid[k];          // (unmasked: the broadcast it multiplies is, 8 selects instead of 32)
  const bool mid = nd == TW_MID;
  double z[NRHS][4];
#pragma unroll
  for (int q = 0; q < NRHS; q++) {
    const double bm0 = lane_bcast(y[q][0], TW_MID - 1), bm1 = lane_bcast(y[q][1], TW_MID - 1), bm2 = lane_bcast(y[q][2], TW_MID - 1), bm3 = lane_bcast(y[q][3], TW_MID - 1);
    const double m0 = mid ? bm0 : 0.0, m1 = mid ? bm1 : 0.0, m2 = mid ? bm2 : 0.0, m3 = mid ? bm3 : 0.0;
#pragma unroll
    for (int rr = 0; rr < 4; rr++) y[q][rr] = PL_NMS4(y[q][rr], Lm[rr * 4], m0, Lm[rr * 4 + 1], m1, Lm[rr * 4 + 2], m2, Lm[rr * 4 + 3], m3);
#pragma unroll
    for (int rr = 0; rr < 4; rr++) z[q][rr] = Di[rr * 4] * y[q][0] + Di[rr * 4 + 1] * y[q][1] + Di[rr * 4 + 2] * y[q][2] + Di[rr * 4 + 3] * y[q][3];
#pragma unroll
    for (int rr = 0; rr < 4; rr++) { const double gz = lane_bcast(z[q][rr], tw_lane(TW_MID)); if (lane == TW_MID) z[q][rr] = gz; }   // ghost of the closing node
    for (int k = 0; k < 4; k++) r[q][k] = z[q][k];
  }
  PL_SYNC();
  double G[16];
  {
    OffBlk u = side_blk(S, i, top, alg_only);     // back-substitution block: upper for the top half, lower for the bottom half (zero for the closing node)
    if (!act || nd == TW_MID) { u.ce = u.pc = u.pe = u.pt = u.s = u.Tc = u.Te = u.Ts = u.Tt = 0.0; }
    for (int rr = 0; rr < 4; rr++) {       // G = Dinv U
      const double d0 = Di[rr * 4], d1 = Di[rr * 4 + 1], d2 = Di[rr * 4 + 2], d3 = Di[rr * 4 + 3];
      G[rr * 4 + 0] = d0 * u.ce + d1 * u.pc + d3 * u.Tc;
      G[rr * 4 + 1] = d1 * u.pe + d3 * u.Te;
      G[rr * 4 + 2] = d2 * u.s + d3 * u.Ts;
      G[rr * 4 + 3] = d1 * u.pt + d3 * u.Tt;
    }
    if (far_ahead_nb) {
      for (int rr = 0; rr < 4; rr++) {
        const double t = Di[rr * 4] * cT[0] + Di[rr * 4 + 1] * cT[1] + Di[rr * 4 + 2] * cT[2] + Di[rr * 4 + 3] * cT[3];
        G[rr * 4 + 0] -= t * w0; G[rr * 4 + 1] -= t * w1; G[rr * 4 + 2] -= t * w2;
      }
    }
  }
  if constexpr (STRIDE2T) {                                                // x_n = (z_n - G_n z_{n+1}) + (G_n G_{n+1}) x_{n+2}  ("n+1" = the next lane of the chain)
    double Q[16];
    {
      double Gs[16];
#pragma unroll
      for (int k = 0; k < 16; k++) Gs[k] = shift_down1(G[k]);
#pragma unroll
      for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) Q[a * 4 + b] = ((G[a * 4] * Gs[b] + G[a * 4 + 1] * Gs[4 + b]) + G[a * 4 + 2] * Gs[8 + b]) + G[a * 4 + 3] * Gs[12 + b];
      const double s0 = shift_down1(z[0][0]), s1 = shift_down1(z[0][1]), s2 = shift_down1(z[0][2]), s3 = shift_down1(z[0][3]);
#pragma unroll
      for (int rr = 0; rr < 4; rr++) z[0][rr] = PL_NMS4(z[0][rr], G[rr * 4], s0, G[rr * 4 + 1], s1, G[rr * 4 + 2], s2, G[rr * 4 + 3], s3);
    }
#pragma unroll
    for (int k = 0; k < 4; k++) r[0][k] = z[0][k];
    constexpr int NST2B = (TW_MID + 1) / 2;                                 // lane 0 is TW_MID steps from the ghost of the closing node
#pragma unroll
    for (int it = 0; it < NST2B; it++) {
      const double q0 = row_down2(r[0][0]), q1 = row_down2(r[0][1]), q2 = row_down2(r[0][2]), q3 = row_down2(r[0][3]);
#pragma unroll
      for (int rr = 0; rr < 4; rr++) r[0][rr] = (((z[0][rr] + Q[rr * 4] * q0) + Q[rr * 4 + 1] * q1) + Q[rr * 4 + 2] * q2) + Q[rr * 4 + 3] * q3;
    }
  } else {
#pragma unroll 2
  for (int itr = 0; itr < TW_MID; itr++) {
#pragma unroll
    for (int q = 0; q < NRHS; q++) {
      const double q0 = shift_down1(r[q][0]), q1 = shift_down1(r[q][1]), q2 = shift_down1(r[q][2]), q3 = shift_down1(r[q][3]);
#pragma unroll
      for (int rr = 0; rr < 4; rr++) r[q][rr] = PL_NMS4(z[q][rr], G[rr * 4], q0, G[rr * 4 + 1], q1, G[rr * 4 + 2], q2, G[rr * 4 + 3], q3);
    }
  }
  }
#pragma unroll
  for (int q = 0; q < NRHS; q++) {                          // x_0 -= Dinv_0[:,3] (w . x_2) ; x_29 -= Dinv_29[:,3] (w . x_27)
    const double a0 = lane_bcast(r[q][0], tw_lane(2)), a1 = lane_bcast(r[q][1], tw_lane(2)), a2 = lane_bcast(r[q][2], tw_lane(2));
    const double b0 = lane_bcast(r[q][0], tw_lane(NE - 3)), b1 = lane_bcast(r[q][1], tw_lane(NE - 3)), b2 = lane_bcast(r[q][2], tw_lane(NE - 3));
    if (far_ahead) {
      const double wx = nd == 0 ? w0 * a0 + w1 * a1 + w2 * a2 : w0 * b0 + w1 * b1 + w2 * b2;
      for (int rr = 0; rr < 4; rr++) r[q][rr] -= Di[rr * 4 + 3] * wx;
    }
  }
}

// nodes of the four T rows with a second-neighbour entry, and the node that entry refers to
__device__ __forceinline__ int wb_src_node(int k) { return k == 0 ? 2 : (k == 1 ? NP - 3 : (k == 2 ? NP + NS + 2 : NE - 3)); }
__device__ __forceinline__ int wb_row_node(int k) { return k == 0 ? 0 : (k == 1 ? NP - 1 : (k == 2 ? NP + NS : NE - 1)); }

template <class M>
PL_DEV_FACTOR void thermal_factor(CellLDS<M>& S, LaneRegs& R, const Tables* __restrict__ tb, double cj, int mode, bool alg_only) {
  PL_MODEL(M);
  const int lane = lane_id();
  const CellConst& c = S.cc;
  auto& TP = S.th;
  const int r = lane % NR, g = lane < CS_LANES ? lane / NR : CS_G - 1;
  PL_TICD();
  // 1. particle resolvents in spectral form: A_p^-1 = V diag(1 / (kappa_p lam_m - cj)) W.  Lane (g, r) owns mode r of its particles and forms the reciprocals of ITS modes
  //    only (CS_PASS divisions per lane; r03 divided inside the sums: N_r CS_PASS of them).  They stay in registers for the solves (R.rcp) and are shared with the other
  //    lanes of the particle through the c_s section of S.yy, which is dead between a residual evaluation and the next form_iterate (thermal_solve uses it the same way).
  if (!alg_only) {
    if (lane < NJ) TP.kapF[lane] = TP.kapP[lane];
    if constexpr (PL_THROWB) {
      // r06, ROW layout: lane (row q, rr) owns mode rr of particle pass * 4 + q.  The reciprocals and W c / d of a particle's modes sit in the lanes of its DPP row: the two
      // sums over the modes take them through row_newbcast (rowb_fmac) -- no round trip through the c_s section of S.yy, no phase separator
      const int q = lane >> 4, rr = lane & 15, rc = rr < NR ? rr : NR - 1;
      const double lam_r = NR_EQ ? PL_RADIAL_LAM[rc] : tb->LAMp(0)[rc];
      [[maybe_unused]] const double lam_rN = NR_EQ ? 0.0 : tb->LAMp(1)[rc];
      double VW[NR], VL[NR];                                  // V[r][m] W[m][last] and V[r][m] lam_m: the constant factors of the two sums
      [[maybe_unused]] double VWN[NR_EQ ? 1 : NR], VLN[NR_EQ ? 1 : NR];
      if constexpr (NR_EQ) { for (int m = 0; m < NR; m++) { const double v = S.Mr[S.OFF_VR + rc * NR + m]; VW[m] = v * PL_RADIAL_W[m * NR + NR - 1]; VL[m] = v * PL_RADIAL_LAM[m]; } }
      else for (int m = 0; m < NR; m++) {
        const double v = S.Mr[S.OFF_VR + rc * NR + m], vn = S.Mr[S.mr_el(1) + S.OFF_VR + rc * NR + m];
        VW[m] = v * tb->Wp(0)[m * NR + NRP - 1]; VL[m] = v * tb->LAMp(0)[m]; VWN[m] = vn * tb->Wp(1)[m * NR + NRN - 1]; VLN[m] = vn * tb->LAMp(1)[m];
      }
      int pp[CSD_PASS]; double rcv[CSD_PASS], wq[CSD_PASS], ae[CSD_PASS], aq[CSD_PASS], dk[CSD_PASS];
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) {
        const int p0 = pass * CSD_G + q, pq = p0 < NJ ? p0 : NJ - 1; pp[pass] = pq;
        R.rcp[pass] = pl_rcp(TP.kapP[pq] * ((NR_EQ || pq < NP) ? lam_r : lam_rN) - cj);
        rcv[pass] = R.rcp[pass]; wq[pass] = TP.AinvQ[pq][rc] * R.rcp[pass];      // AinvQ still holds W c
        dk[pass] = TP.dkapP[pq]; ae[pass] = 0.0; aq[pass] = 0.0;
      }
      csd_settle(rcv); csd_settle(wq);
      static_for<0, NR>([&](auto mc) {
        constexpr int m = decltype(mc)::value;
#pragma unroll
        for (int pass = 0; pass < CSD_PASS; pass++) {
          if constexpr (NR_EQ) { rowb_fmac<m>(ae[pass], rcv[pass], VW[m]); rowb_fmac<m>(aq[pass], wq[pass], VL[m]); }
          else { rowb_fmac<m>(ae[pass], rcv[pass], pp[pass] < NP ? VW[m] : VWN[m]); rowb_fmac<m>(aq[pass], wq[pass], pp[pass] < NP ? VL[m] : VLN[m]); }
        }
      });
      PL_SYNC();                                             // every lane has read W c before it is overwritten
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) {
        const int p0 = pass * CSD_G + q;
        if (p0 < NJ && rr < NR) { TP.AinvE[pp[pass]][rr] = ae[pass]; TP.AinvQ[pp[pass]][rr] = aq[pass] * dk[pass]; }
      }
    } else {
    // (N_r_p != N_r_n: eigenvalues and W[:, last] of the particle's own electrode from the padded tables -- a padded mode has lam = 0, its reciprocal -1/cj meets V = 0 --; the
    //  reciprocals are shared through the particle's own c_s entries of S.yy, S.yy[O_CS + cs_off(p) + mode])
    const double lam_r = NR_EQ ? PL_RADIAL_LAM[r] : tb->LAMp(0)[r];
    [[maybe_unused]] const double lam_rN = NR_EQ ? 0.0 : tb->LAMp(1)[r];
    const int cs0 = PL_OPAQUE_IDX(g * NR + r);
    double wc[CS_PASS];
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p0 = pass * CS_G + g, p = p0 < NJ ? p0 : NJ - 1;
      R.rcp[pass] = 1.0 / (TP.kapP[p] * ((NR_EQ || p < NP) ? lam_r : lam_rN) - cj);
      wc[pass] = (&TP.AinvQ[0][0])[(p0 < NJ ? pass * CS_G * NR : (NJ - 1 - g) * NR) + cs0];      // AinvQ still holds W c  (clamped like p)
    }
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p0 = pass * CS_G + g;
      if (lane < CS_LANES && p0 < NJ) {
        if constexpr (NR_EQ) S.yy[O_CS + pass * CS_G * NR + cs0] = R.rcp[pass];
        else { if (r < nr_of(p0)) S.yy[O_CS + cs_off(p0) + r] = R.rcp[pass]; }
        (&TP.AinvQ[0][0])[pass * CS_G * NR + cs0] = wc[pass] * R.rcp[pass];
      }
    }
    PL_SYNC();
    double VW[NR], VL[NR];                                  // V[r][m] W[m][last] and V[r][m] lam_m: the constant factors of the two sums
    [[maybe_unused]] double VWN[NR_EQ ? 1 : NR], VLN[NR_EQ ? 1 : NR];
    if constexpr (NR_EQ) { for (int m = 0; m < NR; m++) { const double v = S.Mr[S.OFF_VR + r * NR + m]; VW[m] = v * PL_RADIAL_W[m * NR + NR - 1]; VL[m] = v * PL_RADIAL_LAM[m]; } }
    else for (int m = 0; m < NR; m++) {
      const double v = S.Mr[S.OFF_VR + r * NR + m], vn = S.Mr[S.mr_el(1) + S.OFF_VR + r * NR + m];
      VW[m] = v * tb->Wp(0)[m * NR + NRP - 1]; VL[m] = v * tb->LAMp(0)[m]; VWN[m] = vn * tb->Wp(1)[m * NR + NRN - 1]; VLN[m] = vn * tb->LAMp(1)[m];
    }
    double ae[CS_PASS], aq[CS_PASS];
    const int pg0 = PL_OPAQUE_IDX(g * NR);
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p0 = pass * CS_G + g;
      const int base = p0 < NJ ? pass * CS_G * NR + pg0 : (NJ - 1) * NR;
      ae[pass] = 0.0; aq[pass] = 0.0;
      if constexpr (NR_EQ) {
#pragma unroll
        for (int m = 0; m < NR; m++) { ae[pass] += VW[m] * S.yy[O_CS + base + m]; aq[pass] += VL[m] * (&TP.AinvQ[0][0])[base + m]; }
      } else {
        const int pc = p0 < NJ ? p0 : NJ - 1;
        for (int m = 0; m < NR; m++) { ae[pass] += (pc < NP ? VW[m] : VWN[m]) * S.yy[O_CS + cs_off(pc) + m]; aq[pass] += (pc < NP ? VL[m] : VLN[m]) * (&TP.AinvQ[0][0])[base + m]; }
      }
    }
    PL_SYNC();                                             // every lane has read the shared reciprocals and W c / d before they are overwritten
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p0 = pass * CS_G + g, p = p0 < NJ ? p0 : NJ - 1;
      if (lane < CS_LANES && p0 < NJ) { TP.AinvE[p][r] = ae[pass]; TP.AinvQ[p][r] = aq[pass] * TP.dkapP[p]; }
    }
    }
    // 2. collector chains: (aL, aD - cj, aU) x = rhs by the Thomas algorithm, lane 32 + k = collector node k (systolic DPP chains as in thermal_solve: one LDS load per
    //    operand and lane).  Two fixed right-hand sides: the coupling to T of the neighbouring cell node (Al: last row, through aU; Cu: first row, through aL) and the column of I
    {
      const int ck = lane - 32;
      const bool cact = ck >= 0 && ck < NA + NZ;
      const int kq = cact ? ck : 0, q = kq < NA ? 0 : 1, kk = q == 0 ? kq : kq - NA, ic = q == 0 ? kk : NA + NE + kk, nq = q == 0 ? NA : NZ;
      const double l_aL = TP.aL[ic], l_aU = TP.aU[ic], l_aD = thermal_aD(TP, ic), l_qij = TP.qIJ[q];
      const bool head = !cact || kk == 0;
      const double aLk = head ? 0.0 : l_aL, up = (cact && kk < nq - 1) ? l_aU : 0.0, dk = l_aD - cj, qij = cact ? l_qij : 0.0;
      const double aU_prev = shift_up1(l_aU);
      const double rcpl = !cact ? 0.0 : ((q == 0 && kk == NA - 1) ? l_aU : ((q == 1 && kk == 0) ? l_aL : 0.0));
      constexpr int NC = NA > NZ ? NA : NZ;
      double cp = pl_rcp(dk), fc = rcpl, fi = qij;
#pragma unroll 1
      for (int st = 1; st < NC; st++) {
        const double cpp = shift_up1(cp), fcp = shift_up1(fc), fip = shift_up1(fi);
        const double mlt = aLk * cpp;
        cp = pl_rcp(dk - mlt * aU_prev); fc = rcpl - mlt * fcp; fi = qij - mlt * fip;
      }
      double xc = fc * cp, xi = fi * cp;
#pragma unroll
      for (int st = 1; st < NC; st++) { const double xcn = shift_down1(xc), xin = shift_down1(xi); xc = (fc - up * xcn) * cp; xi = (fi - up * xin) * cp; }
      if (cact) { TP.cP[q][kk] = cp; TP.zc[q][kk] = xc; TP.zI[q][kk] = xi; }
    }
  }
  PL_SYNC();
  PL_TOCD(S, 2);
  // 3. node-local elimination of j (and the particle / collector Schur complements)
  if (lane < NE) {
    const int i = lane, sc = sec_of(i);
    double t0 = 0, t1 = 0, t2 = 0, t3 = 0, p0 = 0, p1 = 0, p2 = 0, p3 = 0, cI2 = 0, cI3 = 0;
    if (sc != 1) {
      const int jx = sc == 0 ? i : i - NS;
      const double bj = sc == 0 ? c.bj_p : c.bj_n;
      const double sig = alg_only ? 0.0 : TP.AinvE[jx][nr_of(jx) - 1], tau = alg_only ? 0.0 : TP.AinvQ[jx][nr_of(jx) - 1];
      const double d = -1.0 - S.gcs[jx] * sig * bj;
      const double rd = pl_rcp(d);
      S.dj[jx] = rd;
      p0 = alg_only ? 0.0 : S.gce[jx] * rd; p1 = S.gpe[jx] * rd; p2 = S.gps[jx] * rd; p3 = alg_only ? 0.0 : (TP.gT[jx] - S.gcs[jx] * tau) * rd;
      t0 = alg_only ? 0.0 : S.ceJ[i]; t1 = S.peJ[i]; t2 = S.psJ[jx]; t3 = alg_only ? 0.0 : TP.TJ[jx] - TP.Tcs[jx] * sig * bj;
      if (i == 0) cI2 = c.JI0;
      if (i == NE - 1) cI2 = c.JI29;
    }
    if (!alg_only) {                                       // the collectors' column of I, folded onto T of the neighbouring node
      if (i == 0) cI3 = -TP.aL[NA] * TP.zI[0][NA - 1];
      if (i == NE - 1) cI3 = -TP.aU[NA + NE - 1] * TP.zI[1][0];
    }
    (void)t0; (void)t1; (void)t2; (void)p0; (void)p1; (void)p2;      // (formed again where they are used, from ceJ / peJ / psJ and gce / gpe / gps, dj: thermal_tq / thermal_phi)
    TP.tq3[i] = t3; TP.phi3[i] = p3;
    if (i == 0) { TP.cI4[0] = cI2; TP.cI4[1] = cI3; }
    if (i == NE - 1) { TP.cI4[2] = cI2; TP.cI4[3] = cI3; }
  }
  if (lane == 0) TP.cjf = cj;
  PL_SYNC();
  PL_TOCD(S, 3);
  // 4. twisted block-Thomas factorisation with 4x4 blocks (lane layout and recurrences as in iso_factor step 3)
  const int nd = tw_node(lane);
  const bool act = nd >= 0, top = lane < TW_MID;
  const int i = act ? nd : 0;
  {
    const int sc = sec_of(i);
    const bool elec = sc != 1;
    const int jx = sc == 0 ? i : i - NS;
    double D[16], Dinv[16], LDm[16], Dn[16];
    for (int k = 0; k < 16; k++) { D[k] = 0.0; LDm[k] = 0.0; }
    if (alg_only) { D[0] = 1.0; D[15] = 1.0; }
    else {
      D[0] = S.ceD[i] - cj; D[4] = S.pcD[i]; D[7] = TP.ptD[i];
      D[12] = TP.TcD[i]; D[13] = TP.TeD[i]; D[14] = TP.TsD[i]; D[15] = thermal_aD(TP, NA + i) - cj + TP.TtD[i];
      if (elec) D[15] -= TP.Tcs[jx] * TP.AinvQ[jx][nr_of(jx) - 1];
      if (i == 0) D[15] -= TP.aL[NA] * TP.zc[0][NA - 1];
      if (i == NE - 1) D[15] -= TP.aU[NA + NE - 1] * TP.zc[1][0];
    }
    D[5] = S.peD[i];
    D[10] = 1.0;
    if (elec) {
      const bool first = (i == 0) || (i == NP + NS), last = (i == NP - 1) || (i == NE - 1);
      D[10] = (first || last) ? -1.0 : -2.0;
      {
        const double rd = S.dj[jx];
        const double tqv[4] = {alg_only ? 0.0 : S.ceJ[i], S.peJ[i], S.psJ[jx], TP.tq3[i]}, phv[4] = {alg_only ? 0.0 : S.gce[jx] * rd, S.gpe[jx] * rd, S.gps[jx] * rd, TP.phi3[i]};
        for (int rr = 0; rr < 4; rr++) for (int cc = 0; cc < 4; cc++) D[rr * 4 + cc] -= tqv[rr] * phv[cc];
      }
    }
    // a = left block (L_n top / U_n bottom), b = right block (U_{n-1} top / L_{n+1} bottom); zero at the chain heads and in idle lanes
    OffBlk a = top ? lower_blk(S, i, alg_only) : upper_blk(S, i, alg_only);
    const int nb = top ? (i > 0 ? i - 1 : 0) : (i < NE - 1 ? i + 1 : NE - 1);
    if (!act) { a.ce = a.pc = a.pe = a.pt = a.s = a.Tc = a.Te = a.Ts = a.Tt = 0.0; }
    for (int k = 0; k < 16; k++) Dn[k] = D[k];
    // second-neighbour T-row entries (see thermal_sweeps): nodes 2 / 27 see U_1 / L_28 modified by -fv (x) fw ; nodes 9 / 20 get a modified
    // lower / upper block once the factor of node 7 / 22 is final
    double fv[4] = {0.0, 0.0, 0.0, 0.0}, fw[3] = {0.0, 0.0, 0.0};
    constexpr int FAR_T = NP - 3, FAR_B = NN - 3;          // stage after which the factor of node N_p - 3 (top chain) / N_p + N_s + 2 (bottom chain) is final
    inv4(D, Dinv);
    // the node's own block D and the neighbour's off-diagonal block b stay in registers across the stages (r04: with MachineLICM off -- __graft_entry__.py -- nothing
    // spills any more, and the two LDS round trips per stage that re-read them were 2.5 % of C3; until then they were re-read to keep 50 registers free)
    const OffBlk b = top ? upper_blk(S, nb, alg_only) : lower_blk(S, nb, alg_only);
    double Dk[16];
    for (int k = 0; k < 16; k++) Dk[k] = D[k];
#pragma unroll 1
    for (int itr = 1; itr < TW_FWD; itr++) {
      PL_SYNC();                                            // (also keeps the reloads below inside the loop)
      {
        double P[16];
        for (int k = 0; k < 16; k++) P[k] = shift_up1(Dinv[k]);
        for (int k = 0; k < 4; k++) {
          LDm[k] = a.ce * P[k];
          LDm[4 + k] = a.pc * P[k] + a.pe * P[4 + k] + a.pt * P[12 + k];
          LDm[8 + k] = a.s * P[8 + k];
          LDm[12 + k] = a.Tc * P[k] + a.Te * P[4 + k] + a.Ts * P[8 + k] + a.Tt * P[12 + k];
        }
      }
      const double* D = Dk;
      for (int rr = 0; rr < 4; rr++) {
        const double a0 = LDm[rr * 4], a1 = LDm[rr * 4 + 1], a2 = LDm[rr * 4 + 2], a3 = LDm[rr * 4 + 3];
        Dn[rr * 4 + 0] = ((D[rr * 4 + 0] - a0 * b.ce) - a1 * b.pc) - a3 * b.Tc;
        Dn[rr * 4 + 1] = (D[rr * 4 + 1] - a1 * b.pe) - a3 * b.Te;
        Dn[rr * 4 + 2] = (D[rr * 4 + 2] - a2 * b.s) - a3 * b.Ts;
        Dn[rr * 4 + 3] = (D[rr * 4 + 3] - a1 * b.pt) - a3 * b.Tt;
      }
      if (nd == 2 || nd == NE - 3) {
        for (int rr = 0; rr < 4; rr++) {
          const double t = LDm[rr * 4] * fv[0] + LDm[rr * 4 + 1] * fv[1] + LDm[rr * 4 + 2] * fv[2] + LDm[rr * 4 + 3] * fv[3];
          Dn[rr * 4 + 0] += t * fw[0]; Dn[rr * 4 + 1] += t * fw[1]; Dn[rr * 4 + 2] += t * fw[2];
        }
      }
      inv4(Dn, Dinv);
      if (itr == 1 && !alg_only) {                          // LD_1 (UD_28) is final: its T column modifies U_1 (L_28) as seen by node 2 (27)
        for (int k = 0; k < 4; k++) {
          const double c1 = lane_bcast(LDm[4 * k + 3], tw_lane(1)), c28 = lane_bcast(LDm[4 * k + 3], tw_lane(NE - 2));
          if (nd == 2) fv[k] = c1;
          if (nd == NE - 3) fv[k] = c28;
        }
        if (nd == 2) for (int cc = 0; cc < 3; cc++) fw[cc] = TP.TX2[0][cc];
        if (nd == NE - 3) for (int cc = 0; cc < 3; cc++) fw[cc] = TP.TX2[3][cc];
      }
      if ((itr == FAR_T || itr == FAR_B) && !alg_only) {    // the factor of node N_p - 3 (N_p + N_s + 2) is final: fold node N_p - 1's (N_p + N_s's) entry on it into its L (U)
        double q7[4] = {0.0, 0.0, 0.0, 0.0}, q22[4] = {0.0, 0.0, 0.0, 0.0};
        for (int cc = 0; cc < 3; cc++) for (int k = 0; k < 4; k++) {
          const double d7 = lane_bcast(Dinv[cc * 4 + k], tw_lane(NP - 3)), d22 = lane_bcast(Dinv[cc * 4 + k], tw_lane(NP + NS + 2));
          q7[k] += TP.TX2[1][cc] * d7; q22[k] += TP.TX2[2][cc] * d22;
        }
        if ((nd == NP - 1 && itr == FAR_T) || (nd == NP + NS && itr == FAR_B)) {
          const double* q = nd == NP - 1 ? q7 : q22;
          const OffBlk e = nd == NP - 1 ? upper_blk(S, NP - 3, alg_only) : lower_blk(S, NP + NS + 2, alg_only);
          a.Tc -= q[0] * e.ce + q[1] * e.pc + q[3] * e.Tc;
          a.Te -= q[1] * e.pe + q[3] * e.Te;
          a.Ts -= q[2] * e.s + q[3] * e.Ts;
          a.Tt -= q[1] * e.pt + q[3] * e.Tt;
          for (int k = 0; k < 4; k++) TP.qfar[nd == NP - 1 ? 0 : 1][k] = q[k];
        }
      }
    }
    {   // closing node TW_MID: D'' = D'_mid - L_mid D'^-1_{mid-1} U_{mid-1}
      double P[16], L2[16], Dm[16], Dmi[16];
      for (int k = 0; k < 16; k++) P[k] = lane_bcast(Dinv[k], TW_MID - 1);
      const OffBlk c = lower_blk(S, TW_MID, alg_only), e = upper_blk(S, TW_MID - 1, alg_only);
      for (int k = 0; k < 4; k++) {
        L2[k] = c.ce * P[k];
        L2[4 + k] = c.pc * P[k] + c.pe * P[4 + k] + c.pt * P[12 + k];
        L2[8 + k] = c.s * P[8 + k];
        L2[12 + k] = c.Tc * P[k] + c.Te * P[4 + k] + c.Ts * P[8 + k] + c.Tt * P[12 + k];
      }
      for (int rr = 0; rr < 4; rr++) {
        const double a0 = L2[rr * 4], a1 = L2[rr * 4 + 1], a2 = L2[rr * 4 + 2], a3 = L2[rr * 4 + 3];
        Dm[rr * 4 + 0] = ((Dn[rr * 4 + 0] - a0 * e.ce) - a1 * e.pc) - a3 * e.Tc;
        Dm[rr * 4 + 1] = (Dn[rr * 4 + 1] - a1 * e.pe) - a3 * e.Te;
        Dm[rr * 4 + 2] = (Dn[rr * 4 + 2] - a2 * e.s) - a3 * e.Ts;
        Dm[rr * 4 + 3] = (Dn[rr * 4 + 3] - a1 * e.pt) - a3 * e.Tt;
      }
      inv4(Dm, Dmi);
      if (nd == TW_MID) for (int k = 0; k < 16; k++) { Dinv[k] = Dmi[k]; S.LDmid[k] = PL_F32(L2[k]); }
    }
    if (act) for (int k = 0; k < 16; k++) { S.Dinv[k][i] = PL_F32(Dinv[k]); S.LD[k][i] = PL_F32(LDm[k]); }
  }
  PL_TOCD(S, 4);
  // 6a. control row over the node unknowns (computed in the lane = node layout: the twin needs neighbour shifts), stored in TP.vB
  double dI = 0.0;
  {
    const int ln = lane < NE ? lane : NE - 1;
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    if (mode == PLH_MODE_I) dI = 1.0;
    else if (mode == PLH_MODE_V) { if (lane == 0) v2 = 1.0; if (lane == NE - 1) v2 = -1.0; }
    else if (mode == PLH_MODE_P) { if (lane == 0) v2 = S.ctrlJ[0]; if (lane == NE - 1) v2 = -S.ctrlJ[0]; dI = S.ctrlJ[1]; }
    else if (mode == PLH_MODE_ETA_P) { if (lane == NP + NS) { v2 = 1.0; v1 = -1.0; } }
    else if (mode == PLH_MODE_DT) {
      if (lane < NE) v3 = -cj * TP.wT5[tsec_of(NA + ln)];
      // collector part: sum_k vc_k dT_k with dT = zb - zc dT_end - zI xI
      double vc = 0.0, vi = 0.0;
      if (lane == 0) for (int k = 0; k < NA; k++) { vc += TP.wT5[tsec_of(k)] * TP.zc[0][k]; vi += TP.wT5[tsec_of(k)] * TP.zI[0][k]; }
      if (lane == NE - 1) for (int k = 0; k < NZ; k++) { vc += TP.wT5[tsec_of(NA + NE + k)] * TP.zc[1][k]; vi += TP.wT5[tsec_of(NA + NE + k)] * TP.zI[1][k]; }
      v3 -= -cj * vc;                                       // -(-cj w) zc
      dI = wave_sum((lane == 0 || lane == NE - 1) ? cj * vi : 0.0);   // -(-cj w) zI
    } else {                                                // PL_MODE_DT_TWIN: -(sum_i w_i d rhs_T,i / d y_alg)
      // T row i depends on Phi_e, Phi_s of nodes i-1, i, i+1 (TeL/D/U, TsL/D/U), on j_i (TJ) and, for the four one-sided stencils, on a second neighbour
      const double wi = lane < NE ? TP.wT5[tsec_of(NA + ln)] : 0.0;
      const double eU = shift_up1(lane < NE ? wi * TP.TeU[ln] : 0.0), eL = shift_down1(lane < NE ? wi * TP.TeL[ln] : 0.0);
      const double sU = shift_up1(lane < NE ? wi * TP.TsU[ln] : 0.0), sL = shift_down1(lane < NE ? wi * TP.TsL[ln] : 0.0);
      if (lane < NE) {
        v1 = -(eU + wi * TP.TeD[ln] + (lane < NE - 1 ? eL : 0.0));
        v2 = -(sU + wi * TP.TsD[ln] + (lane < NE - 1 ? sL : 0.0));
        for (int k = 0; k < 4; k++) if (lane == wb_src_node(k)) { const double wr = TP.wT5[tsec_of(NA + wb_row_node(k))]; v1 -= wr * TP.TX2[k][1]; v2 -= wr * TP.TX2[k][2]; }
        const int sc = sec_of(ln);
        if (sc != 1) {                                     // j eliminated: v_x -= v_j phi
          const int jx = sc == 0 ? ln : ln - NS;
          const double vj = -wi * TP.TJ[jx];
          v1 -= vj * (S.gpe[jx] * S.dj[jx]); v2 -= vj * (S.gps[jx] * S.dj[jx]);
          v0 = vj;                                         // kept for the right-hand side (b_I -= v_j beta); slot 0 is free in the algebraic system
        }
      }
      // the collector rows depend on I only (Joule heat): -(sum_k w_k) d rhs_k/dI
      dI = -(NA * TP.wT5[tsec_of(0)] * TP.qIJ[0] + NZ * TP.wT5[tsec_of(NT - 1)] * TP.qIJ[1]);
    }
    if (lane < NE) { TP.vB[0][ln] = v0; TP.vB[1][ln] = v1; TP.vB[2][ln] = v2; TP.vB[3][ln] = v3; }
  }
  PL_SYNC();
  PL_SYNC();
  // 5. border: x2 = B^-1 (column of I) and the pivot d - v.x2
  if (mode != PLH_MODE_I) {
    double ra[1][4] = {{0.0, 0.0, 0.0, 0.0}};
    if (nd == 0) { ra[0][2] = TP.cI4[0]; ra[0][3] = TP.cI4[1]; }
    if (nd == NE - 1) { ra[0][2] = TP.cI4[2]; ra[0][3] = TP.cI4[3]; }
    thermal_sweeps<1>(S, alg_only, ra);
    if (act) for (int cc = 0; cc < 4; cc++) TP.x2[cc][i] = ra[0][cc];
    const double vx = wave_sum(act ? TP.vB[1][i] * ra[0][1] + TP.vB[2][i] * ra[0][2] + TP.vB[3][i] * ra[0][3] + (mode == PL_MODE_DT_TWIN ? 0.0 : TP.vB[0][i] * ra[0][0]) : 0.0);
    if (lane == 0) { TP.bord[0] = dI - vx; TP.bord[1] = dI; }
  } else if (lane == 0) { TP.bord[0] = 1.0; TP.bord[1] = 1.0; }
  PL_SYNC();
  PL_TOCD(S, 5);
}

template <class M>
PL_DEV void thermal_solve(CellLDS<M>& S, LaneRegs& R, const Tables* __restrict__ tb, double* b, int mode, bool alg_only) {
  PL_MODEL(M);
  constexpr int O_T = M::O_T;
  const int lane = lane_id();
  const CellConst& c = S.cc;
  auto& TP = S.th;
  const int r = lane % NR, g = lane < CS_LANES ? lane / NR : CS_G - 1;
  PL_TICE(2);
  // a. particle partial solutions w = A_p^-1 b_cs  (two mat-vecs through the spectral form, the diagonal from R.rcp); collector forward/backward substitution
  double zbk = 0.0;                                        // lane 32 + k: chain solution of collector node k (T^-1 b_T restricted to the chain)
  if (!alg_only) {
    auto collectors = [&]() {
      // collector chains: lane 32 + k owns collector node k (aluminium 0 .. N_a - 1, then copper), the mapping of the residual rows and of phase e.  Both Thomas recurrences
      // run as systolic DPP chains (every lane re-evaluates its stage until its predecessor is final, as in the block sweeps): ONE LDS load per operand and lane instead of
      // the 4 N_a loads per lane of a single-lane recurrence -- the LDS array, shared by the four cells of the CU, is what the particle phases wait for -- and straight-line
      // code the scheduler interleaves with the mat-vecs.  The chain heads have a zero multiplier, so the copper chain ignores the lane below it.
      const int ck = lane - 32;
      const bool cact = ck >= 0 && ck < NA + NZ;
      const int kq = cact ? ck : 0, q = kq < NA ? 0 : 1, kk = q == 0 ? kq : kq - NA, ic = q == 0 ? kk : NA + NE + kk, nq = q == 0 ? NA : NZ;
      const double l_bv = b[O_T + ic], l_aL = TP.aL[ic], l_aU = TP.aU[ic], l_cp = TP.cP[q][kk];
      const double cp_prev = shift_up1(l_cp);
      const double cm = (cact && kk > 0) ? l_aL * cp_prev : 0.0, up = (cact && kk < nq - 1) ? l_aU : 0.0, bv = cact ? l_bv : 0.0, cpk = cact ? l_cp : 0.0;
      constexpr int NC = NA > NZ ? NA : NZ;
      double f = bv;
#pragma unroll
      for (int st = 1; st < NC; st++) { const double fp = shift_up1(f); f = bv - cm * fp; }
      double x = f * cpk;
#pragma unroll
      for (int st = 1; st < NC; st++) { const double xn = shift_down1(x); x = (f - up * xn) * cpk; }
      zbk = x;
    };
    if constexpr (PL_THROWB) {
      double yvr[CSD_PASS];                                 // W b / d per pass, handed from the first mat-vec to the second in registers
        // r06, ROW layout: y = diag(1 / (kappa lam - cj)) W b with b through row_newbcast; the second mat-vec (below, behind the collector chains) takes y from the lanes of
        // the row as well -- r05 wrote y to the c_s section of S.yy and read it back behind a phase separator
        const int q = lane >> 4, rr = lane & 15, rc = rr < NR ? rr : NR - 1;
        double Wr_[NR];
        for (int k = 0; k < NR; k++) Wr_[k] = S.Mr[S.OFF_WR + rc * NR + k];
        [[maybe_unused]] double WrN[NR_EQ ? 1 : NR];
        if constexpr (!NR_EQ) for (int k = 0; k < NR; k++) WrN[k] = S.Mr[S.mr_el(1) + S.OFF_WR + rc * NR + k];
        int pp[CSD_PASS]; double bc[CSD_PASS];
  #pragma unroll
        for (int pass = 0; pass < CSD_PASS; pass++) {
          const int p0 = pass * CSD_G + q; pp[pass] = p0 < NJ ? p0 : NJ - 1; yvr[pass] = 0.0;
          const int rk = rc < nr_of(pp[pass]) ? rc : nr_of(pp[pass]) - 1;
          bc[pass] = b[O_CS + cs_off(pp[pass]) + rk];
        }
        csd_settle(bc);
        static_for<0, NR>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
  #pragma unroll
          for (int pass = 0; pass < CSD_PASS; pass++) {
            if constexpr (NR_EQ) rowb_fmac<k>(yvr[pass], bc[pass], Wr_[k]);
            else rowb_fmac<k>(yvr[pass], bc[pass], pp[pass] < NP ? Wr_[k] : WrN[k]);
          }
        });
  #pragma unroll
        for (int pass = 0; pass < CSD_PASS; pass++) yvr[pass] *= R.rcp[pass];
      collectors();
      // second mat-vec: w = V y with y from the lanes of the row (registers: no LDS round trip, no phase separator)
      {
        const int q = lane >> 4, rr = lane & 15, rc = rr < NR ? rr : NR - 1;
        double Vr_[NR];
        for (int m = 0; m < NR; m++) Vr_[m] = S.Mr[S.OFF_VR + rc * NR + m];
        [[maybe_unused]] double VrN[NR_EQ ? 1 : NR];
        if constexpr (!NR_EQ) for (int m = 0; m < NR; m++) VrN[m] = S.Mr[S.mr_el(1) + S.OFF_VR + rc * NR + m];
        double w[CSD_PASS];
#pragma unroll
        for (int pass = 0; pass < CSD_PASS; pass++) w[pass] = 0.0;
        csd_settle(yvr);
        static_for<0, NR>([&](auto mc) {
          constexpr int m = decltype(mc)::value;
#pragma unroll
          for (int pass = 0; pass < CSD_PASS; pass++) {
            const int p0 = pass * CSD_G + q, pc = p0 < NJ ? p0 : NJ - 1;
            if constexpr (NR_EQ) rowb_fmac<m>(w[pass], yvr[pass], Vr_[m]);
            else rowb_fmac<m>(w[pass], yvr[pass], pc < NP ? Vr_[m] : VrN[m]);
          }
        });
#pragma unroll
        for (int pass = 0; pass < CSD_PASS; pass++) {
          const int p0 = pass * CSD_G + q;
          if (p0 < NJ && rr == nr_of(p0 < NJ ? p0 : NJ - 1) - 1) S.w9[p0] = w[pass];
          R.wreg[pass] = w[pass];
        }
      }
    } else {
    double Wrow[NR], Vrow[NR];
    for (int k = 0; k < NR; k++) { Wrow[k] = S.Mr[S.OFF_WR + r * NR + k]; Vrow[k] = S.Mr[S.OFF_VR + r * NR + k]; }
    // (one address register per array for the lane's particle group; pass and column go into the offset fields.  The last pass may reach beyond the last particle: its
    //  lanes read the last particle instead -- same values as a clamped index, selected per lane on the ADDRESS)
    constexpr int LASTP = (CS_PASS - 1) * CS_G;             // first particle of the last pass
    const bool over = LASTP + g >= NJ;
    constexpr bool A16 = NR % 2 == 0 && O_CS % 2 == 0;      // particle rows start on 16-byte boundaries (b is one of the 16-byte aligned vectors of CellLDS)
    const lds_cptr bg = PL_LDS_BASE_A(A16, (const double*)b + O_CS + g * NR), bl = PL_LDS_BASE_A(A16, (const double*)b + O_CS + (over ? NJ - 1 : LASTP + g) * NR);
    // (N_r_p != N_r_n: per-electrode rows of W and V, zero-padded; the particles sit at cs_off(p) in b and in S.yy, plain indexing instead of the affine address tricks)
    [[maybe_unused]] double WrowN[NR_EQ ? 1 : NR], VrowN[NR_EQ ? 1 : NR];
    if constexpr (!NR_EQ) for (int k = 0; k < NR; k++) { WrowN[k] = S.Mr[S.mr_el(1) + S.OFF_WR + r * NR + k]; VrowN[k] = S.Mr[S.mr_el(1) + S.OFF_VR + r * NR + k]; }
    double yv[CS_PASS];
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      double y = 0.0;
      if constexpr (NR_EQ) {
        const lds_cptr bb = pass == CS_PASS - 1 ? bl : bg + pass * CS_G * NR;
#pragma unroll
        for (int k = 0; k < NR; k++) y += Wrow[k] * bb[k];
      } else {
        const int p0 = pass * CS_G + g, pc = p0 < NJ ? p0 : NJ - 1;
        for (int k = 0; k < NR; k++) y += (pc < NP ? Wrow[k] : WrowN[k]) * b[O_CS + cs_off(pc) + k];
      }
      yv[pass] = y * R.rcp[pass];
    }
    const lds_ptr yg = PL_LDS_BASE(S.yy + O_CS + g * NR + r);
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      if constexpr (NR_EQ) { if (lane < CS_LANES && pass * CS_G + g < NJ) yg[pass * CS_G * NR] = yv[pass]; }   // S.yy is dead between a residual and the next form_iterate
      else { const int p0 = pass * CS_G + g; if (lane < CS_LANES && p0 < NJ && r < nr_of(p0)) S.yy[O_CS + cs_off(p0) + r] = yv[pass]; }
    }
    collectors();
    PL_SYNC();
    const lds_cptr yr = PL_LDS_BASE_A(A16, (const double*)S.yy + O_CS + g * NR), yl = PL_LDS_BASE_A(A16, (const double*)S.yy + O_CS + (over ? NJ - 1 : LASTP + g) * NR);
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p0 = pass * CS_G + g;
      double w = 0.0;
      if constexpr (NR_EQ) {
        const lds_cptr yb = pass == CS_PASS - 1 ? yl : yr + pass * CS_G * NR;
#pragma unroll
        for (int m = 0; m < NR; m++) w += Vrow[m] * yb[m];
      } else {
        const int pc = p0 < NJ ? p0 : NJ - 1;
        for (int m = 0; m < NR; m++) w += (pc < NP ? Vrow[m] : VrowN[m]) * S.yy[O_CS + cs_off(pc) + m];
      }
      if (lane < CS_LANES && p0 < NJ && r == nr_of(p0) - 1) S.w9[p0] = w;
      R.wreg[pass] = w;
    }
    }
  }
  PL_SYNC();
  PL_TOCE(S, 2, 0);
  // b. node right-hand sides
  // (all LDS operands of phases b, d, e are loaded unconditionally up front -- clamped indices for the lanes / nodes that do not use them -- and selected afterwards: loads under
  //  the nested `if`s (node lane? electrode node? current mode?) were dependent exec-masked round trips, see iso_solve)
  const int nd = tw_node(lane);                   // twisted lane layout of thermal_sweeps
  const bool act = nd >= 0;
  const int i = act ? nd : 0;
  const int sc = sec_of(i);
  const bool elec = act && sc != 1;
  const int jx = sc == 0 ? i : (sc == 2 ? i - NS : 0);
  const double l_ce = b[O_CE + i], l_pe = b[O_PE + i], l_T = b[O_T + NA + i], l_j = b[O_J + jx], l_ps = b[O_PS + jx], l_bI = b[O_I];
  const double l_w9 = S.w9[jx], l_gcs = S.gcs[jx], l_Tcs = TP.Tcs[jx], l_dj = S.dj[jx];
  const double l_ceJ = S.ceJ[i], l_tq1 = S.peJ[i], l_tq2 = S.psJ[jx], l_tq3 = TP.tq3[i], l_tq0 = alg_only ? 0.0 : l_ceJ;
  const double l_cI0 = i == 0 ? TP.cI4[0] : (i == NE - 1 ? TP.cI4[2] : 0.0), l_cI1 = i == 0 ? TP.cI4[1] : (i == NE - 1 ? TP.cI4[3] : 0.0);
  const double l_zbA = lane_bcast(zbk, 32 + NA - 1), l_zbZ = lane_bcast(zbk, 32 + NA), l_aL = TP.aL[NA], l_aU = TP.aU[NA + NE - 1];
  const double l_gce = S.gce[jx], l_gpe = S.gpe[jx], l_gps = S.gps[jx], l_ph3 = TP.phi3[i];
  const double l_ph0 = alg_only ? 0.0 : l_gce * l_dj, l_ph1 = l_gpe * l_dj, l_ph2 = l_gps * l_dj;
  double beta = 0.0;
  double y[4] = {0.0, 0.0, 0.0, 0.0};
  if (act) {
    y[0] = alg_only ? 0.0 : l_ce; y[1] = l_pe; y[2] = 0.0; y[3] = alg_only ? 0.0 : l_T;
    if (elec) {
      const double w9 = alg_only ? 0.0 : l_w9;
      const double bjp = l_j - l_gcs * w9;
      y[2] = l_ps;
      if (!alg_only) y[3] -= l_Tcs * w9;
      beta = bjp * l_dj;
      y[0] -= l_tq0 * beta; y[1] -= l_tq1 * beta; y[2] -= l_tq2 * beta; y[3] -= l_tq3 * beta;
    }
    if (!alg_only) {
      if (i == 0) y[3] -= l_aL * l_zbA;
      if (i == NE - 1) y[3] -= l_aU * l_zbZ;
    }
  }
  double xI = 0.0;
  if (mode == PLH_MODE_I) {
    xI = l_bI;
    if (act) { y[2] -= l_cI0 * xI; y[3] -= l_cI1 * xI; }
  }
  PL_TOCE(S, 2, 1);
  // c. block-Thomas sweeps + Woodbury correction
  {
    double ra[1][4] = {{y[0], y[1], y[2], y[3]}};
    thermal_sweeps<1>(S, alg_only, ra);
    for (int cc = 0; cc < 4; cc++) y[cc] = ra[0][cc];
  }
  PL_TOCE(S, 2, 2);
  // d. border
  if (mode != PLH_MODE_I) {
    const double v0 = TP.vB[0][i], v1 = TP.vB[1][i], v2 = TP.vB[2][i], v3 = TP.vB[3][i];
    const double x20 = TP.x2[0][i], x21 = TP.x2[1][i], x22 = TP.x2[2][i], x23 = TP.x2[3][i], bd = TP.bord[0];
    double vy = 0.0;
    if (act) {
      vy = v1 * y[1] + v2 * y[2] + v3 * y[3];
      if (mode == PL_MODE_DT_TWIN) vy += v0 * beta;           // v_j (b_j'/d): the eliminated j part of the twin row
      else vy += v0 * y[0];
    }
    if (mode == PLH_MODE_DT) {                                          // collector T's in the control row: -cj w_k zb_k
      const int k = lane >= 32 && lane < 32 + NA + NZ ? lane - 32 : 0, q = k < NA ? 0 : 1, kk = k < NA ? k : k - NA;
      const double wk = TP.wT5[tsec_of(q == 0 ? kk : NA + NE + kk)], cjf = TP.cjf;
      if (lane >= 32 && lane < 32 + NA + NZ) vy += -cjf * wk * zbk;
    }
    const double vsum = wave_sum(vy);
    xI = pl_div(l_bI - vsum, bd);
    if (act) { y[0] -= xI * x20; y[1] -= xI * x21; y[2] -= xI * x22; y[3] -= xI * x23; }
  }
  PL_SYNC();
  // e. write node unknowns, back-substitute j and the collectors
  if (act) {
    if (!alg_only) { b[O_CE + i] = y[0]; b[O_T + NA + i] = y[3]; }
    b[O_PE + i] = y[1];
    if (elec) {
      b[O_PS + jx] = y[2];
      b[O_J + jx] = beta - (l_ph0 * y[0] + l_ph1 * y[1] + l_ph2 * y[2] + l_ph3 * y[3]);
    }
  }
  if (lane == 0) b[O_I] = xI;
  if (!alg_only) {
    const double T0n = lane_bcast(y[3], tw_lane(0)), T29n = lane_bcast(y[3], tw_lane(NE - 1));
    const int k = lane >= 32 && lane < 32 + NA + NZ ? lane - 32 : 0, q = k < NA ? 0 : 1, kk = k < NA ? k : k - NA;
    const double zck = TP.zc[q][kk], zIk = TP.zI[q][kk];
    if (lane >= 32 && lane < 32 + NA + NZ) b[O_T + (q == 0 ? kk : NA + NE + kk)] = zbk - zck * (q == 0 ? T0n : T29n) - zIk * xI;
  }
  PL_SYNC();
  PL_TOCE(S, 2, 3);
  // f. particles: dc = w - A^-1 e_last bj dj - A^-1 q dT
  if constexpr (PL_THROWB) {
    if (!alg_only) {       // ROW layout
      const int q = lane >> 4, rr = lane & 15, rc = rr < NR ? rr : NR - 1;
      double ae[CSD_PASS], aq[CSD_PASS], dj[CSD_PASS], dT[CSD_PASS];
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) {
        const int p0 = pass * CSD_G + q, p = p0 < NJ ? p0 : NJ - 1, ndp = p < NP ? p : p + NS;
        ae[pass] = TP.AinvE[p][rc]; aq[pass] = TP.AinvQ[p][rc]; dj[pass] = b[O_J + p]; dT[pass] = b[O_T + NA + ndp];
      }
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) {
        const int p0 = pass * CSD_G + q, p = p0 < NJ ? p0 : NJ - 1;
        const double bj = p < NP ? c.bj_p : c.bj_n;
        const double v = R.wreg[pass] - ae[pass] * bj * dj[pass] - aq[pass] * dT[pass];
        if (p0 < NJ && rr < nr_of(p)) b[O_CS + cs_off(p) + rr] = v;
      }
    }
  } else
  if (!alg_only) {
    // (unconditional clamped loads first, guarded stores last -- see iso_solve)
    double ae[CS_PASS], aq[CS_PASS], dj[CS_PASS], dT[CS_PASS];
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p0 = pass * CS_G + g, p = p0 < NJ ? p0 : NJ - 1, nd = p < NP ? p : p + NS;
      ae[pass] = TP.AinvE[p][r]; aq[pass] = TP.AinvQ[p][r]; dj[pass] = b[O_J + p]; dT[pass] = b[O_T + NA + nd];
    }
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p = pass * CS_G + g;
      const double bj = p < NP ? c.bj_p : c.bj_n;
      const double v = R.wreg[pass] - ae[pass] * bj * dj[pass] - aq[pass] * dT[pass];
      if (lane < CS_LANES && p < NJ && r < nr_of(p)) b[O_CS + cs_off(p) + r] = v;
    }
  }
  PL_SYNC();
  PL_TOCE(S, 2, 4);
}

}  // namespace pl

namespace pl {

// Jacobian entry types that exist only with temperature (decode word as in iso_jac_entry: type<<24 | a<<16 | b<<8 | c)
enum JTT { TT_CS_T = 64, TT_J_T, TT_PE_TL, TT_PE_TD, TT_PE_TU, TT_T_TL, TT_T_TD, TT_T_TU, TT_T_CL, TT_T_CD, TT_T_CU, TT_T_EL, TT_T_ED, TT_T_EU,
           TT_T_SL, TT_T_SD, TT_T_SU, TT_T_J, TT_T_CS, TT_T_X2, TT_T_I, TT_CTRL_T };

template <bool FROZEN, class M>
PL_DEV double thermal_jac_entry(const CellLDS<M>& S, const Tables* __restrict__ tb, unsigned w, double cj) {
  PL_MODEL(M);
  const int t = w >> 24, a = (w >> 16) & 255, bb = (w >> 8) & 255, cc = w & 255;
  const auto& TP = S.th;
  switch (t) {
    case JT_CS_CS: return (FROZEN ? TP.kapF[a] : TP.kapP[a]) * tb->Mp(a < NP ? 0 : 1)[bb * NR + cc] - (bb == cc ? cj : 0.0);   // (kapP follows every residual pass, kapF is the factored one)
    case TT_CS_T: {                                      // d(kappa_p(T) (M c)_r)/dT
      double acc = 0.0;
      if (FROZEN) {                                      // q = kappa' M c of the last factorisation, rebuilt from A^-1 q: q = (kappa M - cj I) (A^-1 q)
        for (int k = 0; k < nr_of(a); k++) acc += (TP.kapF[a] * tb->Mp(a < NP ? 0 : 1)[bb * NR + k] - (bb == k ? TP.cjf : 0.0)) * TP.AinvQ[a][k];
        return acc;
      }
      for (int k = 0; k < nr_of(a); k++) acc += tb->Mp(a < NP ? 0 : 1)[bb * NR + k] * S.yy[O_CS + cs_off(a) + k];      // evaluated at the state in S.yy (plh_jacobian)
      return TP.dkapP[a] * acc;
    }
    case TT_J_T: return TP.gT[a];
    case TT_PE_TL: return TP.ptL[a];
    case TT_PE_TD: return TP.ptD[a];
    case TT_PE_TU: return TP.ptU[a];
    case TT_T_TL: return TP.aL[a];
    case TT_T_TD: return thermal_aD(TP, a) - cj + ((a >= NA && a < NA + NE) ? TP.TtD[a - NA] : 0.0);
    case TT_T_TU: return TP.aU[a];
    case TT_T_CL: return TP.TcL[a];
    case TT_T_CD: return TP.TcD[a];
    case TT_T_CU: return TP.TcU[a];
    case TT_T_EL: return TP.TeL[a];
    case TT_T_ED: return TP.TeD[a];
    case TT_T_EU: return TP.TeU[a];
    case TT_T_SL: return TP.TsL[a];
    case TT_T_SD: return TP.TsD[a];
    case TT_T_SU: return TP.TsU[a];
    case TT_T_J: return TP.TJ[a];
    case TT_T_CS: return TP.Tcs[a];
    case TT_T_X2: return TP.TX2[a][bb];
    case TT_T_I: return TP.qIJ[a];
    case TT_CTRL_T: return -cj * TP.wT5[tsec_of(a)];
  }
  return iso_jac_entry(S, tb, w, cj);
}

}  // namespace pl

// petlion_kernels.h -- the kernels of one model variant and its VariantOps table (included by variant_tu.hip once per variant; the test-only
// wave-emulator build includes it once for all variants).
//
// One workgroup = one 64-lane wavefront = one cell.  Grid = n_cells workgroups; ~38 KB of LDS per workgroup, so four cells are
// resident per CU (one per SIMD) and 1024 cells fill the 256 CUs of an MI355X in a single wave of workgroups.
#pragma once
#include "dfn_integrate.h"
#include "plh_host.h"

#ifndef PL_WAVE_EMU
#define PL_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif

namespace pl {

// Registers.  The 301-state kernels run one wavefront per SIMD (LDS: >= 37 kB per cell), so the compiler may use the whole 512-entry register file of a lane.  What it makes of
// that -- VGPRs, AGPRs, spills, scratch bytes per lane and LDS per workgroup of EVERY instantiation of every variant -- is read out of the code objects after each build
// (tools/kernel_resources.py -> petlion.jl_amd/libpetlion_hip.so.resources.json) and committed for the validated binary in profiles/validated_build.json
// ("kernel_resources"); tests/test_build_records.py fails when a plain benchmark kernel of the isothermal / SEI models uses scratch.  No figures are typed in here: r05's
// said "0 B/lane" for the thermal kernel while the shipped object had 28.  (rocprofv3's `accum_vgpr_count` reads 0 on this unified-file part and is not the figure to read.)
#if !defined(PL_WAVE_EMU) && !defined(PL_NO_WAVES_ATTR) && defined(PL_WAVES_PER_EU)
// -DPL_WAVES_PER_EU=n: experiment builds (tools/experiments/occupancy.py)
#define PL_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(PL_WAVES_PER_EU, PL_WAVES_PER_EU)))
#elif !defined(PL_WAVE_EMU) && !defined(PL_NO_WAVES_ATTR)      /* (PL_NO_WAVES_ATTR: experiment builds of tools/experiments/build_modes.py) */
// One cell per SIMD (one wave with 512 registers; M::W2: its two waves with 256 each) for every model whose cell fills a quarter of the CU's LDS -- the 301-state models.  A cell
// of <= 26 624 B leaves room for at least six on a CU: those kernels are compiled for TWO cells per SIMD (256 registers per lane).  Measured r05 (tools/experiments/occupancy.py):
// quadratic particles (20.7 kB, 7 cells per CU) +42 %, polynomial +39 %, the (2, 2, 2, 10) grid (12.2 kB, 8 per CU) +63 %; three per SIMD +25 % only (168 registers).  The
// 301-state models reach 26.2 kB only with the history in global memory (-DPL_OCC2, ModelT::PHI_GLOBAL) and lose 5 % there: they stay at one.
#define PL_CELLS_PER_SIMD(M) ((sizeof(CellLDS<M>) <= (M::PHI_GLOBAL ? 32768 : 26624) && !M::W2) ? 2 : 1)      /* (the PL_OCC2 experiment layouts: five cells per CU count too) */
#define PL_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(M::NWAVES * PL_CELLS_PER_SIMD(M), M::NWAVES * PL_CELLS_PER_SIMD(M))))
#else
#define PL_ONE_WAVE_PER_SIMD
#endif

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
template <class M> __device__ __forceinline__ void load_vec(double* dst, const double* __restrict__ src) {
  const int lane = lane_id(), wv = wave_id();
  _Pragma("unroll") for (int k__ = 0; k__ < M::NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wv); vokg<M>(k__, lane, wv)) dst[n] = src[n];
}
template <class M> __device__ __forceinline__ void store_vec(double* __restrict__ dst, const double* src) {
  const int lane = lane_id(), wv = wave_id();
  _Pragma("unroll") for (int k__ = 0; k__ < M::NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wv); vokg<M>(k__, lane, wv)) dst[n] = src[n];
}

template <class M> __global__ __launch_bounds__(64 * M::NWAVES) PL_ONE_WAVE_PER_SIMD void k_initial_guess(const Tables* tb, int n_cells, const double* theta, const double* SOC, double* Y) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  cell_initial_guess(S, S.yy, SOC[cell]);
  store_vec<M>(Y + (size_t)cell * NST, S.yy);
}

// F[cell][nrows] = rows row0 .. row0+nrows-1 of the residual (the whole vector, or the f_diff! / f_alg! slices of seam 1)
template <class M> __global__ __launch_bounds__(64 * M::NWAVES) PL_ONE_WAVE_PER_SIMD void k_residual(const Tables* tb, int n_cells, const double* theta, const double* Y, const double* YP,
                                                 int mode, double value, double* F, int row0, int nrows) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  load_vec<M>(S.yy, Y + (size_t)cell * NST); load_vec<M>(S.yp, YP + (size_t)cell * NST);
  PL_XSYNC();
  cell_residual(S, R, S.yy, S.yp, S.delta, mode, value);
  PL_XSYNC();
  for (int n = (int)threadIdx.x; n < nrows; n += WAVE * M::NWAVES) F[(size_t)cell * nrows + n] = S.delta[row0 + n];
}

template <class M> __global__ __launch_bounds__(64 * M::NWAVES) PL_ONE_WAVE_PER_SIMD void k_jacobian(const Tables* tb, int n_cells, const double* theta, const double* Y, const double* YP,
                                                 double cj, int mode, double* nz, const int* sel, int nsel) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  load_vec<M>(S.yy, Y + (size_t)cell * NST); load_vec<M>(S.yp, YP + (size_t)cell * NST);
  PL_XSYNC();
  cell_res_jac(S, R, S.yy, S.yp, S.delta, mode, 0.0);
  PL_XSYNC();
  const int nnz = sel ? nsel : tb->nnz[mode];             // sel: positions (in the CSC order of the mode) of the entries to export, e.g. the J_y_alg! block
  const unsigned* code = tb->csc_code[mode];
  double* out = nz + (size_t)cell * nnz;
  for (int k = (int)threadIdx.x; k < nnz; k += WAVE * M::NWAVES) out[k] = jac_entry(S, tb, code[sel ? sel[k] : k], cj);
}

template <class M> __global__ __launch_bounds__(64 * M::NWAVES) PL_ONE_WAVE_PER_SIMD void k_linear_solve(const Tables* tb, int n_cells, const double* theta, const double* Y, const double* YP,
                                                     double cj, int mode, double* b, int nref) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  load_vec<M>(S.yy, Y + (size_t)cell * NST); load_vec<M>(S.yp, YP + (size_t)cell * NST); load_vec<M>(S.delta, b + (size_t)cell * NST);
  PL_XSYNC();
  cell_res_jac(S, R, S.yy, S.yp, S.phi[1], mode, 0.0);
  cell_factor(S, R, tb, cj, mode, false);
  if (nref > 0) cell_solve_refined(S, R, tb, S.delta, S.phi[0], cj, mode, false, nref);
  else cell_solve(S, R, S.delta, mode, false);
  store_vec<M>(b + (size_t)cell * NST, S.delta);
}

template <class M> __global__ __launch_bounds__(64 * M::NWAVES) PL_ONE_WAVE_PER_SIMD void k_init_consistent(const Tables* tb, int n_cells, const double* theta, int mode, double value,
                                                        double reltol_init, double* Y, double* YP, int* status, int* iters, int nref) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  load_vec<M>(S.yy, Y + (size_t)cell * NST);
  PL_XSYNC();
  Counters cnt; for (int k = 0; k < 10; k++) cnt.v[k] = 0;
  PL_SYNC();
  const int rc = cell_init_consistent<GF_REFINE>(S, R, tb, S.yy, S.yp, S.delta, S.phi[1], mode, value, reltol_init, cnt, S.phi[0], nref);
  store_vec<M>(Y + (size_t)cell * NST, S.yy); store_vec<M>(YP + (size_t)cell * NST, S.yp);
  PL_SYNC();
  if (threadIdx.x == 0) { if (status) status[cell] = rc; if (iters) iters[cell] = cnt.v[C_INIT]; }
}


template <class M, int F> __global__ __launch_bounds__(64 * M::NWAVES) PL_ONE_WAVE_PER_SIMD void k_integrate(IntegrateArgs a) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= a.n_cells) return;
  cell_setup(S, R, a.tb, a.theta + (size_t)cell * a.tb->P);
  if constexpr ((F & GF_EXPR) != 0) { if (threadIdx.x == 0) S.theta_row = a.theta + (size_t)cell * a.tb->P; }     // closure inputs read theta again (general instantiation only)
  Counters cnt; for (int k = 0; k < 10; k++) cnt.v[k] = 0;
#ifdef PL_PHASE_TIMERS
  if (threadIdx.x < 8) S.cyc[threadIdx.x] = 0;
#endif
  PL_XSYNC();
  PL_TIC(); PL_TIC_TOTAL();
  CellOut co;
  const size_t off = (size_t)cell * a.out.max_pts;
  co.max_pts = a.out.max_pts;
  co.t = a.out.t ? a.out.t + off : nullptr; co.V = a.out.V ? a.out.V + off : nullptr; co.I = a.out.I ? a.out.I + off : nullptr;
  co.SOC = a.out.SOC ? a.out.SOC + off : nullptr; co.T = a.out.T_avg ? a.out.T_avg + off : nullptr;
  co.Yall = a.out.Y_all ? a.out.Y_all + off * NST : nullptr;
  cell_simulate<F>(S, R, a.tb, a.SOC0[cell], a.Y_init ? a.Y_init + (size_t)cell * NST : nullptr, a.t_init ? a.t_init[cell] : 0.0, a.n_runs, a.runs, a.opts, co, a.out.n_pts ? a.out.n_pts + cell : nullptr,
                a.out.run_info + (size_t)cell * a.n_runs, cnt,
                a.out.Y_final ? a.out.Y_final + (size_t)cell * NST : nullptr, a.out.YP_final ? a.out.YP_final + (size_t)cell * NST : nullptr,
                a.scratch + (size_t)cell * 2 * NST, a.scratch + (size_t)cell * 2 * NST + NST, cell, a.genW ? a.genW + (size_t)cell * NST : nullptr, a.sens, a.theta + (size_t)cell * a.tb->P,
                (M::PHI_GLOBAL && a.phig) ? a.phig + (size_t)cell * (MAXORD + 1 - M::PHI_LDS) * M::NPADG : nullptr, a.Y_init && !a.t_init);
  PL_TOC_TOTAL(S);
  PL_SYNC();
  if (threadIdx.x == 0 && a.out.counters) {
    plh_counters* c = a.out.counters + cell;
#ifdef PL_PHASE_TIMERS
    for (int k = 0; k < 8; k++) c->cyc[k] = S.cyc[k];
#else
    for (int k = 0; k < 8; k++) c->cyc[k] = 0;
#endif
    c->n_steps = cnt.v[C_STEPS]; c->n_res = cnt.v[C_RES]; c->n_jac = cnt.v[C_JAC]; c->n_fact = cnt.v[C_FACT]; c->n_solve = cnt.v[C_SOLVE];
    c->n_newton = cnt.v[C_NEWTON]; c->n_errfail = cnt.v[C_ERRFAIL]; c->n_convfail = cnt.v[C_CONVFAIL]; c->sum_kp2 = cnt.v[C_SUMKP2]; c->n_init_iters = cnt.v[C_INIT];
  }
}

// decode word of the structural Jacobian entry (r, c), 0 if structurally zero
template <class M>
unsigned classify(const Tables& tb, int mode, int r, int c) {
  PL_MODEL(M);
  auto W = [](int t, int a, int b, int cc) { return (unsigned)((t << 24) | (a << 16) | (b << 8) | cc); };
  auto node_of_j = [](int jx) { return jx < NP ? jx : jx + NS; };
  if (r == O_I) {
    if (mode == PLH_MODE_I) return c == O_I ? W(JT_CTRL_P1, 0, 0, 0) : 0;
    if (mode == PLH_MODE_V) return c == O_PS ? W(JT_CTRL_P1, 0, 0, 0) : (c == O_PS + NJ - 1 ? W(JT_CTRL_M1, 0, 0, 0) : 0);
    if (M::THERMAL && mode == PLH_MODE_DT) return (c >= M::O_T && c < M::O_T + NT) ? W(TT_CTRL_T, c - M::O_T, 0, 0) : 0;
    if (mode == PLH_MODE_P) return c == O_PS ? W(JT_CTRL_PA, 0, 0, 0) : (c == O_PS + NJ - 1 ? W(JT_CTRL_PB, 0, 0, 0) : (c == O_I ? W(JT_CTRL_PI, 0, 0, 0) : 0));
    if (mode == PLH_MODE_ETA_P) return c == O_PE + NP + NS ? W(JT_CTRL_M1, 0, 0, 0) : (c == O_PS + NP ? W(JT_CTRL_P1, 0, 0, 0) : 0);
    return 0;
  }
  if constexpr (M::THERMAL) {                       // entries that exist only with temperature; everything else falls through
    constexpr int O_T = M::O_T;
    const bool cT = c >= O_T && c < O_T + NT;
    const int ct = c - O_T;                         // T node of the column
    if (r >= O_CS && r < N_CECS && cT) { const int p = cs_particle(r - O_CS); return ct == NA + node_of_j(p) ? W(TT_CS_T, p, r - O_CS - cs_off(p), 0) : 0; }
    if (r >= O_J && r < O_PE && cT) { const int jx = r - O_J; return ct == NA + node_of_j(jx) ? W(TT_J_T, jx, 0, 0) : 0; }
    if (r >= O_PE && r < O_PS && cT) {
      const int i = r - O_PE, k = ct - NA;
      if (i == NE - 1) return 0;
      if (k == i - 1 && i > 0) return W(TT_PE_TL, i, 0, 0);
      if (k == i) return W(TT_PE_TD, i, 0, 0);
      if (k == i + 1) return W(TT_PE_TU, i, 0, 0);
      return 0;
    }
    if (r >= O_T && r < O_T + NT) {                 // T row (residuals_T!)
      const int it = r - O_T;
      if (cT) { if (ct == it - 1) return W(TT_T_TL, it, 0, 0); if (ct == it) return W(TT_T_TD, it, 0, 0); if (ct == it + 1) return W(TT_T_TU, it, 0, 0); return 0; }
      if (it < NA || it >= NA + NE) return c == O_I ? W(TT_T_I, it < NA ? 0 : 1, 0, 0) : 0;
      const int i = it - NA, sc = sec_of(i);
      const bool first = (i == 0) || (i == NP + NS), last = (i == NP - 1) || (i == NE - 1);
      const int far = (i == 0 || i == NP + NS) ? i + 2 : i - 2;      // second neighbour of the one-sided stencils
      const int xk = i == 0 ? 0 : (i == NP - 1 ? 1 : (i == NP + NS ? 2 : 3));
      if (c < O_CS) {                               // c_e columns
        if (c == i - 1 && i > 0) return W(TT_T_CL, i, 0, 0);
        if (c == i) return W(TT_T_CD, i, 0, 0);
        if (c == i + 1 && i < NE - 1) return W(TT_T_CU, i, 0, 0);
        if ((i == 0 || i == NE - 1) && c == far) return W(TT_T_X2, xk, 0, 0);
        return 0;
      }
      if (c >= O_PE && c < O_PS) {                  // Phi_e columns (the diagonal only through the reaction heat: electrodes, and the one-sided ends)
        const int k = c - O_PE;
        if (k == i - 1 && i > 0) return W(TT_T_EL, i, 0, 0);
        if (k == i && sc != 1) return W(TT_T_ED, i, 0, 0);
        if (k == i + 1 && i < NE - 1) return W(TT_T_EU, i, 0, 0);
        if ((i == 0 || i == NE - 1) && k == far) return W(TT_T_X2, xk, 1, 0);
        return 0;
      }
      if (sc == 1) return 0;
      const int jx = sc == 0 ? i : i - NS;
      if (c >= O_PS && c < O_PS + NJ) {
        const int k = c - O_PS, kf = (i == 0 || i == NP + NS) ? jx + 2 : jx - 2;
        if (k == jx - 1 && !first) return W(TT_T_SL, i, 0, 0);
        if (k == jx) return W(TT_T_SD, i, 0, 0);
        if (k == jx + 1 && !last) return W(TT_T_SU, i, 0, 0);
        if ((first || last) && k == kf) return W(TT_T_X2, xk, 2, 0);
        return 0;
      }
      if (c == O_J + jx) return W(TT_T_J, jx, 0, 0);
      if (c == O_CS + cs_surf(jx)) return W(TT_T_CS, jx, 0, 0);
      return 0;
    }
    if (cT) return 0;
  }
  if (r < O_CS) {                                   // c_e row i
    const int i = r, sc = sec_of(i);
    if (c < O_CS) { if (c == i - 1) return W(JT_CE_L, i, 0, 0); if (c == i) return W(JT_CE_D, i, 0, 0); if (c == i + 1) return W(JT_CE_U, i, 0, 0); return 0; }
    if (sc != 1 && c == O_J + (sc == 0 ? i : i - NS)) return W(JT_CE_J, i, 0, 0);
    if (M::SEI && sc == 2 && c == O_JS + (i - NP - NS)) return W(JT_CE_JS, i, 0, 0);
    return 0;
  }
  if constexpr (M::SD != 0) {                       // quadratic / polynomial particles: c_avg row (and Q row) per electrode node
    if (r >= O_CS && r < N_CECS) { const int p = r - O_CS; if (c == r) return W(JT_CSA_D, p, 0, 0); if (c == O_J + p) return W(JT_CS_J, p, 0, 0); return 0; }
    if (M::SD == 2 && r >= O_Q && r < O_Q + NJ) { const int p = r - O_Q; if (c == r) return W(JT_Q_Q, p, 0, 0); if (c == O_J + p) return W(JT_Q_J, p, 0, 0); return 0; }
  }
  if (r < N_CECS) {                                 // c_s row (p, rr)
    const int p = cs_particle(r - O_CS), rr = r - O_CS - cs_off(p);
    if (c >= O_CS && c < N_CECS && cs_particle(c - O_CS) == p) { const int cc = c - O_CS - cs_off(p); return (tb.Mp(p < NP ? 0 : 1)[rr * NR + cc] != 0.0 || rr == cc) ? W(JT_CS_CS, p, rr, cc) : 0; }
    if (rr == nr_of(p) - 1 && c == O_J + p) return W(JT_CS_J, p, 0, 0);
    return 0;
  }
  if (M::SEI && r < O_J) {                          // film rows (residuals_film!) and the SOH row (residuals_SOH!)
    if (r < O_SOH) { const int k = r - O_FILM; if (c == r) return W(JT_F_F, k, 0, 0); if (c == O_JS + k) return W(JT_F_JS, k, 0, 0); return 0; }
    if (c == r) return W(JT_SOH_SOH, 0, 0, 0);
    if (c >= O_JS && c < O_JS + NN) return W(JT_SOH_JS, c - O_JS, 0, 0);
    return 0;
  }
  if (r < O_PE) {                                   // j row
    const int jx = r - O_J, nd = node_of_j(jx);
    if (M::SEI && jx >= NP && c == O_FILM + jx - NP) return W(JT_J_F, jx - NP, 0, 0);
    if (c == O_CE + nd) return W(JT_J_CE, jx, 0, 0);
    if (c == (M::SD == 0 ? O_CS + cs_surf(jx) : O_CS + jx)) return W(JT_J_CS, jx, 0, 0);
    if (M::SD == 2 && c == O_Q + jx) return W(JT_J_Q, jx, 0, 0);
    if (c == r) return W(JT_J_J, jx, 0, 0);
    if (c == O_PE + nd) return W(JT_J_PE, jx, 0, 0);
    if (c == O_PS + jx) return W(JT_J_PS, jx, 0, 0);
    return 0;
  }
  if (r < O_PS) {                                   // Phi_e row i
    const int i = r - O_PE, sc = sec_of(i);
    if (i == NE - 1) return c == r ? W(JT_PE_D, i, 0, 0) : 0;
    if (c < O_CS) { if (c == i - 1) return W(JT_PE_CL, i, 0, 0); if (c == i) return W(JT_PE_CD, i, 0, 0); if (c == i + 1) return W(JT_PE_CU, i, 0, 0); return 0; }
    if (c >= O_PE && c < O_PS) { const int k = c - O_PE; if (k == i - 1) return W(JT_PE_L, i, 0, 0); if (k == i) return W(JT_PE_D, i, 0, 0); if (k == i + 1) return W(JT_PE_U, i, 0, 0); return 0; }
    if (sc != 1 && c == O_J + (sc == 0 ? i : i - NS)) return W(JT_PE_J, i, 0, 0);
    if (M::SEI && sc == 2 && c == O_JS + (i - NP - NS)) return W(JT_PE_JS, i, 0, 0);
    return 0;
  }
  if (M::SEI && r >= O_JS) {                        // j_s row k (residuals_j_s!)
    const int k = r - O_JS, jx = NP + k, nd = NP + NS + k;
    if (c == O_PS + jx) return W(JT_JS_PS, k, 0, 0);
    if (c == O_PE + nd) return W(JT_JS_PE, k, 0, 0);
    if (c == O_J + jx) return W(JT_JS_J, k, 0, 0);
    if (c == r) return W(JT_JS_JS, k, 0, 0);
    if (c == O_FILM + k) return W(JT_JS_F, k, 0, 0);
    if (c == O_I) return W(JT_JS_I, k, 0, 0);
    return 0;
  }
  {                                                 // Phi_s row jx
    const int jx = r - O_PS;
    const bool first = (jx == 0) || (jx == NP), last = (jx == NP - 1) || (jx == NJ - 1);
    if (M::SEI && jx >= NP && c == O_JS + jx - NP) return W(JT_PS_JS, jx, 0, 0);
    if (c >= O_PS && c < O_PS + NJ) { const int k = c - O_PS; if (k == jx - 1 && !first) return W(JT_PS_L, jx, 0, 0); if (k == jx) return W(JT_PS_D, jx, (first || last) ? 1 : 0, 0); if (k == jx + 1 && !last) return W(JT_PS_U, jx, 0, 0); return 0; }
    if (c == O_J + jx) return W(JT_PS_J, jx, 0, 0);
    if (c == O_I && jx == 0) return W(JT_PS_I, 0, 0, 0);
    if (c == O_I && jx == NJ - 1) return W(JT_PS_I, 1, 0, 0);
    return 0;
  }
}

template <class M>
int sections_of(SectionInfo* o) {
  int k = 0;
  o[k++] = {"c_e", O_CE, NE}; o[k++] = {"c_s_avg", O_CS, M::NCS};
  if (M::THERMAL) o[k++] = {"T", M::O_T, NT};
  if (M::SEI) { o[k++] = {"film", M::O_FILM, NN}; o[k++] = {"SOH", M::O_SOH, 1}; }
  if (M::SD == 2) o[k++] = {"Q", M::O_Q, NJ};
  o[k++] = {"j", M::O_J, NJ}; o[k++] = {"Φ_e", M::O_PE, NE}; o[k++] = {"Φ_s", M::O_PS, NJ};
  if (M::SEI) o[k++] = {"j_s", M::O_JS, NN};
  o[k++] = {"I", M::O_I, 1};
  return k;
}

// ---------------------------------------------------------------------------------------------------------------------
// the variant's operations table
// ---------------------------------------------------------------------------------------------------------------------
template <class M> struct OpsOf {
  static void initial_guess(hipStream_t st, const Tables* tb, int n, const double* theta, const double* SOC, double* Y) {
    PL_LAUNCH(k_initial_guess<M>, n, WAVE * M::NWAVES, st, tb, n, theta, SOC, Y);
  }
  static void residual(hipStream_t st, const Tables* tb, int n, const double* theta, const double* Y, const double* YP, int mode, double value, double* F, int row0, int nrows) {
    PL_LAUNCH(k_residual<M>, n, WAVE * M::NWAVES, st, tb, n, theta, Y, YP, mode, value, F, row0, nrows);
  }
  static void jacobian(hipStream_t st, const Tables* tb, int n, const double* theta, const double* Y, const double* YP, double cj, int mode, double* nz, const int* sel, int nsel) {
    PL_LAUNCH(k_jacobian<M>, n, WAVE * M::NWAVES, st, tb, n, theta, Y, YP, cj, mode, nz, sel, nsel);
  }
  static void linear_solve(hipStream_t st, const Tables* tb, int n, const double* theta, const double* Y, const double* YP, double cj, int mode, double* b, int nref) {
    PL_LAUNCH(k_linear_solve<M>, n, WAVE * M::NWAVES, st, tb, n, theta, Y, YP, cj, mode, b, nref);
  }
  static void init_consistent(hipStream_t st, const Tables* tb, int n, const double* theta, int mode, double value, double reltol_init, double* Y, double* YP, int* status,
                              int* iters, int nref) {
    PL_LAUNCH(k_init_consistent<M>, n, WAVE * M::NWAVES, st, tb, n, theta, mode, value, reltol_init, Y, YP, status, iters, nref);
  }
  static void integrate(hipStream_t st, const IntegrateArgs& a, int features) {     // features: GenFlag bits the call needs; the smallest instantiation that has them all
#ifdef PL_CLOSURE_COMPILED      /* a closure library holds the two closure instantiations only (plh_integrate sends it nothing else) */
    if (features & GF_GENROW) PL_LAUNCH((k_integrate<M, GF_STOPS | GF_FUNC | GF_EXPR | GF_GENROW>), a.n_cells, WAVE * M::NWAVES, st, a);
    else if (features & GF_EXPR) PL_LAUNCH((k_integrate<M, GF_STOPS | GF_FUNC | GF_EXPR>), a.n_cells, WAVE * M::NWAVES, st, a);
#else
    if (features & GF_SENS) { if constexpr (!M::W2) PL_LAUNCH((k_integrate<M, GF_STOPS | GF_SENS>), a.n_cells, WAVE * M::NWAVES, st, a); }
    else if (features & GF_REFINE) PL_LAUNCH((k_integrate<M, GF_STOPS | GF_FUNC | GF_EXPR | GF_GENROW | GF_REFINE>), a.n_cells, WAVE * M::NWAVES, st, a);
    else if (features & GF_GENROW) PL_LAUNCH((k_integrate<M, GF_STOPS | GF_FUNC | GF_EXPR | GF_GENROW>), a.n_cells, WAVE * M::NWAVES, st, a);
    else if (features & GF_EXPR) PL_LAUNCH((k_integrate<M, GF_STOPS | GF_FUNC | GF_EXPR>), a.n_cells, WAVE * M::NWAVES, st, a);
    else if (features & GF_FUNC) PL_LAUNCH((k_integrate<M, GF_STOPS | GF_FUNC>), a.n_cells, WAVE * M::NWAVES, st, a);
    else if (features & GF_STOPS) PL_LAUNCH((k_integrate<M, GF_STOPS>), a.n_cells, WAVE * M::NWAVES, st, a);
    else PL_LAUNCH((k_integrate<M, 0>), a.n_cells, WAVE * M::NWAVES, st, a);
#endif
  }
  static const VariantOps* table(int id) {
#if !defined(PL_PHASE_TIMERS) && !defined(PL_WAVE_EMU)
    // four cells per CU (160 kB of LDS, one wave per SIMD) is what every built-in kernel is tuned for: one byte over 40 960 drops the CU to three cells (-25 %)
    static_assert(!GRID_DEFAULT || sizeof(CellLDS<M>) <= 40960, "built-in variant: LDS per cell above 40 960 B, only three cells per CU would be resident");
#endif
    static const VariantOps ops = {id, M::CHEM, M::SEI ? 1 : 0, M::THERMAL ? 1 : 0, M::PREC, M::SD, M::TF, M::RXN, M::W2 ? 1 : 0, M::NST, M::NDIFF, {NP, NS, NN, NRP, NA, NZ, NRN},
                                   {PL_RADIAL_M, PL_RADIAL_M_N}, {PL_RADIAL_LAM, PL_RADIAL_LAM_N}, {PL_RADIAL_V, PL_RADIAL_V_N}, {PL_RADIAL_W, PL_RADIAL_W_N}, {PL_RADIAL_BJ_FACTOR, PL_RADIAL_BJ_FACTOR_N}, sizeof(CellLDS<M>), (int)(sizeof(CellLDS<M>) / sizeof(double)) + M::NWAVES * WAVE * 2 * LR_PASS, M::PHI_GLOBAL ? (MAXORD + 1 - M::PHI_LDS) * M::NPADG : 0, &classify<M>, &sections_of<M>,
                                   &initial_guess, &residual, &jacobian, &linear_solve, &init_consistent, &integrate};
    return &ops;
  }
};

}  // namespace pl

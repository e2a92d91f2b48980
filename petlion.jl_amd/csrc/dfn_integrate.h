// dfn_integrate.h -- device-side time stepping for one cell per wavefront.
//
//  * consistent initialisation  = newtons_method!                      (reference src/model_evaluation.jl:430-480)
//  * variable-order (1..5) variable-step BDF with modified Newton      = what the reference gets from SUNDIALS IDA through
//    Sundials.jl's step! (reference src/model_evaluation.jl:259-287, 312-333); algorithm restated from the IDA documentation
//    (fixed-leading-coefficient BDF, DASSL lineage): IDASetCoeffs / IDANls / IDATestError / IDAHandleNFlag /
//    IDACompleteStep / IDAGetSolution and the ONE_STEP_TSTOP stop tests.  SUNDIALS' documented default constants are used.
//  * run logic: tstops, SOC trapezoid, stop conditions with direction guards, back-interpolation of the last point
//                                                                       (reference src/checks.jl:1-249, src/model_evaluation.jl:174-232, 288-382,
//                                                                        src/physics_equations/scalar_residual.jl:103-111)
// All scalars of the integrator are wave-uniform (every lane holds the same value); vectors live in LDS and are processed
// lane-strided (element n -> lane n % 64).
#pragma once
#include "dfn_cell.h"

namespace pl {

// most trips of a lane through a state vector over the variants of this grid (thermal: NT more states; SEI: 2 NN + 1 more; polynomial: NJ more); 6 on the default grid
constexpr int NTRIP_MAX = (2 * NE + NJ * NR + 2 * NJ + 1 + (NT > 2 * NN + 1 ? NT : 2 * NN + 1) + WAVE - 1) / WAVE;
struct IdaScalars {   // the coefficient arrays psi/alpha/beta/sigma/gamma live in CellLDS (S.ida_*)
  double tn, hh, hused, cj, cjlast, cjold, cjratio, ss, rr, h0_forced, rtol, atol;
  double ew[NTRIP_MAX];      // per-lane error weights of this step
  double ph[4][NTRIP_MAX];
  double ee[NTRIP_MAX];      // per-lane accumulated Newton correction of the step
  double pa[NTRIP_MAX], pb[NTRIP_MAX];   // predictor y_n(0), y'_n(0) of the step (models with M::PRED_REGS)   // per-lane entries of the BDF history vectors that are not kept in LDS (models with M::PHI_LDS < 6): [j - PHI_LDS][trip]
  int kk, kused, knew, phase, ns, maxord;
  int nst;
  double* phg;               // models with M::PHI_GLOBAL: this cell's block of BDF history orders 2 .. 5 in global memory, [4][NPAD], entry n of a vector in lane n % 64
};

// Feature flags of the integrate kernel's instantiations.  The step loop keeps ~60 per-lane values live next to the cell's LDS block, and code that is merely PRESENT in it costs
// registers and instruction cache whether it runs or not -- so each call pays only for what it asks for (plh_integrate picks the instantiation):
//   0                      constant / hold / rest inputs, per-step scalars out                                   (the benchmark path)
//   GF_STOPS               + opts.tstops / tdiscon stop times, the per-step state dump (outputs = :all), yp_alg_zero
//   GF_STOPS | GF_FUNC     + inputs that are functions of time given as tables (run_function), check_reinitialization!
//   ... | GF_EXPR          + closure inputs of (t, Y, YP, theta): the postfix interpreter inside every residual evaluation
//   ... | GF_REFINE        + iterative refinement of every linear solve (plh_opts.refine, the parity diagnostic)
//   ... | GF_GENROW        + a closure of the state with derivative programs: the general control row (GenRow, dfn_cell.h); separate from GF_EXPR because its
//                           presence alone costs the plain-closure kernel 9 % (registers around the Newton loop)
//   GF_STOPS | GF_SENS     + forward parameter sensitivities dY/dtheta_k next to the states (dfn_sens.h; constant-input protocols)
enum GenFlag { GF_STOPS = 1, GF_FUNC = 2, GF_EXPR = 4, GF_REFINE = 8, GF_GENROW = 16, GF_SENS = 32 };

// device counters: wave-uniform registers (every call site uses a compile-time index), written out once at the end; indices:
enum Cnt { C_STEPS, C_RES, C_JAC, C_FACT, C_SOLVE, C_NEWTON, C_ERRFAIL, C_CONVFAIL, C_SUMKP2, C_INIT };
struct Counters { int v[10]; };
__device__ __forceinline__ void cnt_add(Counters& c, int k, int v = 1) { c.v[k] += v; }

// lane-strided sweep over the N state entries: exactly 5 trips (301 = 4*64 + 45), fully unrolled so that the LDS loads of all
// trips are issued back to back (one latency instead of five); only the last trip is predicated.
// BDF history access inside a PL_VEC loop (k__ = compile-time trip index): vectors j < M::PHI_LDS are LDS arrays, the rest registers in I.ph
#define PHI_RD(j, n) (M::PHI_GLOBAL ? ((j) < M::PHI_LDS ? S.phi[(j) < M::PHI_LDS ? (j) : 0][n] : I.phg[((j) - M::PHI_LDS) * M::NPADG + (n)]) : PHI_RD_R(j, n))
#define PHI_WR(j, n, v) do { if constexpr (M::PHI_GLOBAL) { if ((j) < M::PHI_LDS) S.phi[(j) < M::PHI_LDS ? (j) : 0][n] = (v); else I.phg[((j) - M::PHI_LDS) * M::NPADG + (n)] = (v); } else PHI_WR_R(j, n, v); } while (0)
#define PHI_RD_R(j, n) ((M::PHI_LDS > MAXORD || (j) < M::PHI_LDS) ? S.phi[(M::PHI_LDS > MAXORD || (j) < M::PHI_LDS) ? (j) : 0][n] : ((j) == M::PHI_LDS ? I.ph[0][k__] : ((j) == M::PHI_LDS + 1 ? I.ph[1][k__] : ((j) == M::PHI_LDS + 2 ? I.ph[2][k__] : I.ph[3][k__]))))
#define PHI_WR_R(j, n, v) do { if (M::PHI_LDS > MAXORD || (j) < M::PHI_LDS) S.phi[(M::PHI_LDS > MAXORD || (j) < M::PHI_LDS) ? (j) : 0][n] = (v); \
                             else if ((j) == M::PHI_LDS) I.ph[0][k__] = (v); else if ((j) == M::PHI_LDS + 1) I.ph[1][k__] = (v); \
                             else if ((j) == M::PHI_LDS + 2) I.ph[2][k__] = (v); else I.ph[3][k__] = (v); } while (0)
// accumulated correction ee inside a PL_VEC loop
#define EE(n) I.ee[k__]
// some BDF history orders live in registers (thermal model): the step-control passes then index the history with compile-time orders under wave-uniform branches
template <class M> constexpr bool PHI_REGS = M::PHI_LDS <= MAXORD && !M::PHI_GLOBAL;      // (history in global memory: a runtime order is just an address, as with the whole history in LDS)
#ifdef PL_EXP_BRANCHY_PHI
constexpr bool PL_BRANCHY_PHI = true;
#else
constexpr bool PL_BRANCHY_PHI = false;
#endif
// r06: the step-control passes over the BDF history of the models that keep the WHOLE history in LDS are written without control flow around the LDS accesses: orders 1 .. 3 are
// always processed with wave-uniform 0 / 1 factors (p * 1.0, fma(0.0, p, a), fma(1.0, p, a) are exact: the sums are those of the branching form bit for bit -- the device of the
// register-resident orders of the thermal model, r03), orders 4 and 5 under ONE wave-uniform branch, every load of a pass is issued before its arithmetic and the stores follow
// at the end.  r05's form had a branch per order (and, inside the error test, per trip): each arm waited for its own LDS round trip, and with one wavefront per SIMD nothing
// hides that -- the phase timers charged 2.2 k cycles per Newton iteration to the iterate / norm passes and 3.6 k per step to IDACompleteStep.  Cells are zero-initialised
// (cell_setup) so that an order that has never been written is finite.  -DPL_NO_FLAT: the r05 form (A/B builds).
#ifdef PL_NO_FLAT
template <class M> constexpr bool PL_FLAT = false;
#else
template <class M> constexpr bool PL_FLAT = M::PHI_LDS > MAXORD && !M::PHI_GLOBAL && !M::W2;
#endif
#define PL_VEC(n) _Pragma("unroll") for (int k__ = 0; k__ < NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wave_id()); vok<M>(k__, lane, wave_id()))
// the same for statements that touch a vector in GLOBAL memory (NST entries: no padding there): always masked
#define PL_VECG(n) _Pragma("unroll") for (int k__ = 0; k__ < NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wave_id()); vokg<M>(k__, lane, wave_id()))

template <class M>
__device__ __forceinline__ double wrms(const double* v, const double* w) {
  PL_MODEL(M);
  const int lane = lane_id();
  double s = 0.0;
  PL_VEC(n) { const double p = v[n] * w[n]; s += p * p; }
  return sqrt(wave_sum(s) * (1.0 / NST));      // (one-wave models only; unused by the integrator)
}

// value of a tabulated input at run-local time t (reference run_function: method(Y,p) - run.func(t,Y,YP,p), scalar_residual.jl:169-170);
// wave-uniform: every lane walks the (small) table in HBM through scalar loads
PL_DEV double tab_eval(const plh_run& r, double t) {
  const int n = r.n_tab; const double* tt = r.tab_t; const double* vv = r.tab_v;
  if (n <= 0) return 0.0;
  if (t < tt[0]) return vv[0];
  int lo = 0, hi = n;                                                  // last knot with t_k <= t (a repeated knot time is a jump, right-continuous):
  while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (tt[mid] <= t) lo = mid; else hi = mid; }   // bisection, O(log n) loads per evaluation (drive cycles)
  const int k = lo;
  if (k == n - 1) return vv[n - 1];
  const double dt = tt[k + 1] - tt[k];
  return dt > 0.0 ? vv[k] + (vv[k + 1] - vv[k]) * ((t - tt[k]) / dt) : vv[k + 1];
}
// value of a closure input (PLH_VAL_EXPR): a postfix program over t, Y, YP, theta (include/petlion_hip.h); wave-uniform, every lane runs it (scalar loads of the program,
// broadcast LDS reads of the states).  The value stack is an LDS array -- every lane stores the same value at the same address -- because a runtime-indexed private array
// would live in scratch memory, once per inlined copy of this function.
// A closure library (petlion.jl_amd/closure_lib.py) compiles this file with the programs of one protocol written out as C++ (PL_CLOSURE_HEADER): expr_eval is then a call of
// straight-line code -- `which` = -1 the closure itself, k its derivative program k -- and the interpreter below is not instantiated.
#ifdef PL_CLOSURE_HEADER
#include PL_CLOSURE_HEADER
#endif
// the interpreter: instructions [k0, k1) of the program (opcodes ops[], operands args[])
template <class M>
PL_DEV double prog_eval(CellLDS<M>& S, const double* __restrict__ ops, const double* __restrict__ args, int k0, int k1, double t, const double* Y, const double* YP) {
  double* st = S.xstk + PLH_EXPR_STACK * wave_id(); const double* th = S.theta_row; int sp = 0;
  for (int k = k0; k < k1; k++) {
    const int op = (int)ops[k]; const double a = args[k];
    if (op <= PLH_OP_THETA) {
      double v;
      switch (op) { case PLH_OP_CONST: v = a; break; case PLH_OP_T: v = t; break; case PLH_OP_Y: v = Y[(int)a]; break; case PLH_OP_YP: v = YP[(int)a]; break; default: v = th[(int)a]; }
      st[sp++] = v;
    } else if (op == PLH_OP_SELECT) { const double b = st[sp - 1], x = st[sp - 2], c = st[sp - 3]; sp -= 2; st[sp - 1] = c != 0.0 ? x : b; }
    else if (op == PLH_OP_NEG || op == PLH_OP_SIN || op == PLH_OP_COS || op == PLH_OP_EXP || op == PLH_OP_LOG || op == PLH_OP_SQRT || op == PLH_OP_ABS || op == PLH_OP_TANH) {
      const double x = st[sp - 1]; double v;
      switch (op) { case PLH_OP_NEG: v = -x; break; case PLH_OP_SIN: v = sin(x); break; case PLH_OP_COS: v = cos(x); break; case PLH_OP_EXP: v = exp(x); break;
                    case PLH_OP_LOG: v = log(x); break; case PLH_OP_SQRT: v = sqrt(x); break; case PLH_OP_ABS: v = fabs(x); break; default: v = tanh(x); }
      st[sp - 1] = v;
    } else {
      const double y = st[sp - 1], x = st[sp - 2]; sp--; double v;
      switch (op) { case PLH_OP_ADD: v = x + y; break; case PLH_OP_SUB: v = x - y; break; case PLH_OP_MUL: v = x * y; break; case PLH_OP_DIV: v = x / y; break;
                    case PLH_OP_POW: v = pow(x, y); break; case PLH_OP_MIN: v = x < y ? x : y; break; case PLH_OP_MAX: v = x > y ? x : y; break;
                    case PLH_OP_LT: v = x < y ? 1.0 : 0.0; break; case PLH_OP_LE: v = x <= y ? 1.0 : 0.0; break; case PLH_OP_GT: v = x > y ? 1.0 : 0.0; break; default: v = x >= y ? 1.0 : 0.0; }
      st[sp - 1] = v;
    }
  }
  return st[0];
}
template <class M>
PL_DEV double expr_eval(CellLDS<M>& S, const plh_run& r, double t, const double* Y, const double* YP, int k0 = 0, int k1 = -1, int which = -1) {      // instructions [k0, k1); default: the main program
#ifdef PL_CLOSURE_COMPILED
  (void)k0; (void)k1;
  return pl_closure_compiled(r.closure_id, which, t, Y, YP, S.theta_row);
#else
  (void)which;
  if (k1 < 0) k1 = r.n_tab;
  return prog_eval(S, r.tab_t, r.tab_v, k0, k1, t, Y, YP);
#endif
}
// value of the closure of a run as the control residual uses it: method(Y) - f for the input modes; for PLH_MODE_RES (method_res = 0; run_residual,
// scalar_residual.jl:172: res = theta[:_residual_val] - f) the row is -(f - x) with x = plh_run.value
template <class M>
PL_DEV double closure_input(CellLDS<M>& S, const plh_run& r, double t, const double* Y, const double* YP) {
  if (r.mode == PLH_MODE_DSTATE) return YP[r.dstate] - r.value;      // state_deriv_func(ind) as the residual x - YP[ind] (r.dstate = the index once the run has started; r.value = x)
  const double f = expr_eval(S, r, t, Y, YP);
  return r.mode == PLH_MODE_RES ? f - r.value : f;
}
// input of a run_function run at run-local time t with the iterate (Y, YP): table or closure
template <int F, class M>
PL_DEV double run_input(CellLDS<M>& S, const plh_run& r, double t, const double* Y, const double* YP) {
  if constexpr ((F & GF_EXPR) != 0) return r.value_kind == PLH_VAL_EXPR ? closure_input(S, r, t, Y, YP) : tab_eval(r, t);
  else return tab_eval(r, t);
}
// Jacobian refresh of a run with a general control row (GenRow, dfn_cell.h): factor in current mode, evaluate the row at (t, Y, YP) -- the input method's own entries
// (scalar_jacobian!, scalar_residual.jl:174-202) and minus the closure's derivative programs --, one solve for W = J_I^-1 e_I into the free LDS vector `tmp`, border pivot g.W.
// alg_only: the algebraic block of the consistent initialisation (columns >= NDIFF of the row, J_vec[N.diff+1:end], scalar_residual.jl:369-371).
template <class M>
PL_DEV void gen_factor(CellLDS<M>& S, LaneRegs& R, const Tables* tb, double cj, int mode, bool alg_only, GenRow& g, double t, const double* Y, const double* YP, double* tmp) {
  PL_MODEL(M);
  const int lane = lane_id();
  const plh_run& r = *g.run;
  double gv = 0.0; int gcol = 0, ng = 0;
  auto entry = [&](int c, double v) { if (!(alg_only && c < NDIFF)) { if (lane == ng) { gv = v; gcol = c; } ng++; } };      // (wave-uniform c, v; entry k lives in lane k)
  if (r.mode == PLH_MODE_DSTATE) {
    // x - YP[ind] (state_deriv_func, input_methods.jl:190-247): the integration row is -cj at column ind; the consistent-initialisation row has YP[ind] replaced by the
    // differential equation of that state (scalar_residual.jl:335-362), i.e. minus the entries of row ind of dF/dY in the algebraic columns -- walked through the
    // decode words of the exported pattern (current mode's: the rows other than the last do not depend on the mode)
    const int ind = r.dstate;
    if (!alg_only) entry(ind, -cj);
    else {
      const int* __restrict__ ptr = tb->csr_ptr[PLH_MODE_I]; const unsigned* __restrict__ code = tb->csr_code[PLH_MODE_I]; const unsigned short* __restrict__ ccol = tb->csr_col[PLH_MODE_I];
      for (int k = ptr[ind]; k < ptr[ind + 1]; k++) { const int c = ccol[k]; if (c >= NDIFF) entry(c, -jac_entry<false>(S, tb, code[k], 0.0)); }
    }
  } else {
    if (mode == PLH_MODE_I) entry(O_I, 1.0);                                                                 // scalar_jacobian! of the input method (scalar_residual.jl:174-202)
    else if (mode == PLH_MODE_V) { entry(O_PS, 1.0); entry(O_PS + NJ - 1, -1.0); }
    else if (mode == PLH_MODE_P) { const double iI = Y[O_I] * S.cc.I1C; entry(O_PS, iI); entry(O_PS + NJ - 1, -iI); entry(O_I, (Y[O_PS] - Y[O_PS + NJ - 1]) * S.cc.I1C); }
    else if (mode == PLH_MODE_ETA_P) { entry(O_PE + NP + NS, -1.0); entry(O_PS + NP, 1.0); }
    // (PLH_MODE_RES: method_res = 0, the row is the closure's alone)
    // Evaluation point of the derivative programs in the consistent initialisation: the closure itself is evaluated there with YP -> rhs(Y) = F_diff(Y, YP) + YP
    // (input_value, cell_init_consistent_impl), so its partials are too -- for a closure nonlinear in YP (or with Y / YP cross terms) the row evaluated at the init's raw
    // YP = 0 would be a different row (res = x - YP[i]^2 gives 2 YP[i] = 0: a zero border).  `tmp` is free until the W solve below.
    const double* YPe = YP;
    if (alg_only && r.n_dcol > 0 && r.dcol[r.n_dcol - 1] >= NST) {
      cell_residual(S, R, Y, YP, tmp, PLH_MODE_RES, 0.0);
      PL_XSYNC();
      PL_VEC(n) tmp[n] = n < NDIFF ? tmp[n] + YP[n] : YP[n];
      PL_XSYNC();
      YPe = tmp;
    }
    for (int k = 0; k < r.n_dcol; k++) {
      const int c = r.dcol[k];
      const double v = -expr_eval(S, r, t, Y, YPe, r.dofs[k], r.dofs[k + 1], k);
      if (c < NST) entry(c, v);                                                                               // - d f / d Y[c]
      else {                                                                                                    // - d f / d YP[i] of the differential state i = c - NST:
        const int i = c - NST;
        if (!alg_only) entry(i, cj * v);                                                                      //   times cj in the integration row;
        else {                                                                                                  //   YP[i] -> rhs_i(Y) in the consistent-initialisation row: chain through row i of dF/dY
          const int* __restrict__ ptr = tb->csr_ptr[PLH_MODE_I]; const unsigned* __restrict__ code = tb->csr_code[PLH_MODE_I]; const unsigned short* __restrict__ ccol = tb->csr_col[PLH_MODE_I];
          for (int q = ptr[i]; q < ptr[i + 1]; q++) { const int c2 = ccol[q]; if (c2 >= NDIFF) entry(c2, v * jac_entry<false>(S, tb, code[q], 0.0)); }
        }
      }
    }
  }
  g.gv = gv; g.gcol = gcol; g.ng = ng;
  cell_factor(S, R, tb, cj, PLH_MODE_I, alg_only);
  PL_XSYNC();
  PL_VEC(n) tmp[n] = n == O_I ? 1.0 : 0.0;
  PL_XSYNC();
  cell_solve(S, R, tmp, PLH_MODE_I, alg_only);
  PL_XSYNC();
  PL_VECG(n) g.W[n] = tmp[n];
  g.bord = g.dot(tmp, alg_only ? NDIFF : 0);
  PL_XSYNC();
}

// ---- consistent initialisation (newtons_method!) : Y (LDS, in/out), YP (LDS, out).  returns 0 / PLH_ERR_INIT ----
// Returns the number of Newton iterations (>= 1) or PLH_ERR_INIT.  cell_simulate has exactly ONE call site (the re-initialisation of a
// function input loops back to it): a second inlined copy costs 4-5 % of the step loop in instruction-cache misses, and a real call
// spills the ~100 live registers of the step loop.
// `frun` (general instantiation only): the run when its input is a closure of the state -- re-evaluated with the iterate before every residual evaluation, at run-local time t_fun
template <int F, class M>
PL_DEV int cell_init_consistent_impl(CellLDS<M>& S, const Tables* tb, double* Y, double* YP, double* res, double* Ytmp,
                                                      int mode, double value, double reltol_init, double* bsave = nullptr, int nref = 0, const plh_run* frun = nullptr, double t_fun = 0.0,
                                                      GenRow* g = nullptr) {
  LaneRegs R;                                                        // (the algebraic solves do not touch the particle registers)
  for (int k = 0; k < LR_PASS; k++) { R.wreg[k] = 0.0; R.rcp[k] = 0.0; }
  int iters = 0;
  PL_MODEL(M);
  const int lane = lane_id();
  PL_VEC(n) YP[n] = 0.0;
  PL_XSYNC();
  // the dT control row contains YP_T; the algebraic system uses its twin with YP_T -> rhs_T(Y) (scalar_residual.jl:347-372)
  if (M::THERMAL && mode == PLH_MODE_DT) mode = PL_MODE_DT_TWIN;
  // value of a closure input with the iterate; PLH_MODE_DSTATE (x - YP[ind]): YP[ind] is replaced by the differential equation of that state (the reference's
  // consistent-initialisation form of a residual that contains YP, scalar_residual.jl:335-362): rhs_ind(Yv) = F_ind(Yv, YP) + YP[ind]
  auto input_value = [&](const double* Yv) -> double {
    if constexpr ((F & GF_GENROW) != 0) {
      if (frun->mode == PLH_MODE_DSTATE) {
        cell_residual(S, R, Yv, YP, res, PLH_MODE_RES, 0.0);
        PL_XSYNC();
        const double v = res[frun->dstate] + YP[frun->dstate] - frun->value;
        PL_XSYNC();
        return v;
      }
    }
    if constexpr ((F & GF_GENROW) != 0) {
      if (frun->n_dcol > 0 && frun->dcol[frun->n_dcol - 1] >= NST) {      // the closure reads YP of differential states: evaluated with YP = rhs(Yv) = F_diff(Yv, YP) + YP
        cell_residual(S, R, Yv, YP, res, PLH_MODE_RES, 0.0);
        PL_XSYNC();
        PL_VEC(n) if (n < NDIFF) res[n] += YP[n];
        PL_XSYNC();
        const double v = closure_input(S, *frun, t_fun, Yv, res);
        PL_XSYNC();
        return v;
      }
    }
    return closure_input(S, *frun, t_fun, Yv, YP);
  };
  int ok = 0;
  for (int iter = 1; iter <= 100; iter++) {
    if constexpr ((F & GF_EXPR) != 0) { if (frun) value = input_value(Y); }
    cell_node_pass<true, true>(S, Y, YP, res, mode, value);      // R_alg + J_alg partials (differential rows ignored)
    PL_SYNC();
    bool gen = false;
    if constexpr ((F & GF_GENROW) != 0) gen = g && g->on();
    if constexpr ((F & GF_GENROW) != 0) { if (gen) gen_factor(S, R, tb, 0.0, mode, true, *g, t_fun, Y, YP, Ytmp); }        // (Ytmp is free until the Newton iteration is over)
    if (!gen) cell_factor(S, R, tb, 0.0, mode, true);
    if ((F & GF_REFINE) && nref > 0) cell_solve_refined(S, R, tb, res, bsave, 0.0, mode, true, nref, gen ? g : nullptr);     // (GF_REFINE instantiations only: plh_opts.refine)
    else if (gen) gen_solve(S, R, res, true, *g);
    else cell_solve(S, R, res, mode, true);
    iters++;
    double s = 0.0;
    if (!M::W2 || wave_id() == 0) for (int n = NDIFF + lane; n < NST; n += WAVE) { const double d = res[n]; Y[n] -= d; s += d * d; }      // (the algebraic rows are wave 0's)
    const double nrm = sqrt(block_sum<M>(S, s));
    PL_XSYNC();
    if (nrm < reltol_init) { ok = 1; break; }
    if (!(nrm == nrm)) break;
  }
  if (!ok) return PLH_ERR_INIT;
  // YP_diff = rhs_diff(Y)   (R_diff with YP = 0)
  if constexpr ((F & GF_EXPR) != 0) { if (frun) value = input_value(Y); }
  cell_residual(S, R, Y, YP, res, mode == PL_MODE_DT_TWIN ? PLH_MODE_DT : mode, value);
  PL_VEC(n) if (n < NDIFF) YP[n] = res[n];
  PL_SYNC();
  // finite-difference estimate of the algebraic derivatives (model_evaluation.jl:462-477)
  const double ce0 = S.cc.ce0;
  const double epsce = nextafter(ce0, 1e300) - ce0;
  double dt = sqrt(epsce);
  if (10.0 * reltol_init > dt) dt = 10.0 * reltol_init;
  PL_VEC(n) Ytmp[n] = Y[n] + dt * YP[n];
  PL_XSYNC();
  if constexpr ((F & GF_EXPR) != 0) { if (frun) value = input_value(Ytmp); }
  cell_node_pass<true, false>(S, Ytmp, YP, res, mode, value);
  PL_SYNC();
  bool gen = false;
  if constexpr ((F & GF_GENROW) != 0) gen = g && g->on();
  if ((F & GF_REFINE) && nref > 0) cell_solve_refined(S, R, tb, res, bsave, 0.0, mode, true, nref, gen ? g : nullptr);
  else if (gen) gen_solve(S, R, res, true, *g);
  else cell_solve(S, R, res, mode, true);
  if (!M::W2 || wave_id() == 0) for (int n = NDIFF + lane; n < NST; n += WAVE) YP[n] = -res[n] / dt;
  PL_XSYNC();
  return iters;
}
template <int F = 0, class M>
__device__ __forceinline__ int cell_init_consistent(CellLDS<M>& S, LaneRegs& R, const Tables* tb, double* Y, double* YP, double* res, double* Ytmp,
                                                    int mode, double value, double reltol_init, Counters& cnt, double* bsave = nullptr, int nref = 0,
                                                    const plh_run* frun = nullptr, double t_fun = 0.0, GenRow* g = nullptr) {
  (void)R;
  const int it = cell_init_consistent_impl<F>(S, tb, Y, YP, res, Ytmp, mode, value, reltol_init, bsave, nref, frun, t_fun, g);
  if (it < 0) { cnt_add(cnt, C_INIT, 100); return it; }
  cnt_add(cnt, C_RES, it + 2); cnt_add(cnt, C_JAC, it); cnt_add(cnt, C_FACT, it); cnt_add(cnt, C_SOLVE, it + 1); cnt_add(cnt, C_INIT, it);
  return 0;
}

// ---- IDA pieces ----
template <class M>
PL_DEV void ida_reinit(CellLDS<M>& S, IdaScalars& I, const double* y0, const double* yp0, int maxord, double t_start = 0.0) {
  PL_MODEL(M);
  const int lane = lane_id();
  I.tn = t_start; I.nst = 0; I.kk = 0; I.kused = 0; I.hused = 0.0; I.hh = 0.0; I.maxord = maxord;
  I.cjratio = 1.0; I.ss = 20.0; I.phase = 0; I.ns = 0; I.h0_forced = 0.0; I.cj = 0.0; I.cjlast = 0.0; I.cjold = 0.0; I.rr = 0.0; I.knew = 0;
  if (lane <= MAXORD && wave_id() == 0) { S.ida_psi[lane] = 0; S.ida_alpha[lane] = 0; S.ida_beta[lane] = 0; S.ida_sigma[lane] = 0; S.ida_gamma[lane] = 0; }
  PL_VEC(n) { S.phi[0][n] = y0[n]; S.phi[1][n] = yp0[n]; }
  for (int q = 0; q < NTRIP_MAX; q++) { I.ph[0][q] = 0.0; I.ph[1][q] = 0.0; I.ph[2][q] = 0.0; I.ph[3][q] = 0.0; }
  PL_XSYNC();
}

// error weights: IDA evaluates ewt from phi[0] = y_n at the start of every step.  Kept in six registers per lane (I.ew[trip]); EWT(n) reads
// it inside a PL_VEC loop.
template <class M>
PL_DEV void set_ewt(CellLDS<M>& S, IdaScalars& I, double rtol, double atol) {
  PL_MODEL(M);
  const int lane = lane_id();
  PL_VEC(n) { const double w = pl_rcp(rtol * fabs(S.phi[0][n]) + atol); I.ew[k__] = w; }
  PL_SYNC();
}
#define EWT(n) I.ew[k__]

template <class M>
PL_DEV double ida_set_coeffs(CellLDS<M>& S, IdaScalars& I) {
  PL_MODEL(M);
  const int lane = lane_id();
  const int kk = I.kk; const double hh = I.hh;
  if (hh != I.hused || kk != I.kused) I.ns = 0;
  I.ns = (I.ns + 1 < I.kused + 2) ? I.ns + 1 : I.kused + 2;
  double alphas = 0.0, alpha0 = 0.0, ak;
  const double rinv[MAXORD + 1] = {1.0, 1.0 / 2, 1.0 / 3, 1.0 / 4, 1.0 / 5, 1.0 / 6};     // (compile-time quotients: identical values, no runtime division)
  if (kk + 1 >= I.ns) {
    // IDASetCoeffs recurrences; the divisions are done by lanes 0..kk in parallel, the (division-free) prefix products / sums by every lane from register broadcasts
    // (no LDS round trip per order), each lane m <= kk then stores entry m:
    //   psi_new[0] = h, psi_new[i] = psi_old[i-1] + h ; alpha[i] = h/psi_new[i] ; beta[i] = prod_{m<=i} psi_new[m-1]/psi_old[m-1] ;
    //   sigma[i] = i sigma[i-1] alpha[i] ; gamma[i] = gamma[i-1] + alpha[i-1]/h
    // (loads with clamped indices and all four quotients unconditionally, selected afterwards: a guarded load or quotient is an exec-masked branch of its own, and four
    //  quotients in four basic blocks are four dependent chains one after the other instead of four interleaved ones)
    const int i = lane <= kk ? lane : 0;
    const double ld1 = S.ida_psi[i > 0 ? i - 1 : 0], ld2 = S.ida_psi[i > 1 ? i - 2 : 0];
    const double po_im1 = i > 0 ? ld1 : 1.0, po_im2 = i > 1 ? ld2 : 1.0;
    const double pn_i = i > 0 ? po_im1 + hh : hh;
    const double pn_im1 = i > 1 ? po_im2 + hh : hh;
    const double d_al = pl_div(hh, pn_i), d_q = pl_div(pn_im1, po_im1), d_ap = pl_div(hh, pn_im1);
    const double al = i > 0 ? d_al : 1.0;
    const double q = i > 0 ? d_q : 1.0;
    const double al_prev = i > 1 ? d_ap : 1.0;
    const double d_g = pl_div(al_prev, hh);
    const double g = i > 0 ? d_g : 0.0;
    double bm = 1.0, sg = 1.0, gm = 0.0, myb = q, mys = 1.0, myg = g;
    _Pragma("unroll") for (int m = 1; m <= MAXORD; m++) if (m <= kk) {
      const double qm = lane_bcast(q, m), am = lane_bcast(al, m), gmm = lane_bcast(g, m);
      bm *= qm; sg = m * sg * am; gm += gmm;
      if (lane == m) { myb = bm; mys = sg; myg = gm; }
    }
    _Pragma("unroll") for (int m = 0; m < MAXORD; m++) if (m < kk) { alphas -= rinv[m]; alpha0 -= lane_bcast(al, m); }
    ak = lane_bcast(al, kk);                         // alpha[kk] (from the register of lane kk: no LDS round trip behind the stores below)
    PL_XSYNC();                                      // (every lane has read the old psi)
    if (lane <= kk && wave_id() == 0) { S.ida_psi[lane] = pn_i; S.ida_alpha[lane] = al; S.ida_beta[lane] = myb; S.ida_sigma[lane] = mys; S.ida_gamma[lane] = myg; }
    PL_XSYNC();
  } else {
    for (int m = 0; m < kk; m++) { alphas -= (m == 0 ? rinv[0] : m == 1 ? rinv[1] : m == 2 ? rinv[2] : m == 3 ? rinv[3] : rinv[4]); alpha0 -= S.ida_alpha[m]; }
    ak = S.ida_alpha[kk];
  }
  I.cjlast = I.cj; I.cj = pl_div(-alphas, hh);
  double ck = fabs(ak + alphas - alpha0); if (ck < ak) ck = ak;
  // (IDASetCoeffs' rescaling phi[m] *= beta[m], m = ns .. kk, is done by the first form_iterate of the step, in the pass that sums the predictor anyway: same
  //  products, one pass over the history less; ida_restore undoes it from the same beta / ns)
  I.tn += hh;
  PL_SYNC();
  return ck;
}

// yy = ypred + ee, yp = yppred + cj ee with the predictor re-summed from phi (no separate predictor storage)
template <class M>
PL_DEV void form_iterate(CellLDS<M>& S, IdaScalars& I, bool first = true) {
  PL_MODEL(M);
  const int lane = lane_id();
  PL_AMARK("form_iterate");
  if (M::PRED_REGS && !first) {          // the predictor of this step is already in registers
    PL_VEC(n) { const double e = EE(n); S.yy[n] = I.pa[k__] + e; S.yp[n] = I.pb[k__] + I.cj * e; }
    PL_XSYNC();
    return;
  }
  // history vector outermost, the lane's trips innermost: the LDS loads of one order are issued back to back (a runtime-bounded inner
  // loop over the orders would expose one LDS round trip per order and trip); same summation order as before
  double a[NTRIP], b[NTRIP];
  if constexpr (PL_FLAT<M>) {
    const int kk = I.kk, ns = I.ns;
    double gm[MAXORD + 1], bt[MAXORD + 1];
    _Pragma("unroll") for (int j = 1; j <= MAXORD; j++) { gm[j] = S.ida_gamma[j]; bt[j] = S.ida_beta[j]; }
    double p[3][NTRIP];
    PL_VEC(n) { a[k__] = S.phi[0][n]; b[k__] = 0.0; }
    _Pragma("unroll") for (int j = 1; j <= 3; j++) PL_VEC(n) p[j - 1][k__] = S.phi[j][n];
    bool resc[4];
    _Pragma("unroll") for (int j = 1; j <= 3; j++) {
      const bool on = j <= kk; resc[j] = on && first && j >= ns;
      const double f = resc[j] ? bt[j] : 1.0, m = on ? 1.0 : 0.0, g = on ? gm[j] : 0.0;
      PL_VEC(n) { const double q = p[j - 1][k__] * f; p[j - 1][k__] = q; a[k__] += m * q; b[k__] += g * q; }
    }
    if (kk > 3) {
      double p4[NTRIP], p5[NTRIP];
      PL_VEC(n) { p4[k__] = S.phi[4][n]; p5[k__] = S.phi[5][n]; }
      const bool on5 = kk > 4, r4 = first && 4 >= ns, r5 = on5 && first && 5 >= ns;
      const double f4 = r4 ? bt[4] : 1.0, f5 = r5 ? bt[5] : 1.0, m5 = on5 ? 1.0 : 0.0, g5 = on5 ? gm[5] : 0.0;
      PL_VEC(n) { const double q = p4[k__] * f4; p4[k__] = q; a[k__] += q; b[k__] += gm[4] * q; }
      PL_VEC(n) { const double q = p5[k__] * f5; p5[k__] = q; a[k__] += m5 * q; b[k__] += g5 * q; }
      if (r4) { PL_VEC(n) S.phi[4][n] = p4[k__]; }
      if (r5) { PL_VEC(n) S.phi[5][n] = p5[k__]; }
    }
    // (the rescaled orders go back last: stores only under the wave-uniform branches)
    if (resc[1]) { PL_VEC(n) S.phi[1][n] = p[0][k__]; }
    if (resc[2]) { PL_VEC(n) S.phi[2][n] = p[1][k__]; }
    if (resc[3]) { PL_VEC(n) S.phi[3][n] = p[2][k__]; }
  } else {
  PL_VEC(n) { a[k__] = S.phi[0][n]; b[k__] = 0.0; }
  if constexpr (PHI_REGS<M> && !PL_BRANCHY_PHI) {
    // History orders that live in registers (thermal model): NO control flow around them.  A wave-uniform branch whose arm rewrites a register-resident order makes the
    // compiler reconcile the whole register array at the join (hundreds of v_mov per call, r03: 12 cycles per instruction in the step-control phases); instead every order is
    // processed with wave-uniform 0 / 1 factors -- p * 1.0, fma(0.0, p, a) and fma(1.0, p, a) are exact, so the sums are the ones of the branching form, bit for bit.
    _Pragma("unroll") for (int j = 1; j <= MAXORD; j++) {
      if (j < M::PHI_LDS) {                      // (orders kept in LDS: j <= kk always holds for them, kk >= 1)
        const double g = S.ida_gamma[j];
        if (first && j >= I.ns) { const double bt = S.ida_beta[j]; PL_VEC(n) { const double p = S.phi[j][n] * bt; S.phi[j][n] = p; a[k__] += p; b[k__] += g * p; } }
        else PL_VEC(n) { const double p = S.phi[j][n]; a[k__] += p; b[k__] += g * p; }
      } else {
        const bool on = j <= I.kk;
        const double m = on ? 1.0 : 0.0, mg = on ? S.ida_gamma[j <= MAXORD ? j : 0] : 0.0, bt = (on && first && j >= I.ns) ? S.ida_beta[j] : 1.0;
        PL_VEC(n) { const double p = PHI_RD(j, n) * bt; PHI_WR(j, n, p); a[k__] += m * p; b[k__] += mg * p; }
      }
    }
  } else
  _Pragma("unroll") for (int j = 1; j <= MAXORD; j++) if (j <= I.kk) {
    const double g = S.ida_gamma[j];
    if (first && j >= I.ns) {            // the rescaling of IDASetCoeffs (see ida_set_coeffs)
      const double bt = S.ida_beta[j];
      PL_VEC(n) { const double p = PHI_RD(j, n) * bt; PHI_WR(j, n, p); a[k__] += p; b[k__] += g * p; }
    } else PL_VEC(n) { const double p = PHI_RD(j, n); a[k__] += p; b[k__] += g * p; }
  }
  }
  PL_VEC(n) { const double e = EE(n); S.yy[n] = a[k__] + e; S.yp[n] = b[k__] + I.cj * e; if (M::PRED_REGS) { I.pa[k__] = a[k__]; I.pb[k__] = b[k__]; } }
  PL_XSYNC();
  PL_AMARK("form_iterate end");
}

// IDANls + Newton + convergence test.  0 ok, 1 recoverable failure
template <int F, class M>
PL_DEV int ida_nls(CellLDS<M>& S, LaneRegs& R, const Tables* tb, IdaScalars& I, int mode, double value, int jac_every_step, Counters& cnt, int nref, const plh_run* xrun = nullptr,
                   double* xvalue = nullptr, GenRow* g = nullptr) {
  // (xrun / xvalue: a closure input is re-evaluated inside every residual and leaves the value of its last evaluation behind -- run.value[] of scalar_residual.jl:170 -- for
  //  check_reinitialization!)
  PL_MODEL(M);
  const int lane = lane_id();
  const double epsNewt = 0.33, toldel = 0.0001 * epsNewt;
  int callLSetup = 0;
  if (I.nst == 0) { I.cjold = I.cj; I.ss = 20.0; callLSetup = 1; }
  else {
    I.cjratio = pl_div(I.cj, I.cjold);
    const double temp1 = (1.0 - 0.25) / (1.0 + 0.25), temp2 = 1.0 / temp1;
    if (I.cjratio < temp1 || I.cjratio > temp2) callLSetup = 1;
    if (I.cj != I.cjlast) I.ss = 100.0;
    if (jac_every_step) callLSetup = 1;
  }
  PL_VEC(n) EE(n) = 0.0;
  PL_SYNC();
  // One loop, one site per phase (form_iterate, residual + Jacobian + factorisation, residual, solve): every device function is inlined into the
  // kernel, so each extra call site would be another copy of the node pass in the instruction stream of the hot loop.
  int jcur = 0, ret = 0, m = 0; bool done = false; double oldnrm = 0.0;
  for (bool first = true;; first = false) {
    { PL_TIC(); form_iterate(S, I, first); PL_TOC(S, PH_NEWTVEC); }
    if (done) break;                                      // the iterate (yy, yp) now includes the last correction
    if constexpr ((F & GF_EXPR) != 0) { if (xrun) { value = closure_input(S, *xrun, I.tn, S.yy, S.yp); *xvalue = value; } }       // closure input: run.func(t, Y, YP, p) inside every residual (scalar_residual.jl:169-170)
    if (callLSetup) {
      PL_TIC();
#ifndef PL_EXP_NO_JAC
      cell_res_jac(S, R, S.yy, S.yp, S.delta, mode, value);
#else
      cell_residual(S, R, S.yy, S.yp, S.delta, mode, value);
#endif
#ifndef PL_EXP_NO_FACTOR
      bool genf = false;
      if constexpr ((F & GF_GENROW) != 0) { if (g && g->on()) { genf = true; gen_factor(S, R, tb, I.cj, mode, false, *g, I.tn, S.yy, S.yp, S.yp); } }   // (S.yp is dead until the next form_iterate; the row is evaluated before W overwrites it)
      if (!genf) cell_factor(S, R, tb, I.cj, mode, false);
#endif
      cnt_add(cnt, C_RES); cnt_add(cnt, C_JAC); cnt_add(cnt, C_FACT);
      I.cjold = I.cj; I.cjratio = 1.0; I.ss = 20.0; jcur = 1; callLSetup = 0;
      PL_TOC(S, PH_JACFACT);
    } else {
      PL_TIC();
      cell_residual(S, R, S.yy, S.yp, S.delta, mode, value);
      cnt_add(cnt, C_RES);
      PL_TOC(S, PH_RES);
    }
    cnt_add(cnt, C_NEWTON); cnt_add(cnt, C_SOLVE);
    { PL_TIC();
#ifndef PL_EXP_NO_SOLVE
    bool gens = false;
    if constexpr ((F & GF_GENROW) != 0) gens = g && g->on();
    if ((F & GF_REFINE) && nref > 0) cell_solve_refined(S, R, tb, S.delta, S.yp, I.cjold, mode, false, nref, gens ? g : nullptr);   // (S.yp is dead until the next form_iterate; cjold = cj of the factors)
    else if (gens) gen_solve(S, R, S.delta, false, *g);
    else cell_solve(S, R, S.delta, mode, false);           // x = J^-1 F ; the Newton correction is -x
#endif
    PL_TOC(S, PH_SOLVE); }
    PL_TIC();
    const double sc = (I.cjratio != 1.0) ? -2.0 * pl_rcp(1.0 + I.cjratio) : -1.0;
    double s = 0.0;
    PL_VEC(n) { const double d = S.delta[n] * sc; EE(n) += d; const double p = d * EWT(n); s += p * p; }
    const double delnrm = pl_sqrt(block_sum<M>(S, s) * (1.0 / NST));
    PL_SYNC();
    ret = 2;
    if (m == 0) { oldnrm = delnrm; if (delnrm <= toldel) ret = 0; }
    else { const double q = pl_div(delnrm, oldnrm); const double rate = (m == 1) ? q : pow(q, 1.0 / m); if (rate > 0.9) ret = 1; else I.ss = pl_div(rate, 1.0 - rate); }
    if (ret == 2 && I.ss * delnrm <= epsNewt) ret = 0;
    if (!(delnrm == delnrm)) ret = 1;
    PL_TOC(S, PH_NEWTVEC);
    if (ret == 2) { m++; if (m < 4) continue; ret = 1; }  // another iteration with the same matrix
    if (ret == 1 && !jcur) { callLSetup = 1; m = 0; PL_VEC(n) EE(n) = 0.0; PL_SYNC(); continue; }   // failed with a stale Jacobian: refresh and restart
    done = true;                                            // converged (0) or failed with a current Jacobian (1)
  }
  return ret;
}

template <class M>
PL_DEV int ida_test_error(CellLDS<M>& S, IdaScalars& I, double ck, double& err_k, double& err_km1) {
  PL_MODEL(M);
  const int lane = lane_id();
  const int kk = I.kk;
  PL_AMARK("test_error");
  double s0 = 0, s1 = 0, s2 = 0;
#ifdef PL_WAVE_EMU
  if (getenv("PL_EMU_TRACE_EE") && blockIdx.x == 0 && I.nst < atoi(getenv("PL_EMU_TRACE_EE"))) { PL_VEC(n) fprintf(stderr, "dev ee %d %d %.6e %.6e\n", I.nst + 1, n, EE(n), EE(n) * EWT(n)); }
#endif
  // phi[kk], phi[kk-1] through a compile-time order index (wave-uniform branches): the history orders that live in registers (thermal model) are then read directly
  // instead of through a select chain per element
  // (with the whole history in LDS a runtime order is just an address, and the extra branches cost 2 % of the isothermal kernels: PHI_REGS selects the form)
  if constexpr (PL_FLAT<M>) {
    // phi[kk] and phi[kk - 1] through clamped runtime indices, all three sums unconditionally (those of the orders the decision does not look at are computed and ignored)
    const int j1 = kk, j2 = kk > 1 ? kk - 1 : 0;
    double pk[NTRIP], pkm1[NTRIP];
    PL_VEC(n) { pk[k__] = S.phi[j1][n]; pkm1[k__] = S.phi[j2][n]; }
    PL_VEC(n) {
      const double w = EWT(n), e = EE(n);
      double p = e * w; s0 += p * p;
      const double d1 = pk[k__] + e; p = d1 * w; s1 += p * p;
      const double d2 = d1 + pkm1[k__]; p = d2 * w; s2 += p * p;
    }
  } else if constexpr (PHI_REGS<M>) {
    double pk[NTRIP], pkm1[NTRIP];
    PL_VEC(n) { pk[k__] = 0.0; pkm1[k__] = 0.0; }
    _Pragma("unroll") for (int j = 1; j <= MAXORD; j++) {
      if (j == kk && kk > 1) { PL_VEC(n) pk[k__] = PHI_RD(j, n); }
      else if (j == kk - 1 && kk > 2) { PL_VEC(n) pkm1[k__] = PHI_RD(j, n); }
    }
    PL_VEC(n) {
      const double w = EWT(n), e = EE(n);
      double p = e * w; s0 += p * p;
      if (kk > 1) { const double d1 = pk[k__] + e; p = d1 * w; s1 += p * p;
        if (kk > 2) { const double d2 = d1 + pkm1[k__]; p = d2 * w; s2 += p * p; } }
    }
  } else {
    PL_VEC(n) {
      const double w = EWT(n), e = EE(n);
      double p = e * w; s0 += p * p;
      if (kk > 1) { const double d1 = PHI_RD(kk, n) + e; p = d1 * w; s1 += p * p;
        if (kk > 2) { const double d2 = d1 + PHI_RD(kk - 1, n); p = d2 * w; s2 += p * p; } }
    }
  }
  // (the three sigma entries are read before the reduction: clamped indices, unconditional)
  const double sg_k = S.ida_sigma[kk], sg_km1 = S.ida_sigma[kk > 1 ? kk - 1 : 0], sg_km2 = S.ida_sigma[kk > 2 ? kk - 2 : 0];
  block_sum3<M>(S, s0, s1, s2);                         // (all three at once: one pair of barriers with two waves per cell)
  const double enorm_k = pl_sqrt(s0 * (1.0 / NST));
  err_k = sg_k * enorm_k; const double terr_k = (kk + 1) * err_k;
  I.knew = kk; err_km1 = 0.0;
  if (kk > 1) {
    const double enorm_km1 = pl_sqrt(s1 * (1.0 / NST)); err_km1 = sg_km1 * enorm_km1; const double terr_km1 = kk * err_km1;
    if (kk > 2) {
      const double enorm_km2 = pl_sqrt(s2 * (1.0 / NST)); const double err_km2 = sg_km2 * enorm_km2; const double terr_km2 = (kk - 1) * err_km2;
      if ((terr_km1 > terr_km2 ? terr_km1 : terr_km2) <= terr_k) I.knew = kk - 1;
    } else if (terr_km1 <= 0.5 * terr_k) I.knew = kk - 1;
  }
  PL_AMARK("test_error end");
  return (ck * enorm_k > 1.0) ? 1 : 0;
}

template <class M>
PL_DEV void ida_restore(CellLDS<M>& S, IdaScalars& I, double saved_t) {
  PL_MODEL(M);
  const int lane = lane_id();
  I.tn = saved_t;
  if (lane == 0 && wave_id() == 0) for (int j = 1; j <= I.kk; j++) S.ida_psi[j - 1] = S.ida_psi[j] - I.hh;
  if constexpr (PHI_REGS<M>) {
    _Pragma("unroll") for (int j = 0; j <= MAXORD; j++) if (j >= I.ns && j <= I.kk) { const double b = 1.0 / S.ida_beta[j]; PL_VEC(n) PHI_WR(j, n, PHI_RD(j, n) * b); }
  } else {
    if (I.ns <= I.kk) for (int j = I.ns; j <= I.kk; j++) { const double b = 1.0 / S.ida_beta[j]; PL_VEC(n) PHI_WR(j, n, PHI_RD(j, n) * b); }
  }
  PL_XSYNC();
}

// IDACompleteStep.  Returns true when it has also produced the solution at the new time, y(tn) -> S.yy, y'(tn) -> S.yp: unless the step ended on tstop (the caller then
// interpolates to tstop), IDAGetSolution(tn) is  y = phi[0]  and  y' = sum_j d_{j-1} phi[j]  (delt = 0 makes every c_j with j >= 1 vanish), i.e. by-products of the
// running sums that update the history -- one pass over the history instead of two.  The coefficients d are formed by IDAGetSolution's own recurrence; y' is summed from the
// top order down (IDAGetSolution sums upwards: last-bit differences in the REPORTED y' only -- the integrator continues from phi, not from y').
template <class M>
PL_DEV bool ida_complete_step(CellLDS<M>& S, IdaScalars& I, double err_k, double err_km1, double tstop) {
  PL_MODEL(M);
  const int lane = lane_id();
  PL_AMARK("complete_step");
  I.nst++;
  const int kdiff = I.kk - I.kused; I.kused = I.kk; I.hused = I.hh;
  if (I.knew == I.kk - 1 || I.kk == I.maxord) I.phase = 1;
  if (I.phase == 0) { if (I.nst > 1) { I.kk++; I.hh *= 2.0; } }
  else {
    int action = 0;
    double err_kp1 = 0.0, err_knew;
    if (I.knew == I.kk - 1) action = 1;
    else if (I.kk == I.maxord) action = 2;
    else if (I.kk + 1 >= I.ns || kdiff == 1) action = 2;
    if (action == 0) {
      double s = 0.0;
      if constexpr (PHI_REGS<M>) {
        _Pragma("unroll") for (int j = 2; j <= MAXORD; j++) if (j == I.kk + 1) { PL_VEC(n) { const double p = (EE(n) - PHI_RD(j, n)) * EWT(n); s += p * p; } }
      } else { PL_VEC(n) { const double p = (EE(n) - PHI_RD(I.kk + 1, n)) * EWT(n); s += p * p; } }
      const double enorm = pl_sqrt(block_sum<M>(S, s) * (1.0 / NST)); err_kp1 = pl_div(enorm, (double)(I.kk + 2));
      const double terr_k = (I.kk + 1) * err_k, terr_kp1 = (I.kk + 2) * err_kp1;
      if (I.kk == 1) action = (terr_kp1 >= 0.5 * terr_k) ? 2 : 3;
      else { const double terr_km1 = I.kk * err_km1;
        if (terr_km1 <= (terr_k < terr_kp1 ? terr_k : terr_kp1)) action = 1; else if (terr_kp1 >= terr_k) action = 2; else action = 3; }
    }
    if (action == 3) { I.kk++; err_knew = err_kp1; } else if (action == 1) { I.kk--; err_knew = err_km1; } else err_knew = err_k;
    double hnew = I.hh; I.rr = pl_inv_root(2.0 * err_knew + 0.0001, I.kk + 1);   // = (2 err + 1e-4)^(-1/(k+1))
    if (I.rr >= 2.0) hnew = 2.0 * I.hh;
    else if (I.rr <= 1.0) { I.rr = I.rr < 0.9 ? I.rr : 0.9; I.rr = I.rr > 0.5 ? I.rr : 0.5; hnew = I.hh * I.rr; }
    I.hh = hnew;
  }
  const int ku = I.kused;
  const bool at_tstop = fabs(I.tn - tstop) <= 100.0 * 2.220446049250313e-16 * (fabs(I.tn) + fabs(I.hh));     // (ida_step's test, with the step size chosen above)
  // d_{j-1} of IDAGetSolution at t = tn (delt = 0: c_j = 0 for j >= 1, gam_j = psi[j-1] / psi[j]); the reciprocals of psi by lanes 0..ku in parallel as in ida_get_solution
  double dc1 = 0, dc2 = 0, dc3 = 0, dc4 = 0, dc5 = 0;
  if (!at_tstop) {
    const double ps0 = S.ida_psi[0], ps1 = S.ida_psi[1], ps2 = S.ida_psi[2], ps3 = S.ida_psi[3], ps4 = S.ida_psi[4];      // (one batch of loads: the steps below are wave-uniform branches)
    const double ps_mine = S.ida_psi[lane <= MAXORD ? lane : MAXORD];
    pl_pin(ps0, ps1, ps2, ps3, ps4, ps_mine);
    const double ps_of[5] = {ps0, ps1, ps2, ps3, ps4};
    const double rp_mine = pl_rcp(ps_mine);
    const double rp0 = lane_bcast(rp_mine, 0), rp1 = lane_bcast(rp_mine, 1), rp2 = lane_bcast(rp_mine, 2), rp3 = lane_bcast(rp_mine, 3), rp4 = lane_bcast(rp_mine, 4),
                 rp5 = lane_bcast(rp_mine, 5);
    const double delt = 0.0;
    double c = 1.0, d = 0.0, gam = delt * rp0;
#define PL_GS_STEP(J, RPJM1, RPJ, DV) if (J <= ku) { d = d * gam + c * RPJM1; c = c * gam; gam = (delt + ps_of[J - 1]) * RPJ; DV = d; }
    PL_GS_STEP(1, rp0, rp1, dc1) PL_GS_STEP(2, rp1, rp2, dc2) PL_GS_STEP(3, rp2, rp3, dc3) PL_GS_STEP(4, rp3, rp4, dc4) PL_GS_STEP(5, rp4, rp5, dc5)
#undef PL_GS_STEP
  }
  {   // phi update (running sums from the top order down), orders outermost / trips innermost as in form_iterate; y' accumulates d_{j-1} * (new phi[j]) on the way
    double acc[NTRIP], sp[NTRIP];
    const double dku = ku == 1 ? dc1 : ku == 2 ? dc2 : ku == 3 ? dc3 : ku == 4 ? dc4 : dc5;
    // (the order index of every history access is a compile-time constant under a wave-uniform branch: direct register access for the orders kept in registers)
    if constexpr (PL_FLAT<M>) {
      // the running sum starts as ee and runs from the top order down; order j joins it iff j <= ku (factor 1.0 / 0.0: exact), takes the sum back iff j <= ku, takes ee iff
      // j = ku + 1 <= maxord, and enters y' with d_{j-1} (dc_j is 0 beyond ku).  Orders 4, 5 under one branch (ku >= 3), orders 0 .. 3 always: loads, sums, then the stores.
      (void)dku;
      const bool grow = ku < I.maxord;
      PL_VEC(n) { acc[k__] = EE(n); sp[k__] = 0.0; }
      if (ku >= 3) {
        double q5[NTRIP], q4[NTRIP];
        PL_VEC(n) { q5[k__] = S.phi[5][n]; q4[k__] = S.phi[4][n]; }
        const double m5 = ku >= 5 ? 1.0 : 0.0, m4 = ku >= 4 ? 1.0 : 0.0;
        PL_VEC(n) { acc[k__] += m5 * q5[k__]; q5[k__] = acc[k__]; sp[k__] += dc5 * acc[k__]; acc[k__] += m4 * q4[k__]; q4[k__] = acc[k__]; sp[k__] += dc4 * acc[k__]; }
        if (ku >= 5) { PL_VEC(n) S.phi[5][n] = q5[k__]; } else if (grow && ku == 4) { PL_VEC(n) S.phi[5][n] = EE(n); }
        if (ku >= 4) { PL_VEC(n) S.phi[4][n] = q4[k__]; } else if (grow) { PL_VEC(n) S.phi[4][n] = EE(n); }      // (ku == 3 here)
      }
      double q[4][NTRIP];
      _Pragma("unroll") for (int j = 3; j >= 0; j--) PL_VEC(n) q[j][k__] = S.phi[j][n];
      const double m3 = ku >= 3 ? 1.0 : 0.0, m2 = ku >= 2 ? 1.0 : 0.0;
      PL_VEC(n) {
        acc[k__] += m3 * q[3][k__]; q[3][k__] = acc[k__]; sp[k__] += dc3 * acc[k__];
        acc[k__] += m2 * q[2][k__]; q[2][k__] = acc[k__]; sp[k__] += dc2 * acc[k__];
        acc[k__] += q[1][k__]; q[1][k__] = acc[k__]; sp[k__] += dc1 * acc[k__];
        acc[k__] += q[0][k__];
      }
      if (ku >= 3) { PL_VEC(n) S.phi[3][n] = q[3][k__]; } else if (grow && ku == 2) { PL_VEC(n) S.phi[3][n] = EE(n); }
      if (ku >= 2) { PL_VEC(n) S.phi[2][n] = q[2][k__]; } else if (grow) { PL_VEC(n) S.phi[2][n] = EE(n); }           // (ku == 1 here)
      PL_VEC(n) { S.phi[1][n] = q[1][k__]; S.phi[0][n] = acc[k__]; }
    } else if constexpr (PHI_REGS<M> && PL_BRANCHY_PHI) {      // (r03 form, kept for same-box A/B builds: tools/experiments/build_modes.py th_branchy)
      _Pragma("unroll") for (int j = MAXORD; j >= 0; j--) {
        if (j == ku + 1 && ku < I.maxord) { PL_VEC(n) PHI_WR(j, n, EE(n)); }
        else if (j == ku) { PL_VEC(n) { acc[k__] = PHI_RD(j, n) + EE(n); PHI_WR(j, n, acc[k__]); sp[k__] = dku * acc[k__]; } }
        else if (j < ku) {
          const double dj = j == 1 ? dc1 : j == 2 ? dc2 : j == 3 ? dc3 : j == 4 ? dc4 : 0.0;       // (j = 0: phi[0] does not enter y')
          PL_VEC(n) { acc[k__] += PHI_RD(j, n); PHI_WR(j, n, acc[k__]); if (j > 0) sp[k__] += dj * acc[k__]; }
        }
      }
    } else if constexpr (PHI_REGS<M>) {
      // branch-free over the register-resident orders (see form_iterate): the running sum starts as ee, order j joins it iff j <= ku (factor 1.0 / 0.0: exact), takes the
      // sum back iff j <= ku, takes ee iff j = ku + 1 < = maxord, and enters y' with d_{j-1} (0 for the orders that are not part of the step: dc_j is 0 beyond ku)
      (void)dku;
      const bool grow = ku < I.maxord;
      PL_VEC(n) { acc[k__] = EE(n); sp[k__] = 0.0; }
      _Pragma("unroll") for (int j = MAXORD; j >= 0; j--) {
        const double dj = j == 1 ? dc1 : j == 2 ? dc2 : j == 3 ? dc3 : j == 4 ? dc4 : j == 5 ? dc5 : 0.0;       // (j = 0: phi[0] does not enter y')
        if (j < M::PHI_LDS) {                    // LDS orders 0 .. PHI_LDS - 1 <= ku: always part of the sum
          PL_VEC(n) { acc[k__] += S.phi[j][n]; S.phi[j][n] = acc[k__]; if (j > 0) sp[k__] += dj * acc[k__]; }
        } else {
          const bool in = j <= ku, take_ee = grow && j == ku + 1;
          const double m = in ? 1.0 : 0.0;
          PL_VEC(n) {
            const double ph = PHI_RD(j, n), e = EE(n);
            const double an = acc[k__] + m * ph;           // (contracts to one fma on the GPU; m * ph is exact either way)
            acc[k__] = an;
            PHI_WR(j, n, in ? an : (take_ee ? e : ph));
            sp[k__] += dj * an;
          }
        }
      }
    } else {
      PL_VEC(n) { const double e = EE(n); if (ku < I.maxord) PHI_WR(ku + 1, n, e); acc[k__] = PHI_RD(ku, n) + e; PHI_WR(ku, n, acc[k__]); sp[k__] = dku * acc[k__]; }
      _Pragma("unroll") for (int j = MAXORD - 1; j >= 0; j--) if (j < ku) {
        const double dj = j == 1 ? dc1 : j == 2 ? dc2 : j == 3 ? dc3 : j == 4 ? dc4 : 0.0;       // (j = 0: phi[0] does not enter y')
        PL_VEC(n) { acc[k__] += PHI_RD(j, n); PHI_WR(j, n, acc[k__]); if (j > 0) sp[k__] += dj * acc[k__]; }
      }
    }
    if (!at_tstop) { PL_VEC(n) { S.yy[n] = acc[k__]; S.yp[n] = sp[k__]; } }
  }
  PL_XSYNC();
  PL_AMARK("complete_step end");
  return !at_tstop;
}

// IDAGetSolution(t): y -> yo, y' -> ypo (LDS vectors)
template <class M>
PL_DEV void ida_get_solution(CellLDS<M>& S, const IdaScalars& I, double t, double* yo, double* ypo) {
  PL_MODEL(M);
  const int lane = lane_id();
  int kord = I.kused; if (kord == 0) kord = 1;
  const double delt = t - I.tn;
  // reciprocals of psi[0..kord] by lanes 0..kord in parallel (the recurrence below is then division-free)
  const double rp_mine = pl_rcp(S.ida_psi[lane <= MAXORD ? lane : MAXORD]);
  const double rp0 = lane_bcast(rp_mine, 0), rp1 = lane_bcast(rp_mine, 1), rp2 = lane_bcast(rp_mine, 2), rp3 = lane_bcast(rp_mine, 3),
               rp4 = lane_bcast(rp_mine, 4), rp5 = lane_bcast(rp_mine, 5);
  double c = 1.0, d = 0.0, gam = delt * rp0;
  double c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, d0 = 0, d1 = 0, d2 = 0, d3 = 0, d4 = 0;
#define PL_GS_STEP(J, RPJM1, RPJ, CV, DV) if (J <= kord) { d = d * gam + c * RPJM1; c = c * gam; gam = (delt + S.ida_psi[J - 1]) * RPJ; CV = c; DV = d; }
  PL_GS_STEP(1, rp0, rp1, c1, d0) PL_GS_STEP(2, rp1, rp2, c2, d1) PL_GS_STEP(3, rp2, rp3, c3, d2) PL_GS_STEP(4, rp3, rp4, c4, d3) PL_GS_STEP(5, rp4, rp5, c5, d4)
#undef PL_GS_STEP
  PL_VEC(n) {
    // (history in global memory: orders beyond the one in use have never been written)
    const double p1 = S.phi[1][n], p2 = (!M::PHI_GLOBAL || kord >= 2) ? PHI_RD(2, n) : 0.0, p3 = (!M::PHI_GLOBAL || kord >= 3) ? PHI_RD(3, n) : 0.0,
                 p4 = (!M::PHI_GLOBAL || kord >= 4) ? PHI_RD(4, n) : 0.0, p5 = (!M::PHI_GLOBAL || kord >= 5) ? PHI_RD(5, n) : 0.0;
    double s = S.phi[0][n] + c1 * p1, sp = d0 * p1;
    if (kord >= 2) { s += c2 * p2; sp += d1 * p2; }
    if (kord >= 3) { s += c3 * p3; sp += d2 * p3; }
    if (kord >= 4) { s += c4 * p4; sp += d3 * p4; }
    if (kord >= 5) { s += c5 * p5; sp += d4 * p5; }
    yo[n] = s; ypo[n] = sp;
  }
  PL_XSYNC();
}

// next tstop after run-local time t: the sorted set opts.tstops U {tdiscon - reltol/2} U {1.0 if continuation} U {tf} of postfix_integrator!
// (model_evaluation.jl:288-310) walked without storing it
PL_DEV double next_tstop(const plh_opts& o, double t, bool continuation, double tf) {
  double best = tf;
  if (continuation && 1.0 > t && 1.0 < best) best = 1.0;
  if (o.n_tstops > 0) {                                                // user tstops (model_evaluation.jl:292-294; sorted device copy): first entry > max(t, 0)
    const double lim = t > 0.0 ? t : 0.0;
    int lo = -1, hi = o.n_tstops;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (o.tstops[mid] > lim) hi = mid; else lo = mid; }
    if (hi < o.n_tstops && o.tstops[hi] < best) best = o.tstops[hi];
  }
  if (o.n_tdiscon > 0) {                                               // o.tdiscon is sorted ascending (plh_integrate stages a sorted copy): first entry with tdiscon - reltol/2 > max(t, 0)
    const double lim = (t > 0.0 ? t : 0.0) + o.reltol / 2;
    int lo = -1, hi = o.n_tdiscon;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (o.tdiscon[mid] > lim) hi = mid; else lo = mid; }
    // (start one entry early: the bisection compares tdiscon > t + reltol/2, the test below tdiscon - reltol/2 > t)
    for (int q = hi > 0 ? hi - 1 : 0; q < o.n_tdiscon; q++) { const double s = o.tdiscon[q] - o.reltol / 2; if (s > t && s > 0.0) { if (s < best) best = s; break; } }
  }
  return best;
}

// one IDASolve(ONE_STEP_TSTOP) call: advances, returns y(tret), y'(tret) in S.yy / S.yp.  0 ok, <0 failure
template <int F, class M>
PL_DEV int ida_step(CellLDS<M>& S, LaneRegs& R, const Tables* tb, IdaScalars& I, double tstop, double& tret, int mode, double& value,
                               const plh_opts& o, Counters& cnt, const plh_run* frun = nullptr, GenRow* g = nullptr) {
  PL_MODEL(M);
  const int lane = lane_id();
  const double uround = 2.220446049250313e-16;
  I.rtol = o.reltol; I.atol = o.abstol;
  if (I.nst == 0) {
    set_ewt(S, I, o.reltol, o.abstol);
    const double tdist = fabs(tstop - I.tn);
    double hh = I.h0_forced != 0.0 ? I.h0_forced : o.init_step;
    if (hh == 0.0) {
      hh = 0.001 * tdist;
      double sy = 0.0;
      PL_VEC(n) { const double pq = S.phi[1][n] * EWT(n); sy += pq * pq; }
      const double ypnorm = sqrt(block_sum<M>(S, sy) * (1.0 / NST));
      if (ypnorm > 0.5 / hh) hh = 0.5 / ypnorm;
    }
    if ((I.tn + hh - tstop) * hh > 0.0) hh = (tstop - I.tn) * (1.0 - 4.0 * uround);
    I.hh = hh; I.kk = 0; I.kused = 0;
    PL_VEC(n) S.phi[1][n] *= hh;
    PL_SYNC();
  } else {
    const double troundoff = 100.0 * uround * (fabs(I.tn) + fabs(I.hh));
    if (fabs(I.tn - tstop) <= troundoff) { ida_get_solution(S, I, tstop, S.yy, S.yp); tret = tstop; return 0; }
    if ((I.tn + I.hh - tstop) * I.hh > 0.0) I.hh = (tstop - I.tn) * (1.0 - 4.0 * uround);
    set_ewt(S, I, o.reltol, o.abstol);
  }
  const double saved_t = I.tn; int ncf = 0, nef = 0; double err_k = 0, err_km1 = 0;
  if (I.nst == 0) { I.kk = 1; I.kused = 0; I.hused = 0.0; if (lane == 0 && wave_id() == 0) S.ida_psi[0] = I.hh; I.cj = 1.0 / I.hh; I.phase = 0; I.ns = 0; PL_XSYNC(); }
  for (;;) {
    double ck; { PL_TIC(); PL_TICE(3); ck = ida_set_coeffs(S, I); PL_TOC(S, PH_STEPCTL); PL_TOCE(S, 3, 0); }
    if constexpr ((F & GF_FUNC) != 0) { if (frun && frun->mode != PLH_MODE_DSTATE) value = run_input<F>(S, *frun, I.tn, S.yy, S.yp); }                           // every residual of this step attempt is evaluated at t = tn
    const int nflag = ida_nls<F>(S, R, tb, I, mode, value, o.jac_every_step, cnt, o.refine, ((F & GF_EXPR) && frun && (frun->value_kind == PLH_VAL_EXPR || frun->mode == PLH_MODE_DSTATE)) ? frun : nullptr, &value, g);
    int errfail = 0;
    if (nflag == 0) { PL_TIC(); PL_TICE(3); errfail = ida_test_error(S, I, ck, err_k, err_km1); PL_TOC(S, PH_STEPCTL); PL_TOCE(S, 3, 1); }
    if (nflag != 0 || errfail) {
      ida_restore(S, I, saved_t);
      I.phase = 1;
      if (!errfail) {
        cnt_add(cnt, C_CONVFAIL);
        I.rr = 0.25; I.hh *= I.rr; ncf++;
        if (ncf >= 10) return PLH_ERR_STALL;
      } else {
        cnt_add(cnt, C_ERRFAIL); nef++;
        if (nef == 1) { const double err_knew = (I.kk == I.knew) ? err_k : err_km1; I.kk = I.knew;
          I.rr = 0.9 * pl_inv_root(2.0 * err_knew + 0.0001, I.kk + 1); I.rr = I.rr < 0.9 ? I.rr : 0.9; I.rr = I.rr > 0.25 ? I.rr : 0.25; I.hh *= I.rr; }
        else if (nef == 2) { I.kk = I.knew; I.rr = 0.25; I.hh *= I.rr; }
        else if (nef < 10) { I.kk = 1; I.rr = 0.25; I.hh *= I.rr; }
        else return PLH_ERR_STALL;
      }
      const double tscale = fabs(I.tn) > 1.0 ? fabs(I.tn) : 1.0;
      if (fabs(I.hh) < 1e-14 * tscale) return PLH_ERR_STALL;
      if (I.nst == 0) { if (lane == 0 && wave_id() == 0) S.ida_psi[0] = I.hh; const double rr = I.rr; PL_VEC(n) S.phi[1][n] *= rr; PL_XSYNC(); }
      continue;
    }
    break;
  }
  cnt_add(cnt, C_STEPS); cnt_add(cnt, C_SUMKP2, I.kk + 2);
#ifdef PL_WAVE_EMU
  if (getenv("PL_EMU_TRACE") && lane == 0 && blockIdx.x == 0) fprintf(stderr, "dev step %d tn %.9g h %.6g k %d knew %d phase %d ns %d err_k %.6e err_km1 %.6e nef %d ncf %d\n", I.nst + 1, I.tn, I.hh, I.kk, I.knew, I.phase, I.ns, err_k, err_km1, nef, ncf);
#endif
  PL_TIC(); PL_TICE(3);
  const bool have_sol = ida_complete_step(S, I, err_k, err_km1, tstop);
  PL_TOCE(S, 3, 2);
  if (!have_sol) { ida_get_solution(S, I, tstop, S.yy, S.yp); tret = tstop; PL_TOC(S, PH_STEPCTL); PL_TOCE(S, 3, 3); return 0; }      // the step ended on tstop (|tn - tstop| <= troundoff)
  if ((I.tn + I.hh - tstop) * I.hh > 0.0) I.hh = (tstop - I.tn) * (1.0 - 4.0 * uround);
  tret = I.tn;                                                                      // y(tn), y'(tn) are in S.yy / S.yp (ida_complete_step)
  PL_TOC(S, PH_STEPCTL); PL_TOCE(S, 3, 3);
  return 0;
}

// ---- stop conditions (check_simulation_stop!, src/checks.jl:1-224); scalars are wave-uniform ----
struct PrevVals { double frac, V, SOC, I, c_s_n, c_e_min, eta_pl, dfilm, T, g; };     // (g: the caller's stop function, plh_opts.stop_ops)

template <class M>
__device__ __forceinline__ double cellV(const double* Y) { return Y[M::O_PS] - Y[M::O_PS + NJ - 1]; }

// calc_T_avg / temperature_weighting (aux...jl:649-679): wave-uniform, every lane must call it
template <class M>
__device__ __forceinline__ double cellTavg(const CellLDS<M>& S, const double* Y) {
  if constexpr (M::THERMAL) { const int lane = lane_id(); return wave_sum(lane < NT ? S.th.wT5[tsec_of(lane)] * Y[M::O_T + lane] : 0.0); }
  else return S.cc.T0;
}

template <int F = 0, class M>
PL_DEV void check_stop(CellLDS<M>& S, const plh_run& run, const plh_opts& o, double t, double tf, const double* Y, const double* YP,
                                  double SOC, PrevVals& pv, int& flag) {
  PL_MODEL(M);
  const double eps = t < 1.0 ? o.reltol : 0.0;
  [[maybe_unused]] double Tav = 0.0, dTav = 0.0;
  if constexpr (M::THERMAL) { Tav = cellTavg<M>(S, Y); dTav = cellTavg<M>(S, YP); }     // (all lanes, before any early return)
  // r06: every operand of the tests that run at every accepted step -- the bounds of the run (LDS copy of the descriptor) and the handful of state entries they look at --
  // is loaded here, together; r05 loaded each where its test used it: ten dependent LDS round trips per step (1.5 k cycles by the phase timers for a dozen comparisons)
  const plh_bounds& b = run.bounds;
  double bImax = b.I_max, bImin = b.I_min, bVmin = b.V_min, bVmax = b.V_max, bSmin = b.SOC_min, bSmax = b.SOC_max, bcs = b.c_s_n_max, bce = b.c_e_min;
  double bep = b.eta_plating_min, Ic = Y[O_I], dI = YP[O_I], V = cellV<M>(Y), dV = cellV<M>(YP), ep = Y[O_PS + NP] - Y[O_PE + NP + NS], dep = YP[O_PS + NP] - YP[O_PE + NP + NS];
  [[maybe_unused]] double bTmax = M::THERMAL ? b.T_max : 0.0;
  const int rmode = run.mode, vkind = run.value_kind;
  pl_pin(bImax, bImin, bVmin, bVmax, bSmin, bSmax, bcs, bce); pl_pin(bep, Ic, dI, V, dV, ep, dep, bTmax);
  if (t >= tf) { flag = 0; return; }
  if (!o.check_bounds || vkind == PLH_VAL_REST) return;
  if (rmode != PLH_MODE_I) {                                                            // check_stop_I, checks.jl:31-54
    if ((Ic - bImax > eps) && dI > 0) { const double f = (pv.I - bImax) / (pv.I - Ic); if (f < pv.frac) { pv.frac = f; flag = 7; } }
    else if ((bImin - Ic > eps) && dI < 0) { const double f = (pv.I - bImin) / (pv.I - Ic); if (f < pv.frac) { pv.frac = f; flag = 8; } }
    pv.I = Ic;
  }
  if (rmode != PLH_MODE_V) {                                                            // check_stop_V, checks.jl:56-81
    if ((bVmin - V > eps) && dV < 0) { const double f = (pv.V - bVmin) / (pv.V - V); if (f < pv.frac) { pv.frac = f; flag = 1; } }
    else if ((V - bVmax > eps) && dV > 0) { const double f = (pv.V - bVmax) / (pv.V - V); if (f < pv.frac) { pv.frac = f; flag = 2; } }
    pv.V = V;
  }
  {                                                                                     // check_stop_SOC, checks.jl:83-104
    if ((bSmin - SOC > eps) && Ic < 0) { const double f = (pv.SOC - bSmin) / (pv.SOC - SOC); if (f < pv.frac) { pv.frac = f; flag = 3; } }
    else if ((SOC - bSmax > eps) && Ic > 0) { const double f = (pv.SOC - bSmax) / (pv.SOC - SOC); if (f < pv.frac) { pv.frac = f; flag = 4; } }
    pv.SOC = SOC;
  }
  if constexpr (M::THERMAL) {                                                           // check_stop_T, checks.jl:106-124
    if (bTmax == bTmax && rmode != PLH_MODE_DT) {
      if (Tav - bTmax > eps && dTav > 0) { const double f = (pv.T - bTmax) / (pv.T - Tav); if (f < pv.frac) { pv.frac = f; flag = 5; } }
      pv.T = Tav;
    }
  }
  if (bcs == bcs) {                                                                     // check_stop_c_s_surf, checks.jl:141-161
    double cm = -1e300; for (int i = 0; i < NN; i++) { const double v = M::SD == 0 ? Y[O_CS + cs_surf(NP + i)] : Y[O_CS + NP + i]; cm = v > cm ? v : cm; }     // (c_s_n_maximum, checks.jl:125-139)
    const double lim = bcs * S.cc.cmaxn;
    if (Ic > 0 && cm - lim > eps) { const double f = (pv.c_s_n - lim) / (pv.c_s_n - cm); if (f < pv.frac) { pv.frac = f; flag = 6; } }
    pv.c_s_n = cm;
  }
  if (bce == bce) {                                                                     // check_stop_c_e, checks.jl:163-183
    double cm = 1e300; for (int i = 0; i < NE; i++) { const double v = Y[O_CE + i]; cm = v < cm ? v : cm; }
    if (bce - cm > eps) { const double f = (pv.c_e_min - bce) / (pv.c_e_min - cm); if (f < pv.frac) { pv.frac = f; flag = 9; } }
    pv.c_e_min = cm;
  }
  if (bep == bep) {                                                                     // check_stop_η_plating, checks.jl:185-201
    if (bep - ep > eps && dep < 0) { const double f = (pv.eta_pl - bep) / (pv.eta_pl - ep); if (f < pv.frac) { pv.frac = f; flag = 11; } }
    pv.eta_pl = ep;
  }
  if constexpr (M::SEI) {                                                               // check_stop_dfilm, checks.jl:203-224
    const double bdf = b.dfilm_max;
    double dm = -1e300; for (int i = 0; i < NN; i++) { const double v = YP[O_FILM + i]; dm = v > dm ? v : dm; }
    if (bdf == bdf && dm - bdf > eps) { const double f = (pv.dfilm - bdf) / (pv.dfilm - dm); if (f < pv.frac) { pv.frac = f; flag = 10; } }
    pv.dfilm = dm;
  }
  if constexpr ((F & GF_EXPR) != 0) {                                                   // opts.stop_function, checks.jl:26: after the built-in checks (plh_opts.stop_ops)
    if (o.n_stop > 0) {
      const double g = prog_eval(S, o.stop_ops, o.stop_args, 0, o.n_stop, t, Y, YP);
      if (g > eps) { const double f = pv.g / (pv.g - g); if (f < pv.frac) { pv.frac = f; flag = PLH_FLAG_STOP_FUNCTION; } }
      pv.g = g;
    }
  }
}

#include "dfn_sens.h"

struct CellOut {
  double *t, *V, *I, *SOC, *T, *Yall;
  int max_pts;
};

// the whole protocol for one cell.  Yprev/YPprev: per-cell scratch in HBM.  The back-interpolation of a run that ends on a bound (interp_final_points!) needs the
// PREVIOUS accepted point.  Its Y is not stored: after IDACompleteStep the BDF history holds phi[0] = y_n and phi[1] = y_n - y_(n-1) (the first modified divided
// difference), so y_(n-1) = phi[0] - phi[1] to the last bit of y_n -- formed only when a bound fires, instead of N stores per step (r03: 1.0 MB of the 1.8 MB written per C3
// trajectory).  Its YP is not a linear function of the current history once the order has dropped, so YPprev IS stored per step -- but only when the caller asked for
// YP_final (the reference interpolates YP only with var_keep.YP, model_evaluation.jl:369-373; without it nothing reads YP of a run end: the next run starts from
// newtons_method!, which resets it).  Yprev also keeps the point a run (re)starts from, for check_solve's first-step retry (checks.jl:227-237).
// F: feature flags (GenFlag): constant-input protocols (the benchmark path) carry none of the general code
template <int F, class M>
PL_DEV void cell_simulate(CellLDS<M>& S, LaneRegs& R, const Tables* tb, double SOC0, const double* Yinit, double t_init, int n_runs, const plh_run* runs, const plh_opts& o,
                                     const CellOut& out, int* n_pts_out, plh_run_info* info, Counters& cnt, double* Yfin, double* YPfin,
                                     double* Yprev, double* YPprev, int cell, double* genW = nullptr, SensArgs sens = SensArgs(), const double* th0 = nullptr, double* phig = nullptr,
                                     bool from_states = false) {
  PL_MODEL(M);
  const int lane = lane_id();
  IdaScalars I;
  I.phg = phig;
  [[maybe_unused]] SensCell<M> SX;
  if constexpr ((F & GF_SENS) != 0) { SX.a = sens; SX.th0 = th0; SX.cell = cell; SX.P = tb->P; SX.max_pts = out.max_pts; SX.first = true; SX.n_it = 0; SX.n_fail = 0; SX.n_refresh = 0; }
  int nout = 0;
  bool have_prev = false;
  const double T0 = S.cc.T0;
  // (what a run inherits from the one before -- SOC, end time, V / I / eta_plating -- travels through S.carry: see CellLDS)
  if (Yinit && !from_states) {                                          // simulate!(sol, ...): continue from sol.Y[end]
    PL_VECG(n) S.yy[n] = Yinit[n];
    PL_XSYNC();
    have_prev = true;
    if (lane == 0 && wave_id() == 0) { S.carry[0] = SOC0; S.carry[1] = t_init; S.carry[2] = cellV<M>(S.yy); S.carry[3] = S.yy[O_I]; S.carry[4] = S.yy[O_PS + NP] - S.yy[O_PE + NP + NS]; }
  } else {
    // simulate(p, ...; initial_states = Y) (model_evaluation.jl:15, 102-110, 193-199): a NEW solution -- t0 = 0, no tstop at 1 s, nothing to :hold -- that starts from the caller's
    // state vector instead of initial_guess! (the algebraic part is re-solved by newtons_method! as always; SOC0 is the caller's calc_SOC(Y), scalar_residual.jl:95-102)
    if (Yinit) { PL_VECG(n) S.yy[n] = Yinit[n]; PL_XSYNC(); }
    if (lane == 0 && wave_id() == 0) { S.carry[0] = SOC0; S.carry[1] = 0.0; S.carry[2] = 0.0; S.carry[3] = 0.0; S.carry[4] = 0.0; }
  }
  PL_XSYNC();
  double* const outp = wave_id() != 0 ? nullptr : (lane == 0 ? out.t : (lane == 1 ? out.V : (lane == 2 ? out.I : (lane == 3 ? out.SOC : (lane == 4 ? out.T : nullptr)))));
  auto save_pt = [&](int idx, double tt, const double* Y, double soc) {
    const double Tav = (M::THERMAL && out.T) ? cellTavg<M>(S, Y) : T0;
    if constexpr ((F & GF_STOPS) != 0) if (out.Yall && idx < out.max_pts) { PL_VECG(n) out.Yall[(size_t)idx * NST + n] = Y[n]; }      // outputs = :all
    double Vv = cellV<M>(Y), Iv = Y[O_I];                              // (read by every lane AHEAD of the store: no LDS round trip under the lane mask)
    pl_pin(Vv, Iv);
    // r06: lanes 0 .. 4 each own ONE of the five per-point output arrays (pointer in the lane: outp) and store their value with one instruction; r05 had lane 0 walk five
    // `if (pointer) store` blocks whose addresses were rebuilt from spilled scalar registers every step (~100 instructions per saved point)
    const double val = lane == 0 ? tt : (lane == 1 ? Vv : (lane == 2 ? Iv : (lane == 3 ? soc : Tav)));
    if (outp && idx < out.max_pts) outp[idx] = val;
  };
  for (int r = 0; r < n_runs; r++) {
    // the run descriptor is staged in LDS once per run: reading it from global memory in every step costs ~1.4 k cycles per step (scalar
    // loads that cannot be hoisted past the global stores), and a by-value register copy proved fragile under register pressure
    PL_XSYNC();                                                       // (the other wave may still be reading the previous run's descriptor)
    if (lane == 0 && wave_id() == 0) { S.runc = runs[r]; if (S.runc.value_cell) S.runc.value = S.runc.value_cell[cell]; if (S.runc.tf_cell) S.runc.tf = S.runc.tf_cell[cell]; }
    PL_XSYNC();
    const plh_run& run = S.runc;
    double SOC = S.carry[0];
    const double t_global = S.carry[1], prev_V = S.carry[2], prev_I = S.carry[3], prev_etap = S.carry[4];
    // PLH_MODE_DSTATE (x - YP[ind] = 0) runs as a control residual with no method part (PLH_MODE_RES) whose "closure" is YP[ind] (closure_input, gen_factor)
    const bool dstate = (F & GF_GENROW) && run.mode == PLH_MODE_DSTATE;
    const int mode = dstate ? PLH_MODE_RES : run.mode;
    if constexpr ((F & GF_GENROW) != 0) {
      if (dstate) {      // which state: the extreme surface / electrolyte concentration of the state the run starts from (input_methods.jl:195-247; argmax / argmin: the first extreme)
        const int kind = run.dstate;
        const int first = kind <= PLH_DSTATE_CS_P_MIN ? O_CS + (M::SD == 0 ? cs_surf(0) : 0) : (kind <= PLH_DSTATE_CS_N_MIN ? O_CS + (M::SD == 0 ? cs_surf(NP) : NP) : O_CE);
        const int count = kind <= PLH_DSTATE_CS_P_MIN ? NP : (kind <= PLH_DSTATE_CS_N_MIN ? NN : NE);
        const int stride = M::SD != 0 ? 1 : (kind <= PLH_DSTATE_CS_P_MIN ? NRP : (kind <= PLH_DSTATE_CS_N_MIN ? NRN : 1));
        const bool want_max = (kind & 1) != 0;
        int best = first; double bv = S.yy[first];
        for (int q = 1; q < count; q++) { const double v = S.yy[first + q * stride]; if (want_max ? v > bv : v < bv) { bv = v; best = first + q * stride; } }
        PL_XSYNC();
        if (lane == 0 && wave_id() == 0) S.runc.dstate = best;
        PL_XSYNC();
      }
    }
    const bool new_run = !have_prev;
    double t0;
    // S.yy holds the current state Y, S.yp the current YP between steps
    if (new_run) { t0 = 0.0; if (!(Yinit && from_states)) cell_initial_guess(S, S.yy, SOC0); SOC = SOC0; }
    else t0 = nextafter(t_global, 1e300);                               // initial_time, model_evaluation.jl:112
    // initial_current! (input_methods.jl:11-74)
    double value = run.value, Iguess;
    const bool is_tab = (F & GF_FUNC) && (run.value_kind == PLH_VAL_TABLE || run.value_kind == PLH_VAL_EXPR);      // run_function
    const bool is_fun = is_tab && run.mode != PLH_MODE_RES;              // run_function; a `res` closure is a run_residual (checks.jl:226: stall test, no check_reinitialization!)
    if (is_tab) {                                                       // run_function: initial_current!, input_methods.jl:28-34, 65-76, 104-107, 143-153
      value = run_input<F>(S, run, 0.0, S.yy, S.yp);
      if (mode == PLH_MODE_I) Iguess = value;
      else if (mode == PLH_MODE_P) Iguess = value / (cellV<M>(S.yy) * S.cc.I1C);
      else if (mode == PLH_MODE_RES) Iguess = have_prev ? prev_I : 1.0;                                  // input_methods.jl:171-176 (res_I_guess = nothing)
      else if (have_prev) Iguess = prev_I;
      else { const double OCV = cellV<M>(S.yy); Iguess = value > OCV ? 1.0 : -1.0; }
    } else
    if (dstate) {                                                       // run_residual: custom_res! (:hold -> 0), initial_current! (input_methods.jl:171-176)
      if (run.value_kind == PLH_VAL_HOLD) { value = 0.0; if (lane == 0 && wave_id() == 0) S.runc.value = 0.0; }
      Iguess = have_prev ? prev_I : 1.0;
    } else
    if (mode == PLH_MODE_I) {
      if (run.value_kind == PLH_VAL_HOLD) value = have_prev ? prev_I : 0.0;
      else if (run.value_kind == PLH_VAL_REST) value = 0.0;
      Iguess = value;
    } else if (mode == PLH_MODE_V) {
      if (run.value_kind == PLH_VAL_HOLD) { value = prev_V; Iguess = prev_V; }
      else if (have_prev && prev_I != 0.0) Iguess = prev_I;
      else { const double OCV = cellV<M>(S.yy); Iguess = value > OCV ? 1.0 : -1.0; }
    } else if (mode == PLH_MODE_P) {                                    // input_methods.jl:86-103
      if (run.value_kind == PLH_VAL_HOLD) { value = prev_I * S.cc.I1C * prev_V; Iguess = prev_I; }
      else if (run.value_kind == PLH_VAL_REST) { value = 0.0; Iguess = 0.0; }
      else Iguess = value / (cellV<M>(S.yy) * S.cc.I1C);
    } else if (mode == PLH_MODE_ETA_P) {                                // input_methods.jl:120-142
      if (run.value_kind == PLH_VAL_HOLD) { value = prev_etap; Iguess = prev_I; }
      else if (have_prev) Iguess = prev_I;
      else { const double OCV = cellV<M>(S.yy); Iguess = value > OCV ? 1.0 : -1.0; }
    } else {                                                            // dT: custom_res! (model_evaluation.jl:155-172): :hold -> 0 K/s
      if (run.value_kind == PLH_VAL_HOLD) value = 0.0;
      Iguess = have_prev ? prev_I : 1.0;                                // input_methods.jl:171-176
    }
    PL_XSYNC();
    if (lane == 0 && wave_id() == 0) S.yy[O_I] = Iguess;
    PL_XSYNC();
    GenRow grow;                                                        // closure of the state with derivative programs: general control row (dfn_cell.h)
    if constexpr ((F & GF_GENROW) != 0) { if (((run.value_kind == PLH_VAL_EXPR && run.n_dcol > 0) || dstate) && genW) { grow.run = &run; grow.W = genW; } }
    int flag = PLH_FLAG_RUNNING;
    plh_run_info ri; ri.flag = PLH_FLAG_RUNNING; ri.iterations = 0; ri.t_end = t_global; ri.V = 0; ri.I = 0; ri.SOC = SOC; ri.T_avg = T0;
    // tstops = {tdiscon - reltol/2} U {1.0 if continuation} U {tf}   (postfix_integrator!, model_evaluation.jl:288-310)
    const bool continuation = !new_run;
    PrevVals pv; pv.frac = 1.0; pv.V = -1; pv.SOC = -1; pv.I = -1; pv.c_s_n = -1; pv.c_e_min = -1; pv.eta_pl = -1; pv.dfilm = -1; pv.T = -1; pv.g = -1;
    double tprev = 0.0, t = 0.0, t_prev_saved = t0; int iter = 1; bool stalled_once = false;
    double I_prev_pt = 0.0, t_restart = 0.0;
    [[maybe_unused]] double soc_nm1_s = 0.0, dt_step_s = 0.0, soc_n_s = 0.0;      // (sensitivities: the trapezoid SOC at the last two accepted points and the step between them)
    bool first_init = true, again = false, init_failed = false;
    [[maybe_unused]] int steps_since_restart = 2;                       // (accepted steps since a check_reinitialization! restart; 2 = "more than one")
    // (re)initialise -> integrate.  check_reinitialization! sends a run with a function input back to the consistent initialisation; until r05 that was an outer do-while
    // around this block and the step loop.  Every per-run scalar (SOC, t, the previous point's values) was then carried around TWO nested loops, and the table-input
    // instantiation was the one kernel that kept coming out of the compiler with a garbage SOC or first step: r03 (LCO), r05 under three other instruction schedulers
    // (thermal; tools/experiments/miscompile_repro.py).  Now there is ONE loop: the block is a lambda, called before the loop where the input is constant and at the top of an
    // iteration (need_init) where it is a function.
    auto init_block = [&]() -> bool {                                   // false: the initialisation failed (flag / ri.flag / init_failed are set)
    again = false;
    int ierr; { PL_TIC(); ierr = cell_init_consistent<F>(S, R, tb, S.yy, S.yp, S.delta, S.phi[1], mode, value, o.reltol_init, cnt, S.phi[0], o.refine,
                                                                       ((F & GF_EXPR) && (run.value_kind == PLH_VAL_EXPR || dstate)) ? &run : nullptr, t_restart, &grow); PL_TOC(S, PH_INIT); }
    if (ierr != 0) { if (first_init) init_failed = true; else flag = ierr; ri.flag = ierr; return false; }
    if constexpr ((F & GF_STOPS) != 0) { if (o.yp_alg_zero) { PL_VEC(n) if (n >= NDIFF) S.yp[n] = 0.0; PL_XSYNC(); } }     // plh_opts.yp_alg_zero
    ida_reinit(S, I, S.yy, S.yp, first_init ? (o.max_order > 0 && o.max_order <= MAXORD ? o.max_order : MAXORD) : I.maxord, t_restart);
    if (first_init) {
      first_init = false;
      save_pt(nout, t0, S.yy, SOC); nout++;
      check_stop<F>(S, run, o, 0.0, run.tf, S.yy, S.yp, SOC, pv, flag);
      PL_VECG(n) { Yprev[n] = S.yy[n]; YPprev[n] = S.yp[n]; }
      PL_SYNC();
      I_prev_pt = S.yy[O_I];
      if constexpr ((F & GF_SENS) != 0) sens_init(S, SX, mode, value, new_run, SOC0, o.reltol, o.abstol, nout - 1, run.value_kind == PLH_VAL_HOLD && have_prev, prev_V, prev_I);
    }
    return true;
    };
    bool init_ok = true;
    [[maybe_unused]] bool need_init = true;
    if constexpr ((F & GF_FUNC) == 0) init_ok = init_block();
    while (init_ok && flag == PLH_FLAG_RUNNING) {
      if constexpr ((F & GF_FUNC) != 0) {
        if (need_init) { need_init = false; if (!init_block()) break; if (flag != PLH_FLAG_RUNNING) break; }      // (a bound can fire on the point the run starts from)
      }
      double tret = t; tprev = t;
      double tstop_now;
      if constexpr ((F & GF_STOPS) != 0) tstop_now = next_tstop(o, t, continuation, run.tf);
      else tstop_now = (continuation && run.tf > 1.0 && t < 1.0) ? 1.0 : run.tf;
      // two waves per cell: both evaluate the stop checks, SOC and I_prev_pt from S.yy redundantly; the wave that is ahead must not start overwriting S.yy (predictor of
      // the next step) while the other still reads the accepted point -- with a varying current its SOC, hence its stop flag, would differ
      if constexpr (M::W2) __syncthreads();
      const int sf = ida_step<F>(S, R, tb, I, tstop_now, tret, mode, value, o, cnt, (is_tab || dstate) ? &run : nullptr, &grow);
      if (sf != 0) {
        if (I.nst == 0 && !stalled_once) {                              // check_solve, checks.jl:227-237
          stalled_once = true;
          PL_VECG(n) { S.yy[n] = Yprev[n]; S.yp[n] = YPprev[n]; }
          PL_XSYNC();
          ida_reinit(S, I, S.yy, S.yp, I.maxord); I.h0_forced = o.reltol; iter++; t = tprev;
          // the reference's solve! has already pushed this (repeated) point and run the stop checks when check_solve shortens the first step
          // (model_evaluation.jl:319-327, checks.jl:227-231): run.info.iterations stays equal to the number of saved points of the run
          save_pt(nout, t + t0, S.yy, SOC); nout++;
          if constexpr ((F & GF_SENS) != 0) sens_repeat_point(S, SX, nout - 1);      // (the repeated point: the sensitivities of the point it repeats)
          check_stop<F>(S, run, o, t, run.tf, S.yy, S.yp, SOC, pv, flag);
          continue;
        }
        flag = sf; break;
      }
      iter++; t = tret;
      if constexpr ((F & GF_FUNC) != 0) steps_since_restart++;
      PL_TIC(); PL_TICE(3);
      const double SOC_new = SOC + pl_div(0.5 * ((t + t0) - t_prev_saved) * (S.yy[O_I] + I_prev_pt), 3600.0);   // calc_SOC, scalar_residual.jl:103-111
      [[maybe_unused]] const double dt_saved = (t + t0) - t_prev_saved, soc_before = SOC;
      if constexpr ((F & GF_SENS) != 0) { soc_nm1_s = soc_before; dt_step_s = dt_saved; }
      SOC = SOC_new;
      save_pt(nout, t + t0, S.yy, SOC); nout++;
      PL_TOCE(S, 3, 4);
      check_stop<F>(S, run, o, t, run.tf, S.yy, S.yp, SOC, pv, flag);
      PL_TOCE(S, 3, 5);
      if constexpr ((F & GF_SENS) != 0) sens_step(S, R, I, SX, mode, value, nout - 1, dt_saved);      // (the solution point in S.yy / S.yp is saved and restored around it)
      if (!is_fun && t == tprev) { flag = PLH_ERR_STALL; break; }      // (run_function has no stall test, checks.jl:251-269; run_residual -- res, dT, d<state> -- has: checks.jl:226)
      if (iter == o.maxiters) { flag = PLH_ERR_MAXITERS; break; }
      if (nout >= out.max_pts && out.max_pts > 0 && flag == PLH_FLAG_RUNNING) { flag = PLH_ERR_OUTPUT_FULL; break; }
      if (flag == PLH_FLAG_RUNNING) {
#ifdef PL_EXP_STORE_PREV     /* (A/B build: r03's per-step copy of the whole previous point) */
        PL_VECG(n) { Yprev[n] = S.yy[n]; YPprev[n] = S.yp[n]; }
#else
        if (YPfin) { PL_VECG(n) YPprev[n] = S.yp[n]; }                // fire-and-forget: read back only when a bound fires (wave-uniform condition)
#endif
        t_prev_saved = t + t0; I_prev_pt = S.yy[O_I];
        if constexpr ((F & GF_FUNC) != 0) if (is_fun && t - tprev < 1e-3 * o.reltol) {                    // check_reinitialization!, checks.jl:341-364
          const double t_new = t + o.reltol, v_new = run_input<F>(S, run, t_new, S.yy, S.yp);
          const double big = fabs(value) > fabs(v_new) ? fabs(value) : fabs(v_new);
          const double tolv = o.abstol > o.reltol * big ? o.abstol : o.reltol * big;
          if (!(fabs(value - v_new) <= tolv)) {
            value = v_new; t_restart = t_new; again = true;
            // the last saved point: check_solve's retry restores it, and if a bound fires on the FIRST step after the re-initialisation it is the previous point of the
            // back-interpolation (the history then starts at the re-initialised state, whose algebraic part is not the saved one)
            PL_VECG(n) { Yprev[n] = S.yy[n]; YPprev[n] = S.yp[n]; }
            steps_since_restart = 0;
          }
        }
      }
      PL_TOC(S, PH_OUTPUT); PL_TOCE(S, 3, 6);
      if constexpr ((F & GF_FUNC) != 0) { if (again) need_init = true; }         // back to the consistent initialisation at t_restart
    }
    if (init_failed) { if (lane == 0) info[r] = ri; for (int q = r + 1; q < n_runs; q++) if (lane == 0) { plh_run_info z = ri; z.flag = PLH_FLAG_RUNNING; info[q] = z; } break; }
    double t_end = t + t0;
    if constexpr ((F & GF_SENS) != 0) soc_n_s = SOC;
    if (flag > 0 && o.interp_final && t > 1.0) {                        // interp_final_points!, model_evaluation.jl:369-382
      const double fr = pv.frac;
      const double ti = fr * (t - tprev) + tprev;
      PL_XSYNC();
#ifdef PL_EXP_STORE_PREV
      PL_VECG(n) { S.yy[n] = fr * (S.yy[n] - Yprev[n]) + Yprev[n]; S.yp[n] = fr * (S.yp[n] - YPprev[n]) + YPprev[n]; }
#else
      // (t > 1 excludes the point a run starts from: at least one step has been completed, phi[0] / phi[1] are those of the step that crossed the bound)
      if ((F & GF_FUNC) && steps_since_restart == 1) { PL_VECG(n) { const double yprev = Yprev[n]; S.yy[n] = fr * (S.yy[n] - yprev) + yprev; } }
      else { PL_VEC(n) { const double yprev = S.phi[0][n] - S.phi[1][n]; S.yy[n] = fr * (S.yy[n] - yprev) + yprev; } }
      if (YPfin) { PL_VECG(n) { const double ypp = YPprev[n]; S.yp[n] = fr * (S.yp[n] - ypp) + ypp; } }
#endif
      PL_XSYNC();
      SOC = SOC + 0.5 * ((ti + t0) - (t + t0)) * (S.yy[O_I] + S.yy[O_I]) / 3600.0;
      t_end = ti + t0;
      save_pt(nout - 1, t_end, S.yy, SOC);
    }
    if constexpr ((F & GF_SENS) != 0) sens_finish(S, SX, flag > 0 && o.interp_final && t > 1.0, pv.frac, flag < 0, nout - 1, flag, run.bounds, mode, soc_n_s, soc_nm1_s, dt_step_s, S.yy[O_I]);
    ri.flag = flag; ri.iterations = iter; ri.t_end = t_end; ri.V = cellV<M>(S.yy); ri.I = S.yy[O_I]; ri.SOC = SOC; ri.T_avg = cellTavg<M>(S, S.yy);
    if (lane == 0 && wave_id() == 0) info[r] = ri;
    have_prev = true;
    PL_XSYNC();
    if (lane == 0 && wave_id() == 0) { S.carry[0] = SOC; S.carry[1] = t_end; S.carry[2] = ri.V; S.carry[3] = ri.I; S.carry[4] = S.yy[O_PS + NP] - S.yy[O_PE + NP + NS]; }
    if (flag < 0) { for (int q = r + 1; q < n_runs; q++) if (lane == 0) { plh_run_info z = ri; z.flag = PLH_FLAG_RUNNING; z.iterations = 0; info[q] = z; } break; }
    PL_XSYNC();
  }
  if (lane == 0 && wave_id() == 0 && n_pts_out) *n_pts_out = nout < out.max_pts ? nout : out.max_pts;
  PL_XSYNC();
  if constexpr ((F & GF_SENS) != 0) { if (lane == 0 && wave_id() == 0 && sens.stat) { sens.stat[3 * cell] = SX.n_it; sens.stat[3 * cell + 1] = SX.n_fail; sens.stat[3 * cell + 2] = SX.n_refresh; } }
  if (Yfin) PL_VECG(n) Yfin[n] = S.yy[n];
  if (YPfin) PL_VECG(n) YPfin[n] = S.yp[n];
}

}  // namespace pl

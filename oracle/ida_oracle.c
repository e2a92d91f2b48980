/*
 * ORACLE (test infrastructure, NOT product code).
 *
 * Plain-C, one-cell-at-a-time CPU restatement of the reference's time-stepping path:
 *   simulate()/simulate!()            reference src/model_evaluation.jl:11-97
 *   initialize_simulation!            reference src/model_evaluation.jl:174-232
 *   newtons_method! (consistent init) reference src/model_evaluation.jl:430-480
 *   retrieve/create/postfix integrator (IDA + KLU, tstops)  reference src/model_evaluation.jl:234-310
 *   solve! step loop                  reference src/model_evaluation.jl:312-333
 *   check_simulation_stop!/check_stop_*  reference src/checks.jl:1-224, check_solve 226-249
 *   interp_final_points!              reference src/model_evaluation.jl:369-382
 *   calc_SOC trapezoid                reference src/physics_equations/scalar_residual.jl:103-111
 *   control rows                      reference src/physics_equations/scalar_residual.jl:167-229
 *
 * The arithmetic the reference delegates to third-party libraries that are NOT in /root/reference is restated
 * from their published algorithms:
 *   - SUNDIALS IDA (Sundials.jl "4" -> SUNDIALS 5.x/6.x): fixed-leading-coefficient variable-order (1..5) BDF with
 *     modified-Newton corrector (Brenan/Campbell/Petzold DASSL lineage; IDA user guide "Mathematical
 *     considerations"): IDASetCoeffs / IDANls / IDATestError / IDAHandleNFlag / IDACompleteStep / IDAGetSolution,
 *     tstop handling of IDASolve in IDA_ONE_STEP_TSTOP mode (what Sundials.jl's step! uses).
 *   - SuiteSparse KLU (KLU.jl "0.6"): left-looking (Gilbert-Peierls style) sparse LU with threshold partial
 *     pivoting that prefers the diagonal (tol 1e-3), fill-reducing ordering on A+A' (minimum degree), and the
 *     numeric-only "refactor" that reuses the pivot sequence (klu_refactor), which is what klu!() and
 *     SUNLinSol_KLU call after the first factorization.
 * Parity at that IDA/KLU boundary is UNPINNED (no reference tests, no reference binaries here): the
 * constants below are SUNDIALS' documented defaults; accuracy is proven by tolerance tightening and the notebook
 * known-answers (tests/golden/notebook_kats.json), not by matching IDA step for step.
 *
 * The model functions (f_diff, f_alg, J_y, J_y_alg, initial_guess) are the straight-line C emitted by
 * oracle/codegen.py from oracle/dfn_model.py (the analogue of the reference's generated functions,
 * src/generate_functions.jl:124-158).
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call into this library.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------ */
/* model registry                                                                                               */
/* ------------------------------------------------------------------------------------------------------------ */
typedef void (*fres_t)(double*, const double*, const double*, const double*);
typedef void (*fjac_t)(double*, const double*, const double*, double, const double*);
typedef void (*fig_t)(double*, double, const double*);

typedef struct {
  const char* name;
  int N, Nd, nnz, nnz_alg, P;
  const int *colptr, *rowval, *acolptr, *arowval;
  const char* const* theta_keys;
  fres_t f_diff, f_alg;
  fjac_t jac, jac_alg;
  fig_t initial_guess;
  /* layout (0-based), reference src/external.jl:275-365 */
  int Np, Ns, Nn, Na, Nz, Nrp, Nrn, thermal, aging, has_Q;
  int o_ce, o_cs, o_T, o_film, o_SOH, o_j, o_pe, o_ps, o_js, o_I;
  /* thermal extras */
  int nnz_twin; const int* twin_cols; fres_t dT_twin; fjac_t dT_twin_jac; void (*dT_weights)(double*, const double*);
} orc_model;

#define DECL_VARIANT(v) \
  extern const int orc_##v##_N, orc_##v##_NDIFF, orc_##v##_NNZ, orc_##v##_NNZ_ALG, orc_##v##_P; \
  extern const int orc_##v##_colptr[], orc_##v##_rowval[], orc_##v##_alg_colptr[], orc_##v##_alg_rowval[]; \
  extern const char* const orc_##v##_theta_keys[]; \
  void orc_##v##_f_diff(double*, const double*, const double*, const double*); \
  void orc_##v##_f_alg(double*, const double*, const double*, const double*); \
  void orc_##v##_jac(double*, const double*, const double*, double, const double*); \
  void orc_##v##_jac_alg(double*, const double*, const double*, double, const double*); \
  void orc_##v##_initial_guess(double*, double, const double*);
#define DECL_THERMAL(v) \
  extern const int orc_##v##_NNZ_DT_TWIN; extern const int orc_##v##_dT_twin_cols[]; \
  void orc_##v##_dT_twin(double*, const double*, const double*, const double*); \
  void orc_##v##_dT_twin_jac(double*, const double*, const double*, double, const double*); \
  void orc_##v##_dT_weights(double*, const double*);

#ifdef ORC_HAVE_lco_iso
DECL_VARIANT(lco_iso)
#endif
#ifdef ORC_HAVE_lco_thermal
DECL_VARIANT(lco_thermal) DECL_THERMAL(lco_thermal)
#endif
#ifdef ORC_HAVE_lco_thermal_g8_6_7_11_5_7
DECL_VARIANT(lco_thermal_g8_6_7_11_5_7) DECL_THERMAL(lco_thermal_g8_6_7_11_5_7)
#endif
#ifdef ORC_HAVE_lco_thermal_g8_6_7_11_5_7_rn13
DECL_VARIANT(lco_thermal_g8_6_7_11_5_7_rn13) DECL_THERMAL(lco_thermal_g8_6_7_11_5_7_rn13)
#endif
#ifdef ORC_HAVE_lgm50_thermal
DECL_VARIANT(lgm50_thermal) DECL_THERMAL(lgm50_thermal)
#endif
#ifdef ORC_HAVE_lco_thermal_tdiff
DECL_VARIANT(lco_thermal_tdiff) DECL_THERMAL(lco_thermal_tdiff)
#endif
#ifdef ORC_HAVE_lco_thermal_quiet
DECL_VARIANT(lco_thermal_quiet) DECL_THERMAL(lco_thermal_quiet)
#endif
#ifdef ORC_HAVE_lco_iso_quiet
DECL_VARIANT(lco_iso_quiet)
#endif
#ifdef ORC_HAVE_nmc_iso_sei_quiet
DECL_VARIANT(nmc_iso_sei_quiet)
#endif
#ifdef ORC_HAVE_lco_iso_sei
DECL_VARIANT(lco_iso_sei)
#endif
#ifdef ORC_HAVE_nmc_iso_sei
DECL_VARIANT(nmc_iso_sei)
#endif
#ifdef ORC_HAVE_nmc_iso
DECL_VARIANT(nmc_iso)
#endif
#ifdef ORC_HAVE_lgm50_iso
DECL_VARIANT(lgm50_iso)
#endif
#ifdef ORC_HAVE_lco_iso_g12_7_9_11
DECL_VARIANT(lco_iso_g12_7_9_11)
#endif
#ifdef ORC_HAVE_lco_iso_g7_6_8_12_rn10
DECL_VARIANT(lco_iso_g7_6_8_12_rn10)
#endif
#ifdef ORC_HAVE_nmc_iso_sei_g6_5_8_13
DECL_VARIANT(nmc_iso_sei_g6_5_8_13)
#endif
#ifdef ORC_HAVE_lco_iso_quad
DECL_VARIANT(lco_iso_quad)
#endif
#ifdef ORC_HAVE_lco_iso_poly
DECL_VARIANT(lco_iso_poly)
#endif
#ifdef ORC_HAVE_lco_iso_nu
DECL_VARIANT(lco_iso_nu)
#endif
#ifdef ORC_HAVE_lco_iso_mhc
DECL_VARIANT(lco_iso_mhc)
#endif

static void set_layout(orc_model* m, int thermal, int aging) {
  if (m->Np == 0) m->Np = m->Ns = m->Nn = 10;   /* (another discretisation: FILL_VARIANT_GRID) */
  if (m->Na == 0) m->Na = m->Nz = 10;           /* (FILL_THERMAL_GRID sets the collector discretisation) */
  if (m->Nrp == 0) m->Nrp = m->Nrn = 10;      /* (1 for the quadratic / polynomial solid-diffusion approximations: one c_s_avg per particle) */
  m->thermal = thermal; m->aging = aging;
  int o = 0;
  m->o_ce = o; o += m->Np + m->Ns + m->Nn;
  m->o_cs = o; o += m->Np * m->Nrp + m->Nn * m->Nrn;
  m->o_T = -1; m->o_film = -1; m->o_SOH = -1; m->o_js = -1;
  if (thermal) { m->o_T = o; o += m->Na + m->Np + m->Ns + m->Nn + m->Nz; }
  if (aging) { m->o_film = o; o += m->Nn; m->o_SOH = o; o += 1; }
  if (m->has_Q) o += m->Np + m->Nn;            /* Q, polynomial approximation only (states_definition.jl:60-67) */
  m->o_j = o; o += m->Np + m->Nn;
  m->o_pe = o; o += m->Np + m->Ns + m->Nn;
  m->o_ps = o; o += m->Np + m->Nn;
  if (aging) { m->o_js = o; o += m->Nn; }
  m->o_I = o; o += 1;
}

#define FILL_VARIANT(m, v, th_, ag_) do { memset(m, 0, sizeof(*m)); (m)->name = #v; (m)->N = orc_##v##_N; (m)->Nd = orc_##v##_NDIFF; \
  (m)->nnz = orc_##v##_NNZ; (m)->nnz_alg = orc_##v##_NNZ_ALG; (m)->P = orc_##v##_P; (m)->colptr = orc_##v##_colptr; \
  (m)->rowval = orc_##v##_rowval; (m)->acolptr = orc_##v##_alg_colptr; (m)->arowval = orc_##v##_alg_rowval; \
  (m)->theta_keys = orc_##v##_theta_keys; (m)->f_diff = orc_##v##_f_diff; (m)->f_alg = orc_##v##_f_alg; \
  (m)->jac = orc_##v##_jac; (m)->jac_alg = orc_##v##_jac_alg; (m)->initial_guess = orc_##v##_initial_guess; \
  (m)->Nrp = (m)->Nrn = 0; (m)->has_Q = 0; set_layout(m, th_, ag_); } while (0)
#define FILL_VARIANT_GRID(m, v, ag_, np_, ns_, nn_, nr_) do { FILL_VARIANT(m, v, 0, ag_); (m)->Np = np_; (m)->Ns = ns_; (m)->Nn = nn_; (m)->Nrp = (m)->Nrn = nr_; \
  set_layout(m, 0, ag_); } while (0)
#define FILL_THERMAL_GRID(m, v, np_, ns_, nn_, nr_, na_, nz_) do { FILL_VARIANT(m, v, 1, 0); (m)->Np = np_; (m)->Ns = ns_; (m)->Nn = nn_; (m)->Nrp = (m)->Nrn = nr_; \
  (m)->Na = na_; (m)->Nz = nz_; set_layout(m, 1, 0); (m)->nnz_twin = orc_##v##_NNZ_DT_TWIN; (m)->twin_cols = orc_##v##_dT_twin_cols; \
  (m)->dT_twin = orc_##v##_dT_twin; (m)->dT_twin_jac = orc_##v##_dT_twin_jac; (m)->dT_weights = orc_##v##_dT_weights; } while (0)
#define FILL_VARIANT_SD(m, v, nr_, q_) do { FILL_VARIANT(m, v, 0, 0); (m)->Nrp = (m)->Nrn = nr_; (m)->has_Q = q_; set_layout(m, 0, 0); } while (0)

static int get_model(const char* name, orc_model* m) {
#ifdef ORC_HAVE_lco_iso
  if (!strcmp(name, "lco_iso")) { FILL_VARIANT(m, lco_iso, 0, 0); return 0; }
#endif
#ifdef ORC_HAVE_lco_iso_quiet
  if (!strcmp(name, "lco_iso_quiet")) { FILL_VARIANT(m, lco_iso_quiet, 0, 0); return 0; }
#endif
#ifdef ORC_HAVE_lco_thermal
  if (!strcmp(name, "lco_thermal")) { FILL_VARIANT(m, lco_thermal, 1, 0);
    m->nnz_twin = orc_lco_thermal_NNZ_DT_TWIN; m->twin_cols = orc_lco_thermal_dT_twin_cols;
    m->dT_twin = orc_lco_thermal_dT_twin; m->dT_twin_jac = orc_lco_thermal_dT_twin_jac; m->dT_weights = orc_lco_thermal_dT_weights;
    return 0; }
#endif
#ifdef ORC_HAVE_lco_thermal_tdiff
  if (!strcmp(name, "lco_thermal_tdiff")) { FILL_VARIANT(m, lco_thermal_tdiff, 1, 0);
    m->nnz_twin = orc_lco_thermal_tdiff_NNZ_DT_TWIN; m->twin_cols = orc_lco_thermal_tdiff_dT_twin_cols;
    m->dT_twin = orc_lco_thermal_tdiff_dT_twin; m->dT_twin_jac = orc_lco_thermal_tdiff_dT_twin_jac; m->dT_weights = orc_lco_thermal_tdiff_dT_weights;
    return 0; }
#endif
#ifdef ORC_HAVE_lco_thermal_quiet
  if (!strcmp(name, "lco_thermal_quiet")) { FILL_VARIANT(m, lco_thermal_quiet, 1, 0);
    m->nnz_twin = orc_lco_thermal_quiet_NNZ_DT_TWIN; m->twin_cols = orc_lco_thermal_quiet_dT_twin_cols;
    m->dT_twin = orc_lco_thermal_quiet_dT_twin; m->dT_twin_jac = orc_lco_thermal_quiet_dT_twin_jac; m->dT_weights = orc_lco_thermal_quiet_dT_weights;
    return 0; }
#endif
#ifdef ORC_HAVE_lgm50_thermal
  if (!strcmp(name, "lgm50_thermal")) { FILL_VARIANT(m, lgm50_thermal, 1, 0);
    m->nnz_twin = orc_lgm50_thermal_NNZ_DT_TWIN; m->twin_cols = orc_lgm50_thermal_dT_twin_cols;
    m->dT_twin = orc_lgm50_thermal_dT_twin; m->dT_twin_jac = orc_lgm50_thermal_dT_twin_jac; m->dT_weights = orc_lgm50_thermal_dT_weights;
    return 0; }
#endif
#ifdef ORC_HAVE_lco_thermal_g8_6_7_11_5_7
  if (!strcmp(name, "lco_thermal_g8_6_7_11_5_7")) { FILL_THERMAL_GRID(m, lco_thermal_g8_6_7_11_5_7, 8, 6, 7, 11, 5, 7); return 0; }
#endif
#ifdef ORC_HAVE_lco_iso_sei
  if (!strcmp(name, "lco_iso_sei")) { FILL_VARIANT(m, lco_iso_sei, 0, 1); return 0; }
#endif
#ifdef ORC_HAVE_nmc_iso_sei
  if (!strcmp(name, "nmc_iso_sei")) { FILL_VARIANT(m, nmc_iso_sei, 0, 1); return 0; }
#endif
#ifdef ORC_HAVE_nmc_iso_sei_quiet
  if (!strcmp(name, "nmc_iso_sei_quiet")) { FILL_VARIANT(m, nmc_iso_sei_quiet, 0, 1); return 0; }
#endif
#ifdef ORC_HAVE_nmc_iso
  if (!strcmp(name, "nmc_iso")) { FILL_VARIANT(m, nmc_iso, 0, 0); return 0; }
#endif
#ifdef ORC_HAVE_lgm50_iso
  if (!strcmp(name, "lgm50_iso")) { FILL_VARIANT(m, lgm50_iso, 0, 0); return 0; }
#endif
#ifdef ORC_HAVE_lco_iso_g7_6_8_12_rn10      /* N_r_p = 12, N_r_n = 10 (reference src/params.jl:124-136: independent options) */
  if (!strcmp(name, "lco_iso_g7_6_8_12_rn10")) { FILL_VARIANT_GRID(m, lco_iso_g7_6_8_12_rn10, 0, 7, 6, 8, 12); m->Nrn = 10; set_layout(m, 0, 0); return 0; }
#endif
#ifdef ORC_HAVE_lco_thermal_g8_6_7_11_5_7_rn13
  if (!strcmp(name, "lco_thermal_g8_6_7_11_5_7_rn13")) { FILL_THERMAL_GRID(m, lco_thermal_g8_6_7_11_5_7_rn13, 8, 6, 7, 11, 5, 7); m->Nrn = 13; set_layout(m, 1, 0); return 0; }
#endif
#ifdef ORC_HAVE_lco_iso_g12_7_9_11
  if (!strcmp(name, "lco_iso_g12_7_9_11")) { FILL_VARIANT_GRID(m, lco_iso_g12_7_9_11, 0, 12, 7, 9, 11); return 0; }
#endif
#ifdef ORC_HAVE_nmc_iso_sei_g6_5_8_13
  if (!strcmp(name, "nmc_iso_sei_g6_5_8_13")) { FILL_VARIANT_GRID(m, nmc_iso_sei_g6_5_8_13, 1, 6, 5, 8, 13); return 0; }
#endif
#ifdef ORC_HAVE_lco_iso_quad
  if (!strcmp(name, "lco_iso_quad")) { FILL_VARIANT_SD(m, lco_iso_quad, 1, 0); return 0; }
#endif
#ifdef ORC_HAVE_lco_iso_poly
  if (!strcmp(name, "lco_iso_poly")) { FILL_VARIANT_SD(m, lco_iso_poly, 1, 1); return 0; }
#endif
#ifdef ORC_HAVE_lco_iso_nu
  if (!strcmp(name, "lco_iso_nu")) { FILL_VARIANT(m, lco_iso_nu, 0, 0); return 0; }
#endif
#ifdef ORC_HAVE_lco_iso_mhc
  if (!strcmp(name, "lco_iso_mhc")) { FILL_VARIANT(m, lco_iso_mhc, 0, 0); return 0; }
#endif
  return -1;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* KLU-like sparse LU                                                                                           */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  int n;
  int* q;        /* column order (fill reducing) */
  int* prow;     /* prow[k] = original row chosen as k-th pivot */
  int* pinv;     /* pinv[row] = k */
  int *Lp, *Li;  double* Lx;   /* L columns (unit diagonal implied), row indices are ORIGINAL rows */
  int *Up, *Ui;  double* Ux;   /* U columns: Ui = pivot positions j<k (ascending); diagonal in Ud */
  double* Ud;
  int Lcap, Ucap;
  double* x; int* mark;
  int factored;
  double rcond;
} splu;

static void splu_free(splu* f) {
  free(f->q); free(f->prow); free(f->pinv); free(f->Lp); free(f->Li); free(f->Lx); free(f->Up); free(f->Ui);
  free(f->Ux); free(f->Ud); free(f->x); free(f->mark); memset(f, 0, sizeof(*f));
}

/* minimum-degree ordering on the pattern of A+A' (explicit elimination graph; n is a few hundred) */
static void min_degree_order(int n, const int* cp, const int* ri, int* q) {
  int W = (n + 63) / 64;
  uint64_t* adj = (uint64_t*)calloc((size_t)n * W, sizeof(uint64_t));
  char* done = (char*)calloc(n, 1);
  for (int c = 0; c < n; c++) for (int k = cp[c]; k < cp[c + 1]; k++) { int r = ri[k]; if (r != c) {
    adj[(size_t)r * W + c / 64] |= 1ull << (c % 64); adj[(size_t)c * W + r / 64] |= 1ull << (r % 64); } }
  for (int step = 0; step < n; step++) {
    int best = -1, bestdeg = 1 << 30;
    for (int v = 0; v < n; v++) if (!done[v]) { int d = 0; for (int w = 0; w < W; w++) d += __builtin_popcountll(adj[(size_t)v * W + w]);
      if (d < bestdeg) { bestdeg = d; best = v; } }
    q[step] = best; done[best] = 1;
    uint64_t* nb = &adj[(size_t)best * W];
    for (int u = 0; u < n; u++) if (!done[u] && (nb[u / 64] >> (u % 64) & 1)) {
      uint64_t* au = &adj[(size_t)u * W];
      for (int w = 0; w < W; w++) au[w] |= nb[w];
      au[u / 64] &= ~(1ull << (u % 64)); au[best / 64] &= ~(1ull << (best % 64));
    }
    for (int u = 0; u < n; u++) adj[(size_t)u * W + best / 64] &= ~(1ull << (best % 64));
  }
  free(adj); free(done);
}

static void splu_init(splu* f, int n, const int* cp, const int* ri) {
  memset(f, 0, sizeof(*f));
  f->n = n; f->q = (int*)malloc(n * sizeof(int)); f->prow = (int*)malloc(n * sizeof(int)); f->pinv = (int*)malloc(n * sizeof(int));
  f->Lp = (int*)calloc(n + 1, sizeof(int)); f->Up = (int*)calloc(n + 1, sizeof(int)); f->Ud = (double*)calloc(n, sizeof(double));
  f->x = (double*)calloc(n, sizeof(double)); f->mark = (int*)calloc(n, sizeof(int));
  f->Lcap = f->Ucap = 16 * (cp[n] + n);
  f->Li = (int*)malloc(f->Lcap * sizeof(int)); f->Lx = (double*)malloc(f->Lcap * sizeof(double));
  f->Ui = (int*)malloc(f->Ucap * sizeof(int)); f->Ux = (double*)malloc(f->Ucap * sizeof(double));
  min_degree_order(n, cp, ri, f->q);
}

/* full factorization with threshold partial pivoting (diagonal preferred, tol = 1e-3 like KLU) */
static int splu_factor(splu* f, const int* cp, const int* ri, const double* ax) {
  int n = f->n; double* x = f->x; int* mark = f->mark;
  for (int i = 0; i < n; i++) { f->pinv[i] = -1; mark[i] = 0; x[i] = 0.0; }
  int lnz = 0, unz = 0; double umin = 1e300, umax = 0.0;
  for (int k = 0; k < n; k++) {
    int c = f->q[k];
    f->Lp[k] = lnz; f->Up[k] = unz;
    for (int p = cp[c]; p < cp[c + 1]; p++) { x[ri[p]] = ax[p]; mark[ri[p]] = 1; }
    /* structural left-looking update: ascending pivot order is a valid topological order */
    for (int j = 0; j < k; j++) {
      int pr = f->prow[j];
      if (!mark[pr]) continue;
      double xj = x[pr];
      if (unz >= f->Ucap) return -2;
      f->Ui[unz] = j; f->Ux[unz] = xj; unz++;
      for (int p = f->Lp[j]; p < f->Lp[j + 1]; p++) { int i = f->Li[p]; x[i] -= f->Lx[p] * xj; mark[i] = 1; }
      x[pr] = 0.0; mark[pr] = 0;
    }
    /* pivot among non-pivotal marked rows */
    double amax = 0.0; int imax = -1;
    for (int i = 0; i < n; i++) if (mark[i] && f->pinv[i] < 0) { double a = fabs(x[i]); if (a > amax) { amax = a; imax = i; } }
    if (imax < 0 || amax == 0.0) { for (int i = 0; i < n; i++) { x[i] = 0; mark[i] = 0; } return -1; }
    int piv = imax;
    if (mark[c] && f->pinv[c] < 0 && fabs(x[c]) >= 1e-3 * amax) piv = c;   /* prefer the diagonal */
    double pv = x[piv];
    f->prow[k] = piv; f->pinv[piv] = k; f->Ud[k] = pv;
    if (fabs(pv) < umin) umin = fabs(pv);
    if (fabs(pv) > umax) umax = fabs(pv);
    for (int i = 0; i < n; i++) if (mark[i]) {
      if (i != piv && f->pinv[i] < 0) { if (lnz >= f->Lcap) return -2; f->Li[lnz] = i; f->Lx[lnz] = x[i] / pv; lnz++; }
      x[i] = 0.0; mark[i] = 0;
    }
  }
  f->Lp[n] = lnz; f->Up[n] = unz; f->factored = 1; f->rcond = umin / umax;
  return 0;
}

/* numeric-only refactorization with the stored pattern and pivot sequence (klu_refactor) */
static int splu_refactor(splu* f, const int* cp, const int* ri, const double* ax) {
  if (!f->factored) return splu_factor(f, cp, ri, ax);
  int n = f->n; double* x = f->x; double umin = 1e300, umax = 0.0;
  for (int k = 0; k < n; k++) {
    int c = f->q[k];
    for (int p = cp[c]; p < cp[c + 1]; p++) x[ri[p]] = ax[p];
    for (int p = f->Up[k]; p < f->Up[k + 1]; p++) {
      int j = f->Ui[p]; int pr = f->prow[j]; double xj = x[pr]; f->Ux[p] = xj; x[pr] = 0.0;
      for (int t = f->Lp[j]; t < f->Lp[j + 1]; t++) x[f->Li[t]] -= f->Lx[t] * xj;
    }
    double pv = x[f->prow[k]]; x[f->prow[k]] = 0.0; f->Ud[k] = pv;
    if (pv == 0.0 || pv != pv) { for (int i = 0; i < n; i++) x[i] = 0; f->factored = 0; return -1; }
    if (fabs(pv) < umin) umin = fabs(pv);
    if (fabs(pv) > umax) umax = fabs(pv);
    for (int p = f->Lp[k]; p < f->Lp[k + 1]; p++) { int i = f->Li[p]; f->Lx[p] = x[i] / pv; x[i] = 0.0; }
  }
  f->rcond = umin / umax;
  return 0;
}

/* solve A z = b, b overwritten by z */
static void splu_solve(const splu* f, double* b) {
  int n = f->n; double* y = f->x;
  /* forward: L y = P b  (L stored by columns, rows original) */
  for (int k = 0; k < n; k++) { double yk = b[f->prow[k]]; y[k] = yk; if (yk != 0.0) for (int p = f->Lp[k]; p < f->Lp[k + 1]; p++) b[f->Li[p]] -= f->Lx[p] * yk; }
  /* backward: U w = y, U stored by columns */
  for (int k = n - 1; k >= 0; k--) { double wk = y[k] / f->Ud[k]; y[k] = wk; for (int p = f->Up[k]; p < f->Up[k + 1]; p++) y[f->Ui[p]] -= f->Ux[p] * wk; }
  for (int k = 0; k < n; k++) b[f->q[k]] = y[k];
  for (int k = 0; k < n; k++) y[k] = 0.0;
}

/* klu!() / SUNLinSol_KLU setup after the first call = klu_refactor: numeric refactorisation with the stored pivot
 * sequence (reference src/model_evaluation.jl:417-428; its rcond-triggered re-pivoting is commented out there).  A full
 * factorisation with fresh pivoting is done the first time and whenever the refactorisation breaks down. */
static int splu_setup(splu* f, const int* cp, const int* ri, const double* ax) {
  int rc = splu_refactor(f, cp, ri, ax);
  if (rc != 0) { f->factored = 0; rc = splu_factor(f, cp, ri, ax); }
  return rc;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* public structs                                                                                               */
/* ------------------------------------------------------------------------------------------------------------ */
enum { ORC_MODE_I = 0, ORC_MODE_V = 1, ORC_MODE_DT = 2, ORC_MODE_P = 3, ORC_MODE_ETAP = 4,
       ORC_MODE_RES = 5,   /* method_res (input_methods.jl:155-175): res[end] = theta[:_residual_val] - run.func(t, Y, YP, p) (run_residual, scalar_residual.jl:172), a closure with derivative programs */
       ORC_MODE_DSTATE = 6, /* x - YP[ind] = 0 with ind the extreme surface / electrolyte concentration at the start of the run (dc_s_p_max ... dc_e_min, input_methods.jl:190-247:
                               state_deriv_func(ind) as a run_residual); orc_run.dstate = 1..6 in that order */
       ORC_NMODES = 7 };
enum { ORC_VAL_CONST = 0, ORC_VAL_HOLD = 1, ORC_VAL_REST = 2, ORC_VAL_TABLE = 3, ORC_VAL_EXPR = 4 };
/* ORC_VAL_EXPR: the input closure run.func(t, Y, YP, p) (scalar_residual.jl:169-170) as a postfix program, instruction k = (opcode tab_t[k], operand tab_v[k]); the opcode
   numbering is the C ABI's (include/petlion_hip.h PLH_OP_*), restated here: */
enum { OP_CONST = 0, OP_T, OP_Y, OP_YP, OP_THETA, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_SIN, OP_COS, OP_EXP, OP_LOG, OP_SQRT, OP_POW, OP_ABS, OP_MIN, OP_MAX,
       OP_LT, OP_LE, OP_GT, OP_GE, OP_SELECT, OP_TANH };

typedef struct {   /* reference boundary_stop_conditions, src/structures.jl:237-250 ; NaN disables a bound */
  double V_max, V_min, SOC_max, SOC_min, T_max, c_s_n_max, I_max, I_min, eta_plating_min, c_e_min, dfilm_max;
} orc_bounds;

typedef struct {   /* one run of a protocol = one simulate()/simulate!() call */
  int mode;        /* ORC_MODE_* */
  int value_kind;  /* ORC_VAL_* */
  double value;    /* C-rate / V / K/s */
  double tf;       /* run length in local time (reference default 1e6) */
  orc_bounds bounds;
  /* ORC_VAL_TABLE: the input is a function of the run-local time (reference run_function, structures.jl), given as a piecewise-linear
     table; a repeated knot time is a jump (right-continuous); beyond the last knot the last value holds */
  int n_tab; const double* tab_t; const double* tab_v;
  /* ORC_VAL_EXPR of the state: the symbolic derivative of the closure, d f / d Y[dcol[k]] = instructions [dofs[k], dofs[k+1]) of the same arrays -- what the reference's
     differentiate_residual_func (scalar_residual.jl:276-416) compiles into J_scalar_func; n_dcol = 0: _get_method_funcs_no_differentiation (:248-274) */
  int n_dcol; const int* dcol; const int* dofs;
  int dstate;      /* ORC_MODE_DSTATE: which state (1 c_s_p max, 2 c_s_p min, 3 c_s_n max, 4 c_s_n min, 5 c_e max, 6 c_e min) */
} orc_run;

typedef struct {   /* reference options_simulation, src/structures.jl:266-285 */
  double abstol, reltol, abstol_init, reltol_init;
  int maxiters;
  int check_bounds, interp_final;
  int max_order;      /* 5 */
  int jac_every_step; /* 0 = IDA policy (default); 1 = refresh each step (ablation) */
  double init_step;   /* 0 = automatic; > 0 = IDASetInitStep */
  int n_tdiscon; const double* tdiscon;   /* opts.tdiscon: known discontinuities of a function input (structures.jl:279) */
  int refine;         /* 0 = plain LU solves (default); n > 0 = n steps of iterative refinement of every linear solve against the matrix that
                         was factored (parity mode: makes the solve independent of the elimination order to ~1e-13, see tools/solve_accuracy.py) */
  /* reproducibility probe (tests only): the algebraic residual of the finite-difference estimate of the algebraic derivatives
     (model_evaluation.jl:462-477) gets an evaluation-rounding-sized perturbation, res_i += fd_perturb * u_i * sum_c |J_ic Y_c|, u_i in [-1, 1) from
     splitmix64(perturb_seed, i).  With fd_perturb = 2.2e-16 this is what two correct fp64 evaluations of the same row differ by (another summation
     order, products formed before or after a difference); the spread of the results over a few seeds is the reproducibility floor any second fp64
     implementation of the reference algorithm sits in. */
  double fd_perturb; int perturb_seed;
  int n_tstops; const double* tstops;     /* opts.tstops: user stop times in run-local time, appended to the integrator's tstops (model_evaluation.jl:292-294) */
  /* bounded experiment on the step history of the reference's notebook (tests/test_oracle_golden.py::test_step_history_of_the_2C_charge_notebook): 1 = the integrator is
     started with YP_alg = 0, i.e. without newtons_method!'s finite-difference estimate of the algebraic derivatives (model_evaluation.jl:462-477) */
  int exp_yp_alg_zero;
  /* reproducibility probe (tests only), the integrator's counterpart of fd_perturb: EVERY residual evaluation of the corrector gets the same evaluation-rounding-sized
     perturbation, res_i += res_perturb * u * sum_c |J_ic Y_c| with the entries of the last evaluated Newton matrix and a fresh u in [-1, 1) per row and evaluation
     (splitmix64 of perturb_seed, an evaluation counter and the row).  A second correct fp64 implementation differs from this one in exactly that way in every evaluation
     (flux form or matrix form of a stencil, the order of a sum), not only in the one evaluation fd_perturb touches. */
  double res_perturb;
  /* opts.stop_function (src/structures.jl:283; called after the built-in checks at every accepted step, src/checks.jl:26) as a postfix program g(t, Y, YP, theta) -- the
     form in which the product's C ABI takes it (plh_opts.stop_ops): the run ends when g > 0, flag 12, interpolation fraction g_prev / (g_prev - g); n_stop = 0: none */
  int n_stop; const double* stop_ops; const double* stop_args;
} orc_opts;

typedef struct {
  int flag;            /* reference exit flags 0..11 (src/checks.jl); negative = error paths */
  int iterations;      /* run.info.iterations */
  double t_end;        /* global time at the end of the run */
  double V, I, SOC, T_avg;
} orc_runinfo;

typedef struct {
  long n_steps, n_res, n_jac, n_fact, n_solve, n_newton, n_errfail, n_convfail, sum_kp2, n_init_iters;
} orc_counters;

#define ORC_ERR_INIT (-1)      /* "Could not initialize DAE", model_evaluation.jl:456 */
#define ORC_ERR_STALL (-2)     /* "Model failed to converge at t = ...", checks.jl:233,236 */
#define ORC_ERR_MAXITERS (-3)  /* "Reached max iterations", checks.jl:239 */
#define ORC_ERR_LINSOL (-4)

/* ------------------------------------------------------------------------------------------------------------ */
/* evaluator bundle: R_full/R_alg/R_diff/J_full/J_alg  (reference scalar_residual.jl:435-487, 558-602)          */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct {
  orc_model m;
  const double* th;
  int mode; double value;
  /* full Jacobian CSC (N x N) = base rows + control row */
  int *cp, *ri; double* ax; int nnz; int* base_map; int n_ctrl; int ctrl_pos[128]; int ctrl_col[128];
  /* algebraic Jacobian CSC (N_alg x N_alg) */
  int *acp, *ari; double* aax; int annz; int* abase_map; int an_ctrl; int actrl_pos[128]; int actrl_col[128];
  /* closure with derivative programs (drun->n_dcol > 0): the control row = the input method's own entries (the first n_base / an_base columns above) minus d f / d Y, whose
     columns not already in the row follow them (scalar_residual.jl:300-303: J_sp_scalar[J_vec.nzind] .= 1); dpos[k] / adpos[k] = position of column dcol[k] (-1: not in the block) */
  const orc_run* drun; int n_base, an_base; int dpos[64], adpos[64];
  int dind;                /* ORC_MODE_DSTATE: the state whose derivative is held (-1 otherwise); its twin row takes the algebraic entries of row dind of the base Jacobian */
  int n_tw; int tw_src[64]; /* positions of those entries in the base CSC values (the control-row entries actrl_pos[0..n_tw) of the algebraic block) */
  double* tmp_diff;        /* N_diff work vector (f_diff for the twin) */
  /* closure columns dcol[k] = N + i: d f / d YP[i] of a differential state i.  Integration row: times cj at column i.  Algebraic block (consistent initialisation): the closure is
     evaluated with YP -> rhs(Y) and the column chains through row i of dF/dY (scalar_residual.jl:335-362): records (program k, source position in the base CSC values, control entry) */
  int has_yp, n_yp; int yp_k[512], yp_src[512], yp_dst[512]; double* yp_sub;
  double *tmp_nz, *w;
  double *ax_f, *aax_f, *rtmp, *xtmp;   /* the matrices as last factored (iterative refinement), work vectors */
  int ax_valid;                          /* ax holds the values of an evaluated Newton matrix (perturb_residual) */
  double I1C;
  const orc_run* frun;   /* != NULL: the control value is frun's table / closure evaluated at the current time */
  double t_fun;          /* run-local time of the residual evaluations (closure inputs) */
  splu lu, alu;
  orc_counters* cnt;
} evalb;

static int ctrl_columns(const orc_model* m, int mode, int* cols, int alg_only) {
  int n = 0;
  if (mode == ORC_MODE_I) cols[n++] = m->o_I;
  else if (mode == ORC_MODE_V) { cols[n++] = m->o_ps; cols[n++] = m->o_ps + m->Np + m->Nn - 1; }
  else if (mode == ORC_MODE_P) { cols[n++] = m->o_ps; cols[n++] = m->o_ps + m->Np + m->Nn - 1; cols[n++] = m->o_I; }   /* scalar_residual.jl:212-218 */
  else if (mode == ORC_MODE_ETAP) { cols[n++] = m->o_pe + m->Np + m->Ns; cols[n++] = m->o_ps + m->Np; }              /* scalar_residual.jl:219-225 */
  else if (mode == ORC_MODE_DT) {
    if (!alg_only) { int nt = m->Na + m->Np + m->Ns + m->Nn + m->Nz; for (int i = 0; i < nt; i++) cols[n++] = m->o_T + i; }
    else { for (int i = 0; i < m->nnz_twin; i++) cols[n++] = m->Nd + m->twin_cols[i]; }
  }
  return n;
}

static void build_pattern(int N, int rows_base, const int* bcp, const int* bri, int col0, int nctrl, const int* ccols,
                          int** cp_o, int** ri_o, int* nnz_o, int** bmap_o, int* cpos) {
  /* columns col0..col0+N-1 of the base pattern (row offset already removed) + a last row */
  int bn = bcp[N];
  int* cp = (int*)malloc((N + 1) * sizeof(int)); int* ri = (int*)malloc((bn + nctrl) * sizeof(int)); int* bmap = (int*)malloc((bn > 0 ? bn : 1) * sizeof(int));
  int nz = 0; cp[0] = 0;
  for (int c = 0; c < N; c++) {
    for (int p = bcp[c]; p < bcp[c + 1]; p++) { ri[nz] = bri[p]; bmap[p] = nz; nz++; }
    for (int k = 0; k < nctrl; k++) if (ccols[k] - col0 == c) { ri[nz] = rows_base; cpos[k] = nz; nz++; }
    cp[c + 1] = nz;
  }
  *cp_o = cp; *ri_o = ri; *nnz_o = nz; *bmap_o = bmap;
}

static int find_key(const orc_model* m, const char* k);
/* value of a tabulated input at run-local time t (scalar_residual.jl:169-170: method(Y,p) - run.func(t,Y,YP,p)) */
static double tab_eval(const orc_run* r, double t) {
  const int n = r->n_tab; const double* tt = r->tab_t; const double* vv = r->tab_v;
  if (n <= 0) return 0.0;
  if (t < tt[0]) return vv[0];
  int k = 0; for (int q = 1; q < n; q++) if (tt[q] <= t) k = q;      /* last knot with t_k <= t */
  if (k == n - 1) return vv[n - 1];
  const double dt = tt[k + 1] - tt[k];
  return dt > 0.0 ? vv[k] + (vv[k + 1] - vv[k]) * ((t - tt[k]) / dt) : vv[k + 1];
}
static double prog_eval_range(const double* ops, const double* args, double t, const double* Y, const double* YP, const double* th, int k0, int k1);
static double expr_eval_range(const orc_run* r, double t, const double* Y, const double* YP, const double* th, int k0, int k1) {
  return prog_eval_range(r->tab_t, r->tab_v, t, Y, YP, th, k0, k1);
}
static double prog_eval_range(const double* ops, const double* args, double t, const double* Y, const double* YP, const double* th, int k0, int k1) {
  double st[64]; int sp = 0;
  for (int k = k0; k < k1; k++) {
    const int op = (int)ops[k]; const double a = args[k];
    switch (op) {
      case OP_CONST: st[sp++] = a; break; case OP_T: st[sp++] = t; break; case OP_Y: st[sp++] = Y[(int)a]; break; case OP_YP: st[sp++] = YP[(int)a]; break;
      case OP_THETA: st[sp++] = th[(int)a]; break;
      case OP_NEG: st[sp - 1] = -st[sp - 1]; break; case OP_SIN: st[sp - 1] = sin(st[sp - 1]); break; case OP_COS: st[sp - 1] = cos(st[sp - 1]); break;
      case OP_EXP: st[sp - 1] = exp(st[sp - 1]); break; case OP_LOG: st[sp - 1] = log(st[sp - 1]); break; case OP_SQRT: st[sp - 1] = sqrt(st[sp - 1]); break;
      case OP_ABS: st[sp - 1] = fabs(st[sp - 1]); break; case OP_TANH: st[sp - 1] = tanh(st[sp - 1]); break;
      case OP_SELECT: { const double b = st[sp - 1], x = st[sp - 2], c = st[sp - 3]; sp -= 2; st[sp - 1] = c != 0.0 ? x : b; break; }
      default: { const double y = st[sp - 1], x = st[sp - 2]; sp--; double v = 0.0;
        switch (op) { case OP_ADD: v = x + y; break; case OP_SUB: v = x - y; break; case OP_MUL: v = x * y; break; case OP_DIV: v = x / y; break; case OP_POW: v = pow(x, y); break;
                      case OP_MIN: v = x < y ? x : y; break; case OP_MAX: v = x > y ? x : y; break; case OP_LT: v = x < y; break; case OP_LE: v = x <= y; break;
                      case OP_GT: v = x > y; break; case OP_GE: v = x >= y; break; }
        st[sp - 1] = v; }
    }
  }
  return st[0];
}
static double expr_eval(const orc_run* r, double t, const double* Y, const double* YP, const double* th) { return expr_eval_range(r, t, Y, YP, th, 0, r->n_tab); }
static double run_input(const orc_run* r, double t, const double* Y, const double* YP, const double* th) {
  return r->value_kind == ORC_VAL_EXPR ? expr_eval(r, t, Y, YP, th) : tab_eval(r, t);
}
/* calc_I1C, reference auxiliary_states_and_coefficients.jl:632-647 */
static double calc_I1C_c(const orc_model* m, const double* th) {
  const char* nm[10] = {"ϵ_fp", "ϵ_p", "ϵ_fn", "ϵ_n", "l_p", "l_n", "c_max_p", "c_max_n", "θ_min_p", "θ_max_p"};
  double v[12];
  for (int i = 0; i < 10; i++) { int k = find_key(m, nm[i]); v[i] = k >= 0 ? th[k] : NAN; }
  { int k = find_key(m, "θ_max_n"); v[10] = k >= 0 ? th[k] : NAN; k = find_key(m, "θ_min_n"); v[11] = k >= 0 ? th[k] : NAN; }
  const double eps_sp = 1.0 - (v[0] + v[1]), eps_sn = 1.0 - (v[2] + v[3]);
  const double a = eps_sp * v[4] * v[6] * (v[8] - v[9]), b = eps_sn * v[5] * v[7] * (v[10] - v[11]);
  return (96485.3321233 / 3600.0) * (a < b ? a : b);
}

static int evalb_init_x(evalb* e, const orc_model* m, const double* th, int mode, double value, orc_counters* cnt, const orc_run* drun, int dind) {
  memset(e, 0, sizeof(*e));
  e->m = *m; e->th = th; e->mode = mode; e->value = value; e->cnt = cnt; e->drun = drun; e->dind = dind;
  int N = m->N, Nd = m->Nd, Na = N - Nd;
  if (mode == ORC_MODE_DT && !m->thermal) return -1;
  e->n_ctrl = e->n_base = ctrl_columns(m, mode, e->ctrl_col, 0);
  e->an_ctrl = e->an_base = ctrl_columns(m, mode, e->actrl_col, 1);
  if (mode == ORC_MODE_DSTATE) {          /* integration row: -cj at column dind; twin: row dind of dF/dY in the algebraic columns (scalar_residual.jl:335-362) */
    if (dind < 0 || dind >= Nd) return -1;
    e->ctrl_col[0] = dind; e->n_ctrl = e->n_base = 1;
    e->n_tw = 0;
    for (int c = Nd; c < N; c++) for (int q = m->colptr[c]; q < m->colptr[c + 1]; q++) if (m->rowval[q] == dind && e->n_tw < 64) { e->actrl_col[e->n_tw] = c; e->tw_src[e->n_tw] = q; e->n_tw++; }
    e->an_ctrl = e->an_base = e->n_tw;
  }
  if (drun && mode != ORC_MODE_DSTATE) {
    if (drun->n_dcol > 64 || mode == ORC_MODE_DT) return -1;
    for (int k = 0; k < drun->n_dcol; k++) {
      int c = drun->dcol[k]; int q;
      if (c >= N) {                       /* d f / d YP[i] */
        const int i = c - N;
        if (i >= Nd) return -1;
        e->has_yp = 1;
        for (q = 0; q < e->n_ctrl; q++) if (e->ctrl_col[q] == i) break;
        if (q == e->n_ctrl) e->ctrl_col[e->n_ctrl++] = i;
        e->dpos[k] = q; e->adpos[k] = -1;
        for (int c2 = Nd; c2 < N; c2++) for (int src = m->colptr[c2]; src < m->colptr[c2 + 1]; src++) if (m->rowval[src] == i && e->n_yp < 512) {
          int d; for (d = 0; d < e->an_ctrl; d++) if (e->actrl_col[d] == c2) break;
          if (d == e->an_ctrl) e->actrl_col[e->an_ctrl++] = c2;
          e->yp_k[e->n_yp] = k; e->yp_src[e->n_yp] = src; e->yp_dst[e->n_yp] = d; e->n_yp++;
        }
        continue;
      }
      for (q = 0; q < e->n_ctrl; q++) if (e->ctrl_col[q] == c) break;
      if (q == e->n_ctrl) e->ctrl_col[e->n_ctrl++] = c;
      e->dpos[k] = q; e->adpos[k] = -1;
      if (c >= Nd) { for (q = 0; q < e->an_ctrl; q++) if (e->actrl_col[q] == c) break; if (q == e->an_ctrl) e->actrl_col[e->an_ctrl++] = c; e->adpos[k] = q; }
    }
  }
  build_pattern(N, N - 1, m->colptr, m->rowval, 0, e->n_ctrl, e->ctrl_col, &e->cp, &e->ri, &e->nnz, &e->base_map, e->ctrl_pos);
  build_pattern(Na, Na - 1, m->acolptr, m->arowval, Nd, e->an_ctrl, e->actrl_col, &e->acp, &e->ari, &e->annz, &e->abase_map, e->actrl_pos);
  e->ax = (double*)calloc(e->nnz, sizeof(double)); e->aax = (double*)calloc(e->annz, sizeof(double));
  e->tmp_nz = (double*)calloc(m->nnz + 64, sizeof(double)); e->w = (double*)calloc(N, sizeof(double)); e->tmp_diff = (double*)calloc(Nd > 0 ? Nd : 1, sizeof(double));
  e->yp_sub = (double*)calloc(N, sizeof(double));
  e->ax_f = (double*)calloc(e->nnz, sizeof(double)); e->aax_f = (double*)calloc(e->annz, sizeof(double)); e->rtmp = (double*)calloc(N, sizeof(double)); e->xtmp = (double*)calloc(N, sizeof(double));
  if (m->thermal) m->dT_weights(e->w, th);
  e->I1C = calc_I1C_c(m, th);
  splu_init(&e->lu, N, e->cp, e->ri); splu_init(&e->alu, Na, e->acp, e->ari);
  return 0;
}
static int evalb_init_d(evalb* e, const orc_model* m, const double* th, int mode, double value, orc_counters* cnt, const orc_run* drun) { return evalb_init_x(e, m, th, mode, value, cnt, drun, -1); }
static int evalb_init(evalb* e, const orc_model* m, const double* th, int mode, double value, orc_counters* cnt) { return evalb_init_d(e, m, th, mode, value, cnt, NULL); }
/* minus the closure's derivative programs into the control row (J_scalar_func of differentiate_residual_func); alg: the columns of the algebraic block only
   (J_vec[N.diff+1:end], scalar_residual.jl:369-371) */
static void ctrl_row_derivatives(const evalb* e, const double* Y, const double* YP, double cj, double* ax, const int* cpos, int n_base, int n_ctrl, int alg) {
  const orc_run* r = e->drun;
  if (!r) return;
  const int N = e->m.N;
  const double* YPe = YP;
  if (alg && e->has_yp) {                  /* the closure with YP -> rhs(Y); row i of dF/dY at cj = 0 for the chain rule */
    e->m.f_diff(e->tmp_diff, Y, YP, e->th);
    for (int i = 0; i < N; i++) e->yp_sub[i] = i < e->m.Nd ? e->tmp_diff[i] + YP[i] : YP[i];
    YPe = e->yp_sub;
    e->m.jac(e->tmp_nz, Y, YP, 0.0, e->th);
  }
  for (int q = n_base; q < n_ctrl; q++) ax[cpos[q]] = 0.0;
  for (int k = 0; k < r->n_dcol; k++) {
    const double a = expr_eval_range(r, e->t_fun, Y, YPe, e->th, r->dofs[k], r->dofs[k + 1]);
    if (r->dcol[k] >= N) {
      if (!alg) ax[cpos[e->dpos[k]]] -= cj * a;
      else for (int w = 0; w < e->n_yp; w++) if (e->yp_k[w] == k) ax[cpos[e->yp_dst[w]]] -= a * e->tmp_nz[e->yp_src[w]];
      continue;
    }
    const int q = alg ? e->adpos[k] : e->dpos[k];
    if (q >= 0) ax[cpos[q]] -= a;
  }
}
static void evalb_free(evalb* e) {
  free(e->cp); free(e->ri); free(e->ax); free(e->base_map); free(e->acp); free(e->ari); free(e->aax); free(e->abase_map);
  free(e->tmp_nz); free(e->w); free(e->tmp_diff); free(e->yp_sub); free(e->ax_f); free(e->aax_f); free(e->rtmp); free(e->xtmp); splu_free(&e->lu); splu_free(&e->alu);
}

static double ctrl_residual(evalb* e, const double* Y, const double* YP) {
  /* closure input: val = run.func(t, Y, YP, p) with the iterate, in every residual, and run.value[] = val (scalar_residual.jl:169-170) */
  if (e->frun && e->frun->value_kind == ORC_VAL_EXPR) e->value = expr_eval(e->frun, e->t_fun, Y, YP, e->th);
  const orc_model* m = &e->m;
  if (e->mode == ORC_MODE_RES) return e->frun->value - e->value;                                  /* run_residual: _residual_val - f */
  if (e->mode == ORC_MODE_DSTATE) return e->value - YP[e->dind];                                  /* state_deriv_func(ind): _residual_val - YP[ind] */
  if (e->mode == ORC_MODE_I) return Y[m->o_I] - e->value;                                         /* method_I */
  if (e->mode == ORC_MODE_V) return Y[m->o_ps] - Y[m->o_ps + m->Np + m->Nn - 1] - e->value;       /* method_V */
  if (e->mode == ORC_MODE_P) return Y[m->o_I] * e->I1C * (Y[m->o_ps] - Y[m->o_ps + m->Np + m->Nn - 1]) - e->value;   /* method_P = calc_P, scalar_residual.jl:87 */
  if (e->mode == ORC_MODE_ETAP) return Y[m->o_ps + m->Np] - Y[m->o_pe + m->Np + m->Ns] - e->value;                  /* method_η_p = calc_η_plating, :92 */
  double s = 0.0; int nt = m->Na + m->Np + m->Ns + m->Nn + m->Nz;                                 /* dT */
  for (int i = 0; i < nt; i++) s += e->w[i] * YP[m->o_T + i];
  return e->value - s;
}
/* R_full (scalar_residual.jl:558-583) */
static void R_full(evalb* e, double* res, const double* Y, const double* YP) {
  const orc_model* m = &e->m;
  m->f_diff(res, Y, YP, e->th); m->f_alg(res + m->Nd, Y, YP, e->th);
  res[m->N - 1] = ctrl_residual(e, Y, YP);
  if (e->cnt) e->cnt->n_res++;
}
/* R_alg: algebraic rows + (twin of) the control row; YP is zero in the init Newton */
static void R_alg(evalb* e, double* res /*N_alg*/, const double* Y, const double* YP) {
  const orc_model* m = &e->m; int Na = m->N - m->Nd;
  m->f_alg(res, Y, YP, e->th);
  if (e->mode == ORC_MODE_DT) { double tw; m->dT_twin(&tw, Y, YP, e->th); res[Na - 1] = e->value + tw; }
  else if (e->mode == ORC_MODE_DSTATE) { m->f_diff(e->tmp_diff, Y, YP, e->th); res[Na - 1] = e->value - (e->tmp_diff[e->dind] + YP[e->dind]); }   /* YP[ind] -> rhs_ind(Y) = F_ind + YP[ind] */
  else if (e->drun && e->has_yp) {          /* a closure of YP: YP -> rhs(Y) = F_diff(Y, YP) + YP in the consistent-initialisation row (scalar_residual.jl:335-362) */
    m->f_diff(e->tmp_diff, Y, YP, e->th);
    for (int i = 0; i < m->N; i++) e->yp_sub[i] = i < m->Nd ? e->tmp_diff[i] + YP[i] : YP[i];
    res[Na - 1] = ctrl_residual(e, Y, e->yp_sub);
  }
  else res[Na - 1] = ctrl_residual(e, Y, YP);
  if (e->cnt) e->cnt->n_res++;
}
/* scalar_jacobian! for the P and η_p rows (scalar_residual.jl:189-202); pos = positions of the ctrl_columns() entries */
static void ctrl_row_P_etap(const evalb* e, const double* Y, double* ax, const int* pos) {
  const orc_model* m = &e->m;
  if (e->mode == ORC_MODE_P) {
    const double I = Y[m->o_I] * e->I1C, V = Y[m->o_ps] - Y[m->o_ps + m->Np + m->Nn - 1];
    ax[pos[0]] = I; ax[pos[1]] = -I; ax[pos[2]] = V * e->I1C;
  } else { ax[pos[0]] = -1.0; ax[pos[1]] = 1.0; }
}
/* J_full (scalar_residual.jl:588-602, 174-202) */
static void J_full(evalb* e, const double* Y, const double* YP, double cj) {
  const orc_model* m = &e->m;
  m->jac(e->tmp_nz, Y, YP, cj, e->th);
  for (int p = 0; p < m->nnz; p++) e->ax[e->base_map[p]] = e->tmp_nz[p];
  if (e->mode == ORC_MODE_I) e->ax[e->ctrl_pos[0]] = 1.0;
  else if (e->mode == ORC_MODE_V) { e->ax[e->ctrl_pos[0]] = 1.0; e->ax[e->ctrl_pos[1]] = -1.0; }
  else if (e->mode == ORC_MODE_P || e->mode == ORC_MODE_ETAP) ctrl_row_P_etap(e, Y, e->ax, e->ctrl_pos);
  else if (e->mode == ORC_MODE_DT) for (int k = 0; k < e->n_ctrl; k++) e->ax[e->ctrl_pos[k]] = -cj * e->w[k];
  else if (e->mode == ORC_MODE_DSTATE) e->ax[e->ctrl_pos[0]] = -cj;
  ctrl_row_derivatives(e, Y, YP, cj, e->ax, e->ctrl_pos, e->n_base, e->n_ctrl, 0);
  if (e->cnt) e->cnt->n_jac++;
  e->ax_valid = 1;
}
static void J_alg(evalb* e, const double* Y, const double* YP) {
  const orc_model* m = &e->m;
  m->jac_alg(e->tmp_nz, Y, YP, 0.0, e->th);
  for (int p = 0; p < m->nnz_alg; p++) e->aax[e->abase_map[p]] = e->tmp_nz[p];
  if (e->mode == ORC_MODE_I) e->aax[e->actrl_pos[0]] = 1.0;
  else if (e->mode == ORC_MODE_V) { e->aax[e->actrl_pos[0]] = 1.0; e->aax[e->actrl_pos[1]] = -1.0; }
  else if (e->mode == ORC_MODE_P || e->mode == ORC_MODE_ETAP) ctrl_row_P_etap(e, Y, e->aax, e->actrl_pos);
  else if (e->mode == ORC_MODE_DT) { m->dT_twin_jac(e->tmp_nz, Y, YP, 0.0, e->th); for (int k = 0; k < e->an_ctrl; k++) e->aax[e->actrl_pos[k]] = e->tmp_nz[k]; }
  else if (e->mode == ORC_MODE_DSTATE) { m->jac(e->tmp_nz, Y, YP, 0.0, e->th); for (int k = 0; k < e->n_tw; k++) e->aax[e->actrl_pos[k]] = -e->tmp_nz[e->tw_src[k]]; }
  ctrl_row_derivatives(e, Y, YP, 0.0, e->aax, e->actrl_pos, e->an_base, e->an_ctrl, 1);
  if (e->cnt) e->cnt->n_jac++;
}


/* factor + remember the matrix; solve with `nref` steps of iterative refinement  x += A^-1 (b - A x)  against that matrix */
static int lu_setup_keep(splu* f, int n, const int* cp, const int* ri, const double* ax, double* ax_keep) {
  memcpy(ax_keep, ax, (size_t)cp[n] * sizeof(double));
  return splu_setup(f, cp, ri, ax);
}
static void lu_solve_refined(const splu* f, int n, const int* cp, const int* ri, const double* ax_keep, double* b, double* r, double* x0, int nref) {
  if (nref <= 0) { splu_solve(f, b); return; }
  memcpy(r, b, n * sizeof(double));                 /* r = b (kept) */
  splu_solve(f, b);                                 /* b = x */
  for (int it = 0; it < nref; it++) {
    memcpy(x0, b, n * sizeof(double));
    /* b <- r - A x  (CSC) */
    for (int i = 0; i < n; i++) b[i] = r[i];
    for (int c = 0; c < n; c++) { const double xc = x0[c]; for (int p = cp[c]; p < cp[c + 1]; p++) b[ri[p]] -= ax_keep[p] * xc; }
    splu_solve(f, b);
    for (int i = 0; i < n; i++) b[i] += x0[i];
  }
}

/* ------------------------------------------------------------------------------------------------------------ */
/* consistent initialisation: newtons_method!  (reference src/model_evaluation.jl:430-480)                      */
/* ------------------------------------------------------------------------------------------------------------ */
static int newtons_method(evalb* e, double* Y, double* YP, const orc_opts* o, double c_e0) {
  const orc_model* m = &e->m; int N = m->N, Nd = m->Nd, Na = N - Nd;
  double* res = (double*)calloc(N, sizeof(double)); double* Ynew = (double*)calloc(N, sizeof(double));
  for (int i = 0; i < N; i++) YP[i] = 0.0;
  int ok = 0;
  for (int iter = 1; iter <= 100; iter++) {
    R_alg(e, res, Y, YP); J_alg(e, Y, YP);
    if (lu_setup_keep(&e->alu, Na, e->acp, e->ari, e->aax, e->aax_f) != 0) { free(res); free(Ynew); return ORC_ERR_LINSOL; }
    if (e->cnt) { e->cnt->n_fact++; e->cnt->n_solve++; e->cnt->n_init_iters++; }
    lu_solve_refined(&e->alu, Na, e->acp, e->ari, e->aax_f, res, e->rtmp, e->xtmp, o->refine);
    double nrm = 0.0;
    for (int i = 0; i < Na; i++) { Y[Nd + i] -= res[i]; nrm += res[i] * res[i]; }
    if (sqrt(nrm) < o->reltol_init) { ok = 1; break; }
  }
  if (!ok) { free(res); free(Ynew); return ORC_ERR_INIT; }
  /* YP_diff = rhs (R_diff with YP = 0) */
  m->f_diff(YP, Y, YP, e->th);
  if (e->cnt) e->cnt->n_res++;
  /* finite-difference estimate of YP_alg (model_evaluation.jl:462-477) */
  double dt = fmax(10.0 * o->reltol_init, sqrt(nextafter(c_e0, INFINITY) - c_e0));
  for (int i = 0; i < N; i++) Ynew[i] = Y[i] + dt * YP[i];
  R_alg(e, res, Ynew, YP);
  if (o->fd_perturb != 0.0) {
    /* reproducibility probe: res_i += fd_perturb * u_i * sum_c |J_ic Y_c| -- an evaluation-rounding-sized perturbation of every algebraic row (each
       row is a sum of terms of that magnitude; two correct fp64 evaluations of it differ by a few ulps of the largest term) */
    const int Na_ = N - Nd;
    double* term = (double*)calloc(Na_, sizeof(double));
    for (int c = 0; c < Na_; c++) for (int q = e->acp[c]; q < e->acp[c + 1]; q++) term[e->ari[q]] += fabs(e->aax_f[q] * Ynew[Nd + c]);
    for (int i = 0; i < Na_; i++) {
      uint64_t z = ((uint64_t)o->perturb_seed * 0x9E3779B97F4A7C15ull) ^ (uint64_t)i; z += 0x9E3779B97F4A7C15ull;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
      res[i] += o->fd_perturb * term[i] * (2.0 * ((double)(z >> 11) / 9007199254740992.0) - 1.0);
    }
    free(term);
  }
  lu_solve_refined(&e->alu, Na, e->acp, e->ari, e->aax_f, res, e->rtmp, e->xtmp, o->refine);
  if (e->cnt) e->cnt->n_solve++;
  for (int i = 0; i < Na; i++) YP[Nd + i] = -res[i] / dt;
  free(res); free(Ynew);
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* IDA-style integrator                                                                                         */
/* ------------------------------------------------------------------------------------------------------------ */
#define MXORDP1 6
typedef struct {
  int N; evalb* e; const orc_opts* o;
  double *phi[MXORDP1], *ewt, *yy, *yp, *ee, *delta, *ypred, *yppred, *tmp;
  double psi[MXORDP1], alpha[MXORDP1], beta[MXORDP1], sigma[MXORDP1], gamma[MXORDP1];
  double tn, hh, hused, cj, cjlast, cjold, cjratio, ss, rr, epsNewt, toldel, tstop;
  int kk, kused, knew, phase, ns, maxord, tstopset;
  long nst;
  double uround;
  double h0_forced;   /* IDASetInitStep (src/checks.jl:231) */
} ida_t;

static double wrms(int N, const double* v, const double* w) { double s = 0.0; for (int i = 0; i < N; i++) { double p = v[i] * w[i]; s += p * p; } return sqrt(s / N); }
static void set_ewt(ida_t* I, const double* y) { for (int i = 0; i < I->N; i++) I->ewt[i] = 1.0 / (I->o->reltol * fabs(y[i]) + I->o->abstol); }

static void ida_alloc(ida_t* I, int N) {
  memset(I, 0, sizeof(*I)); I->N = N;
  for (int j = 0; j < MXORDP1; j++) I->phi[j] = (double*)calloc(N, sizeof(double));
  I->ewt = (double*)calloc(N, sizeof(double)); I->yy = (double*)calloc(N, sizeof(double)); I->yp = (double*)calloc(N, sizeof(double));
  I->ee = (double*)calloc(N, sizeof(double)); I->delta = (double*)calloc(N, sizeof(double)); I->ypred = (double*)calloc(N, sizeof(double));
  I->yppred = (double*)calloc(N, sizeof(double)); I->tmp = (double*)calloc(N, sizeof(double));
}
static void ida_free(ida_t* I) { for (int j = 0; j < MXORDP1; j++) free(I->phi[j]); free(I->ewt); free(I->yy); free(I->yp); free(I->ee); free(I->delta); free(I->ypred); free(I->yppred); free(I->tmp); }

/* IDAReInit(mem, 0.0, Y0, YP0) + IDASStolerances (reference src/model_evaluation.jl:247-251) */
static void ida_reinit_at(ida_t* I, evalb* e, const orc_opts* o, const double* y0, const double* yp0, double t_start);
static void ida_reinit(ida_t* I, evalb* e, const orc_opts* o, const double* y0, const double* yp0) { ida_reinit_at(I, e, o, y0, yp0, 0.0); }
static void ida_reinit_at(ida_t* I, evalb* e, const orc_opts* o, const double* y0, const double* yp0, double t_start) {
  I->e = e; I->o = o; I->tn = t_start; I->nst = 0; I->kk = 0; I->kused = 0; I->hused = 0.0; I->hh = 0.0;
  I->maxord = o->max_order > 0 ? o->max_order : 5; I->epsNewt = 0.33; I->toldel = 0.0001 * I->epsNewt; I->uround = 2.220446049250313e-16;
  I->cjratio = 1.0; I->ss = 20.0; I->tstopset = 0; I->phase = 0; I->ns = 0; I->h0_forced = 0.0;
  memcpy(I->phi[0], y0, I->N * sizeof(double)); memcpy(I->phi[1], yp0, I->N * sizeof(double));
  memcpy(I->yy, y0, I->N * sizeof(double)); memcpy(I->yp, yp0, I->N * sizeof(double));
  e->lu.factored = e->lu.factored;   /* the KLU symbolic/pivot data is kept across runs, like the cached integrator */
}

static double ida_set_coeffs(ida_t* I) {
  int kk = I->kk; double hh = I->hh;
  if (hh != I->hused || kk != I->kused) I->ns = 0;
  I->ns = (I->ns + 1 < I->kused + 2) ? I->ns + 1 : I->kused + 2;
  if (kk + 1 >= I->ns) {
    I->beta[0] = 1.0; I->alpha[0] = 1.0; double temp1 = hh; I->gamma[0] = 0.0; I->sigma[0] = 1.0;
    for (int i = 1; i <= kk; i++) {
      double temp2 = I->psi[i - 1]; I->psi[i - 1] = temp1; I->beta[i] = I->beta[i - 1] * I->psi[i - 1] / temp2; temp1 = temp2 + hh;
      I->alpha[i] = hh / temp1; I->sigma[i] = i * I->sigma[i - 1] * I->alpha[i]; I->gamma[i] = I->gamma[i - 1] + I->alpha[i - 1] / hh;
    }
    I->psi[kk] = temp1;
  }
  double alphas = 0.0, alpha0 = 0.0;
  for (int i = 0; i < kk; i++) { alphas -= 1.0 / (i + 1); alpha0 -= I->alpha[i]; }
  I->cjlast = I->cj; I->cj = -alphas / hh;
  double ck = fabs(I->alpha[kk] + alphas - alpha0); if (ck < I->alpha[kk]) ck = I->alpha[kk];
  for (int i = I->ns; i <= kk; i++) { double b = I->beta[i]; double* p = I->phi[i]; for (int n = 0; n < I->N; n++) p[n] *= b; }
  I->tn += hh;
  return ck;
}

/* nonlinear solve: IDANls + SUNNonlinSol_Newton + idaNlsConvTest + IDALs scaling.  returns 0 ok, >0 recoverable, <0 fatal */
static unsigned long long g_res_perturb_calls = 0;      /* (one trajectory per thread-local integrator; the counter only decorrelates the draws) */
static void perturb_residual(ida_t* I, double* res, const double* Y) {
  evalb* e = I->e; const int N = I->N; const double eps = I->o->res_perturb;
  double* term = (double*)calloc(N, sizeof(double));
  for (int c = 0; c < N; c++) for (int q = e->cp[c]; q < e->cp[c + 1]; q++) term[e->ri[q]] += fabs(e->ax[q] * Y[c]);
  const uint64_t call = (uint64_t)__atomic_add_fetch(&g_res_perturb_calls, 1ull, __ATOMIC_RELAXED);
  for (int i = 0; i < N; i++) {
    uint64_t z = ((uint64_t)I->o->perturb_seed * 0x9E3779B97F4A7C15ull) ^ (call * 0xD1B54A32D192ED03ull) ^ (uint64_t)i; z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; z ^= z >> 31;
    res[i] += eps * term[i] * (2.0 * ((double)(z >> 11) / 9007199254740992.0) - 1.0);
  }
  free(term);
}

static int ida_nls(ida_t* I) {
  evalb* e = I->e; int N = I->N; orc_counters* cnt = e->cnt;
  if (e->frun) { e->value = run_input(e->frun, I->tn, I->phi[0], I->phi[1], e->th); e->t_fun = I->tn; }     /* every residual of this step is evaluated at t = tn */
  int callLSetup = 0;
  if (I->nst == 0) { I->cjold = I->cj; I->ss = 20.0; callLSetup = 1; }
  else {
    I->cjratio = I->cj / I->cjold;
    const double temp1 = (1.0 - 0.25) / (1.0 + 0.25), temp2 = 1.0 / temp1;
    if (I->cjratio < temp1 || I->cjratio > temp2) callLSetup = 1;
    if (I->cj != I->cjlast) I->ss = 100.0;
    if (I->o->jac_every_step) callLSetup = 1;
  }
  /* predictor */
  for (int n = 0; n < N; n++) { I->ypred[n] = I->phi[0][n]; I->yppred[n] = 0.0; }
  for (int j = 1; j <= I->kk; j++) { double g = I->gamma[j]; const double* p = I->phi[j]; for (int n = 0; n < N; n++) { I->ypred[n] += p[n]; I->yppred[n] += g * p[n]; } }
  for (int n = 0; n < N; n++) I->ee[n] = 0.0;
  int jcur = 0, ret = 0;
  for (;;) {
    for (int n = 0; n < N; n++) { I->yy[n] = I->ypred[n] + I->ee[n]; I->yp[n] = I->yppred[n] + I->cj * I->ee[n]; }
    R_full(e, I->delta, I->yy, I->yp);
    if (callLSetup) {
      J_full(e, I->yy, I->yp, I->cj);
      if (lu_setup_keep(&e->lu, N, e->cp, e->ri, e->ax, e->ax_f) != 0) return 1;     /* treat as recoverable */
      if (cnt) cnt->n_fact++;
      I->cjold = I->cj; I->cjratio = 1.0; I->ss = 20.0; jcur = 1;
    }
    if (I->o->res_perturb != 0.0 && e->ax_valid) perturb_residual(I, I->delta, I->yy);
    int m = 0; double oldnrm = 0.0;
    for (;;) {
      if (cnt) { cnt->n_newton++; cnt->n_solve++; }
      for (int n = 0; n < N; n++) I->delta[n] = -I->delta[n];
      lu_solve_refined(&e->lu, N, e->cp, e->ri, e->ax_f, I->delta, e->rtmp, e->xtmp, I->o->refine);
      if (I->cjratio != 1.0) { double s = 2.0 / (1.0 + I->cjratio); for (int n = 0; n < N; n++) I->delta[n] *= s; }
      for (int n = 0; n < N; n++) I->ee[n] += I->delta[n];
      double delnrm = wrms(N, I->delta, I->ewt);
      ret = 2; /* continue */
      if (m == 0) { oldnrm = delnrm; if (delnrm <= I->toldel) ret = 0; }
      else { double rate = pow(delnrm / oldnrm, 1.0 / m); if (rate > 0.9) ret = 1; else I->ss = rate / (1.0 - rate); }
      if (ret == 2 && I->ss * delnrm <= I->epsNewt) ret = 0;
      if (delnrm != delnrm) ret = 1;
      if (ret == 0) { jcur = 0; break; }
      if (ret != 2) break;
      m++; if (m >= 4) { ret = 1; break; }
      for (int n = 0; n < N; n++) { I->yy[n] = I->ypred[n] + I->ee[n]; I->yp[n] = I->yppred[n] + I->cj * I->ee[n]; }
      R_full(e, I->delta, I->yy, I->yp);
      if (I->o->res_perturb != 0.0) perturb_residual(I, I->delta, I->yy);
    }
    if (ret > 0 && !jcur) { callLSetup = 1; for (int n = 0; n < N; n++) I->ee[n] = 0.0; continue; }
    break;
  }
  for (int n = 0; n < N; n++) { I->yy[n] = I->ypred[n] + I->ee[n]; I->yp[n] = I->yppred[n] + I->cj * I->ee[n]; }
  return ret;
}

static int ida_test_error(ida_t* I, double ck, double* err_k, double* err_km1) {
  int N = I->N, kk = I->kk;
  double enorm_k = wrms(N, I->ee, I->ewt);
  if (getenv("ORC_TRACE_EE") && I->nst < atoi(getenv("ORC_TRACE_EE"))) for (int n = 0; n < N; n++) fprintf(stderr, "orc ee %d %d %.6e %.6e\n", (int)I->nst + 1, n, I->ee[n], I->ee[n] * I->ewt[n]);
  *err_k = I->sigma[kk] * enorm_k; double terr_k = (kk + 1) * (*err_k);
  I->knew = kk; *err_km1 = 0.0;
  if (kk > 1) {
    for (int n = 0; n < N; n++) I->delta[n] = I->phi[kk][n] + I->ee[n];
    double enorm_km1 = wrms(N, I->delta, I->ewt); *err_km1 = I->sigma[kk - 1] * enorm_km1; double terr_km1 = kk * (*err_km1);
    if (kk > 2) {
      for (int n = 0; n < N; n++) I->delta[n] += I->phi[kk - 1][n];
      double enorm_km2 = wrms(N, I->delta, I->ewt); double err_km2 = I->sigma[kk - 2] * enorm_km2; double terr_km2 = (kk - 1) * err_km2;
      if (fmax(terr_km1, terr_km2) <= terr_k) I->knew = kk - 1;
    } else if (terr_km1 <= 0.5 * terr_k) I->knew = kk - 1;
  }
  return (ck * enorm_k > 1.0) ? 1 : 0;
}

static void ida_restore(ida_t* I, double saved_t) {
  I->tn = saved_t;
  for (int j = 1; j <= I->kk; j++) I->psi[j - 1] = I->psi[j] - I->hh;
  if (I->ns <= I->kk) for (int j = I->ns; j <= I->kk; j++) { double b = 1.0 / I->beta[j]; double* p = I->phi[j]; for (int n = 0; n < I->N; n++) p[n] *= b; }
}

static void ida_complete_step(ida_t* I, double err_k, double err_km1) {
  int N = I->N; I->nst++;
  int kdiff = I->kk - I->kused; I->kused = I->kk; I->hused = I->hh;
  if (I->knew == I->kk - 1 || I->kk == I->maxord) I->phase = 1;
  if (I->phase == 0) { if (I->nst > 1) { I->kk++; I->hh *= 2.0; } }
  else {
    int action = 0; /* 0 unset, 1 lower, 2 maintain, 3 raise */
    double err_kp1 = 0.0, err_knew;
    if (I->knew == I->kk - 1) action = 1;
    else if (I->kk == I->maxord) action = 2;
    else if (I->kk + 1 >= I->ns || kdiff == 1) action = 2;
    if (action == 0) {
      for (int n = 0; n < N; n++) I->tmp[n] = I->ee[n] - I->phi[I->kk + 1][n];
      double enorm = wrms(N, I->tmp, I->ewt); err_kp1 = enorm / (I->kk + 2);
      double terr_k = (I->kk + 1) * err_k, terr_kp1 = (I->kk + 2) * err_kp1;
      if (I->kk == 1) action = (terr_kp1 >= 0.5 * terr_k) ? 2 : 3;
      else { double terr_km1 = I->kk * err_km1;
        if (terr_km1 <= fmin(terr_k, terr_kp1)) action = 1; else if (terr_kp1 >= terr_k) action = 2; else action = 3; }
    }
    if (action == 3) { I->kk++; err_knew = err_kp1; } else if (action == 1) { I->kk--; err_knew = err_km1; } else err_knew = err_k;
    double hnew = I->hh; I->rr = pow(2.0 * err_knew + 0.0001, -1.0 / (I->kk + 1));
    if (I->rr >= 2.0) hnew = 2.0 * I->hh;
    else if (I->rr <= 1.0) { I->rr = fmax(0.5, fmin(0.9, I->rr)); hnew = I->hh * I->rr; }
    I->hh = hnew;
  }
  if (I->kused < I->maxord) memcpy(I->phi[I->kused + 1], I->ee, N * sizeof(double));
  for (int n = 0; n < N; n++) I->phi[I->kused][n] += I->ee[n];
  for (int j = I->kused - 1; j >= 0; j--) for (int n = 0; n < N; n++) I->phi[j][n] += I->phi[j + 1][n];
}

/* IDAGetSolution(t): interpolated y, y' */
static void ida_get_solution(const ida_t* I, double t, double* y, double* yp) {
  int N = I->N; int kord = I->kused; if (kord == 0) kord = 1;
  double delt = t - I->tn, c = 1.0, d = 0.0, gam = delt / I->psi[0];
  double cv[MXORDP1], dv[MXORDP1]; cv[0] = c;
  for (int j = 1; j <= kord; j++) { d = d * gam + c / I->psi[j - 1]; c = c * gam; gam = (delt + I->psi[j - 1]) / I->psi[j]; cv[j] = c; dv[j - 1] = d; }
  for (int n = 0; n < N; n++) { double s = 0.0, sp = 0.0; for (int j = 0; j <= kord; j++) s += cv[j] * I->phi[j][n]; for (int j = 1; j <= kord; j++) sp += dv[j - 1] * I->phi[j][n]; y[n] = s; yp[n] = sp; }
}

/* one IDASolve(..., IDA_ONE_STEP_TSTOP) call.  returns 0 ok (tret set), <0 failure.  t_prev_out mirrors int.tprev */
static int ida_step(ida_t* I, double tstop, double* tret, double* yret, double* ypret) {
  int N = I->N; orc_counters* cnt = I->e->cnt;
  I->tstop = tstop; I->tstopset = 1;
  if (I->nst == 0) {
    set_ewt(I, I->phi[0]);
    double tdist = fabs(tstop - I->tn);
    double hh = I->h0_forced != 0.0 ? I->h0_forced : I->o->init_step;
    if (hh == 0.0) {
      hh = 0.001 * tdist;
      double ypnorm = wrms(N, I->phi[1], I->ewt);
      if (ypnorm > 0.5 / hh) hh = 0.5 / ypnorm;
    }
    if ((I->tn + hh - tstop) * hh > 0.0) hh = (tstop - I->tn) * (1.0 - 4.0 * I->uround);
    I->hh = hh; I->kk = 0; I->kused = 0;
    for (int n = 0; n < N; n++) I->phi[1][n] *= hh;
  } else {
    /* IDAStopTest1 for ONE_STEP_TSTOP */
    double troundoff = 100.0 * I->uround * (fabs(I->tn) + fabs(I->hh));
    if (fabs(I->tn - tstop) <= troundoff) { ida_get_solution(I, tstop, yret, ypret); *tret = tstop; return 0; }
    if ((I->tn + I->hh - tstop) * I->hh > 0.0) I->hh = (tstop - I->tn) * (1.0 - 4.0 * I->uround);
    set_ewt(I, I->phi[0]);
  }
  /* IDAStep */
  double saved_t = I->tn; int ncf = 0, nef = 0; double err_k = 0, err_km1 = 0;
  if (I->nst == 0) { I->kk = 1; I->kused = 0; I->hused = 0.0; I->psi[0] = I->hh; I->cj = 1.0 / I->hh; I->phase = 0; I->ns = 0; }
  for (;;) {
    double ck = ida_set_coeffs(I);
    int nflag = ida_nls(I);
    int errfail = 0;
    if (nflag == 0) { errfail = ida_test_error(I, ck, &err_k, &err_km1); }
    if (nflag != 0 || errfail) {
      ida_restore(I, saved_t);
      I->phase = 1;
      if (!errfail) {
        if (cnt) cnt->n_convfail++;
        if (nflag < 0) return ORC_ERR_STALL;
        I->rr = 0.25; I->hh *= I->rr; ncf++;
        if (ncf >= 10) return ORC_ERR_STALL;
      } else {
        if (cnt) cnt->n_errfail++;
        nef++;
        if (nef == 1) { double err_knew = (I->kk == I->knew) ? err_k : err_km1; I->kk = I->knew;
          I->rr = 0.9 * pow(2.0 * err_knew + 0.0001, -1.0 / (I->kk + 1)); I->rr = fmax(0.25, fmin(0.9, I->rr)); I->hh *= I->rr; }
        else if (nef == 2) { I->kk = I->knew; I->rr = 0.25; I->hh *= I->rr; }
        else if (nef < 10) { I->kk = 1; I->rr = 0.25; I->hh *= I->rr; }
        else return ORC_ERR_STALL;
      }
      if (fabs(I->hh) < 1e-14 * fmax(1.0, fabs(I->tn))) return ORC_ERR_STALL;
      if (I->nst == 0) { I->psi[0] = I->hh; for (int n = 0; n < N; n++) I->phi[1][n] *= I->rr; }
      continue;
    }
    break;
  }
  if (cnt) { cnt->n_steps++; cnt->sum_kp2 += I->kk + 2; }
  if (getenv("ORC_TRACE")) fprintf(stderr, "orc step %d tn %.9g h %.6g k %d knew %d phase %d ns %d err_k %.6e err_km1 %.6e nef %d ncf %d\n", I->nst + 1, I->tn, I->hh, I->kk, I->knew, I->phase, I->ns, err_k, err_km1, nef, ncf);
  ida_complete_step(I, err_k, err_km1);
  /* IDAStopTest2 for ONE_STEP_TSTOP */
  double troundoff = 100.0 * I->uround * (fabs(I->tn) + fabs(I->hh));
  if (fabs(I->tn - tstop) <= troundoff) { ida_get_solution(I, tstop, yret, ypret); *tret = tstop; return 0; }
  if ((I->tn + I->hh - tstop) * I->hh > 0.0) I->hh = (tstop - I->tn) * (1.0 - 4.0 * I->uround);
  ida_get_solution(I, I->tn, yret, ypret); *tret = I->tn;
  return 0;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* PETLION run logic                                                                                            */
/* ------------------------------------------------------------------------------------------------------------ */
typedef struct { double frac, V, SOC, T, c_s_n, I, eta_plating, c_e_min, dfilm, g; } prev_vals;   /* src/structures.jl:174-184 */

static double calc_V(const orc_model* m, const double* Y) { return Y[m->o_ps] - Y[m->o_ps + m->Np + m->Nn - 1]; }
static double calc_Tavg(const orc_model* m, const double* w, const double* Y, double T0) {
  if (!m->thermal) return T0;
  double s = 0.0; int nt = m->Na + m->Np + m->Ns + m->Nn + m->Nz; for (int i = 0; i < nt; i++) s += w[i] * Y[m->o_T + i]; return s;
}

/* check_simulation_stop! (src/checks.jl:1-224).  Updates *flag (stays -1 if nothing fired). */
static void check_stop(const orc_model* m, const evalb* e, const orc_run* run, const orc_opts* o, double t, double tf,
                       const double* Y, const double* YP, double SOC, prev_vals* pv, int* flag, double c_max_n) {
  double eps = t < 1.0 ? o->reltol : 0.0;
  if (t >= tf) { *flag = 0; return; }
  if (!o->check_bounds || run->value_kind == ORC_VAL_REST) return;
  const orc_bounds* b = &run->bounds;
  double I = Y[m->o_I];
  if (run->mode != ORC_MODE_I) {                                     /* check_stop_I, checks.jl:31-54 */
    double dI = YP[m->o_I];
    if ((I - b->I_max > eps) && dI > 0) { double tf_ = (pv->I - b->I_max) / (pv->I - I); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 7; } }
    else if ((b->I_min - I > eps) && dI < 0) { double tf_ = (pv->I - b->I_min) / (pv->I - I); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 8; } }
    pv->I = I;
  }
  if (run->mode != ORC_MODE_V) {                                     /* check_stop_V, checks.jl:56-81 */
    double V = calc_V(m, Y), dV = calc_V(m, YP);
    if ((b->V_min - V > eps) && dV < 0) { double tf_ = (pv->V - b->V_min) / (pv->V - V); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 1; } }
    else if ((V - b->V_max > eps) && dV > 0) { double tf_ = (pv->V - b->V_max) / (pv->V - V); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 2; } }
    pv->V = V;
  }
  {                                                                  /* check_stop_SOC, checks.jl:83-104 */
    if ((b->SOC_min - SOC > eps) && I < 0) { double tf_ = (pv->SOC - b->SOC_min) / (pv->SOC - SOC); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 3; } }
    else if ((SOC - b->SOC_max > eps) && I > 0) { double tf_ = (pv->SOC - b->SOC_max) / (pv->SOC - SOC); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 4; } }
    pv->SOC = SOC;
  }
  if (m->thermal && !isnan(b->T_max) && run->mode != ORC_MODE_DT) {   /* check_stop_T, checks.jl:106-124 */
    double T = calc_Tavg(m, e->w, Y, 0), dT = calc_Tavg(m, e->w, YP, 0);
    if (T - b->T_max > eps && dT > 0) { double tf_ = (pv->T - b->T_max) / (pv->T - T); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 5; } }
    pv->T = T;
  }
  if (!isnan(b->c_s_n_max)) {                                        /* check_stop_c_s_surf, checks.jl:141-161 */
    double cmax = -INFINITY; for (int i = 0; i < m->Nn; i++) cmax = fmax(cmax, Y[m->o_cs + m->Np * m->Nrp + (i + 1) * m->Nrn - 1]);
    if (I > 0 && cmax - b->c_s_n_max * c_max_n > eps) { double tf_ = (pv->c_s_n - b->c_s_n_max * c_max_n) / (pv->c_s_n - cmax); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 6; } }
    pv->c_s_n = cmax;
  }
  if (!isnan(b->c_e_min)) {                                          /* check_stop_c_e, checks.jl:163-183 */
    double cmin = INFINITY; for (int i = 0; i < m->Np + m->Ns + m->Nn; i++) cmin = fmin(cmin, Y[m->o_ce + i]);
    if (b->c_e_min - cmin > eps) { double tf_ = (pv->c_e_min - b->c_e_min) / (pv->c_e_min - cmin); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 9; } }
    pv->c_e_min = cmin;
  }
  if (!isnan(b->eta_plating_min)) {                                  /* check_stop_η_plating, checks.jl:185-201 */
    int is = m->o_ps + m->Np, ie = m->o_pe + m->Np + m->Ns;
    double ep = Y[is] - Y[ie], dep = YP[is] - YP[ie];
    if (b->eta_plating_min - ep > eps && dep < 0) { double tf_ = (pv->eta_plating - b->eta_plating_min) / (pv->eta_plating - ep); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 11; } }
    pv->eta_plating = ep;
  }
  if (m->aging) {                                                    /* check_stop_dfilm, checks.jl:203-224 */
    double dmax = -INFINITY; for (int i = 0; i < m->Nn; i++) dmax = fmax(dmax, YP[m->o_film + i]);
    if (!isnan(b->dfilm_max) && dmax - b->dfilm_max > eps) { double tf_ = (pv->dfilm - b->dfilm_max) / (pv->dfilm - dmax); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 10; } }
    pv->dfilm = dmax;
  }
  if (o->n_stop > 0) {                                               /* opts.stop_function, checks.jl:26 */
    double g = prog_eval_range(o->stop_ops, o->stop_args, t, Y, YP, e->th, 0, o->n_stop);
    if (g > eps) { double tf_ = pv->g / (pv->g - g); if (tf_ < pv->frac) { pv->frac = tf_; *flag = 12; } }
    pv->g = g;
  }
}

static int find_key(const orc_model* m, const char* k) { for (int i = 0; i < m->P; i++) if (!strcmp(m->theta_keys[i], k)) return i; return -1; }

/*
 * One trajectory = a protocol of chained runs on one cell.
 * Outputs one row per saved point (t=0 of a new solution and every accepted step; the last point of a run is
 * replaced by the back-interpolated one), like the reference's default outputs (:t,:V) plus I, SOC, T_avg.
 */
typedef struct { orc_model M; evalb ev[ORC_NMODES]; int ev_ok[ORC_NMODES]; ida_t I; int ida_ok; double* out_Y; } orc_ctx;

static int simulate_core(orc_ctx* ctx, const double* theta, double SOC0, int n_runs, const orc_run* runs, const orc_opts* opts,
                 int max_out, double* out_t, double* out_V, double* out_I, double* out_SOC, double* out_T, int* n_out,
                 double* Y_final, double* YP_final, orc_runinfo* info, orc_counters* counters, const double* Y_init) {
  orc_model M = ctx->M;
  evalb* ev = ctx->ev; int* ev_ok = ctx->ev_ok;
  int N = M.N; orc_counters cz; memset(&cz, 0, sizeof(cz)); orc_counters* cnt = counters ? counters : &cz; memset(cnt, 0, sizeof(*cnt));
  int kc = find_key(&M, "c_e₀"), kT0 = find_key(&M, "T₀"), kcm = find_key(&M, "c_max_n");
  double c_e0 = kc >= 0 ? theta[kc] : 1000.0, T0 = kT0 >= 0 ? theta[kT0] : 298.15, c_max_n = kcm >= 0 ? theta[kcm] : 1.0;
  double* Y = (double*)calloc(N, sizeof(double)); double* YP = (double*)calloc(N, sizeof(double));
  double* Yprev = (double*)calloc(N, sizeof(double)); double* YPprev = (double*)calloc(N, sizeof(double));
  if (!ctx->ida_ok) { ida_alloc(&ctx->I, N); ctx->ida_ok = 1; }
  ida_t* Ip = &ctx->I;
  int nout = 0, rc = 0; double t_global = 0.0, SOC = SOC0; int have_prev = 0; double prev_V = 0, prev_I = 0, prev_etap = 0;
#define SAVE(tt_, Y_, SOC_) do { if (nout < max_out) { \
      if (out_t) { out_t[nout] = (tt_); } \
      if (out_V) { out_V[nout] = calc_V(&M, (Y_)); } \
      if (out_I) { out_I[nout] = (Y_)[M.o_I]; } \
      if (out_SOC) { out_SOC[nout] = (SOC_); } \
      if (out_T) { out_T[nout] = calc_Tavg(&M, cur_w, (Y_), T0); } \
      if (ctx->out_Y) { memcpy(ctx->out_Y + (size_t)nout * N, (Y_), N * sizeof(double)); } }   /* outputs = :all (sol.Y, save_outputs.jl:11-40) */ \
    nout++; } while (0)
#define REPLACE_LAST(tt_, Y_, SOC_) do { nout--; SAVE(tt_, Y_, SOC_); } while (0)
  evalb ev_der; int own_ev = 0; const double* cur_w = NULL;
  for (int r = 0; r < n_runs; r++) {
    const orc_run* run = &runs[r];
    int mode = run->mode;
    if (mode < 0 || mode >= ORC_NMODES) { rc = -101; break; }
    int new_run = !have_prev;
    /* --- initialize_simulation! --- */
    double t0;
    if (new_run) {
      t0 = 0.0;
      if (Y_init) memcpy(Y, Y_init, N * sizeof(double));
      else { M.initial_guess(Y, SOC0, theta); Y[M.o_I] = 0.0; }
      SOC = SOC0;
    } else t0 = nextafter(t_global, INFINITY);       /* initial_time, model_evaluation.jl:112 */
    /* initial_current! (input_methods.jl:11-74) */
    double value = run->value;
    const int is_tab = run->value_kind == ORC_VAL_TABLE || run->value_kind == ORC_VAL_EXPR;
    const int is_fun = is_tab && mode != ORC_MODE_RES;      /* run_function; a `res` closure is a run_residual */
    if (is_tab) {                                /* run_function: initial_current! (input_methods.jl:28-34, 65-76, 104-107, 143-153) */
      value = run_input(run, 0.0, Y, YP, theta);
      if (mode == ORC_MODE_I) Y[M.o_I] = value;
      else if (mode == ORC_MODE_P) Y[M.o_I] = value / (calc_V(&M, Y) * calc_I1C_c(&M, theta));
      else if (mode == ORC_MODE_V || mode == ORC_MODE_ETAP) { if (have_prev) Y[M.o_I] = prev_I; else { double OCV = calc_V(&M, Y); Y[M.o_I] = value > OCV ? 1.0 : -1.0; } }
      else if (mode == ORC_MODE_RES) { if (run->n_dcol < 1) { rc = -103; break; } Y[M.o_I] = have_prev ? prev_I : 1.0; }      /* input_methods.jl:171-176 */
      else { rc = -103; break; }
    } else
    if (mode == ORC_MODE_I) {
      if (run->value_kind == ORC_VAL_HOLD) value = have_prev ? prev_I : 0.0;
      else if (run->value_kind == ORC_VAL_REST) value = 0.0;
      Y[M.o_I] = value;
    } else if (mode == ORC_MODE_V) {
      if (run->value_kind == ORC_VAL_HOLD) { value = prev_V; Y[M.o_I] = prev_V; /* input_methods.jl:58 (guess only) */ }
      else { if (have_prev && prev_I != 0.0) Y[M.o_I] = prev_I; else { double OCV = calc_V(&M, Y); Y[M.o_I] = value > OCV ? 1.0 : -1.0; } }
    } else if (mode == ORC_MODE_P) {          /* input_methods.jl:86-103 */
      if (run->value_kind == ORC_VAL_HOLD) { value = prev_I * calc_I1C_c(&M, theta) * prev_V; Y[M.o_I] = prev_I; }
      else if (run->value_kind == ORC_VAL_REST) { value = 0.0; Y[M.o_I] = 0.0; }
      else Y[M.o_I] = value / (calc_V(&M, Y) * calc_I1C_c(&M, theta));
    } else if (mode == ORC_MODE_ETAP) {       /* input_methods.jl:120-142 */
      if (run->value_kind == ORC_VAL_HOLD) { value = prev_etap; Y[M.o_I] = prev_I; }
      else { if (have_prev) Y[M.o_I] = prev_I; else { double OCV = calc_V(&M, Y); Y[M.o_I] = value > OCV ? 1.0 : -1.0; } }
    } else { /* dT: custom_res! (model_evaluation.jl:155-172): :hold -> hold_val = 0 */
      if (run->value_kind == ORC_VAL_HOLD) value = 0.0;
      if (have_prev) Y[M.o_I] = prev_I; else Y[M.o_I] = 1.0;      /* input_methods.jl:171-176 */
    }
    evalb* e = NULL;
    /* a closure of the state with derivative programs, or the rate of a state chosen at run time, has its own sparsity pattern: its own evaluator bundle */
    if (own_ev) { evalb_free(&ev_der); own_ev = 0; }
    if (mode == ORC_MODE_DSTATE) {
      /* which state: the extreme surface / electrolyte concentration of sol.Y[end] (input_methods.jl:195-247; argmax / argmin return the first extreme) */
      if (!have_prev || run->dstate < 1 || run->dstate > 6) { rc = -103; break; }
      int first, count, stride;                     /* surface entry of particle i: the last of its N_r radial nodes (c_s_indices, aux...jl:688-716; N_r = 1: c_s_avg itself) */
      if (run->dstate <= 2) { first = M.o_cs + M.Nrp - 1; count = M.Np; stride = M.Nrp; }
      else if (run->dstate <= 4) { first = M.o_cs + M.Np * M.Nrp + M.Nrn - 1; count = M.Nn; stride = M.Nrn; }
      else { first = M.o_ce; count = M.Np + M.Ns + M.Nn; stride = 1; }
      int best = first; const int want_max = run->dstate & 1;
      for (int q = 1; q < count; q++) { const int idx = first + q * stride; if (want_max ? Y[idx] > Y[best] : Y[idx] < Y[best]) best = idx; }
      if (evalb_init_x(&ev_der, &M, theta, mode, value, cnt, NULL, best) != 0) { rc = -102; break; }
      own_ev = 1; e = &ev_der;
    } else {
      if (!ev_ok[mode]) { if (evalb_init(&ev[mode], &M, theta, mode, value, cnt) != 0) { rc = -102; break; } ev_ok[mode] = 1; }
      e = &ev[mode];
    }
    if (run->value_kind == ORC_VAL_EXPR && run->n_dcol > 0) { if (evalb_init_d(&ev_der, &M, theta, mode, value, cnt, run) != 0) { rc = -102; break; } own_ev = 1; e = &ev_der; }
    e->value = value; e->th = theta; e->cnt = cnt; e->frun = is_tab ? run : NULL; e->t_fun = 0.0;
    cur_w = e->w;
    if (M.thermal) M.dT_weights(e->w, theta);
    int ierr = newtons_method(e, Y, YP, opts, c_e0);
    orc_runinfo* ri = &info[r]; memset(ri, 0, sizeof(*ri)); ri->flag = -1;
    if (ierr != 0) { ri->flag = ierr; ri->t_end = t_global; rc = 1; break; }
    if (opts->exp_yp_alg_zero) for (int n = M.Nd; n < N; n++) YP[n] = 0.0;
    ida_reinit(Ip, e, opts, Y, YP);
    /* tstops (postfix_integrator!, model_evaluation.jl:288-310): {1.0 if continuation} U {tf} */
    double* tstops = (double*)malloc((opts->n_tdiscon + opts->n_tstops + 4) * sizeof(double)); int nts = 0, its = 0;
    for (int q = 0; q < opts->n_tstops; q++) tstops[nts++] = opts->tstops[q];                          /* model_evaluation.jl:292-294 */
    for (int q = 0; q < opts->n_tdiscon; q++) tstops[nts++] = opts->tdiscon[q] - opts->reltol / 2;     /* model_evaluation.jl:295-297 */
    if (!new_run) tstops[nts++] = 1.0;
    tstops[nts++] = run->tf;
    for (int a = 1; a < nts; a++) { double v = tstops[a]; int b = a - 1; while (b >= 0 && tstops[b] > v) { tstops[b + 1] = tstops[b]; b--; } tstops[b + 1] = v; }   /* sort! */
    { int first = 0; while (first < nts && tstops[first] <= 0.0) first++; if (first) { for (int a = first; a < nts; a++) tstops[a - first] = tstops[a]; nts -= first; } }
    { int w_ = 0; for (int a = 0; a < nts; a++) if (tstops[a] <= run->tf) tstops[w_++] = tstops[a]; nts = w_; }   /* stops beyond tf are never reached */
    prev_vals pv = {1.0, -1, -1, -1, -1, -1, -1, -1, -1, -1};
    int flag = -1;
    /* set_vars! at t=0 of a new solution: a continuation run does not add a point (t0 = nextfloat(t_end)) ...
       the reference does push one (set_vars! is unconditional), so we do too. */
    SAVE(t0, Y, SOC);
    check_stop(&M, e, run, opts, 0.0, run->tf, Y, YP, SOC, &pv, &flag, c_max_n);
    memcpy(Yprev, Y, N * sizeof(double)); memcpy(YPprev, YP, N * sizeof(double));
    double tprev = 0.0, t = 0.0, t_prev_saved = t0; int iter = 1; int stalled_once = 0;
    double SOC_prev_pt = SOC;
    /* --- solve! --- */
    while (flag == -1) {
      double tret; tprev = t;
      int sf = ida_step(Ip, tstops[its], &tret, Y, YP);
      if (sf != 0) {
        /* check_solve, checks.jl:227-237: a stall on the very first step is retried once with h0 = reltol */
        if (Ip->nst == 0 && !stalled_once) { stalled_once = 1; memcpy(Y, Yprev, N * sizeof(double)); memcpy(YP, YPprev, N * sizeof(double));
          ida_reinit(Ip, e, opts, Y, YP); Ip->h0_forced = opts->reltol; iter++; t = tprev; continue; }
        flag = sf; break;
      }
      while (tret >= tstops[its] && its + 1 < nts) its++;
      iter++; t = tret;
      /* calc_SOC trapezoid (scalar_residual.jl:103-111) */
      double SOC_new = SOC + 0.5 * ((t + t0) - t_prev_saved) * (Y[M.o_I] + Yprev[M.o_I]) / 3600.0;
      SOC_prev_pt = SOC; SOC = SOC_new;
      SAVE(t + t0, Y, SOC);
      check_stop(&M, e, run, opts, t, run->tf, Y, YP, SOC, &pv, &flag, c_max_n);
      /* check_solve (checks.jl:226-249) */
      if (!is_fun && t == tprev) { flag = ORC_ERR_STALL; break; }              /* (run_function has no stall test, checks.jl:251-269; run_residual has: checks.jl:226) */
      if (iter == opts->maxiters) { flag = ORC_ERR_MAXITERS; break; }
      if (flag == -1) { memcpy(Yprev, Y, N * sizeof(double)); memcpy(YPprev, YP, N * sizeof(double)); t_prev_saved = t + t0; }
      if (flag == -1 && is_fun && t - tprev < 1e-3 * opts->reltol) {          /* check_reinitialization!, checks.jl:341-364 */
        const double t_new = t + opts->reltol, v_old = e->value, v_new = run_input(run, t_new, Y, YP, theta);
        const double big = fmax(fabs(v_old), fabs(v_new));
        if (!(fabs(v_old - v_new) <= fmax(opts->abstol, opts->reltol * big))) {
          e->value = v_new; e->t_fun = t_new;
          if (newtons_method(e, Y, YP, opts, c_e0) != 0) { flag = ORC_ERR_INIT; break; }
          ida_reinit_at(Ip, e, opts, Y, YP, t_new);
        }
      }
    }
    /* --- exit_simulation! / interp_final_points! (model_evaluation.jl:335-382) --- */
    free(tstops);
    double t_end = t + t0;
    if (flag > 0 && opts->interp_final && t > 1.0) {
      double fr = pv.frac;
      double ti = fr * (t - tprev) + tprev;
      for (int n = 0; n < N; n++) { Y[n] = fr * (Y[n] - Yprev[n]) + Yprev[n]; YP[n] = fr * (YP[n] - YPprev[n]) + YPprev[n]; }
      /* set_vars!(...; modify! = set_var_last!): SOC re-accumulated from the un-interpolated last point with
         Y_prev = the interpolated Y and t_prev = the un-interpolated last time (model_evaluation.jl:379-380) */
      double SOC_i = SOC + 0.5 * ((ti + t0) - (t + t0)) * (Y[M.o_I] + Y[M.o_I]) / 3600.0;
      (void)SOC_prev_pt;
      SOC = SOC_i; t_end = ti + t0;
      REPLACE_LAST(t_end, Y, SOC);
    }
    ri->flag = flag; ri->iterations = iter; ri->t_end = t_end; ri->V = calc_V(&M, Y); ri->I = Y[M.o_I]; ri->SOC = SOC;
    ri->T_avg = calc_Tavg(&M, e->w, Y, T0);
    t_global = t_end; have_prev = 1; prev_V = ri->V; prev_I = ri->I; prev_etap = Y[M.o_ps + M.Np] - Y[M.o_pe + M.Np + M.Ns];
    if (flag < 0) { rc = 1; break; }
  }
  if (own_ev) evalb_free(&ev_der);
  if (n_out) *n_out = nout;
  if (Y_final) memcpy(Y_final, Y, N * sizeof(double));
  if (YP_final) memcpy(YP_final, YP, N * sizeof(double));
  free(Y); free(YP); free(Yprev); free(YPprev);
  return rc;
}

static void ctx_free(orc_ctx* c) { for (int k = 0; k < ORC_NMODES; k++) if (c->ev_ok[k]) evalb_free(&c->ev[k]); if (c->ida_ok) ida_free(&c->I); }

int orc_simulate(const char* variant, const double* theta, double SOC0, int n_runs, const orc_run* runs, const orc_opts* opts,
                 int max_out, double* out_t, double* out_V, double* out_I, double* out_SOC, double* out_T, int* n_out,
                 double* Y_final, double* YP_final, orc_runinfo* info, orc_counters* counters, const double* Y_init) {
  orc_ctx* ctx = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  if (get_model(variant, &ctx->M) != 0) { free(ctx); return -100; }
  int rc = simulate_core(ctx, theta, SOC0, n_runs, runs, opts, max_out, out_t, out_V, out_I, out_SOC, out_T, n_out, Y_final, YP_final, info, counters, Y_init);
  ctx_free(ctx); free(ctx);
  return rc;
}

/* orc_simulate + every saved state vector (outputs = :all): out_Y is [max_out][N] */
int orc_simulate_all(const char* variant, const double* theta, double SOC0, int n_runs, const orc_run* runs, const orc_opts* opts,
                 int max_out, double* out_t, double* out_V, double* out_I, double* out_SOC, double* out_T, double* out_Y, int* n_out,
                 double* Y_final, double* YP_final, orc_runinfo* info, orc_counters* counters, const double* Y_init) {
  orc_ctx* ctx = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  if (get_model(variant, &ctx->M) != 0) { free(ctx); return -100; }
  ctx->out_Y = out_Y;
  int rc = simulate_core(ctx, theta, SOC0, n_runs, runs, opts, max_out, out_t, out_V, out_I, out_SOC, out_T, n_out, Y_final, YP_final, info, counters, Y_init);
  ctx_free(ctx); free(ctx);
  return rc;
}

/* CPU-baseline leg of bench.py: n_traj trajectories, one after another on the calling thread, theta[k % n_theta_sets].
 * The evaluator bundle (CSC pattern, fill-reducing ordering, pivot sequence) is created once and reused, like the
 * reference's cached integrator / KLU symbolic analysis (src/model_evaluation.jl:240-251).  Returns the number of
 * trajectories that ended with a non-negative flag; t_end_sum is a checksum. */
int orc_run_batch(const char* variant, int n_theta_sets, const double* thetas, double SOC0, int n_runs, const orc_run* runs,
                  const orc_opts* opts, int n_traj, double* t_end_sum, orc_counters* total) {
  orc_ctx* ctx = (orc_ctx*)calloc(1, sizeof(orc_ctx));
  if (get_model(variant, &ctx->M) != 0) { free(ctx); return -100; }
  int P = ctx->M.P, ok = 0; double acc = 0.0;
  orc_runinfo* info = (orc_runinfo*)calloc(n_runs, sizeof(orc_runinfo));
  orc_counters c; if (total) memset(total, 0, sizeof(*total));
  for (int k = 0; k < n_traj; k++) {
    int nout = 0;
    int rc = simulate_core(ctx, thetas + (size_t)(k % n_theta_sets) * P, SOC0, n_runs, runs, opts, 0, NULL, NULL, NULL, NULL, NULL, &nout, NULL, NULL, info, &c, NULL);
    if (rc == 0) { ok++; acc += info[n_runs - 1].t_end; }
    if (total) { total->n_steps += c.n_steps; total->n_res += c.n_res; total->n_jac += c.n_jac; total->n_fact += c.n_fact; total->n_solve += c.n_solve;
      total->n_newton += c.n_newton; total->n_errfail += c.n_errfail; total->n_convfail += c.n_convfail; total->sum_kp2 += c.sum_kp2; total->n_init_iters += c.n_init_iters; }
  }
  if (t_end_sum) *t_end_sum = acc;
  free(info); ctx_free(ctx); free(ctx);
  return ok;
}

/* ------------------------------------------------------------------------------------------------------------ */
/* evaluator-level entry points (parity checks of residual / Jacobian / init)                                   */
/* ------------------------------------------------------------------------------------------------------------ */
int orc_info(const char* variant, int* N, int* Nd, int* P, int* nnz_base, int* nnz_alg_base) {
  orc_model M; if (get_model(variant, &M) != 0) return -100;
  *N = M.N; *Nd = M.Nd; *P = M.P; *nnz_base = M.nnz; *nnz_alg_base = M.nnz_alg; return 0;
}
const char* orc_theta_key(const char* variant, int i) { orc_model M; if (get_model(variant, &M) != 0 || i < 0 || i >= M.P) return NULL; return M.theta_keys[i]; }

int orc_residual(const char* variant, const double* theta, int mode, double value, const double* Y, const double* YP, double* res) {
  orc_model M; if (get_model(variant, &M) != 0) return -100;
  evalb e; if (evalb_init(&e, &M, theta, mode, value, NULL) != 0) return -102;
  R_full(&e, res, Y, YP); evalb_free(&e); return 0;
}
int orc_initial_guess(const char* variant, const double* theta, double SOC, double* Y) {
  orc_model M; if (get_model(variant, &M) != 0) return -100;
  M.initial_guess(Y, SOC, theta); Y[M.o_I] = 0.0; return 0;
}
/* full Jacobian in CSC: pass NULL arrays to query nnz */
int orc_jacobian(const char* variant, const double* theta, int mode, double value, const double* Y, const double* YP, double cj,
                 int* nnz, int* colptr, int* rowval, double* nzval) {
  orc_model M; if (get_model(variant, &M) != 0) return -100;
  evalb e; if (evalb_init(&e, &M, theta, mode, value, NULL) != 0) return -102;
  *nnz = e.nnz;
  if (colptr) memcpy(colptr, e.cp, (M.N + 1) * sizeof(int));
  if (rowval) memcpy(rowval, e.ri, e.nnz * sizeof(int));
  if (nzval) { J_full(&e, Y, YP, cj); memcpy(nzval, e.ax, e.nnz * sizeof(double)); }
  evalb_free(&e); return 0;
}
/* consistent initialisation only: Y (in/out), YP (out) */
int orc_init_consistent(const char* variant, const double* theta, int mode, double value, double reltol_init, double* Y, double* YP, int* iters) {
  orc_model M; if (get_model(variant, &M) != 0) return -100;
  orc_counters c; memset(&c, 0, sizeof(c));
  evalb e; if (evalb_init(&e, &M, theta, mode, value, &c) != 0) return -102;
  orc_opts o; memset(&o, 0, sizeof(o)); o.reltol_init = reltol_init;
  int kc = find_key(&M, "c_e₀");
  int rc = newtons_method(&e, Y, YP, &o, kc >= 0 ? theta[kc] : 1000.0);
  if (iters) *iters = (int)c.n_init_iters;
  evalb_free(&e); return rc;
}
/* solve J x = b with the KLU-like LU (used to cross-check the structured device solver) */
int orc_linear_solve_refined(const char* variant, const double* theta, int mode, double value, const double* Y, const double* YP, double cj, double* b, int nref) {
  orc_model M; if (get_model(variant, &M) != 0) return -100;
  evalb e; if (evalb_init(&e, &M, theta, mode, value, NULL) != 0) return -102;
  J_full(&e, Y, YP, cj);
  int rc = lu_setup_keep(&e.lu, M.N, e.cp, e.ri, e.ax, e.ax_f);
  if (rc == 0) lu_solve_refined(&e.lu, M.N, e.cp, e.ri, e.ax_f, b, e.rtmp, e.xtmp, nref);
  evalb_free(&e); return rc;
}
int orc_linear_solve(const char* variant, const double* theta, int mode, double value, const double* Y, const double* YP, double cj, double* b) {
  return orc_linear_solve_refined(variant, theta, mode, value, Y, YP, cj, b, 0);
}

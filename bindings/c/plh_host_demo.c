/*
 * plh_host_demo.c -- a host that is neither Python nor torch on the C ABI of include/petlion_hip.h.
 *
 * What a Julia `ccall` layer does, minus Julia (bindings/julia/PetlionHIP.jl is the same sequence of calls): plain C99, gcc, no HIP headers, host pointers only
 * (PLH_HOST: the library stages everything through the device).  It
 *   1. creates the default LCO model            -- petlion(LCO)                                       (reference src/params.jl:119-174)
 *   2. builds an n_cells x P parameter matrix from the library's own theta table, with the solid-phase diffusivity D_sp varied per cell
 *   3. config C2: one 1C discharge per cell      -- simulate(p, I = -1, SOC = 1)                      (src/model_evaluation.jl:11-49)
 *   4. a 2C charge to V_max = 4.1 V and, in a SECOND plh_integrate call that continues from the first one's end state (Y_init / t_init / SOC0), a voltage hold
 *                                                -- simulate(p, I = 2, SOC = 0, V_max = 4.1); simulate!(sol, p, V = :hold)   (src/model_evaluation.jl:87-97, 206-209)
 * and prints every per-cell result with %a (hexadecimal floating point: bit-exact text), one line per cell and leg.  tests/test_c_host.py runs it and compares the lines
 * bit for bit with the same calls made through the ctypes mirror.
 *
 *   usage: plh_host_demo [n_cells = 1024]
 *   build: gcc -std=c99 -O1 -I include bindings/c/plh_host_demo.c -L petlion.jl_amd -lpetlion_hip -Wl,-rpath,$PWD/petlion.jl_amd -lm -o bindings/c/plh_host_demo
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "petlion_hip.h"

#define CHECK(call) do { int rc__ = (call); if (rc__ != 0) { fprintf(stderr, "%s -> %d: %s\n", #call, rc__, plh_last_error()); return 1; } } while (0)

static plh_bounds default_bounds_LCO(void) {          /* reference boundary_stop_conditions of LCO, src/params.jl:235-250; NaN = unset */
  plh_bounds b;
  b.V_max = 4.3; b.V_min = 2.5; b.SOC_max = 1.0; b.SOC_min = 0.0; b.T_max = 55.0 + 273.15; b.c_s_n_max = NAN; b.I_max = NAN; b.I_min = NAN;
  b.eta_plating_min = NAN; b.c_e_min = NAN; b.dfilm_max = NAN;
  return b;
}
static plh_opts default_opts(void) {                  /* reference options_simulation defaults, src/params.jl:252-285 */
  plh_opts o;
  memset(&o, 0, sizeof o);
  o.abstol = 1e-6; o.reltol = 1e-3; o.abstol_init = 1e-6; o.reltol_init = 1e-3; o.maxiters = 10000; o.check_bounds = 1; o.interp_final = 1; o.max_order = 5;
  return o;
}
static plh_run make_run(int mode, int value_kind, double value, double tf, plh_bounds b) {
  plh_run r;
  memset(&r, 0, sizeof r);
  r.mode = mode; r.value_kind = value_kind; r.value = value; r.tf = tf; r.bounds = b;
  return r;
}
static void print_leg(const char* leg, int n, const plh_run_info* ri, const int* n_pts, const double* Y, int N) {
  for (int c = 0; c < n; c++) {
    double s = 0.0;                                  /* a checksum of the end state: sum of the entries in index order (bit-exact if every entry is) */
    for (int k = 0; k < N; k++) s += Y[(size_t)c * N + k];
    printf("%s cell %d flag %d iterations %d n_pts %d t_end %a V %a I %a SOC %a Ysum %a\n", leg, c, ri[c].flag, ri[c].iterations, n_pts[c], ri[c].t_end, ri[c].V, ri[c].I,
           ri[c].SOC, s);
  }
}

int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1024;
  if (n < 1) { fprintf(stderr, "usage: plh_host_demo [n_cells]\n"); return 2; }
  plh_model_desc d;
  memset(&d, 0, sizeof d);
  d.chemistry = PLH_CHEM_LCO_LIC6; d.N_p = d.N_s = d.N_n = d.N_a = d.N_z = d.N_r_p = d.N_r_n = 10; d.real_bytes = 8; d.precision = PLH_PREC_F64; d.device = -1;
  plh_model_t m;
  CHECK(plh_model_create(&d, &m));
  const int N = plh_n_states(m), P = plh_n_theta(m);
  int iD = -1;
  for (int k = 0; k < P; k++) if (!strcmp(plh_theta_key(m, k), "D_sp")) iD = k;
  if (iD < 0) { fprintf(stderr, "no D_sp in theta_keys\n"); return 1; }
  printf("model N %d N_diff %d P %d LDS %d\n", N, plh_n_diff(m), P, plh_lds_bytes(m));
  double* theta = (double*)malloc(sizeof(double) * (size_t)n * P);
  double* soc = (double*)malloc(sizeof(double) * (size_t)n);
  for (int c = 0; c < n; c++) {
    for (int k = 0; k < P; k++) theta[(size_t)c * P + k] = plh_theta_default(m, k);
    theta[(size_t)c * P + iD] *= 1.0 + 0.125 * (c % 5);         /* (exactly representable factors: the ctypes side forms the same products) */
  }
  const int max_pts = 512;
  plh_run_info* ri = (plh_run_info*)calloc((size_t)n, sizeof *ri);
  plh_counters* cn = (plh_counters*)calloc((size_t)n, sizeof *cn);
  int* n_pts = (int*)calloc((size_t)n, sizeof *n_pts);
  double* Y = (double*)malloc(sizeof(double) * (size_t)n * N);
  double* YP = (double*)malloc(sizeof(double) * (size_t)n * N);
  double* t = (double*)malloc(sizeof(double) * (size_t)n * max_pts);
  double* V = (double*)malloc(sizeof(double) * (size_t)n * max_pts);
  plh_outputs out;
  memset(&out, 0, sizeof out);
  out.max_pts = max_pts; out.t = t; out.V = V; out.n_pts = n_pts; out.Y_final = Y; out.YP_final = YP; out.run_info = ri; out.counters = cn;
  plh_opts o = default_opts();
  plh_bounds b = default_bounds_LCO();

  /* C2: 1C discharge from SOC 1 (ends on SOC_min = 0 at 3600 s, flag 3) */
  for (int c = 0; c < n; c++) soc[c] = 1.0;
  plh_run discharge = make_run(PLH_MODE_I, PLH_VAL_CONST, -1.0, 1e6, b);
  CHECK(plh_integrate(m, n, theta, soc, NULL, NULL, 1, &discharge, &o, &out, PLH_HOST, NULL));
  print_leg("c2", n, ri, n_pts, Y, N);
  long long steps = 0;
  for (int c = 0; c < n; c++) steps += cn[c].n_steps;
  printf("c2 kernel_ms_positive %d steps_total %lld V0 %a\n", plh_last_kernel_ms(m) > 0.0, steps, V[0]);

  /* CC-CV across two calls: leg 1 from SOC 0, leg 2 continues the solution of leg 1 */
  b.V_max = 4.1;
  for (int c = 0; c < n; c++) soc[c] = 0.0;
  plh_run cc = make_run(PLH_MODE_I, PLH_VAL_CONST, 2.0, 1e6, b);
  CHECK(plh_integrate(m, n, theta, soc, NULL, NULL, 1, &cc, &o, &out, PLH_HOST, NULL));
  print_leg("cc", n, ri, n_pts, Y, N);
  double* t_init = (double*)malloc(sizeof(double) * (size_t)n);
  double* Y_init = (double*)malloc(sizeof(double) * (size_t)n * N);
  for (int c = 0; c < n; c++) { t_init[c] = ri[c].t_end; soc[c] = ri[c].SOC; }
  memcpy(Y_init, Y, sizeof(double) * (size_t)n * N);
  plh_run cv = make_run(PLH_MODE_V, PLH_VAL_HOLD, 0.0, 1e6, b);
  CHECK(plh_integrate(m, n, theta, soc, Y_init, t_init, 1, &cv, &o, &out, PLH_HOST, NULL));
  print_leg("cv", n, ri, n_pts, Y, N);

  plh_model_destroy(m);
  free(theta); free(soc); free(ri); free(cn); free(n_pts); free(Y); free(YP); free(t); free(V); free(t_init); free(Y_init);
  return 0;
}

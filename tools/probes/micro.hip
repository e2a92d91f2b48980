// micro-benchmark of the individual device functions (one wave, one cell): shader cycles per call.  GPU box: hipcc ... && ./micro
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include "../../petlion.jl_amd/csrc/dfn_integrate.h"
#include "../../petlion.jl_amd/csrc/radial_tables_nr10.h"
using namespace pl;
#define REP 200
#define TIME(idx, ...) { PL_SYNC(); long long t0 = (long long)__builtin_readcyclecounter(); for (int q = 0; q < REP; q++) { __VA_ARGS__; } PL_SYNC(); long long t1 = (long long)__builtin_readcyclecounter(); if (lane_id() == 0) out[idx] = (double)(t1 - t0) / REP; }
__global__ __launch_bounds__(64) void k_micro(const Tables* tb, const double* th, const double* Y, const double* YP, double* out) {
  __shared__ CellLDS S;
  LaneRegs R;
  const int lane = lane_id();
  cell_setup(S, R, tb, th);
  for (int n = lane; n < NST; n += WAVE) { S.yy[n] = Y[n]; S.yp[n] = YP[n]; S.ee[n] = 0; S.ewt[n] = 1.0 / (1e-3 * fabs(Y[n]) + 1e-6); for (int j = 0; j < 6; j++) S.phi[j][n] = j == 0 ? Y[n] : 1e-3 * Y[n]; }
  if (lane <= MAXORD) { S.ida_psi[lane] = 1.0 + lane; S.ida_gamma[lane] = 0.1; S.ida_alpha[lane] = 0.5; S.ida_beta[lane] = 1.0; S.ida_sigma[lane] = 1.0; }
  PL_SYNC();
  IdaScalars I; memset(&I, 0, sizeof(I)); I.kk = 3; I.kused = 3; I.cj = 0.5; I.hh = 1.0; I.maxord = 5; I.tn = 10.0;
  cell_node_pass<true, true>(S, S.yy, S.yp, S.delta, 0, -1.0); PL_SYNC();
  cell_factor(S, R, tb, 0.5, 0, false);
  TIME(0, (cell_node_pass<true, false>(S, S.yy, S.yp, S.delta, 0, -1.0)));
  TIME(1, (cell_cs_rows(S, R, S.yy, S.yp, S.delta)));
  TIME(2, (cell_node_pass<true, true>(S, S.yy, S.yp, S.delta, 0, -1.0)));
  TIME(3, (cell_factor(S, R, tb, 0.5, 0, false)));
  TIME(4, (cell_solve(S, R, S.delta, 0, false)));
  { double a = S.delta[lane], b = S.delta[lane + 64], c = S.delta[lane + 128]; TIME(5, (thomas_sweeps(S, false, a, b, c))); if (a + b + c == 1.2345) out[20] = a; }
  TIME(6, (form_iterate(S, I)));
  { double acc = 0; TIME(7, (acc += wrms(S.delta, S.ewt))); if (acc == 1.2345) out[21] = acc; }
  { double e1, e2; int r = 0; TIME(8, (r += ida_test_error(S, I, 1.0, e1, e2))); if (r == 12345) out[22] = e1; }
  TIME(9, (ida_get_solution(S, I, 10.0, S.yy, S.yp)));
  { double acc = 0; TIME(10, acc += ida_set_coeffs(S, I); I.tn = 10.0); if (acc == 1.2345) out[23] = acc; }
  TIME(11, (set_ewt(S, 1e-3, 1e-6)));
  { double s = 0; TIME(12, const double sc = -1.0; double ss = 0; PL_VEC(n) { const double d = S.delta[n] * sc; S.ee[n] += d; const double p = d * S.ewt[n]; ss += p * p; } s += sqrt(wave_sum(ss) / NST)); if (s == 1.2345) out[24] = s; }
  { double acc = 1.0001; TIME(13, (acc = exp(-log(2.0 * acc + 0.0001) / 4))); if (acc == 1.2345) out[25] = acc; }
  { double acc = 1.0001; TIME(14, (acc = 1.0 / (acc + 0.5))); if (acc == 1.2345) out[26] = acc; }
  { double acc = 1.0001; TIME(15, (acc = sinh(acc * 0.3))); if (acc == 1.2345) out[27] = acc; }
  { double acc = 1.0001; TIME(16, (acc = acc * 1.000001 + 0.5)); if (acc == 1.2345) out[28] = acc; }
  { double acc = lane; TIME(17, (acc = shift_up1(acc) + 1.0)); if (acc == 1.2345) out[29] = acc; }
  { double acc = lane; TIME(18, (acc = S.delta[((int)acc) & 255] + 1.0)); if (acc == 1.2345) out[30] = acc; }
  { double acc = 0.3; TIME(19, (acc = exp(acc * 0.5))); if (acc == 1.2345) out[39] = acc; }
  { double acc = 0.3; TIME(20, (acc = expm1(acc * 0.5))); if (acc == 1.2345) out[39] = acc; }
  { double acc = 1.3; TIME(21, (acc = sqrt(acc + 1.0))); if (acc == 1.2345) out[39] = acc; }
  { double acc = 0.5, U, dU; TIME(22, ocv_lic6(acc, 298.15, 1, U, dU); acc = 0.5 + 1e-3 * U); if (acc == 1.2345) out[39] = acc; }
  { double acc = 0.7, U, dU; TIME(23, ocv_lco(acc, 298.15, 1, U, dU); acc = 0.7 + 1e-3 * U); if (acc == 1.2345) out[39] = acc; }
  { double acc = 1000.0, K, dK; TIME(24, keff(acc, 298.15, K, dK); acc = 1000.0 + K); if (acc == 1.2345) out[39] = acc; }
  { double acc = 0.3, sh, ch; TIME(25, sinh_cosh(acc, sh, ch); acc = 0.3 + 1e-3 * sh); if (acc == 1.2345) out[39] = acc; }
  { double acc = 1.5; TIME(26, (acc = hmean(0.5, acc, 2.0))); if (acc == 1.2345) out[39] = acc; }
}
int main() {
  Tables tb; memset(&tb, 0, sizeof(tb));
  memcpy(tb.M, PL_RADIAL_M, sizeof(tb.M)); memcpy(tb.LAM, PL_RADIAL_LAM, sizeof(tb.LAM)); memcpy(tb.V, PL_RADIAL_V, sizeof(tb.V)); memcpy(tb.W, PL_RADIAL_W, sizeof(tb.W));
  tb.BJ = PL_RADIAL_BJ_FACTOR; tb.P = K_COUNT; for (int k = 0; k < K_COUNT; k++) tb.thidx[k] = k;
  double th[K_COUNT] = {7.5e-10, 7.5e-10, 7.5e-10, 3.9e-14, 1e-14, 5000.0, 5000.0, 5000.0, 5000.0, 2e-6, 2e-6, 25 + 273.15, 4.0, 4.0, 4.0, 1000.0, 30555.0, 51554.0, 5.0310e-11, 2.334e-11, 88e-6, 80e-6, 25e-6, 0.364, 0.85510, 0.49550, 0.01429, 0.99174, 100.0, 100.0, 0.0326, 0.025, 0.485, 0.385, 0.724};
  std::vector<double> Y(NST), YP(NST, 1e-3);
  for (int n = 0; n < NST; n++) Y[n] = n < 30 ? 1000.0 + n : (n < 130 ? 30000.0 + n : (n < 230 ? 15000.0 + n : (n < 250 ? 1e-5 : (n < 280 ? -0.01 * (n - 250) : (n < 290 ? 4.0 : (n < 300 ? 0.1 : -1.0))))));
  Tables* d_tb; double *d_th, *d_Y, *d_YP, *d_out; double out[40] = {0};
  hipMalloc(&d_tb, sizeof(tb)); hipMalloc(&d_th, sizeof(th)); hipMalloc(&d_Y, NST * 8); hipMalloc(&d_YP, NST * 8); hipMalloc(&d_out, sizeof(out));
  hipMemcpy(d_tb, &tb, sizeof(tb), hipMemcpyHostToDevice); hipMemcpy(d_th, th, sizeof(th), hipMemcpyHostToDevice);
  hipMemcpy(d_Y, Y.data(), NST * 8, hipMemcpyHostToDevice); hipMemcpy(d_YP, YP.data(), NST * 8, hipMemcpyHostToDevice); hipMemset(d_out, 0, sizeof(out));
  k_micro<<<1, 64>>>(d_tb, d_th, d_Y, d_YP, d_out); hipDeviceSynchronize(); hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost);
  const char* nm[27] = {"node_pass<res>", "cs_rows", "node_pass<res+jac>", "cell_factor", "cell_solve", "thomas_sweeps", "form_iterate", "wrms", "ida_test_error", "ida_get_solution",
                        "ida_set_coeffs", "set_ewt", "newton accumulate+norm", "exp(log) chain", "fp64 divide chain", "sinh chain", "fp64 fma chain", "dpp shift+add chain", "LDS load chain", "exp", "expm1", "sqrt", "ocv_lic6", "ocv_lco", "keff", "sinh_cosh", "hmean"};
  for (int k = 0; k < 27; k++) printf("%-24s %10.1f cycles\n", nm[k], out[k]);
  return 0;
}

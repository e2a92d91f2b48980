// petlion_hip.hip -- kernels + the C ABI of include/petlion_hip.h  (libpetlion_hip.so, gfx950).
//
// One workgroup = one 64-lane wavefront = one cell.  Grid = n_cells workgroups; ~38 KB of LDS per workgroup, so four cells are
// resident per CU (one per SIMD) and 1024 cells fill the 256 CUs of an MI355X in a single wave of workgroups.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#ifndef PL_WAVE_EMU
#include <dlfcn.h>
#endif

#include "dfn_integrate.h"
#include "radial_tables_nr10.h"

#ifndef PL_WAVE_EMU
#define PL_LAUNCH(kernel, grid, block, stream, ...) hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, __VA_ARGS__)
#endif

using namespace pl;

// Every kernel runs at most one wavefront per SIMD (LDS: >= 40 kB per single-wave workgroup), so the compiler may use the whole
// 512-entry register file of a lane (256 VGPR + 256 AGPR) instead of spilling to scratch.
#ifndef PL_WAVE_EMU
#define PL_ONE_WAVE_PER_SIMD __attribute__((amdgpu_waves_per_eu(1, 1)))
#else
#define PL_ONE_WAVE_PER_SIMD
#endif

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------
template <class M> __device__ __forceinline__ void load_vec(double* dst, const double* __restrict__ src) {
  const int lane = lane_id();
  _Pragma("unroll") for (int k__ = 0, n = lane; k__ < M::NTRIP; k__++, n += WAVE) if (n < M::NST) dst[n] = src[n];
}
template <class M> __device__ __forceinline__ void store_vec(double* __restrict__ dst, const double* src) {
  const int lane = lane_id();
  _Pragma("unroll") for (int k__ = 0, n = lane; k__ < M::NTRIP; k__++, n += WAVE) if (n < M::NST) dst[n] = src[n];
}

template <class M> __global__ __launch_bounds__(64) PL_ONE_WAVE_PER_SIMD void k_initial_guess(const Tables* tb, int n_cells, const double* theta, const double* SOC, double* Y) {
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

template <class M> __global__ __launch_bounds__(64) PL_ONE_WAVE_PER_SIMD void k_residual(const Tables* tb, int n_cells, const double* theta, const double* Y, const double* YP,
                                                 int mode, double value, double* F) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  load_vec<M>(S.yy, Y + (size_t)cell * NST); load_vec<M>(S.yp, YP + (size_t)cell * NST);
  PL_SYNC();
  cell_residual(S, R, S.yy, S.yp, S.delta, mode, value);
  store_vec<M>(F + (size_t)cell * NST, S.delta);
}

template <class M> __global__ __launch_bounds__(64) PL_ONE_WAVE_PER_SIMD void k_jacobian(const Tables* tb, int n_cells, const double* theta, const double* Y, const double* YP,
                                                 double cj, int mode, double* nz) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  load_vec<M>(S.yy, Y + (size_t)cell * NST); load_vec<M>(S.yp, YP + (size_t)cell * NST);
  PL_SYNC();
  cell_res_jac(S, R, S.yy, S.yp, S.delta, mode, 0.0);
  const int nnz = tb->nnz[mode];
  const unsigned* code = tb->csc_code[mode];
  double* out = nz + (size_t)cell * nnz;
  for (int k = lane_id(); k < nnz; k += WAVE) out[k] = jac_entry(S, tb, code[k], cj);
}

template <class M> __global__ __launch_bounds__(64) PL_ONE_WAVE_PER_SIMD void k_linear_solve(const Tables* tb, int n_cells, const double* theta, const double* Y, const double* YP,
                                                     double cj, int mode, double* b) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  load_vec<M>(S.yy, Y + (size_t)cell * NST); load_vec<M>(S.yp, YP + (size_t)cell * NST); load_vec<M>(S.delta, b + (size_t)cell * NST);
  PL_SYNC();
  cell_res_jac(S, R, S.yy, S.yp, S.phi[1], mode, 0.0);
  cell_factor(S, R, tb, cj, mode, false);
  cell_solve(S, R, S.delta, mode, false);
  store_vec<M>(b + (size_t)cell * NST, S.delta);
}

template <class M> __global__ __launch_bounds__(64) PL_ONE_WAVE_PER_SIMD void k_init_consistent(const Tables* tb, int n_cells, const double* theta, int mode, double value,
                                                        double reltol_init, double* Y, double* YP, int* status, int* iters) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= n_cells) return;
  cell_setup(S, R, tb, theta + (size_t)cell * tb->P);
  load_vec<M>(S.yy, Y + (size_t)cell * NST);
  PL_SYNC();
  Counters cnt; for (int k = 0; k < 10; k++) cnt.v[k] = 0;
  PL_SYNC();
  const int rc = cell_init_consistent(S, R, tb, S.yy, S.yp, S.delta, S.phi[1], mode, value, reltol_init, cnt);
  store_vec<M>(Y + (size_t)cell * NST, S.yy); store_vec<M>(YP + (size_t)cell * NST, S.yp);
  PL_SYNC();
  if (lane_id() == 0) { if (status) status[cell] = rc; if (iters) iters[cell] = cnt.v[C_INIT]; }
}

struct IntegrateArgs {
  const Tables* tb; int n_cells; const double* theta; const double* SOC0; const double* Y_init; const double* t_init; int n_runs; const plh_run* runs; plh_opts opts;
  plh_outputs out; double* scratch;   // scratch: [n_cells][2][NST]
};

template <class M, bool TAB> __global__ __launch_bounds__(64) PL_ONE_WAVE_PER_SIMD void k_integrate(IntegrateArgs a) {
  __shared__ CellLDS<M> S;
  PL_EMU_POISON(S);
  constexpr int NST = M::NST;
  LaneRegs R;
  const int cell = blockIdx.x;
  if (cell >= a.n_cells) return;
  cell_setup(S, R, a.tb, a.theta + (size_t)cell * a.tb->P);
  Counters cnt; for (int k = 0; k < 10; k++) cnt.v[k] = 0;
#ifdef PL_PHASE_TIMERS
  if (lane_id() < 8) S.cyc[lane_id()] = 0;
#endif
  PL_SYNC();
  PL_TIC();
  CellOut co;
  const size_t off = (size_t)cell * a.out.max_pts;
  co.max_pts = a.out.max_pts;
  co.t = a.out.t ? a.out.t + off : nullptr; co.V = a.out.V ? a.out.V + off : nullptr; co.I = a.out.I ? a.out.I + off : nullptr;
  co.SOC = a.out.SOC ? a.out.SOC + off : nullptr; co.T = a.out.T_avg ? a.out.T_avg + off : nullptr;
  co.Yall = a.out.Y_all ? a.out.Y_all + off * NST : nullptr;
  cell_simulate<TAB>(S, R, a.tb, a.SOC0[cell], a.Y_init ? a.Y_init + (size_t)cell * NST : nullptr, a.t_init ? a.t_init[cell] : 0.0, a.n_runs, a.runs, a.opts, co, a.out.n_pts ? a.out.n_pts + cell : nullptr,
                a.out.run_info + (size_t)cell * a.n_runs, cnt,
                a.out.Y_final ? a.out.Y_final + (size_t)cell * NST : nullptr, a.out.YP_final ? a.out.YP_final + (size_t)cell * NST : nullptr,
                a.scratch + (size_t)cell * 2 * NST, a.scratch + (size_t)cell * 2 * NST + NST, cell);
  PL_TOC(S, PH_TOTAL);
  PL_SYNC();
  if (lane_id() == 0 && a.out.counters) {
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

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }
#define HIPCHK(x) do { hipError_t e__ = (x); if (e__ != hipSuccess) return fail(PLH_E_HIP, std::string(#x) + ": " + hipGetErrorString(e__)); } while (0)

// names of the Key enum entries (reference Symbols, UTF-8)
static const char* const KEY_ENUM_NAMES[K_COUNT] = {
    "D_n", "D_p", "D_s", "D_sn", "D_sp", "Ea_D_sn", "Ea_D_sp", "Ea_k_n", "Ea_k_p", "M_n", "R_SEI", "Rp_n", "Rp_p", "T₀", "Uref_s",
    "brugg_n", "brugg_p", "brugg_s", "c_e₀", "c_max_n", "c_max_p", "i_0_jside", "k_n", "k_n_aging", "k_p", "l_n", "l_p", "l_s",
    "t₊", "w", "θ_max_n", "θ_max_p", "θ_min_n", "θ_min_p", "ρ_n", "σ_n", "σ_p", "ϵ_fn", "ϵ_fp", "ϵ_n", "ϵ_p", "ϵ_s",
    "Cp_a", "Cp_n", "Cp_p", "Cp_s", "Cp_z", "T_amb", "h_cell", "l_a", "l_z", "λ_a", "λ_n", "λ_p", "λ_s", "λ_z",
    "ρ_a", "ρ_p", "ρ_s", "ρ_z", "σ_a", "σ_z"};
// per variant: the sorted theta_keys the reference's generated functions would receive (generate_functions.jl:327-363, 387) and
// the chemistry defaults (reference src/params.jl:5-117, 176-226 LCO/LiC6; 295-367, 436-452 NMC/LiC6_NMC)
static const char* const KEYS_LCO_ISO[] = {
    "D_n", "D_p", "D_s", "D_sn", "D_sp", "Ea_D_sn", "Ea_D_sp", "Ea_k_n", "Ea_k_p", "Rp_n", "Rp_p", "T₀", "brugg_n", "brugg_p", "brugg_s",
    "c_e₀", "c_max_n", "c_max_p", "k_n", "k_p", "l_n", "l_p", "l_s", "t₊", "θ_max_n", "θ_max_p", "θ_min_n", "θ_min_p", "σ_n", "σ_p",
    "ϵ_fn", "ϵ_fp", "ϵ_n", "ϵ_p", "ϵ_s"};
static const double DEFAULTS_LCO_ISO[] = {
    7.5e-10, 7.5e-10, 7.5e-10, 3.9e-14, 1e-14, 5000.0, 5000.0, 5000.0, 5000.0, 2e-6, 2e-6, 25 + 273.15, 4.0, 4.0, 4.0,
    1000.0, 30555.0, 51554.0, 5.0310e-11, 2.334e-11, 88e-6, 80e-6, 25e-6, 0.364, 0.85510, 0.49550, 0.01429, 0.99174, 100.0, 100.0,
    0.0326, 0.025, 0.485, 0.385, 0.724};
static const char* const KEYS_NMC_ISO[] = {
    "D_sn", "D_sp", "Ea_D_sn", "Ea_D_sp", "Ea_k_n", "Ea_k_p", "Rp_n", "Rp_p", "T₀", "brugg_n", "brugg_p", "brugg_s", "c_e₀", "c_max_n", "c_max_p",
    "k_n", "k_p", "l_n", "l_p", "l_s", "t₊", "θ_max_n", "θ_max_p", "θ_min_n", "θ_min_p", "σ_n", "σ_p", "ϵ_fn", "ϵ_fp", "ϵ_n", "ϵ_p", "ϵ_s"};
static const double DEFAULTS_NMC_ISO[] = {
    1.5e-14, 2e-14, 4e4, 2.5e4, 3e4, 3e4, 10e-6, 7.5e-6, 25 + 273.15, 1.5, 1.5, 1.5, 1200.0, 31080.0, 51830.0,
    6.3466e-10, 6.3066e-10, 48e-6, 41.6e-6, 25e-6, 0.38, 0.790813, 0.359749, 0.001, 0.955473, 100.0, 100.0, 0.038, 0.12, 0.3, 0.3, 0.4};
// aging = :SEI adds M_n, R_SEI, Uref_s, i_0_jside, k_n_aging, w, rho_n (reference src/params.jl:90,98-110; NMC borrows them)
static const char* const KEYS_LCO_SEI[] = {
    "D_n", "D_p", "D_s", "D_sn", "D_sp", "Ea_D_sn", "Ea_D_sp", "Ea_k_n", "Ea_k_p", "M_n", "R_SEI", "Rp_n", "Rp_p", "T₀", "Uref_s",
    "brugg_n", "brugg_p", "brugg_s", "c_e₀", "c_max_n", "c_max_p", "i_0_jside", "k_n", "k_n_aging", "k_p", "l_n", "l_p", "l_s",
    "t₊", "w", "θ_max_n", "θ_max_p", "θ_min_n", "θ_min_p", "ρ_n", "σ_n", "σ_p", "ϵ_fn", "ϵ_fp", "ϵ_n", "ϵ_p", "ϵ_s"};
static const double DEFAULTS_LCO_SEI[] = {
    7.5e-10, 7.5e-10, 7.5e-10, 3.9e-14, 1e-14, 5000.0, 5000.0, 5000.0, 5000.0, 7.3e-4, 0.01, 2e-6, 2e-6, 25 + 273.15, 0.4,
    4.0, 4.0, 4.0, 1000.0, 30555.0, 51554.0, 1.5e-6, 5.0310e-11, 1.0, 2.334e-11, 88e-6, 80e-6, 25e-6,
    0.364, 2.0, 0.85510, 0.49550, 0.01429, 0.99174, 2500.0, 100.0, 100.0, 0.0326, 0.025, 0.485, 0.385, 0.724};
static const char* const KEYS_NMC_SEI[] = {
    "D_sn", "D_sp", "Ea_D_sn", "Ea_D_sp", "Ea_k_n", "Ea_k_p", "M_n", "R_SEI", "Rp_n", "Rp_p", "T₀", "Uref_s", "brugg_n", "brugg_p", "brugg_s",
    "c_e₀", "c_max_n", "c_max_p", "i_0_jside", "k_n", "k_n_aging", "k_p", "l_n", "l_p", "l_s", "t₊", "w", "θ_max_n", "θ_max_p", "θ_min_n",
    "θ_min_p", "ρ_n", "σ_n", "σ_p", "ϵ_fn", "ϵ_fp", "ϵ_n", "ϵ_p", "ϵ_s"};
static const double DEFAULTS_NMC_SEI[] = {
    1.5e-14, 2e-14, 4e4, 2.5e4, 3e4, 3e4, 7.3e-4, 0.01, 10e-6, 7.5e-6, 25 + 273.15, 0.4, 1.5, 1.5, 1.5,
    1200.0, 31080.0, 51830.0, 1.5e-6, 6.3466e-10, 1.0, 6.3066e-10, 48e-6, 41.6e-6, 25e-6, 0.38, 2.0, 0.790813, 0.359749, 0.001,
    0.955473, 2500.0, 100.0, 100.0, 0.038, 0.12, 0.3, 0.3, 0.4};
// temperature = true adds the heat-equation parameters (reference src/params.jl:40-45, 85-96, 200-226)
static const char* const KEYS_LCO_THERMAL[] = {
    "Cp_a", "Cp_n", "Cp_p", "Cp_s", "Cp_z", "D_n", "D_p", "D_s", "D_sn", "D_sp", "Ea_D_sn", "Ea_D_sp", "Ea_k_n", "Ea_k_p", "Rp_n", "Rp_p",
    "T_amb", "T₀", "brugg_n", "brugg_p", "brugg_s", "c_e₀", "c_max_n", "c_max_p", "h_cell", "k_n", "k_p", "l_a", "l_n", "l_p", "l_s", "l_z",
    "t₊", "θ_max_n", "θ_max_p", "θ_min_n", "θ_min_p", "λ_a", "λ_n", "λ_p", "λ_s", "λ_z", "ρ_a", "ρ_n", "ρ_p", "ρ_s", "ρ_z",
    "σ_a", "σ_n", "σ_p", "σ_z", "ϵ_fn", "ϵ_fp", "ϵ_n", "ϵ_p", "ϵ_s"};
static const double DEFAULTS_LCO_THERMAL[] = {
    897.0, 700.0, 700.0, 700.0, 385.0, 7.5e-10, 7.5e-10, 7.5e-10, 3.9e-14, 1e-14, 5000.0, 5000.0, 5000.0, 5000.0, 2e-6, 2e-6,
    25 + 273.15, 25 + 273.15, 4.0, 4.0, 4.0, 1000.0, 30555.0, 51554.0, 1.0, 5.0310e-11, 2.334e-11, 10e-6, 88e-6, 80e-6, 25e-6, 10e-6,
    0.364, 0.85510, 0.49550, 0.01429, 0.99174, 237.0, 1.7, 2.1, 0.16, 401.0, 2700.0, 2500.0, 2500.0, 1100.0, 8940.0,
    3.55e7, 100.0, 100.0, 5.96e7, 0.0326, 0.025, 0.485, 0.385, 0.724};
struct VariantInfo { int chem, sei, thermal, nkeys; const char* const* keys; const double* defaults; };
enum { V_LCO_ISO = 0, V_NMC_ISO = 1, V_LCO_SEI = 2, V_NMC_SEI = 3, V_LCO_THERMAL = 4, V_COUNT };
static const VariantInfo VARIANTS[V_COUNT] = {
    {PLH_CHEM_LCO_LIC6, 0, 0, 35, KEYS_LCO_ISO, DEFAULTS_LCO_ISO},
    {PLH_CHEM_NMC_LIC6, 0, 0, 32, KEYS_NMC_ISO, DEFAULTS_NMC_ISO},
    {PLH_CHEM_LCO_LIC6, 1, 0, 42, KEYS_LCO_SEI, DEFAULTS_LCO_SEI},
    {PLH_CHEM_NMC_LIC6, 1, 0, 39, KEYS_NMC_SEI, DEFAULTS_NMC_SEI},
    {PLH_CHEM_LCO_LIC6, 0, 1, 56, KEYS_LCO_THERMAL, DEFAULTS_LCO_THERMAL},
};

struct StageBlock { void* p; size_t bytes; bool busy; };
struct plh_model_s {
  plh_model_desc desc;
  int variant = 0;                 // index into the instantiated ModelT<> list (PL_DISPATCH)
  int N = 0, Nd = 0, P = 0;        // states, differential states, theta entries
  const char* const* key_names = nullptr; const double* key_defaults = nullptr;
  Tables h_tb;
  Tables* d_tb = nullptr;
  std::vector<int> colptr[PLH_N_MODES], rowval[PLH_N_MODES];
  std::vector<unsigned> code[PLH_N_MODES];
  unsigned* d_code[PLH_N_MODES] = {};
  double* scratch = nullptr; size_t scratch_cells = 0;
  plh_run* d_runs = nullptr; int runs_cap = 0;
  std::vector<plh_run> runs_on_device;   // the protocol currently in d_runs (a repeated launch with the same protocol uploads nothing and does not synchronise)
  hipEvent_t ev0 = nullptr, ev1 = nullptr; bool timed = false;
  // device staging blocks of the host-pointer (PLH_HOST) path, kept between calls: a repeated call with the same shapes does no hipMalloc / hipFree
  std::vector<StageBlock> stage_cache;
  void* grab(size_t bytes) {
    int best = -1;
    for (size_t k = 0; k < stage_cache.size(); k++)
      if (!stage_cache[k].busy && stage_cache[k].bytes >= bytes && (best < 0 || stage_cache[k].bytes < stage_cache[best].bytes)) best = (int)k;
    if (best >= 0 && stage_cache[best].bytes <= 2 * bytes + 4096) { stage_cache[best].busy = true; return stage_cache[best].p; }
    void* d = nullptr; if (hipMalloc(&d, bytes ? bytes : 8) != hipSuccess) return nullptr;
    stage_cache.push_back({d, bytes, true}); return d;
  }
  void release(void* p) { for (auto& b : stage_cache) if (b.p == p) b.busy = false; }
  // pinned host bounce buffer for the device-to-host copies of the PLH_HOST path: hipMemcpy into fresh pageable pages (a caller that allocates
  // its output arrays per call) costs ~10x the DMA time in page pinning; DMA into this buffer + memcpy does not
  void* pin = nullptr; size_t pin_bytes = 0;
  void* pinned(size_t bytes) {
    if (bytes <= pin_bytes) return pin;
    if (pin) hipHostFree(pin);
    pin = nullptr; pin_bytes = 0;
    if (hipHostMalloc(&pin, bytes, hipHostMallocDefault) != hipSuccess) { pin = nullptr; return nullptr; }
    pin_bytes = bytes; return pin;
  }
  // thermal models: plh_integrate is served by the sibling library (same source, built at -O2: the -O3 pipeline over-unrolls the 4x4 block code of
  // the thermal kernels, C3 97 k -> 115 k trajectories/s); absent sibling = this library's own kernels
  void* sib_lib = nullptr; plh_model_t sib = nullptr; bool sib_last = false;
  int (*sib_integrate)(plh_model_t, int, const double*, const double*, const double*, const double*, int, const plh_run*, const plh_opts*, const plh_outputs*, int, void*) = nullptr;
  double (*sib_kernel_ms)(plh_model_t) = nullptr; void (*sib_destroy)(plh_model_t) = nullptr; const char* (*sib_error)(void) = nullptr;
};

// decode word of the structural Jacobian entry (r, c), 0 if structurally zero
template <class M>
static unsigned classify(const Tables& tb, int mode, int r, int c) {
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
    if (r >= O_CS && r < N_CECS && cT) { const int p = (r - O_CS) / NR; return ct == NA + node_of_j(p) ? W(TT_CS_T, p, (r - O_CS) % NR, 0) : 0; }
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
      if (c == O_CS + jx * NR + NR - 1) return W(TT_T_CS, jx, 0, 0);
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
  if (r < N_CECS) {                                 // c_s row (p, rr)
    const int p = (r - O_CS) / NR, rr = (r - O_CS) % NR;
    if (c >= O_CS && c < N_CECS && (c - O_CS) / NR == p) { const int cc = (c - O_CS) % NR; return (tb.M[rr * NR + cc] != 0.0 || rr == cc) ? W(JT_CS_CS, p, rr, cc) : 0; }
    if (rr == NR - 1 && c == O_J + p) return W(JT_CS_J, p, 0, 0);
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
    if (c == O_CS + jx * NR + NR - 1) return W(JT_J_CS, jx, 0, 0);
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

// ---- staging helpers: host arrays are copied through temporary device buffers ----
struct Stage {
  std::vector<void*> tmp;
  plh_model_s* m; int kind; hipStream_t st;
  Stage(plh_model_s* mm, int k, void* s) : m(mm), kind(k), st((hipStream_t)s) {}
  ~Stage() { for (void* p : tmp) m->release(p); }
  template <class T> const T* in(const T* p, size_t n) {
    if (!p || kind == PLH_DEVICE) return p;
    void* d = m->grab(n * sizeof(T)); if (!d) return nullptr;
    tmp.push_back(d); hipMemcpy(d, p, n * sizeof(T), hipMemcpyHostToDevice); return (const T*)d;
  }
  template <class T> T* buf(T* p, size_t n, bool copy_in) {
    if (!p || kind == PLH_DEVICE) return p;
    void* d = m->grab(n * sizeof(T)); if (!d) return nullptr;
    tmp.push_back(d); if (copy_in) hipMemcpy(d, p, n * sizeof(T), hipMemcpyHostToDevice); return (T*)d;
  }
  template <class T> void back(T* host, const T* dev, size_t n) {
    if (!host || kind == PLH_DEVICE) return;
    void* pb = n * sizeof(T) >= (64u << 10) ? m->pinned(n * sizeof(T)) : nullptr;
    if (pb) { hipMemcpy(pb, dev, n * sizeof(T), hipMemcpyDeviceToHost); memcpy(host, pb, n * sizeof(T)); }
    else hipMemcpy(host, dev, n * sizeof(T), hipMemcpyDeviceToHost);
  }
};
#define CHECK_MODEL(m) do { if (!(m)) return fail(PLH_E_ARG, "null model"); } while (0)
#define CHECK_MODE(mode) do { if ((mode) != PLH_MODE_I && (mode) != PLH_MODE_V && (mode) != PLH_MODE_P && (mode) != PLH_MODE_ETA_P && !((mode) == PLH_MODE_DT && m->desc.temperature)) \
    return fail(PLH_E_UNSUPPORTED, "operating mode not available for this model (I, V, P, eta_p; dT with temperature = true)"); } while (0)
#define FINISH(stage) do { if ((stage).kind != PLH_DEVICE) HIPCHK(hipStreamSynchronize((stage).st)); HIPCHK(hipGetLastError()); } while (0)


// instantiated model variants
// (PL_ONLY_THERMAL: the sibling library libpetlion_hip_thermal.so instantiates the thermal variant only, see plh_model_create)
#ifdef PL_ONLY_THERMAL
#define PL_DISPATCH_ISOTHERMAL(...)
#else
#define PL_DISPATCH_ISOTHERMAL(...) \
    case V_LCO_ISO: { using M = ModelT<PLH_CHEM_LCO_LIC6, false>; __VA_ARGS__; } break; \
    case V_NMC_ISO: { using M = ModelT<PLH_CHEM_NMC_LIC6, false>; __VA_ARGS__; } break; \
    case V_LCO_SEI: { using M = ModelT<PLH_CHEM_LCO_LIC6, true>; __VA_ARGS__; } break; \
    case V_NMC_SEI: { using M = ModelT<PLH_CHEM_NMC_LIC6, true>; __VA_ARGS__; } break;
#endif
#ifdef PL_ONLY_LCO_ISO     /* build experiments (tools/flag_search.sh): one variant, short compile */
#undef PL_DISPATCH_ISOTHERMAL
#define PL_DISPATCH_ISOTHERMAL(...) case V_LCO_ISO: { using M = ModelT<PLH_CHEM_LCO_LIC6, false>; __VA_ARGS__; } break;
#define PL_DISPATCH_THERMAL(...)
#else
#define PL_DISPATCH_THERMAL(...) case V_LCO_THERMAL: { using M = ModelT<PLH_CHEM_LCO_LIC6, false, true>; __VA_ARGS__; } break;
#endif
#define PL_DISPATCH(m, ...) do { switch ((m)->variant) { \
    PL_DISPATCH_ISOTHERMAL(__VA_ARGS__) \
    PL_DISPATCH_THERMAL(__VA_ARGS__) \
    default: return fail(PLH_E_UNSUPPORTED, "model variant not instantiated"); } } while (0)

template <class M> static int build_patterns(plh_model_s* m) {
  Tables& tb = m->h_tb;
  for (int mode = 0; mode < PLH_N_MODES; mode++) {
    if (mode == PLH_MODE_DT && !M::THERMAL) continue;
    m->colptr[mode].assign(M::NST + 1, 0);
    for (int c = 0; c < M::NST; c++) {
      for (int r = 0; r < M::NST; r++) { const unsigned w = classify<M>(tb, mode, r, c); if (w) { m->rowval[mode].push_back(r); m->code[mode].push_back(w); } }
      m->colptr[mode][c + 1] = (int)m->rowval[mode].size();
    }
    tb.nnz[mode] = (int)m->rowval[mode].size();
    if (hipMalloc((void**)&m->d_code[mode], m->code[mode].size() * sizeof(unsigned)) != hipSuccess) return PLH_E_HIP;
    hipMemcpy(m->d_code[mode], m->code[mode].data(), m->code[mode].size() * sizeof(unsigned), hipMemcpyHostToDevice);
    tb.csc_code[mode] = m->d_code[mode];
  }
  m->N = M::NST; m->Nd = M::NDIFF;
  return 0;
}

struct SectionInfo { const char* name; int start, len; };
template <class M>
static int sections_of(SectionInfo* o) {
  int k = 0;
  o[k++] = {"c_e", O_CE, NE}; o[k++] = {"c_s_avg", O_CS, NJ * NR};
  if (M::THERMAL) o[k++] = {"T", M::O_T, NT};
  if (M::SEI) { o[k++] = {"film", M::O_FILM, NN}; o[k++] = {"SOH", M::O_SOH, 1}; }
  o[k++] = {"j", M::O_J, NJ}; o[k++] = {"Φ_e", M::O_PE, NE}; o[k++] = {"Φ_s", M::O_PS, NJ};
  if (M::SEI) o[k++] = {"j_s", M::O_JS, NN};
  o[k++] = {"I", M::O_I, 1};
  return k;
}

extern "C" {

const char* plh_last_error(void) { return g_err.c_str(); }

int plh_model_create(const plh_model_desc* d, plh_model_t* out) {
  if (!d || !out) return fail(PLH_E_ARG, "null argument");
  if (d->real_bytes != 8) return fail(PLH_E_UNSUPPORTED, "only fp64 (real_bytes = 8) is implemented");
  if (d->chemistry != PLH_CHEM_LCO_LIC6 && d->chemistry != PLH_CHEM_NMC_LIC6) return fail(PLH_E_UNSUPPORTED, "unknown chemistry");
  int variant = -1;
  for (int v = 0; v < V_COUNT; v++)
    if (VARIANTS[v].chem == d->chemistry && VARIANTS[v].sei == (d->aging_SEI ? 1 : 0) && VARIANTS[v].thermal == (d->temperature ? 1 : 0)) variant = v;
  if (variant < 0) return fail(PLH_E_UNSUPPORTED, "this chemistry / temperature / aging combination is not instantiated on the device (built: LCO and NMC "
                                                  "isothermal with or without SEI aging, LCO with temperature)");
  if (d->temperature && (d->N_a != NA || d->N_z != NZ)) return fail(PLH_E_UNSUPPORTED, "discretisation: only N_a = N_z = 10 is instantiated");
  if (d->N_p != NP || d->N_s != NS || d->N_n != NN || d->N_r_p != NR || d->N_r_n != NR)
    return fail(PLH_E_UNSUPPORTED, "discretisation: only N_p = N_s = N_n = N_r_p = N_r_n = 10 is instantiated");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) return fail(PLH_E_HIP, "no HIP device visible: the product path has no CPU fallback");
  plh_model_s* m = new plh_model_s();
  m->desc = *d;
  Tables& tb = m->h_tb;
  memset(&tb, 0, sizeof(tb));
  memcpy(tb.M, PL_RADIAL_M, sizeof(tb.M)); memcpy(tb.LAM, PL_RADIAL_LAM, sizeof(tb.LAM));
  memcpy(tb.V, PL_RADIAL_V, sizeof(tb.V)); memcpy(tb.W, PL_RADIAL_W, sizeof(tb.W));
  tb.BJ = PL_RADIAL_BJ_FACTOR; tb.chem = d->chemistry;
  m->variant = variant; m->P = VARIANTS[variant].nkeys; m->key_names = VARIANTS[variant].keys; m->key_defaults = VARIANTS[variant].defaults;
  tb.P = m->P;
  for (int k = 0; k < K_COUNT; k++) {
    tb.thidx[k] = -1;
    for (int q = 0; q < m->P; q++) if (!strcmp(KEY_ENUM_NAMES[k], m->key_names[q])) tb.thidx[k] = q;
  }
  { int rc = 0; PL_DISPATCH(m, rc = build_patterns<M>(m)); if (rc != 0) { delete m; return fail(rc, "pattern construction failed"); } }
  if (hipMalloc((void**)&m->d_tb, sizeof(Tables)) != hipSuccess) { delete m; return fail(PLH_E_HIP, "hipMalloc failed"); }
  hipMemcpy(m->d_tb, &tb, sizeof(Tables), hipMemcpyHostToDevice);
  hipEventCreate(&m->ev0); hipEventCreate(&m->ev1);
#if defined(PL_THERMAL_SIBLING) && !defined(PL_WAVE_EMU)
  if (d->temperature && !getenv("PETLION_HIP_NO_SIBLING")) {
    Dl_info self;
    if (dladdr((void*)&plh_model_create, &self) && self.dli_fname) {
      std::string path(self.dli_fname);
      const size_t slash = path.find_last_of('/');
      path = (slash == std::string::npos ? std::string() : path.substr(0, slash + 1)) + PL_THERMAL_SIBLING;
      if (void* lib = dlopen(path.c_str(), RTLD_NOW | RTLD_LOCAL)) {
        auto create = (int (*)(const plh_model_desc*, plh_model_t*))dlsym(lib, "plh_model_create");
        m->sib_integrate = (decltype(m->sib_integrate))dlsym(lib, "plh_integrate");
        m->sib_kernel_ms = (decltype(m->sib_kernel_ms))dlsym(lib, "plh_last_kernel_ms");
        m->sib_destroy = (decltype(m->sib_destroy))dlsym(lib, "plh_model_destroy");
        m->sib_error = (decltype(m->sib_error))dlsym(lib, "plh_last_error");
        if (create && m->sib_integrate && m->sib_kernel_ms && m->sib_destroy && m->sib_error && create(d, &m->sib) == 0) m->sib_lib = lib;
        else { m->sib = nullptr; dlclose(lib); }
      }
    }
  }
#endif
  *out = m;
  return 0;
}

void plh_model_destroy(plh_model_t m) {
  if (!m) return;
#ifndef PL_WAVE_EMU
  if (m->sib) { m->sib_destroy(m->sib); dlclose(m->sib_lib); }
#endif
  for (int k = 0; k < PLH_N_MODES; k++) if (m->d_code[k]) hipFree(m->d_code[k]);
  if (m->d_tb) hipFree(m->d_tb);
  if (m->scratch) hipFree(m->scratch);
  if (m->d_runs) hipFree(m->d_runs);
  for (auto& b : m->stage_cache) hipFree(b.p);
  if (m->pin) hipHostFree(m->pin);
  if (m->ev0) hipEventDestroy(m->ev0);
  if (m->ev1) hipEventDestroy(m->ev1);
  delete m;
}

int plh_n_states(plh_model_t m) { return m ? m->N : PLH_E_ARG; }
int plh_n_diff(plh_model_t m) { return m ? m->Nd : PLH_E_ARG; }
int plh_n_theta(plh_model_t m) { return m ? m->P : PLH_E_ARG; }
const char* plh_theta_key(plh_model_t m, int i) { return (m && i >= 0 && i < m->P) ? m->key_names[i] : nullptr; }
double plh_theta_default(plh_model_t m, int i) { return (m && i >= 0 && i < m->P) ? m->key_defaults[i] : NAN; }

int plh_n_sections(plh_model_t m) { if (!m) return PLH_E_ARG; SectionInfo s[12]; int n = 0; PL_DISPATCH(m, n = sections_of<M>(s)); return n; }
int plh_section(plh_model_t m, int i, const char** name, int* start, int* len) {
  if (!m) return fail(PLH_E_ARG, "null model");
  SectionInfo s[12]; int n = 0; PL_DISPATCH(m, n = sections_of<M>(s));
  if (i < 0 || i >= n) return fail(PLH_E_ARG, "section index out of range");
  if (name) *name = s[i].name; if (start) *start = s[i].start; if (len) *len = s[i].len;
  return 0;
}

int plh_jac_pattern(plh_model_t m, int mode, int* nnz, int* colptr, int* rowval) {
  if (!m || !nnz) return fail(PLH_E_ARG, "null argument");
  if (mode < 0 || mode >= PLH_N_MODES || m->rowval[mode].empty()) return fail(PLH_E_UNSUPPORTED, "mode not available for this model");
  *nnz = (int)m->rowval[mode].size();
  if (colptr) memcpy(colptr, m->colptr[mode].data(), (m->N + 1) * sizeof(int));
  if (rowval) memcpy(rowval, m->rowval[mode].data(), m->rowval[mode].size() * sizeof(int));
  return 0;
}

int plh_initial_guess(plh_model_t m, int n, const double* theta, const double* SOC, double* Y, int kind, void* stream) {
  CHECK_MODEL(m); if (n <= 0 || !theta || !SOC || !Y) return fail(PLH_E_ARG, "bad argument");
  Stage s(m, kind, stream);
  const double* th = s.in(theta, (size_t)n * m->P); const double* so = s.in(SOC, n); double* y = s.buf(Y, (size_t)n * m->N, false);
  PL_DISPATCH(m, PL_LAUNCH(k_initial_guess<M>, n, WAVE, s.st, m->d_tb, n, th, so, y));
  FINISH(s); s.back(Y, y, (size_t)n * m->N);
  return 0;
}

int plh_residual(plh_model_t m, int n, const double* theta, const double* Y, const double* YP, int mode, double value, double* F, int kind, void* stream) {
  CHECK_MODEL(m); CHECK_MODE(mode); if (n <= 0 || !theta || !Y || !YP || !F) return fail(PLH_E_ARG, "bad argument");
  Stage s(m, kind, stream);
  const double* th = s.in(theta, (size_t)n * m->P); const double* y = s.in(Y, (size_t)n * m->N); const double* yp = s.in(YP, (size_t)n * m->N);
  double* f = s.buf(F, (size_t)n * m->N, false);
  PL_DISPATCH(m, PL_LAUNCH(k_residual<M>, n, WAVE, s.st, m->d_tb, n, th, y, yp, mode, value, f));
  FINISH(s); s.back(F, f, (size_t)n * m->N);
  return 0;
}

int plh_jacobian(plh_model_t m, int n, const double* theta, const double* Y, const double* YP, double cj, int mode, double* nzval, int kind, void* stream) {
  CHECK_MODEL(m); CHECK_MODE(mode); if (n <= 0 || !theta || !Y || !YP || !nzval) return fail(PLH_E_ARG, "bad argument");
  Stage s(m, kind, stream);
  const size_t nnz = m->rowval[mode].size();
  const double* th = s.in(theta, (size_t)n * m->P); const double* y = s.in(Y, (size_t)n * m->N); const double* yp = s.in(YP, (size_t)n * m->N);
  double* z = s.buf(nzval, (size_t)n * nnz, false);
  PL_DISPATCH(m, PL_LAUNCH(k_jacobian<M>, n, WAVE, s.st, m->d_tb, n, th, y, yp, cj, mode, z));
  FINISH(s); s.back(nzval, z, (size_t)n * nnz);
  return 0;
}

int plh_linear_solve(plh_model_t m, int n, const double* theta, const double* Y, const double* YP, double cj, int mode, double* b, int kind, void* stream) {
  CHECK_MODEL(m); CHECK_MODE(mode); if (n <= 0 || !theta || !Y || !YP || !b) return fail(PLH_E_ARG, "bad argument");
  Stage s(m, kind, stream);
  const double* th = s.in(theta, (size_t)n * m->P); const double* y = s.in(Y, (size_t)n * m->N); const double* yp = s.in(YP, (size_t)n * m->N);
  double* bb = s.buf(b, (size_t)n * m->N, true);
  PL_DISPATCH(m, PL_LAUNCH(k_linear_solve<M>, n, WAVE, s.st, m->d_tb, n, th, y, yp, cj, mode, bb));
  FINISH(s); s.back(b, bb, (size_t)n * m->N);
  return 0;
}

int plh_init_consistent(plh_model_t m, int n, const double* theta, int mode, double value, double reltol_init, double* Y, double* YP, int* status,
                        int* iters, int kind, void* stream) {
  CHECK_MODEL(m); CHECK_MODE(mode); if (n <= 0 || !theta || !Y || !YP) return fail(PLH_E_ARG, "bad argument");
  Stage s(m, kind, stream);
  const double* th = s.in(theta, (size_t)n * m->P);
  double* y = s.buf(Y, (size_t)n * m->N, true); double* yp = s.buf(YP, (size_t)n * m->N, false);
  int* st = s.buf(status, n, false); int* it = s.buf(iters, n, false);
  PL_DISPATCH(m, PL_LAUNCH(k_init_consistent<M>, n, WAVE, s.st, m->d_tb, n, th, mode, value, reltol_init, y, yp, st, it));
  FINISH(s);
  s.back(Y, y, (size_t)n * m->N); s.back(YP, yp, (size_t)n * m->N); s.back(status, st, n); s.back(iters, it, n);
  return 0;
}

int plh_integrate(plh_model_t m, int n, const double* theta, const double* SOC0, const double* Y_init, const double* t_init, int n_runs,
                  const plh_run* runs, const plh_opts* opts, const plh_outputs* out, int kind, void* stream) {
  CHECK_MODEL(m);
  if (m->sib) {                                           // thermal: the sibling library's kernels (see plh_model_s)
    const int rc = m->sib_integrate(m->sib, n, theta, SOC0, Y_init, t_init, n_runs, runs, opts, out, kind, stream);
    m->sib_last = true;
    return rc == 0 ? 0 : fail(rc, m->sib_error());
  }
  m->sib_last = false;
  if (n <= 0 || !theta || !SOC0 || n_runs <= 0 || !runs || !opts || !out || !out->run_info) return fail(PLH_E_ARG, "bad argument");
  for (int r = 0; r < n_runs; r++) {
    CHECK_MODE(runs[r].mode);
    if (runs[r].value_kind < 0 || runs[r].value_kind > PLH_VAL_TABLE) return fail(PLH_E_ARG, "bad value_kind");
    if (runs[r].value_kind == PLH_VAL_TABLE) {
      if (runs[r].n_tab < 1 || !runs[r].tab_t || !runs[r].tab_v) return fail(PLH_E_ARG, "PLH_VAL_TABLE needs n_tab >= 1 and both table arrays");
      if (runs[r].mode == PLH_MODE_DT) return fail(PLH_E_UNSUPPORTED, "time-dependent dT inputs are not defined by the reference");
      for (int k = 1; k < runs[r].n_tab; k++) if (!(runs[r].tab_t[k] >= runs[r].tab_t[k - 1])) return fail(PLH_E_ARG, "table times must be non-decreasing");
    }
    if (!(runs[r].tf > 0)) return fail(PLH_E_ARG, "run length tf must be positive");
    if (runs[r].value_cell && runs[r].value_kind != PLH_VAL_CONST) return fail(PLH_E_ARG, "value_cell needs PLH_VAL_CONST");
    if (runs[r].tf_cell) for (int c = 0; c < n; c++) if (!(runs[r].tf_cell[c] > 0)) return fail(PLH_E_ARG, "run length tf_cell must be positive");
  }
  if (out->max_pts < 0) return fail(PLH_E_ARG, "max_pts");
  if ((Y_init == nullptr) != (t_init == nullptr)) return fail(PLH_E_ARG, "Y_init and t_init must be given together");
  Stage s(m, kind, stream);
  if (m->scratch_cells < (size_t)n) {
    if (m->scratch) hipFree(m->scratch);
    HIPCHK(hipMalloc((void**)&m->scratch, (size_t)n * 2 * m->N * sizeof(double)));
    m->scratch_cells = n;
  }
  IntegrateArgs a;
  a.tb = m->d_tb; a.n_cells = n; a.n_runs = n_runs; a.opts = *opts; a.scratch = m->scratch;
  a.theta = s.in(theta, (size_t)n * m->P); a.SOC0 = s.in(SOC0, n);
  a.Y_init = s.in(Y_init, (size_t)n * m->N); a.t_init = s.in(t_init, n);
  // the protocol is always host memory
  if (m->runs_cap < n_runs) { if (m->d_runs) hipFree(m->d_runs); HIPCHK(hipMalloc((void**)&m->d_runs, n_runs * sizeof(plh_run))); m->runs_cap = n_runs; }
  std::vector<plh_run> hruns(runs, runs + n_runs);                    // tables are host arrays: stage them and patch the device copies
  for (int r = 0; r < n_runs; r++) {
    if (hruns[r].value_kind == PLH_VAL_TABLE) {
      Stage hs(m, PLH_HOST, stream);
      const double* dt_ = hs.in(runs[r].tab_t, runs[r].n_tab); const double* dv_ = hs.in(runs[r].tab_v, runs[r].n_tab);
      if (!dt_ || !dv_) return fail(PLH_E_HIP, "hipMalloc failed (input table)");
      hruns[r].tab_t = dt_; hruns[r].tab_v = dv_;
      s.tmp.insert(s.tmp.end(), hs.tmp.begin(), hs.tmp.end()); hs.tmp.clear();     // freed with the call's other staging buffers
    } else { hruns[r].n_tab = 0; hruns[r].tab_t = nullptr; hruns[r].tab_v = nullptr; }
    if (runs[r].value_cell || runs[r].tf_cell) {                       // per-cell protocol values: host arrays like the protocol
      Stage hs(m, PLH_HOST, stream);
      if (runs[r].value_cell) { hruns[r].value_cell = hs.in(runs[r].value_cell, n); if (!hruns[r].value_cell) return fail(PLH_E_HIP, "hipMalloc failed (value_cell)"); }
      if (runs[r].tf_cell) { hruns[r].tf_cell = hs.in(runs[r].tf_cell, n); if (!hruns[r].tf_cell) return fail(PLH_E_HIP, "hipMalloc failed (tf_cell)"); }
      s.tmp.insert(s.tmp.end(), hs.tmp.begin(), hs.tmp.end()); hs.tmp.clear();
    }
  }
  bool plain = true;                                                 // no staged arrays behind the descriptors
  {
    for (int r = 0; r < n_runs; r++) plain = plain && !hruns[r].tab_t && !hruns[r].value_cell && !hruns[r].tf_cell;
    const bool same = plain && (int)m->runs_on_device.size() == n_runs && memcmp(m->runs_on_device.data(), hruns.data(), n_runs * sizeof(plh_run)) == 0;
    if (!same) {
      HIPCHK(hipStreamSynchronize(s.st));                               // an earlier launch on this stream may still be reading d_runs
      HIPCHK(hipMemcpy(m->d_runs, hruns.data(), n_runs * sizeof(plh_run), hipMemcpyHostToDevice));
      if (plain) m->runs_on_device = hruns; else m->runs_on_device.clear();
    }
  }
  a.runs = m->d_runs;
  const size_t np = (size_t)n * out->max_pts;
  a.out = *out;
  a.out.t = s.buf(out->t, np, false); a.out.V = s.buf(out->V, np, false); a.out.I = s.buf(out->I, np, false);
  a.out.SOC = s.buf(out->SOC, np, false); a.out.T_avg = s.buf(out->T_avg, np, false); a.out.n_pts = s.buf(out->n_pts, n, false);
  a.out.Y_all = s.buf(out->Y_all, np * m->N, false);
  a.out.Y_final = s.buf(out->Y_final, (size_t)n * m->N, false); a.out.YP_final = s.buf(out->YP_final, (size_t)n * m->N, false);
  a.out.run_info = s.buf(out->run_info, (size_t)n * n_runs, false); a.out.counters = s.buf(out->counters, n, false);
  hipEventRecord(m->ev0, s.st);
  bool tabular = opts->n_tdiscon > 0 || out->Y_all;     // the general instantiation also carries the per-step state dump (outputs = :all)
  for (int r = 0; r < n_runs; r++) tabular = tabular || runs[r].value_kind == PLH_VAL_TABLE;
  if (tabular) PL_DISPATCH(m, PL_LAUNCH((k_integrate<M, true>), n, WAVE, s.st, a));
  else PL_DISPATCH(m, PL_LAUNCH((k_integrate<M, false>), n, WAVE, s.st, a));
  hipEventRecord(m->ev1, s.st);
  m->timed = true;
  FINISH(s);
  if (!plain && kind == PLH_DEVICE) HIPCHK(hipStreamSynchronize(s.st));   // staged tables / per-cell values are released below: the kernel must be done with them
  s.back(out->t, a.out.t, np); s.back(out->V, a.out.V, np); s.back(out->I, a.out.I, np); s.back(out->SOC, a.out.SOC, np);
  s.back(out->T_avg, a.out.T_avg, np); s.back(out->n_pts, a.out.n_pts, n); s.back(out->Y_all, a.out.Y_all, np * m->N);
  s.back(out->Y_final, a.out.Y_final, (size_t)n * m->N); s.back(out->YP_final, a.out.YP_final, (size_t)n * m->N);
  s.back(out->run_info, a.out.run_info, (size_t)n * n_runs); s.back(out->counters, a.out.counters, n);
  return 0;
}

double plh_last_kernel_ms(plh_model_t m) {
  if (m && m->sib && m->sib_last) return m->sib_kernel_ms(m->sib);
  if (!m || !m->timed) return -1.0;
  if (hipEventSynchronize(m->ev1) != hipSuccess) return -1.0;
  float ms = -1.f;
  if (hipEventElapsedTime(&ms, m->ev0, m->ev1) != hipSuccess) return -1.0;
  return (double)ms;
}

}  // extern "C"

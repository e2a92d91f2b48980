// dfn_cell.h -- device functions: one 64-lane wavefront owns one cell; the whole cell state lives in LDS.
//
// Replaces, for ensembles of independent cells, the reference's generated evaluators and its IDA+KLU stack:
//   residual      f_diff!/f_alg!/scalar_residual!   (reference src/physics_equations/residuals.jl, scalar_residual.jl:167-172,558-583)
//   Jacobian      J_y! / J_y_alg!                    (reference src/generate_functions.jl:289-325)  -- hand-derived, structured
//   linear solve  KLU                                (reference src/model_evaluation.jl:271,417-428) -- particle resolvent + block-Thomas
//   init Newton   newtons_method!                    (reference src/model_evaluation.jl:430-480)
//   integrator    Sundials IDA via step!             (reference src/model_evaluation.jl:234-333)
//   stop logic    check_simulation_stop! etc.        (reference src/checks.jl:1-249, src/model_evaluation.jl:369-382)
//
// Discretisation handled: N_p = N_s = N_n = 10, N_r_p = N_r_n = 10 (the reference defaults, src/params.jl:124-136),
// isothermal, LCO/LiC6 or NMC/LiC6, with or without SEI aging (configs C1/C2/C4/C5 of SURVEY.md 8d).  Finite-volume rows are evaluated in conservative
// edge-flux form (row i = flux_i - flux_{i-1}), which is algebraically identical to the reference's matrix form
// (residuals.jl:6-106, 554-654) and lets 29 lanes own the 29 control-volume edges.
//
// Vocabulary kept deliberately small (threadIdx, __shfl_*, __syncthreads, __shared__) so that the identical source
// also compiles against tests/wave_emu (a test-only lock-step wave emulator used for CPU debugging).
#pragma once
#include <hip/hip_runtime.h>
#include <type_traits>
#include "../../include/petlion_hip.h"

namespace pl {

// Every device function is inlined into its kernel.  The product build does it late (`inline` here + -mllvm -amdgpu-function-calls=false: the AMDGPU
// always-inline pass after the function-level optimisations; 0 B of scratch in the isothermal integrate kernel); tools/experiments/build_modes.py also builds
// "early" (-DPL_DEV='__device__ __forceinline__', no flag) and, through PL_DEV_FACTOR, the thermal factorisation as a real function (-DPL_FACTOR_CALL).
#ifndef PL_DEV
#define PL_DEV __device__ inline
#endif
#if defined(PL_FACTOR_CALL) && !defined(PL_WAVE_EMU)
#define PL_DEV_FACTOR __device__ __attribute__((noinline))
#else
#define PL_DEV_FACTOR PL_DEV
#endif

constexpr int WAVE = 64;
// Discretisation (reference src/params.jl:119-136: N_p, N_s, N_n, N_r_p, N_r_n, N_a, N_z): compile-time constants of a translation unit.  The library's built-in
// variants use the reference default, 10 everywhere; another grid is one more build of variant_tu.hip with -DPL_NP=.. -DPL_NS=.. (and `pl` renamed per grid, so that
// the two builds can live in one process), loaded through plh_register_grid_library (petlion.jl_amd/grids.py drives it; DESIGN.md "other discretisations").
#ifndef PL_NP
#define PL_NP 10
#endif
#ifndef PL_NS
#define PL_NS 10
#endif
#ifndef PL_NN
#define PL_NN 10
#endif
#ifndef PL_NR
#define PL_NR 10
#endif
#ifndef PL_NRN
#define PL_NRN PL_NR
#endif
#ifndef PL_NA
#define PL_NA 10
#endif
#ifndef PL_NZ
#define PL_NZ 10
#endif
// N_r_p != N_r_n (reference src/params.jl:124-136: the two particle grids are independent options): the c_s_avg section keeps the reference's compact layout -- N_p particles
// of NRP entries, then N_n particles of NRN --, the particle phases run on the common lane stride NR = max(NRP, NRN) (lane -> row l % NR of its particle) with the
// radial operator of the smaller electrode zero-padded to NR x NR (cell_setup), so that every sum over k < NR is the sum over the electrode's own rows; lanes whose row
// does not exist in their particle compute and do not store.  With NRP == NRN every expression below folds to what it was for one N_r.
constexpr int NP = PL_NP, NS = PL_NS, NN = PL_NN, NRP = PL_NR, NRN = PL_NRN, NR = NRP > NRN ? NRP : NRN, NE = NP + NS + NN, NJ = NP + NN;
constexpr bool NR_EQ = NRP == NRN;
constexpr int NCS_FICK = NP * NRP + NN * NRN;                      // entries of c_s_avg with Fickian diffusion
__host__ __device__ constexpr int nr_of(int p) { return NR_EQ ? NR : (p < NP ? NRP : NRN); }                             // radial nodes of particle p (0 .. NJ - 1)
__host__ __device__ constexpr int cs_off(int p) { return NR_EQ ? p * NR : (p < NP ? p * NRP : NP * NRP + (p - NP) * NRN); }  // first c_s entry of particle p (relative to O_CS)
__host__ __device__ constexpr int cs_surf(int p) { return cs_off(p) + nr_of(p) - 1; }                                   // its surface node
__host__ __device__ constexpr int cs_particle(int q) { return NR_EQ ? q / NR : (q < NP * NRP ? q / NRP : NP + (q - NP * NRP) / NRN); }   // particle of c_s entry q
constexpr int O_CE = 0, O_CS = NE;                                 // c_e and c_s_avg lead every layout (reference src/external.jl:275-365)
constexpr int NA = PL_NA, NZ = PL_NZ, NT = NA + NE + NZ;         // current collectors; temperature nodes a|p|s|n|z
constexpr bool GRID_DEFAULT = NP == 10 && NS == 10 && NN == 10 && NRP == 10 && NRN == 10 && NA == 10 && NZ == 10;
constexpr int NRMAX = 16;                                          // Tables has room for the radial operator of any supported N_r
}  // namespace pl
#include "radial_tables.h"      // PL_RADIAL_M / LAM / V / W: the radial operator of THIS translation unit's N_r as static const arrays -- device code may index them directly
namespace pl {
// lane maps: the node pass gives control volume i to lane i; the twisted block sweeps put nodes 0 .. NE/2-1 in lanes 0 .. and nodes NE-1 .. NE/2 in lanes 32 .. (tw_node);
// the particle phases give lane l row l % NR of particle pass * CS_G + l / NR
static_assert(NP >= 2 && NS >= 2 && NN >= 2 && NE <= 48, "2 <= N_p, N_s, N_n and N_p + N_s + N_n <= 48 (one lane per node, two 32-lane halves in the sweeps)");
static_assert(NRP >= 10 && NRN >= 10 && NR <= NRMAX, "10 <= N_r_p, N_r_n <= 16 (the radial operator of N_r = 9 has complex eigenvalues: no spectral resolvent; tools/gen_radial_tables.py)");
constexpr int CS_G = WAVE / NR, CS_LANES = CS_G * NR, CS_PASS = (NJ + CS_G - 1) / CS_G;     // particles per pass, lanes in use, passes (default grid: 6, 60, 4)
// r06, isothermal / SEI models: the particle phases in the ROW layout -- one particle per 16-lane DPP row (lane -> row r = lane % 16 of particle pass * 4 + lane / 16, rows
// r < N_r live), so that the matrix-vector products take their vector operand from the row's own lanes through `row_newbcast` (see rowb_fmac) instead of from LDS
constexpr int CSD_G = WAVE / 16, CSD_PASS = (NJ + CSD_G - 1) / CSD_G;                       // default grid: 4 particles per pass, 5 passes
constexpr int LR_PASS = CS_PASS > CSD_PASS ? CS_PASS : CSD_PASS;                            // per-lane registers that persist across phases: one per pass of either layout
constexpr int MAXORD = 5;

// Model traits: state layout  Y = [ c_e | c_s_avg | T (thermal) | film, SOH (SEI) || j | Phi_e | Phi_s | j_s (SEI) | I ]  and closures.
// SD_: solid diffusion -- PLH_SD_FICKIAN (9th-order finite differences, N_r nodes per particle), PLH_SD_QUADRATIC, PLH_SD_POLYNOMIAL (one volume-averaged
//      concentration per particle, the polynomial variant with the extra state Q: residuals.jl:108-127, 237-258, aux...jl:212-248)
// TF_: thermodynamic factor -- 0 linear (nu = 1), 1 nonlinear nu(c_e, T) (custom_functions.jl:177-203);  RXN_: 0 Butler-Volmer, 1 Marcus-Hush-Chidsey (:212-298)
// W2_: two wavefronts per cell (a 128-thread workgroup; isothermal Fickian models): wave 1 owns the particle rows -- c_s residual rows, particle resolvents, particle
//      partial solves and back-substitution, and the c_s entries of every vector phase -- and runs them next to wave 0's node pass / elimination / sweeps; the two waves of
//      a cell share its LDS block and meet at s_barrier (PL_XSYNC).  Two waves per SIMD also means 256 registers per wave instead of 512.
template <int CHEM_, bool SEI_, bool THERMAL_ = false, int PREC_ = 0, int SD_ = 0, int TF_ = 0, int RXN_ = 0, int W2_ = 0> struct ModelT {
  static constexpr int CHEM = CHEM_;                 // PLH_CHEM_LCO_LIC6 / PLH_CHEM_NMC_LIC6
  static constexpr int SD = SD_, TF = TF_, RXN = RXN_;
  static constexpr bool W2 = W2_ != 0;
  static constexpr int NWAVES = W2_ ? 2 : 1;
  static_assert(W2_ == 0 || (SD_ == 0 && !SEI_ && !THERMAL_), "two waves per cell: isothermal Fickian models without aging");
  static_assert(W2_ == 0 || NR_EQ, "two waves per cell: N_r_p = N_r_n");
  static_assert(SD_ == 0 || (!SEI_ && !THERMAL_), "the quadratic / polynomial particle models are instantiated for the isothermal models without aging");
  static constexpr int NCS = SD_ == 0 ? NCS_FICK : NJ;   // entries of c_s_avg
  static constexpr int N_CECS = O_CS + NCS;
  static constexpr bool SEI = SEI_;
  static constexpr bool THERMAL = THERMAL_;
  // PLH_PREC_MIXED (config C5's reduced-precision leg): the LDS-resident factors of the Newton matrix -- block-Thomas D'^-1, L D'^-1 and the particle
  // resolvents -- are STORED in fp32; states, residuals, Jacobian partials, time, error control and all arithmetic stay fp64.  Pure fp32 is not offered: the
  // reltol-1e-3 Newton iteration would still converge, but SOH (1 - 1e-6 per pulse), film (1e-14 m) and t (41 h at 1e-6 s steps) are below fp32 resolution
  // (tools/fp32_study.py).
  static constexpr int PREC = PREC_;                 // plh_model_desc.precision: PLH_PREC_F64 / PLH_PREC_MIXED / PLH_PREC_F64_REFORDER
  static constexpr bool MIXED = PREC_ == PLH_PREC_MIXED;
  // PLH_PREC_F64_REFORDER: the finite-volume rows (c_e, Phi_e, and the heat conduction of the T rows) are evaluated in the REFERENCE's operation order -- matrix form,
  // A x - f with every product A_ik x_k rounded before the sum (residuals.jl:6-106, 554-654, 299-489) -- instead of the conservative edge-flux / difference form of the
  // default build.  Same equations; the rows then carry the reference's evaluation rounding (2e-13 per Phi_e row instead of 5e-15), which is what IDA's start-up order
  // selection in a :hold leg sees.  For users who need the reference's step sequences, and for the A/B that tests that explanation.
  static constexpr bool REFORD = PREC_ == PLH_PREC_F64_REFORDER;
  static_assert(PREC_ >= 0 && PREC_ <= 2, "precision");
  using fact_t = typename std::conditional<MIXED, float, double>::type;
  // LDS diet: the error weights and the accumulated Newton correction live in registers (IdaScalars::ew / ee: they are only touched by the
  // lane-strided vector phases), and so do the BDF history vectors of order >= PHI_LDS
  // r05, two cells per SIMD (PL_OCC2 builds, tools/experiments/occupancy.py): the isothermal Fickian models (with or without SEI) keep BDF history orders >= 2 in
  // GLOBAL memory (per-cell block of 4 vectors, lane-strided and coalesced; L2-resident: 6 cells per CU x 9.7 kB) and read the eigen-decomposition of the radial operator from the
  // model tables at a Jacobian refresh instead of an LDS copy: 37.5 kB -> 26.2 kB per cell, six cells per CU, and the kernels are compiled for two waves per SIMD (256 registers).
  // -DPL_OCC2=4 (r05, second experiment): only the orders 4 and 5 go to global memory -- they are in use in a minority of the steps at the default tolerances (mean order 2.2 on
  // C2) -- and everything else stays as it is: 37.5 -> 32.7 kB, FIVE cells per CU (the SIMD that holds two runs both at 256 registers, which costs this kernel nothing: measured).
#ifdef PL_OCC2
  static constexpr bool PHI_GLOBAL = !THERMAL_ && SD_ == 0 && W2_ == 0 && PREC_ != PLH_PREC_MIXED && (PL_OCC2 + 0 != 4 || !SEI_);
  static constexpr int PHI_GLOBAL_FROM = (PL_OCC2 + 0 == 4) ? 4 : 2;          // first history order kept in global memory
#else
  static constexpr bool PHI_GLOBAL = false;
  static constexpr int PHI_GLOBAL_FROM = 2;
#endif
  static constexpr bool PHI_GLOBAL_ALL = PHI_GLOBAL && PHI_GLOBAL_FROM == 2;  // the 26 kB layout: also no LDS copy of the eigen-decomposition, no predictor registers
  static constexpr int PHI_LDS = THERMAL_ ? 2 : (PHI_GLOBAL ? PHI_GLOBAL_FROM : MAXORD + 1);   // thermal: 40.7 kB -> four cells per CU
  // predictor (y, y') of the step kept in registers across the Newton iteration (else re-summed from phi).  SEI models: since r04 -- with MachineLICM off (__graft_entry__.py) the
  // 24 registers are there (C5 24.6 k -> 25.2 k trajectories/s; with MachineLICM on it cost 1 %).  Thermal model: +0.3 %, within the noise of the boxes, left as it was.
  static constexpr bool PRED_REGS = !THERMAL_ && !PHI_GLOBAL_ALL;      // (the 26 kB layout: 256 registers per lane, the predictor is re-summed)
  static constexpr int NB = THERMAL_ ? 4 : 3;        // node block size of the block-Thomas solve: (c_e, Phi_e, Phi_s[, T])
  static constexpr int O_T = N_CECS;
  static constexpr int O_FILM = N_CECS + (THERMAL_ ? NT : 0), O_SOH = O_FILM + NN;
  static constexpr int O_Q = SEI_ ? O_SOH + 1 : O_FILM;              // Q (polynomial approximation only) closes the differential block
  static constexpr int NDIFF = O_Q + (SD_ == 2 ? NJ : 0);
  static constexpr int O_J = NDIFF, O_PE = O_J + NJ, O_PS = O_PE + NE, O_JS = O_PS + NJ, O_I = SEI_ ? O_JS + NN : O_PS + NJ;
  static constexpr int NST = O_I + 1, NALG = NST - NDIFF;
  // trips of a lane through a state vector: one wave strides the whole vector; with two waves, wave 1 strides the NCS particle entries (4 trips), wave 0 the
  // other NST - NCS entries (2 trips) -- see vrow / vok
  static constexpr int NTRIP = W2_ ? (NCS + WAVE - 1) / WAVE : (NST + WAVE - 1) / WAVE;
  // r06: the LDS state vectors of the isothermal models without aging are padded to whole trips (301 -> 320 entries) and the padding is kept at ZERO (cell_setup
  // clears it once; every vector operation maps 0 to 0; residual / solve / global loads only touch the real entries), so that the lane-strided vector phases run WITHOUT a
  // lane mask: the predicated last trip (lanes < 45) put an exec-mask save / branch / restore around its share of every vector statement and split each phase into
  // basic blocks whose LDS loads could not be issued together.  +1.3 kB of LDS per cell (38.8 of the 40.96 kB that four cells per CU allow); the SEI models (322 states:
  // 62 more entries x 9 vectors) and the thermal ones (at 40.95 kB) do not have the room and keep the mask.  -DPL_NO_VPAD: the r05 layout (A/B builds).
#ifdef PL_NO_VPAD
  static constexpr bool VPAD = false;
#else
  static constexpr bool VPAD = !SEI_ && !THERMAL_ && W2_ == 0 && !PHI_GLOBAL && NTRIP * WAVE - NST <= 24;
#endif
  static constexpr int NPADG = NST + (NST & 1);             // stride of the state-sized vectors kept in GLOBAL memory (sensitivity histories, PHI_GLOBAL block)
  static constexpr int NPAD = VPAD ? NTRIP * WAVE : NPADG;  // length of the LDS state vectors
};
using ModelLcoIso = ModelT<PLH_CHEM_LCO_LIC6, false>;
#define PL_MODEL(M) [[maybe_unused]] constexpr int O_J = M::O_J, O_PE = M::O_PE, O_PS = M::O_PS, O_I = M::O_I, NST = M::NST, NDIFF = M::NDIFF, NTRIP = M::NTRIP, \
                                       O_FILM = M::O_FILM, O_SOH = M::O_SOH, O_JS = M::O_JS, N_CECS = M::N_CECS, O_Q = M::O_Q
constexpr double FAR = 96485.3321233;      // reference src/structures.jl:10
constexpr double RGAS = 8.31446261815324;  // reference src/structures.jl:11
constexpr double TREF = 298.15;

// theta keys the device reads: the union over the model variants; Tables::thidx maps a key to its position in the variant's
// theta vector (the reference's sorted theta_keys) or -1 when the variant does not have it.
enum Key { K_D_n, K_D_p, K_D_s, K_D_sn, K_D_sp, K_Ea_D_sn, K_Ea_D_sp, K_Ea_k_n, K_Ea_k_p, K_M_n, K_R_SEI, K_Rp_n, K_Rp_p, K_T0, K_Uref_s,
           K_brugg_n, K_brugg_p, K_brugg_s, K_c_e0, K_c_max_n, K_c_max_p, K_i_0_jside, K_k_n, K_k_n_aging, K_k_p, K_l_n, K_l_p, K_l_s,
           K_tplus, K_w, K_th_max_n, K_th_max_p, K_th_min_n, K_th_min_p, K_rho_n, K_sig_n, K_sig_p, K_eps_fn, K_eps_fp, K_eps_n, K_eps_p,
           K_eps_s,
           K_Cp_a, K_Cp_n, K_Cp_p, K_Cp_s, K_Cp_z, K_T_amb, K_h_cell, K_l_a, K_l_z, K_lam_a, K_lam_n, K_lam_p, K_lam_s, K_lam_z,
           K_rho_a, K_rho_p, K_rho_s, K_rho_z, K_sig_a, K_sig_z, K_lam_MHC_n, K_lam_MHC_p, K_D_e, K_COUNT };

// read-only model tables in device memory
struct Tables {
  // radial operator M (N_r x N_r, row-major), its eigenvalues LAM and eigenvectors V, W = V^-1, packed back to back for THIS translation unit's N_r (so that the
  // offsets the kernels see do not depend on the largest supported N_r: vector global loads carry a 13-bit immediate offset); the host side fills the block through
  // the strides of the variant's own N_r (plh_model_create)
  // (N_r_p != N_r_n: one block per electrode -- RAD the cathode's, RAD_N at the end of the struct the anode's --, each at the common stride NR = max(N_r_p, N_r_n), the
  //  smaller operator zero-padded; el = 0 cathode, 1 anode.  With N_r_p = N_r_n the kernels read RAD only)
  static constexpr int RADBLK = 3 * NRMAX * NRMAX + NRMAX;
  double RAD[RADBLK];
  __host__ __device__ const double* radp(int el) const { return (NR_EQ || el == 0) ? RAD : RAD_N; }
  __host__ __device__ const double* Mp(int el = 0) const { return radp(el); }
  __host__ __device__ const double* LAMp(int el = 0) const { return radp(el) + NR * NR; }
  __host__ __device__ const double* Vp(int el = 0) const { return radp(el) + NR * NR + NR; }
  __host__ __device__ const double* Wp(int el = 0) const { return radp(el) + 2 * NR * NR + NR; }
  double BJ;                                            // surface-row BC factor (of the cathode's radial grid; BJ_N: the anode's)
  int thidx[K_COUNT];                                   // position of each key in the theta vector (-1: absent)
  int chem;
  int P;
  int nnz[PLH_N_MODES];                                 // full-Jacobian nnz per mode
  const unsigned* csc_code[PLH_N_MODES];                          // per mode: decode word of every CSC entry
  // the same entries in row-major (CSR) order: row pointers, decode words, column indices -- used by the matrix-vector product of the
  // iterative-refinement mode (plh_opts.refine) and by nothing else
  const int* csr_ptr[PLH_N_MODES]; const unsigned* csr_code[PLH_N_MODES]; const unsigned short* csr_col[PLH_N_MODES];
  double BJ_N, RAD_N[RADBLK];                           // the anode's radial grid (last: the offsets of everything above are those of the one-grid layout)
};

struct CellConst {
  double h[3], eps[3], bf[3], Dc[3];
  double a_p, a_n, sig_p, sig_n, kp, kn, cmaxp, cmaxn, kap_p, kap_n, bj_p, bj_n;
  double T0, fRT, Kfac, I1C, tplus, JI0, JI29, ce0;
  double thmin_p, thmax_p, thmin_n, thmax_n;
  double R_SEI, rkag, Mrho, i0F, wexp, Uref;   // SEI: R_SEI, 1/k_n_aging, M_n/rho_n, i_0_jside/F, w, Uref_s
  double rh[3], reps[3], rd_ps, rd_sn, beta_ps, beta_sn, rsg_p, rsg_n, rcm_p, rcm_n, Dh_ps, Dh_sn;   // reciprocals / interface weights used by every node pass
  double EaKp, EaKn, EaDp, EaDn;               // thermal: activation energies / R (kp, kn, kap_p, kap_n then hold the T_ref values)
  double r2h[3], qps_r, qps_l, qsn_r, qsn_l;   // thermal: gradient-stencil factors 1/(2h), 2/(3hp+hs), 2/(hp+3hs), 2/(3hs+hn), 2/(hs+3hn)
  // quadratic / polynomial particle models: c_s* = c_avg + csj j (+ csq Q);  d c_avg/dt = csr j;  dQ/dt = -kappa Q + qj j   (index 0 = p, 1 = n)
  double csj[2], csq[2], csr[2], qj[2];
  double De;     // LGM50: electrolyte diffusivity scale D_e
  double lam[2], mhc_k0[2], rce0;   // MHC: lambda per electrode, k_i / ((1 - erf((lambda - sqrt(1 + sqrt(lambda))) / (2 sqrt(lambda)))) / 2), 1 / c_e0
  double Tamb;   // thermal: ambient temperature (convective ends of the heat equation)
  int iso_ref;   // T0 == TREF exactly (temperature_switch, reference custom_functions.jl:1)
};

// r06: the per-section and per-edge constants of the isothermal node pass as LDS tables that a lane indexes with ITS OWN section (0 = p, 1 = s, 2 = n) and edge kind
// (0 .. 2 = interior edge of section p / s / n, 3 = p|s interface, 4 = s|n interface): one load per constant with a per-lane address instead of three wave-uniform loads
// and a chain of 64-bit selects (r05: 71 v_cndmask per node pass, 10 % of the kernel's static VALU stream).  Same numbers, copied by cell_setup from CellConst; 456 B per
// cell -- the thermal model (at 40 952 of 40 960 B) has its own node pass and does not carry them.
template <bool ON> struct NodeTab {};
template <> struct NodeTab<true> {
  struct alignas(16) Sec { double h, eps, bf, Dc, rh, reps, a, cmax, kk, rsg, rcm, pad; } sec[3];      // (section s: the anode's electrode entries, as the selects of r05 gave it: unused)
  struct alignas(16) Edge { double beta, rdist, Dh, pad; } edge[5];
};

// Jacobian pool of the SEI rows (anode nodes k = 0..NN-1); empty for models without aging so that their LDS footprint is unchanged
template <bool SEI> struct SeiPool {};
template <> struct SeiPool<true> {
  double jjJ[NN], jjF[NN];                                          // d(j row)/d(j, film)
  double jsPS[NN], jsPE[NN], jsJ[NN], jsJS[NN], jsF[NN], jsI[NN];   // d(j_s row)/d(Phi_s, Phi_e, j, j_s, film, I)
  double Wl[9][NN];                                                 // inverse of the node-local (j, j_s, film) block
  double sohw[NN];                                                  // d(rhs_SOH)/d j_s: trapezoid + end-extrapolation weights
  double cjf;                                                       // cj of the current factorisation
};

// pools of the thermal model (temperature = true; functions in dfn_thermal.h).  T lives on NT = 50 nodes (a | p | s | n | z); the
// 30 cell-sandwich nodes carry T as the 4th unknown of the block-Thomas node block, the two collector chains are eliminated
// onto their neighbours, and the four T rows whose one-sided gradient stencils reach a second neighbour (nodes 0, 9, 20, 29)
// are folded into the twisted elimination (modified neighbour blocks, two right-hand-side shares): see thermal_sweeps.
template <bool TH, bool MIXED = false> struct ThermalPool {};
template <bool MIXED> struct ThermalPool<true, MIXED> {
  double Dpark[MIXED ? NE * 16 : 1];                       // (until r04: fp64 parking of the node's own block during the factor sweep, which now keeps it in registers; the member stays so that the pool's layout does not move)
  // heat-conduction stencil of residuals_T! (residuals.jl:299-489), already divided by rho*Cp:  aL T[it-1] + aD T[it] + aU T[it+1] + aC with aD = -(aL + aU) [- aC2 at the
  // two convective ends] (thermal_aD: not stored -- r04's LDS diet: the 2.4 kB freed here hold the radial operator, which the particle phases otherwise fetched from global
  // memory in every residual and every solve, at one wave per SIMD with nothing to hide that latency behind)
  double aL[NT], aU[NT], wT5[5];                            // wT5 = temperature_weighting w_i / L per section a|p|s|n|z (aux...jl:649-679)
  double aC2[2], rc5[5];                                    // convective coefficient h_cell / (h rho Cp) of the two end rows; 1/(rho Cp) per section a|p|s|n|z
  double qI[2], qIJ[2];                                    // collector rows: Joule heat qI * I^2 ; qIJ = d(row)/dI at the last Jacobian pass
  double kapP[NJ], dkapP[NJ];                              // per-particle D_s(T)/Rp^2 and its T derivative
  // Jacobian partials that exist only with temperature
  double ptL[NE], ptD[NE], ptU[NE];                        // Phi_e rows x T
  double gT[NJ];                                           // j rows x T
  double TcL[NE], TcD[NE], TcU[NE], TeL[NE], TeD[NE], TeU[NE], TsL[NE], TsD[NE], TsU[NE], TtD[NE];   // T rows x (c_e, Phi_e, Phi_s, T)
  double TJ[NJ], Tcs[NJ];                                  // T rows x (j, c_s surface)
  double TX2[4][3];                                        // out-of-band T-row entries: rows of nodes 0, 9, 20, 29 x (c_e, Phi_e, Phi_s) of nodes 2, 7, 22, 27
  // particle resolvent in spectral form (per particle: kappa differs with T)
  double AinvE[NJ][NR], AinvQ[NJ][NR];                     // A^-1 e_last, A^-1 q (AinvQ doubles as the store of W c between the Jacobian pass and the factorisation)
  double kapF[NJ];                                         // kappa at the last factorisation (the resolvent is applied in spectral form)
  // node-local elimination
  // t_r phi_c of the eliminated j (D[r][c] -= t_r phi_c): t = (ceJ, peJ, psJ, tq3), phi = (gce, gpe, gps) dj and phi3 -- only the two T entries need storing
  double tq3[NE], phi3[NE];
  double cI4[4];                                           // column of I after the local elimination: (Phi_s, T) components at node 0 and at node NE - 1 (zero elsewhere)
  // collector chains (tridiagonal scalar systems)
  static constexpr int NC = NA > NZ ? NA : NZ;
  double cP[2][NC], zc[2][NC], zI[2][NC];                  // (forward multipliers of the chains: aL[k] cP[k-1], formed where they are used; the chain solution of a solve stays in registers)
  // Woodbury and border
  double x2[4][NE], vB[4][NE];
  double qfar[2][4];                                       // q = (far T-row entry of node 9 / 20) . D'^-1 of node 7 / 22 (right-hand-side share)
  double bord[2];                                          // [0] d2 = d - v.x2, [1] d (direct I entry of the control row)
  double cjf;
};

// 16-byte alignment of the block and of its vectors: with it the compiler can prove that an even-indexed pair of doubles is one aligned 16-byte access and uses ds_read_b128
// (4 LDS-array cycles per wave-instruction, 256 B/clk) instead of ds_read2_b64 (8 cycles for the same 16 bytes per lane: MI355X_MICROARCH.md, LDS table).  The LDS array is
// shared by the four cells of a CU; in the particle phases it, not the VALU, is what a cell waits for.
template <class M> struct alignas(16) CellLDS {
  alignas(16) double phi[M::PHI_LDS][M::NPAD];
  alignas(16) double yy[M::NPAD], yp[M::NPAD], delta[M::NPAD];
  // structured Jacobian pool (cj not included)
  double ceL[NE], ceD[NE], ceU[NE], ceJ[NE];
  double peL[NE], peD[NE], peU[NE], pcL[NE], pcD[NE], pcU[NE], peJ[NE];
  double gce[NJ], gcs[NJ], gpe[NJ], gps[NJ], psJ[NJ];
  // eliminated system
  // (per-node small blocks are stored structure-of-arrays, [element][node]: the lanes of a wave own one node each, so element k of all nodes is
  //  one conflict-free LDS access; [node][element] with a row of 16 doubles puts every lane on the same bank)
  double dj[NJ], nphi[M::THERMAL ? 1 : 3][M::THERMAL ? 1 : NE];      // j pivot; phi_c = omega . A_ux[:,c] (node block D[r][c] -= t_r phi_c, t = (ceJ, peJ, psJ))
  double colI[M::THERMAL ? 1 : 3][M::THERMAL ? 1 : NE];            // column of I after the local elimination (Phi_s ends; + j_s coupling with SEI)
  typename M::fact_t Dinv[M::NB * M::NB][NE], LD[M::NB * M::NB][NE];   // Thomas factors: D'^-1 and L D'^-1(prev)
  typename M::fact_t LDmid[M::NB * M::NB];                             // closing block of the twisted factorisation
  alignas(16) typename M::fact_t Ainv[2][(M::THERMAL || M::SD != 0) ? 1 : NR * NR];          // particle resolvents (kappa M - cj I)^-1 for the p / n electrode (row-major)
  // radial operator (copy of Tables::M) followed by
  // its eigen-decomposition (copies of Tables::V, W, LAM): the resolvents are rebuilt from them at every Jacobian refresh, and reading the tables from HBM there cost
  // 9.7 k cycles per refresh (two dependent rounds of global / scalar loads); from LDS, with the 2 N_r^2 entries spread over the wave, 1 k
  // (one array, so that the models without it -- thermal: 40 952 of the 40 960 B that four cells per CU allow -- pay 8 bytes, not 32)
  static constexpr int MR_BLK = M::PHI_GLOBAL_ALL ? NR * NR : (M::THERMAL ? 3 * NR * NR : 3 * NR * NR + NR);   // (thermal: M, V, W; the eigenvalues are only read at a factorisation; PHI_GLOBAL: M only)
  alignas(16) double Mr[M::SD != 0 ? 1 : (NR_EQ ? 1 : 2) * MR_BLK];                          // (N_r_p != N_r_n: the cathode's block, then the anode's, both at stride NR, zero-padded)
  static constexpr int OFF_VR = NR * NR, OFF_WR = 2 * NR * NR, OFF_LAMR = 3 * NR * NR;
  static __host__ __device__ constexpr int mr_el(int el) { return NR_EQ ? 0 : el * MR_BLK; }  // offset of electrode el's block
  double resp[M::SD != 0 ? NJ : 1], rcjf[M::SD != 0 ? 2 : 1][2];    // quadratic / polynomial particles: d c_s* / d j after eliminating c_avg (and Q); 1/cj and 1/(-kappa - cj) of the factorisation
  double x2[M::THERMAL ? 1 : 3][M::THERMAL ? 1 : NE];      // (the thermal model keeps its own in ThermalPool: one placeholder element here)
  double ctrlJ[2], bord;           // P-mode control row at the last Jacobian pass (I*I1C, V*I1C); border pivot d - v.x2
  double w9[NJ > 2 * NR ? NJ : 2 * NR];
  double sig[2];
  double red[M::W2 ? 16 : 1];      // two waves per cell: partial sums of the cross-wave reductions (ring of 4 x 2 waves x up to 2... see block_sum)
  SeiPool<M::SEI> sei;
  ThermalPool<M::THERMAL, M::MIXED> th;
  // wave-uniform BDF coefficient arrays (dynamically indexed by the order -> LDS, not registers/scratch)
  double ida_psi[MAXORD + 1], ida_alpha[MAXORD + 1], ida_beta[MAXORD + 1], ida_sigma[MAXORD + 1], ida_gamma[MAXORD + 1];
#ifdef PL_PHASE_TIMERS
  long long cyc[8];    // per-phase cycle sums (profiling build only)
#endif
  const Tables* tb;    // model tables (set by cell_setup)
  // what one run hands to the next (SOC, end time, V / I / eta_plating at the end): kept HERE between the runs, read into run-local registers at the top of a run.  As
  // loop-carried 64-bit registers of the run loop these were merged at the loop head from "new solution" and "continuation" paths, and hipcc 7.x twice dropped one HALF of
  // such a merge when the two halves of the pair had been spilled to different places (AGPR / scratch): a run started with a garbage SOC.  A value that is
  // loaded at the top of the run has one definition and nothing to merge.
  double carry[5];     // [0] SOC, [1] t_global, [2] prev_V, [3] prev_I, [4] prev_etap
  plh_run runc;        // the run being integrated (copied from HBM once per run)
  CellConst cc;
  NodeTab<!M::THERMAL> nt;      // (outside the range sens_factor_copy saves; sens_consts saves it with cc)
  // closure inputs (PLH_VAL_EXPR; general instantiation only; kept last so that nothing else moves): the cell's theta row in HBM and the interpreter's value stack
  const double* theta_row;
  double xstk[PLH_EXPR_STACK * M::NWAVES];      // (one stack per wave)
};

// per-lane registers that persist across phases
struct LaneRegs {
  double wreg[LR_PASS]; // particle partial solutions kept across the Thomas phase
  double rcp[LR_PASS];  // thermal model: 1 / (kappa_p lam_r - cj) of the last factorisation for this lane's (particle, radial mode) of each pass -- the spectral resolvent's diagonal
};

// forward parameter sensitivities (dfn_sens.h): what plh_integrate hands the kernel
struct SensArgs {               // (device pointers; part of IntegrateArgs)
  int n_sens;                   // number of theta columns differentiated (0: none)
  const int* cols;              // [n_sens] 0-based positions in theta_keys
  const double* theta_pert;     // [n_cells][n_sens][P]: the cell's theta row with column cols[k] perturbed (relative 1e-7; built by plh_integrate)
  double* hist;                 // [n_cells][n_sens][MAXORD + 1][NPAD]: BDF history of every s_k
  double* dY;                   // [n_cells][n_sens][N] or nullptr: dY/dtheta_k at the end of the last completed run
  double* dV;                   // [n_cells][n_sens][max_pts] or nullptr: dV/dtheta_k at every saved point
  int* stat;                    // [n_cells][3] or nullptr: corrector iterations, solves that did not reach the tolerance, steps that factored their own matrix
  double* cbak;                 // [n_cells][SENS_CBAK]: the cell's theta-derived constants, saved once and copied back after every evaluation with a perturbed theta row
  double* aux;                  // [n_cells][n_sens][4]: per parameter [0] d(held input value)/dtheta of the run being integrated (a :hold run: the previous run's end sensitivity of
                                // the held quantity), [1] dSOC/dtheta (trapezoid of dI/dtheta over the saved points, like calc_SOC), [2] dI/dtheta at the previous saved point,
                                // [3] the last step's increment of [1]
  double* fsave; int fsave_stride;   // [n_cells][fsave_stride]: where a step that factors its own matrix parks the integrator's factorisation (dfn_sens.h, sens_factor_copy)
};
constexpr int SENS_CBAK = 384;

// An index the compiler must treat as opaque: `base + PL_OPAQUE_IDX(lane part)` keeps the lane-dependent part of an LDS address in ONE register and leaves the compile-time part to
// the instruction's offset field.  Without it the constant parts of S.<array>[lane part + const] are folded into one large immediate per access, which ds_read2_b64's 8-bit
// offsets cannot hold: the compiler then materialises a separate address (v_add / v_mad) for every pair of loads -- a quarter of the instructions of the particle phases.
#ifdef PL_WAVE_EMU
#define PL_OPAQUE_IDX(i) (i)
typedef const double* lds_cptr;
typedef double* lds_ptr;
#define PL_LDS_BASE(p) (p)
#define PL_LDS_BASE_A(A16, p) (p)
#else
__device__ __forceinline__ int pl_opaque_idx(int i) { __asm__("" : "+v"(i)); return i; }
#define PL_OPAQUE_IDX(i) pl_opaque_idx(i)
// The same for a whole LDS address: a 32-bit LDS pointer (address_space(3)) the compiler may not look into.  Accesses q[compile-time index] then use q's register plus
// the instruction's offset field -- within the reach of ds_read2_b64 (2040 B) as long as the indices stay below 256.
typedef const __attribute__((address_space(3))) double* lds_cptr;
typedef __attribute__((address_space(3))) double* lds_ptr;
__device__ __forceinline__ lds_cptr pl_lds_base(const double* p) { lds_cptr q = (lds_cptr)p; __asm__("" : "+v"(q)); return q; }
__device__ __forceinline__ lds_ptr pl_lds_base(double* p) { lds_ptr q = (lds_ptr)p; __asm__("" : "+v"(q)); return q; }
#define PL_LDS_BASE(p) pl_lds_base(p)
// the same for an address the CALLER knows to be 16-byte aligned (A16 true; the asm hides that from the compiler): pairs q[2k], q[2k+1] then load as one ds_read_b128
template <bool A16> __device__ __forceinline__ lds_cptr pl_lds_base_a(const double* p) { lds_cptr q = pl_lds_base(p); if constexpr (A16) __builtin_assume(((unsigned)(__UINTPTR_TYPE__)q & 15u) == 0u); return q; }
#define PL_LDS_BASE_A(A16, p) pl_lds_base_a<A16>(p)
#endif

// lane_id_pred(): the lane number for code whose use of it is PREDICATES (which control volume, which section, first / last of an electrode, node of the twisted layout):
// its own copy per call site, so that the compares are made where they are used (one v_cmp each) instead of once at the top of the kernel -- where every lane predicate of
// every inlined phase then lives in an SGPR pair across the whole step loop, i.e. in a spill lane of a VGPR (v_writelane / two v_readlane per use: r05 code object, 1 036 of them)
// (-DPL_LANE_OPAQUE=0: off; =1: every lane_id(), including the particle phases' lane / N_r arithmetic)
#ifndef PL_LANE_OPAQUE
#define PL_LANE_OPAQUE 0
#endif
#if PL_LANE_OPAQUE + 0 == 1 && !defined(PL_WAVE_EMU)
__device__ __forceinline__ int lane_id() { int i = (int)threadIdx.x & (WAVE - 1); __asm__ volatile("" : "+v"(i)); return i; }
#else
__device__ __forceinline__ int lane_id() { return (int)threadIdx.x & (WAVE - 1); }
#endif
#if PL_LANE_OPAQUE + 0 >= 1 && !defined(PL_WAVE_EMU)
__device__ __forceinline__ int lane_id_pred() { int i = (int)threadIdx.x & (WAVE - 1); __asm__ volatile("" : "+v"(i)); return i; }
#else
__device__ __forceinline__ int lane_id_pred() { return (int)threadIdx.x & (WAVE - 1); }
#endif
__device__ __forceinline__ int wave_id() { return (int)threadIdx.x >> 6; }     // 0 for the one-wave kernels; 0 / 1 for M::W2

// Phase separator between LDS producers and consumers.  A workgroup here is exactly ONE wavefront, and the LDS instructions of one
// wave execute in program order, so no s_barrier is needed -- and __syncthreads() would cost a workgroup-scope fence that also
// waits (s_waitcnt vmcnt(0)) for this wave's fire-and-forget global stores (per-step outputs).  On the GPU this is therefore only
// a compiler barrier (the compiler still inserts the lgkmcnt waits that real data dependences need); the lock-step emulator
// yields to the other lanes.
#ifdef PL_WAVE_EMU
#define PL_SYNC() wave_emu::yield()
#else
#define PL_SYNC() __asm__ volatile("" ::: "memory")
#endif
// Hand-over between the two waves of a cell (M::W2): a real workgroup barrier; for the one-wave models it is PL_SYNC.  (`M` must be in scope.)
#define PL_XSYNC() do { if constexpr (M::W2) __syncthreads(); else PL_SYNC(); } while (0)
// row of the state vector a lane handles in trip k of a vector phase, and whether that trip is live
template <class M> __device__ __forceinline__ int vrow(int k, int lane, int wv) {
  if constexpr (!M::W2) return lane + WAVE * k;
  else { const int m = lane + WAVE * k; return wv ? O_CS + m : (m < O_CS ? m : m + M::NCS); }
}
// vokg: trip k of this lane is a real entry (what every access to a vector in GLOBAL memory must test); vok: the same for the LDS vectors -- always true with M::VPAD
template <class M> __device__ __forceinline__ bool vokg(int k, int lane, int wv) {
  if constexpr (!M::W2) return k < M::NST / WAVE || lane + WAVE * k < M::NST;
  else return lane + WAVE * k < (wv ? M::NCS : M::NST - M::NCS);
}
template <class M> __device__ __forceinline__ bool vok(int k, int lane, int wv) {
  if constexpr (M::VPAD) return true;
  else return vokg<M>(k, lane, wv);
}

// Emulator only: with PL_EMU_POISON=1 in the environment the LDS block starts as garbage (on the GPU it holds whatever the previous workgroup
// left there; the emulator's static storage would hide a read of never-written LDS behind zeros).  tests/wave_emu poisons the lane stacks too.
#ifdef PL_WAVE_EMU
#define PL_EMU_POISON(S_) do { if (getenv("PL_EMU_POISON")) { if (threadIdx.x == 0) memset((void*)&(S_), 0x7f, sizeof(S_)); __syncthreads(); } } while (0)
#else
#define PL_EMU_POISON(S_) do {} while (0)
#endif

// store into a factor array (fp32 in the mixed-precision variants, see ModelT::MIXED)
#define PL_F32(x) ((typename M::fact_t)(x))

// optional per-phase cycle accounting (profiling build: -DPL_PHASE_TIMERS)
enum Phase { PH_RES, PH_JACFACT, PH_SOLVE, PH_NEWTVEC, PH_STEPCTL, PH_INIT, PH_OUTPUT, PH_TOTAL };
#if defined(PL_PHASE_TIMERS) && !defined(PL_WAVE_EMU)
#define PL_TIC_() long long pl_t0__ = (long long)__builtin_readcyclecounter()
#define PL_TOC_(S_, ph) do { if (threadIdx.x == 0) (S_).cyc[ph] += (long long)__builtin_readcyclecounter() - pl_t0__; } while (0)
#else
#define PL_TIC_() do {} while (0)
#define PL_TOC_(S_, ph) do {} while (0)
#endif
// -DPL_PHASE_DETAIL (with PL_PHASE_TIMERS): the seven phase slots are re-used for the sub-phases of the Jacobian refresh (PL_TICD / PL_TOCD below); the
// coarse phases are then not recorded (slot 7 stays the total)
#if defined(PL_PHASE_DETAIL) && (PL_PHASE_DETAIL + 0 == 0)
#undef PL_PHASE_DETAIL
#define PL_PHASE_DETAIL 1
#endif
#ifdef PL_PHASE_DETAIL
#define PL_TIC() do {} while (0)
#define PL_TOC(S_, ph) do {} while (0)
#define PL_TIC_TOTAL() PL_TIC_()
#define PL_TOC_TOTAL(S_) PL_TOC_(S_, PH_TOTAL)
#define PL_LAP_(S_, slot) do { const long long now__ = (long long)__builtin_readcyclecounter(); if (threadIdx.x == 0) (S_).cyc[slot] += now__ - pl_t0__; pl_t0__ = now__; } while (0)   /* lap timer */
#if PL_PHASE_DETAIL == 1          /* sub-phases of the Jacobian refresh */
#define PL_TICD() PL_TIC_()
#define PL_TOCD(S_, slot) PL_LAP_(S_, slot)
#else
#define PL_TICD() do {} while (0)
#define PL_TOCD(S_, slot) do {} while (0)
#endif
#define PL_TICE(mode) long long pl_t0__ = (PL_PHASE_DETAIL == (mode)) ? (long long)__builtin_readcyclecounter() : 0   /* mode 2: solve + residual, mode 3: step control + output */
#define PL_TOCE(S_, mode, slot) do { if (PL_PHASE_DETAIL == (mode)) PL_LAP_(S_, slot); } while (0)
#elif defined(PL_ASM_MARKS)
#define PL_AMARK(txt) __asm__ volatile("; PLMARK " txt ::: "memory")      /* named positions for tools/asm/isa.py (nothing in every other build) */      /* ISA inspection builds (tools/asm/): the timer positions become comment markers in the assembly (each one is also a compiler barrier) */
#define PL_MARK_(txt) __asm__ volatile("; PLMARK " txt ::: "memory")
#define PL_STR2_(x) #x
#define PL_STR_(x) PL_STR2_(x)
#define PL_TIC() PL_MARK_("tic")
#define PL_TOC(S_, ph) PL_MARK_("toc " #ph)
#define PL_TIC_TOTAL() do {} while (0)
#define PL_TOC_TOTAL(S_) do {} while (0)
#define PL_TICD() PL_MARK_("ticD")
#define PL_TOCD(S_, slot) PL_MARK_("tocD " #slot)
#define PL_TICE(mode) PL_MARK_("ticE " #mode)
#define PL_TOCE(S_, mode, slot) PL_MARK_("tocE " #mode " " #slot)
#elif defined(PL_PHASE_FENCES) && !defined(PL_PHASE_TIMERS) && !defined(PL_WAVE_EMU)
// Compiler fences (no instruction) at the phase boundaries, i.e. where the profiling builds read the clock and the ISA builds put their marks: memory operations may not be
// moved across them, which keeps the loads of one phase out of the previous one's register budget.  Thermal variants only (__graft_entry__.py VARIANT_FLAGS).
#define PL_FENCE_() __asm__ volatile("" ::: "memory")
#define PL_AMARK(txt) PL_FENCE_()
#define PL_TIC() PL_FENCE_()
#define PL_TOC(S_, ph) PL_FENCE_()
#define PL_TIC_TOTAL() do {} while (0)
#define PL_TOC_TOTAL(S_) do {} while (0)
#define PL_TICD() PL_FENCE_()
#define PL_TOCD(S_, slot) PL_FENCE_()
#define PL_TICE(mode) PL_FENCE_()
#define PL_TOCE(S_, mode, slot) PL_FENCE_()
#else
#define PL_TIC() PL_TIC_()
#define PL_TOC(S_, ph) PL_TOC_(S_, ph)
#define PL_TIC_TOTAL() do {} while (0)
#define PL_TOC_TOTAL(S_) PL_TOC_(S_, PH_TOTAL)
#define PL_TICD() do {} while (0)
#define PL_TOCD(S_, slot) do {} while (0)
#define PL_TICE(mode) do {} while (0)
#define PL_TOCE(S_, mode, slot) do {} while (0)
#endif
#ifndef PL_AMARK
#define PL_AMARK(txt) do {} while (0)
#endif
__host__ __device__ __forceinline__ int sec_of(int i) { return i < NP ? 0 : (i < NP + NS ? 1 : 2); }
// ---- cross-lane primitives.  gfx950: DPP moves (probed on hardware, tools/probes/dpp_probe.hip: wave_shr:1 / wave_shl:1 shift
// across the whole 64-lane wave, boundary lanes keep `old`), v_readlane for broadcasts.  Emulator: __shfl. ----
#ifndef PL_WAVE_EMU
template <int CTRL> __device__ __forceinline__ double dpp_mov0(double v) {     // out-of-range source lanes read 0
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
  hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double shift_up1(double v) { return dpp_mov0<0x138>(v); }     // lane i <- lane i-1 (lane 0 <- 0)
__device__ __forceinline__ double shift_down1(double v) { return dpp_mov0<0x130>(v); }   // lane i <- lane i+1 (lane 63 <- 0)
__device__ __forceinline__ double row_up2(double v) { return dpp_mov0<0x112>(v); }        // lane i <- lane i-2 within its row of 16 lanes (row_shr:2; lanes 0, 1 of a row <- 0)
__device__ __forceinline__ double row_down2(double v) { return dpp_mov0<0x102>(v); }      // lane i <- lane i+2 within its row of 16 lanes (row_shl:2; lanes 14, 15 of a row <- 0)
__device__ __forceinline__ double lane_bcast(double v, int src) {                         // wave-uniform src
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int lane_bcast_i(int v, int src) { return __builtin_amdgcn_readlane(v, src); }                       // wave-uniform src
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov0<0x111>(v); v += dpp_mov0<0x112>(v); v += dpp_mov0<0x114>(v); v += dpp_mov0<0x118>(v);   // row_shr 1,2,4,8: inclusive row scan
  return (lane_bcast(v, 15) + lane_bcast(v, 31)) + (lane_bcast(v, 47) + lane_bcast(v, 63));
}
#else
__device__ __forceinline__ double shift_up1(double v) { const double r = __shfl_up(v, 1); return lane_id() == 0 ? 0.0 : r; }
__device__ __forceinline__ double shift_down1(double v) { const double r = __shfl_down(v, 1); return lane_id() == WAVE - 1 ? 0.0 : r; }
__device__ __forceinline__ double row_up2(double v) { const double r = __shfl_up(v, 2); return (lane_id() & 15) < 2 ? 0.0 : r; }
__device__ __forceinline__ double row_down2(double v) { const double r = __shfl_down(v, 2); return (lane_id() & 15) > 13 ? 0.0 : r; }
__device__ __forceinline__ double lane_bcast(double v, int src) { return __shfl(v, src); }
__device__ __forceinline__ int lane_bcast_i(int v, int src) { return __shfl(v, src); }
__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 1; o < 16; o <<= 1) { const double r = __shfl_up(v, o); if ((lane_id() & 15) >= o) v += r; }
  return (lane_bcast(v, 15) + lane_bcast(v, 31)) + (lane_bcast(v, 47) + lane_bcast(v, 63));
}
#endif

// a set of wave-uniform values the compiler must have in registers HERE: one empty asm that reads them all -- the loads that produce them are issued together above it
// and waited for once (LLVM otherwise sinks every load into the branch that uses it: one exposed LDS round trip per test, and a wave that runs alone on its SIMD has
// nothing to hide it behind)
#ifdef PL_WAVE_EMU
template <class... T> __device__ __forceinline__ void pl_pin(const T&...) {}
#else
// (input-only: the values keep their identity -- and with it the compiler's knowledge that they are wave-uniform, i.e. scalar branches on them; an in/out operand made
//  every test below a divergent one, with exec-mask bookkeeping around each arm)
__device__ __forceinline__ void pl_pin1(const double& a) { __asm__ volatile("" :: "v"(a)); }
template <class... T> __device__ __forceinline__ void pl_pin(const T&... v) { (pl_pin1(v), ...); }
#endif

// ---- reciprocal, quotient and square root without the IEEE corner-case scaffolding (r06) ----
// hipcc lowers an fp64 division to v_div_scale x2 / v_rcp_f64 / 5 fma / v_mul / v_div_fmas / v_div_fixup: 11 dependent VALU instructions of which four only serve operands
// outside the normal range, and `sqrt` to 16 (ldexp scaling, class test, selects).  Every operand of the step loop is a physical quantity well inside the normal range, and a
// single wavefront per SIMD pays every instruction of the chain in full (r05 PMC: 12 % of the VALU instructions of a step were division scaffolding).
//   pl_rcp(x):  v_rcp_f64 (about 23 bits) + two Newton steps: 5 instructions, error below 1 ulp (not correctly rounded)
//   pl_div(a, b): a * pl_rcp(b) + one residual correction: 8 instructions, correctly rounded except in rare half-ulp ties
//   pl_sqrt(x): v_rsq_f64 + Goldschmidt / Newton as the compiler's own lowering, x > 0 normal (x = 0 -> 0 through one select)
// x = 0 / inf / NaN give NaN or inf as they come; the callers either cannot see them or treat a NaN as a failed iteration (ida_nls).  The results differ from the IEEE quotient
// by at most one rounding: far below the 1e-12 residual parity bar (tests/parity.py).  -DPL_IEEE_DIV restores the compiler's division everywhere (same-box A/B builds).
#if defined(PL_WAVE_EMU) || defined(PL_IEEE_DIV)
__device__ __forceinline__ double pl_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ double pl_div(double a, double b) { return a / b; }
__device__ __forceinline__ double pl_sqrt(double x) { return sqrt(x); }
#else
__device__ __forceinline__ double pl_rcp(double x) {
  double r = __builtin_amdgcn_rcp(x);
  double e = __builtin_fma(-x, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-x, r, 1.0);
  return __builtin_fma(r, e, r);
}
__device__ __forceinline__ double pl_div(double a, double b) {
  const double r = pl_rcp(b);
  const double q = a * r;
  return __builtin_fma(__builtin_fma(-b, q, a), r, q);
}
__device__ __forceinline__ double pl_sqrt(double x) {
  const double xs = x > 0.0 ? x : 1.0;
  const double y = __builtin_amdgcn_rsq(xs);
  double g = xs * y, h = 0.5 * y;
  const double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, xs), h, g);
  g = __builtin_fma(__builtin_fma(-g, g, xs), h, g);
  return x > 0.0 ? g : x;          // (0 -> 0, NaN -> NaN, negative: the argument as it came -- the callers clamp first)
}
#endif
// s = sqrt(x) and rs = 1 / sqrt(x) from ONE v_rsq_f64 (x > 0 and normal: the caller clamps): the Goldschmidt pair (g, h) -> (sqrt x, 1 / (2 sqrt x)) carries both
#if defined(PL_WAVE_EMU) || defined(PL_IEEE_DIV)
__device__ __forceinline__ void pl_sqrt_rsqrt(double x, double& s, double& rs) { s = sqrt(x); rs = 1.0 / s; }
#else
__device__ __forceinline__ void pl_sqrt_rsqrt(double x, double& s, double& rs) {
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
  g = __builtin_fma(__builtin_fma(-g, g, x), h, g);
  r = __builtin_fma(-h, g, 0.5);
  h = __builtin_fma(h, r, h);
  s = g; rs = h + h;
}
#endif

// x^(-1/n), n = 2 .. 6: IDA's step-size ratio (2 err + 1e-4)^(-1/(k+1)) (SUNRpowerR; the oracle calls pow).  ocml's fp64 log + exp are ~250 dependent instructions of
// double-double arithmetic -- at every accepted step, for a number that is then clipped to [0.5, 0.9] or compared with 1 and 2.  Here: the fp32 transcendental unit's log2 / exp2
// for seven digits, then two Newton steps on y^-n = x in fp64 (error (n+1)/2 d^2: 4e-7 -> 6e-13 -> rounding): ~25 instructions, within 1-2 ulp of the exact power like pow itself.
#if defined(PL_WAVE_EMU) || defined(PL_IEEE_DIV)
__device__ __forceinline__ double pl_inv_root(double x, int n) { return exp(-log(x) / n); }
#else
__device__ __forceinline__ double pl_inv_root(double x, int n) {
  const double rn = n == 2 ? 0.5 : n == 3 ? 1.0 / 3 : n == 4 ? 0.25 : n == 5 ? 0.2 : 1.0 / 6;          // (n = k + 1 with the BDF order k = 1 .. 5)
  double y = (double)__builtin_amdgcn_exp2f(-__builtin_amdgcn_logf((float)x) * (float)rn);
  _Pragma("unroll") for (int it = 0; it < 2; it++) {
    double p = y * y;
    if (n == 3) p *= y; else if (n == 4) p *= p; else if (n == 5) p = p * p * y; else if (n != 2) p = p * p * p;
    y = __builtin_fma(y * rn, __builtin_fma(-x, p, 1.0), y);
  }
  return y;
}
#endif

// ---- particle mat-vecs without LDS traffic for the vector operand (r06) ----
// acc += (c of lane K of this lane's 16-lane DPP row) * m  as ONE instruction: v_fmac_f64 with the DPP control row_newbcast:K (gfx90a+: the only DPP control the fp64 ALU
// takes).  With one particle per DPP row -- lane r of the row holds entry r of the particle's vector -- row r of a radial operator times that vector is N_r of these, and the
// vector costs ONE LDS load per lane and pass instead of N_r (r05: 40 loads + 40 fma per lane for the four 6-particle passes; now 5 loads + 50 fused broadcast-fma for five
// 4-particle passes).  hipcc has no intrinsic that selects the DPP form of an fp64 fma (it emits v_mov_b64_dpp + v_fma: two instructions), hence the inline assembly; the
// operand that is broadcast comes from an LDS load (csd_settle), never from a VALU result, so the VALU-write -> DPP-read hazard (2 wait states) the assembler cannot see
// does not arise.  All 64 lanes must be active.  K is a compile-time constant (static_for).
#ifdef PL_WAVE_EMU
template <int K> __device__ __forceinline__ void rowb_fmac(double& acc, double c, double m) { acc += __shfl(c, (lane_id() & ~15) + K) * m; }
template <int N> __device__ __forceinline__ void csd_settle(double (&)[N]) {}
#else
template <int K> __device__ __forceinline__ void rowb_fmac(double& acc, double c, double m) {
  __asm__("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(c), "v"(m), "n"(K));
}
// the broadcast operands, after their loads have landed and two wait states later: whatever wrote the registers last, a DPP read is safe from here on
// (the wait states sit in an asm that takes the operands in and hands them out: whatever wrote them -- an LDS load or a VALU instruction -- is done before it, and every
//  rowb_fmac that reads them comes after it BY DATA FLOW, not by the order of statements)
template <int N> __device__ __forceinline__ void csd_settle(double (&c)[N]) {
  if constexpr (N == 5) __asm__ volatile("s_nop 1" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]), "+v"(c[4]));
  else if constexpr (N == 4) __asm__ volatile("s_nop 1" : "+v"(c[0]), "+v"(c[1]), "+v"(c[2]), "+v"(c[3]));
  else { _Pragma("unroll") for (int k = 0; k < N; k++) __asm__ volatile("s_nop 1" : "+v"(c[k])); }
}
#endif
template <int K, int NK, class F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (K < NK) { f(std::integral_constant<int, K>{}); static_for<K + 1, NK>(f); }
}
#ifdef PL_NO_CSDPP
template <class M> constexpr bool PL_CSDPP = false;
#else
template <class M> constexpr bool PL_CSDPP = M::SD == 0 && !M::THERMAL && !M::W2;
#endif
#ifdef PL_NO_THROWB
constexpr bool PL_THROWB = false;
#else
constexpr bool PL_THROWB = true;          // the thermal model's particle phases in the row layout (dfn_thermal.h)
#endif

// a value the compiler must materialise: a product passed through it is ROUNDED before it enters a sum (no fma contraction) -- the reference's operation order of the
// PLH_PREC_F64_REFORDER variants (Julia does not contract a*b + c; neither does the oracle's gcc build for x86-64)
#ifdef PL_WAVE_EMU
__device__ __forceinline__ double pl_rounded(double v) { volatile double r = v; return r; }
#else
__device__ __forceinline__ double pl_rounded(double v) { __asm__ volatile("" : "+v"(v)); return v; }
#endif

// Phi_s row of the PLH_PREC_F64_REFORDER variants: block_tridiag(N) * Phi_s .- f (residuals.jl:656-703) summed in the order of the notebook-pinned oracle's generated code
// (oracle/gen/lco_iso.c: `-j x + Phi[i-1] - 2 Phi[i] + Phi[i+1]`, left to right): the source term src = f / sigma_eff ~ 1e-6 V joins a potential of ~4 V BEFORE the
// Laplacian cancels, so the row is quantised at ulp(Phi_s) = 8.9e-16 V in the cathode (1.4e-17 in the anode).  J^-1 turns that into ~5e-11 V of common-mode noise in Phi_e / Phi_s
// and 1e-9 in I -- 4e-8 ... 4e-7 in the weighted norm, above the local error of the first steps of a :hold leg (h restarts at 1e-3 s) -- and IDA's start-up order selection
// reads it.  The oracle with this order reproduces the reference notebook's 37-point V-hold leg (t_end 2440.61 s); with the Laplacian formed first (what the default build
// does, `lap - src`) it takes 38 points and ends at 2441.33 s: the reference's own generated code has this rounding, and this variant reproduces it (r05).
// The first cathode row is `(-Phi[0] + Phi[1]) - src` in the generated code: clean.
__device__ __forceinline__ double phi_s_row_reford(int i, bool first, bool last, double ps_p, double ps, double ps_n, double src) {
  if (first && i == 0) return pl_rounded(-ps + ps_n) - src;
  const double t = pl_rounded(-src);
  if (first) return pl_rounded(pl_rounded(-ps) + t) + ps_n;                                  // -Phi[i] - src + Phi[i+1]
  if (last) return pl_rounded(t - ps) + ps_p;                                                // -src - Phi[i] + Phi[i-1]
  return pl_rounded(pl_rounded(t + ps_p) - 2.0 * ps) + ps_n;                                 // -src + Phi[i-1] - 2 Phi[i] + Phi[i+1]
}

// sum over all lanes of the cell: one wave, or (M::W2) both waves through two LDS slots and two barriers (every thread of the workgroup must call it)
template <class M, class LDS> __device__ __forceinline__ double block_sum(LDS& S, double v) {
  const double s = wave_sum(v);
  if constexpr (!M::W2) return s;
  else {
    if (lane_id() == 0) S.red[wave_id()] = s;
    __syncthreads();
    const double r = S.red[0] + S.red[1];
    __syncthreads();
    return r;
  }
}
template <class M, class LDS> __device__ __forceinline__ void block_sum3(LDS& S, double& a, double& b, double& c) {
  a = wave_sum(a); b = wave_sum(b); c = wave_sum(c);
  if constexpr (M::W2) {
    if (lane_id() == 0) { S.red[wave_id() * 4] = a; S.red[wave_id() * 4 + 1] = b; S.red[wave_id() * 4 + 2] = c; }
    __syncthreads();
    a = S.red[0] + S.red[4]; b = S.red[1] + S.red[5]; c = S.red[2] + S.red[6];
    __syncthreads();
  }
}

// ------------------------------------------------------------------------------------------------------------------
// closures (reference src/physics_equations/custom_functions.jl)
// ------------------------------------------------------------------------------------------------------------------
// K_eff(c_e, T), custom_functions.jl:96 ; returns K and dK/dc
__device__ __forceinline__ void keff(double c, double T, double& K, double& dK) {
  const double A = -10.5 + 0.668 * 1e-3 * c + 0.494 * 1e-6 * c * c;
  const double B = 0.074 - 1.78 * 1e-5 * c - 8.86 * 1e-10 * c * c;
  const double C = -6.96 * 1e-5 + 2.8 * 1e-8 * c;
  const double P = A + B * T + C * T * T;
  const double dP = (0.668 * 1e-3 + 2 * 0.494 * 1e-6 * c) + (-1.78 * 1e-5 - 2 * 8.86 * 1e-10 * c) * T + 2.8 * 1e-8 * T * T;
  K = 1e-4 * c * P * P;
  dK = 1e-4 * (P * P + 2.0 * c * P * dP);
}

// exp and expm1 for the arguments the model produces (finite, |x| < 700): ocml's algorithm -- n = rint(x / ln 2), r = x - n ln 2 in two FMAs, a polynomial on |r| <= ln 2 / 2,
// ldexp -- without its overflow / underflow / NaN selects, and expm1 as 2^n expm1(r) + (2^n - 1) from the same polynomial (exact for n = 0: the small overpotentials of a
// rest keep their relative accuracy).  exp(r) - 1 - r = r^2 q(r), q = sum_{k = 2..13} r^(k - 2) / k!: truncation 4e-18 relative; against 50-digit values both are within
// 0.82 ulp on [-30, 30] and around zero.  r06 (the instruction diet): same box C2 1.155 -> 1.13 ms, C4 9.29 -> 9.10 ms, C5 33.2 -> 33.0 ms,
// C3 18.23 -> 18.15 ms.  -DPL_OCML_EXP: the library functions (A/B builds); the emulator build keeps libm's.
#if defined(PL_WAVE_EMU) || defined(PL_OCML_EXP)
__device__ __forceinline__ double pl_exp(double x) { return exp(x); }
__device__ __forceinline__ double pl_expm1(double x) { return expm1(x); }
#else
struct PlExpRed { double s, n; };                                                  // expm1(r) of the reduced argument and n (by value: a reference parameter makes
                                                                                    // the late-inlining builds cast a private address to a generic one, and hipcc 7.2
                                                                                    // emits an illegal V_CMP_NE_U32 on src_shared_base for it under the iterative scheduler)
__device__ __forceinline__ PlExpRed pl_expm1_reduced(double x) {
  const double n = __builtin_rint(x * 1.4426950408889634);
  double r = __builtin_fma(n, -0.6931471805599453, x);
  r = __builtin_fma(n, -2.3190468138462996e-17, r);
  double p = 1.6059043836821613e-10;                                              // 1 / 13!
  p = __builtin_fma(p, r, 2.08767569878681e-09);                                  // 1 / 12!
  p = __builtin_fma(p, r, 2.505210838544172e-08);                                 // 1 / 11!
  p = __builtin_fma(p, r, 2.755731922398589e-07);                                 // 1 / 10!
  p = __builtin_fma(p, r, 2.7557319223985893e-06);                                // 1 / 9!
  p = __builtin_fma(p, r, 2.48015873015873e-05);                                  // 1 / 8!
  p = __builtin_fma(p, r, 0.0001984126984126984);                                 // 1 / 7!
  p = __builtin_fma(p, r, 0.001388888888888889);                                  // 1 / 6!
  p = __builtin_fma(p, r, 0.008333333333333333);                                  // 1 / 5!
  p = __builtin_fma(p, r, 0.041666666666666664);                                  // 1 / 4!
  p = __builtin_fma(p, r, 0.16666666666666666);                                   // 1 / 3!
  p = __builtin_fma(p, r, 0.5);
  return {__builtin_fma(r * r, p, r), n};
}
__device__ __forceinline__ double pl_exp(double x) { const PlExpRed e = pl_expm1_reduced(x); return ldexp(1.0 + e.s, (int)e.n); }
__device__ __forceinline__ double pl_expm1(double x) { const PlExpRed e = pl_expm1_reduced(x); const double t = ldexp(1.0, (int)e.n); return __builtin_fma(t, e.s, t - 1.0); }
#endif
// x^w and x^(w - 1) for x > 0 normal (the SEI side reaction's (I / I_1C)^w): exp(w log x) with the classic log -- x = m 2^k, m in [sqrt(1/2), sqrt(2)), f = m - 1,
// s = f / (2 + f), log(1 + f) = f - f^2 / 2 + s (f^2 / 2 + R(s^2)), R of degree 7 (fdlibm's coefficients, < 1 ulp) -- and the second power from the first by one reciprocal.
// ocml's pow carries the logarithm in double-double and three IEEE divisions: ~200 instructions a call, three calls in a Jacobian pass of the anode's nodes.  The result is
// within |w log x| + 1 ulp of the exact power (a few ulp for the arguments of this row); the emulator build and -DPL_OCML_EXP keep the library's pow.
#if defined(PL_WAVE_EMU) || defined(PL_OCML_EXP)
__device__ __forceinline__ void pl_pow_pair(double x, double w, double& pw, double& pwm1) { pw = pow(x, w); pwm1 = pow(x, w - 1.0); }
__device__ __forceinline__ double pl_pow(double x, double w) { return pow(x, w); }
#else
__device__ __forceinline__ double pl_log(double x) {
  double m = __builtin_amdgcn_frexp_mant(x);                                        // [1/2, 1)
  int k = __builtin_amdgcn_frexp_exp(x);
  const bool lo = m < 0.7071067811865476;
  m = lo ? m + m : m; k = lo ? k - 1 : k;
  const double f = m - 1.0, s = f * pl_rcp(2.0 + f), z = s * s, w = z * z, dk = (double)k;
  const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1, hfsq = 0.5 * f * f;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (s * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}
__device__ __forceinline__ double pl_pow(double x, double w) { return pl_exp(w * pl_log(x)); }
__device__ __forceinline__ void pl_pow_pair(double x, double w, double& pw, double& pwm1) { pw = pl_pow(x, w); pwm1 = pw * pl_rcp(x); }
#endif

// OCV_LCO, custom_functions.jl:123-136 ; U(x,T) and dU/dx
__device__ __forceinline__ void ocv_lco(double x, double T, int iso_ref, double& U, double& dUdx) {
  const double x2 = x * x, x4 = x2 * x2, x6 = x4 * x2, x8 = x4 * x4, x10 = x8 * x2;
  const double P = -4.656 + 88.669 * x2 - 401.119 * x4 + 342.909 * x6 - 462.471 * x8 + 433.434 * x10;
  const double Q = -1 + 18.933 * x2 - 79.532 * x4 + 37.311 * x6 - 73.083 * x8 + 95.96 * x10;
  const double dP = x * (2 * 88.669 - 4 * 401.119 * x2 + 6 * 342.909 * x4 - 8 * 462.471 * x6 + 10 * 433.434 * x8);
  const double dQ = x * (2 * 18.933 - 4 * 79.532 * x2 + 6 * 37.311 * x4 - 8 * 73.083 * x6 + 10 * 95.96 * x8);
  const double rQ = pl_rcp(Q);                       // one reciprocal for the value and the derivative: U = P / Q, dU/dx = (P' - U Q') / Q
  U = P * rQ;
  dUdx = (dP - U * dQ) * rQ;
  if (!iso_ref) {
    const double x3 = x2 * x;
    const double n = 0.199521039 - 0.928373822 * x + 1.364550689000003 * x2 - 0.6115448939999998 * x3;
    const double d = 1 - 5.661479886999997 * x + 11.47636191 * x2 - 9.82431213599998 * x3 + 3.048755063 * x4;
    const double dn = -0.928373822 + 2 * 1.364550689000003 * x - 3 * 0.6115448939999998 * x2;
    const double dd = -5.661479886999997 + 2 * 11.47636191 * x - 3 * 9.82431213599998 * x2 + 4 * 3.048755063 * x3;
    const double rd = pl_rcp(d), nd = n * rd;
    U += -0.001 * nd * (T - TREF);
    dUdx += -0.001 * (dn - nd * dd) * rd * (T - TREF);
  }
}

// OCV_LiC6, custom_functions.jl:139-152
__device__ __forceinline__ void ocv_lic6(double x, double T, int iso_ref, double& U, double& dUdx) {
  // one reciprocal (1 / x) and one reciprocal square root (of max(x, 1e-4)) serve the five quotients and two roots of the reference's expression; below x = 1e-4 --
  // an anode emptied to a ten-thousandth of its capacity -- sqrt(x) and its reciprocal take the slow path
  const bool big = x > 1e-4;
  const double xm = big ? x : 1e-4;
  double s1, rs1; pl_sqrt_rsqrt(xm, s1, rs1);
  double s0 = s1, rs0 = rs1;
  if (!big) { s0 = sqrt(x > 0.0 ? x : 0.0); rs0 = 1.0 / s0; }
  const double rx = pl_rcp(x), rx2 = rx * rx;
  const double e1 = pl_exp(0.9 - 15 * x), e2 = pl_exp(0.4465 * x - 0.4108);
  U = 0.7222 + 0.1387 * x + 0.029 * s0 - 0.0172 * rx + 0.0019 * (rs1 * rx) + 0.2808 * e1 - 0.7984 * e2;
  double d = 0.1387 + 0.0172 * rx2 - 0.2808 * 15 * e1 - 0.7984 * 0.4465 * e2;
  if (x > 0.0) d += 0.029 * 0.5 * rs0;
  d += (big ? 0.0019 * (-1.5) : -0.0019) * (rx2 * rs1);
  dUdx = d;
  if (!iso_ref) {
    const double x2 = x * x, x3 = x2 * x, x4 = x2 * x2, x5 = x4 * x, x6 = x3 * x3, x7 = x6 * x, x8 = x4 * x4;
    const double n = 0.001 * (0.005269056 + 3.299265709 * x - 91.79325798 * x2 + 1004.911008 * x3 - 5812.278127 * x4 + 19329.7549 * x5 - 37147.8947 * x6 + 38379.18127 * x7 - 16515.05308 * x8);
    const double q = 1 - 48.09287227 * x + 1017.234804 * x2 - 10481.80419 * x3 + 59431.3 * x4 - 195881.6488 * x5 + 374577.3152 * x6 - 385821.1607 * x7 + 165705.8597 * x8;
    const double dn = 0.001 * (3.299265709 - 2 * 91.79325798 * x + 3 * 1004.911008 * x2 - 4 * 5812.278127 * x3 + 5 * 19329.7549 * x4 - 6 * 37147.8947 * x5 + 7 * 38379.18127 * x6 - 8 * 16515.05308 * x7);
    const double dq = -48.09287227 + 2 * 1017.234804 * x - 3 * 10481.80419 * x2 + 4 * 59431.3 * x3 - 5 * 195881.6488 * x4 + 6 * 374577.3152 * x5 - 7 * 385821.1607 * x6 + 8 * 165705.8597 * x7;
    const double rq = pl_rcp(q), nq = n * rq;
    U += nq * (T - TREF);
    dUdx += (dn - nq * dq) * rq * (T - TREF);
  }
}

// OCV_NMC, custom_functions.jl:154-162 (dU/dT = 0)
__device__ __forceinline__ void ocv_nmc(double x, double& U, double& dUdx) {
  U = -10.72 * x * x * x * x + 23.88 * x * x * x - 16.77 * x * x + 2.595 * x + 4.563;
  dUdx = -4 * 10.72 * x * x * x + 3 * 23.88 * x * x - 2 * 16.77 * x + 2.595;
}
// OCV_LiC6_with_NMC, custom_functions.jl:164-174 (dU/dT = 0)
__device__ __forceinline__ void ocv_lic6_nmc(double x, double& U, double& dUdx) {
  const double e1 = pl_exp(-61.79 * x), e2 = pl_exp(-665.8 * x), e3 = pl_exp(39.42 * x - 41.92);
  const double a1 = 25.59 * x - 4.099, a2 = 32.49 * x - 15.74;
  U = 0.1493 + 0.8493 * e1 + 0.3824 * e2 - e3 - 0.03131 * atan(a1) - 0.009434 * atan(a2);
  dUdx = -61.79 * 0.8493 * e1 - 665.8 * 0.3824 * e2 - 39.42 * e3 - 0.03131 * 25.59 * pl_rcp(1.0 + a1 * a1) - 0.009434 * 32.49 * pl_rcp(1.0 + a2 * a2);
}
// D_eff(c_e, T), custom_functions.jl:83 (NMC system default, params.jl:407): 1e-4 * 10^(-4.43 - 54/(T - 229 - 5e-3 c) - 0.22e-3 c)
__device__ __forceinline__ void deff_nmc(double c, double T, double& D, double& dD) {
  const double LN10 = 2.302585092994046;
  const double u = T - 229 - 5e-3 * c;
  const double ru = pl_rcp(u);
  const double ex = -4.43 - 54.0 * ru - 0.22e-3 * c;
  D = 1e-4 * pl_exp(LN10 * ex);
  dD = D * LN10 * (-54.0 * 5e-3 * (ru * ru) - 0.22e-3);
}

// LGM50 closures (reference src/params.jl:563-572, 627-636, 646, 660; dU/dT = 0)
PL_DEV void ocv_nmc_lgm50(double x, double& U, double& dUdx) {
  const double t1 = tanh(18.5138 * (x - 0.5542)), t2 = tanh(15.7890 * (x - 0.3117)), t3 = tanh(15.9308 * (x - 0.3120));
  U = -0.8090 * x + 4.4875 - 0.0428 * t1 - 17.7326 * t2 + 17.5842 * t3;
  dUdx = -0.8090 - 0.0428 * 18.5138 * (1.0 - t1 * t1) - 17.7326 * 15.7890 * (1.0 - t2 * t2) + 17.5842 * 15.9308 * (1.0 - t3 * t3);
}
PL_DEV void ocv_lic6_lgm50(double x, double& U, double& dUdx) {
  const double e1 = exp(-39.3631 * x), t1 = tanh(29.8538 * (x - 0.1234)), t2 = tanh(14.9159 * (x - 0.2769)), t3 = tanh(30.4444 * (x - 0.6103)), t4 = tanh(17.08 * (x - 1.0));
  U = 1.9793 * e1 + 0.15561 - 0.0909 * t1 - 0.04478 * t2 - 0.0205 * t3 - 0.09259 * t4;
  dUdx = -39.3631 * 1.9793 * e1 - 0.0909 * 29.8538 * (1.0 - t1 * t1) - 0.04478 * 14.9159 * (1.0 - t2 * t2) - 0.0205 * 30.4444 * (1.0 - t3 * t3) - 0.09259 * 17.08 * (1.0 - t4 * t4);
}
PL_DEV void deff_lgm50(double c, double De, double& D, double& dD) {
  const double x = c * 1e-3;
  D = De * (x * x - 4.516715942688196 * x + 5.5287696156470325);
  dD = De * (2.0 * x - 4.516715942688196) * 1e-3;
}
PL_DEV void keff_lgm50(double c, double& K, double& dK) {
  const double x = c * 1e-3, sx = sqrt(x);
  K = 0.1297 * x * x * x - 2.51 * x * sx + 3.329 * x;
  dK = (3.0 * 0.1297 * x * x - 1.5 * 2.51 * sx + 3.329) * 1e-3;
}

// sinh and cosh from ONE expm1 and one division (ocml's sinh alone costs ~670 cycles of dependent latency on gfx950):
//   u = e^x - 1 ;  sinh x = (u + u/(u+1))/2 ,  cosh x = sinh x + 1/(u+1)   -- accurate for small |x| as well (no cancellation)
__device__ __forceinline__ void sinh_cosh(double x, double& sh, double& ch) {
  const double u = pl_expm1(x);
  const double r = pl_rcp(u + 1.0);
  sh = 0.5 * (u + u * r);
  ch = sh + r;
}

// harmonic-mean edge interpolation H(beta; a, b) = ab/(beta b + (1-beta) a), numerical_tools.jl:106-156
__device__ __forceinline__ double hmean(double beta, double a, double b) { return a * b / (beta * b + (1.0 - beta) * a); }

// thermal-model counterparts (dfn_thermal.h); the generic entry points below dispatch to them when M::THERMAL
template <class M, bool INIT = true> PL_DEV void thermal_setup(CellLDS<M>& S, const Tables* __restrict__ tb, const double* __restrict__ th);
template <bool WANT_RES, bool WANT_JAC, class M>
PL_DEV void thermal_node_pass(CellLDS<M>& S, const double* Y, const double* YP, double* Fo, int mode, double value);
template <bool WANT_JAC, class M> PL_DEV void thermal_cs_rows(CellLDS<M>& S, const Tables* __restrict__ tb, const double* Y, const double* YP, double* Fo);
template <class M> PL_DEV_FACTOR void thermal_factor(CellLDS<M>& S, LaneRegs& R, const Tables* __restrict__ tb, double cj, int mode, bool alg_only);
template <class M> PL_DEV void thermal_solve(CellLDS<M>& S, LaneRegs& R, const Tables* __restrict__ tb, double* b, int mode, bool alg_only);
template <bool FROZEN = false, class M> PL_DEV double thermal_jac_entry(const CellLDS<M>& S, const Tables* __restrict__ tb, unsigned w, double cj);
constexpr int PL_MODE_DT_TWIN = 16;   // dT control row with YP_T replaced by rhs_T(Y): the consistent-initialisation form (scalar_residual.jl:347-372)

// ------------------------------------------------------------------------------------------------------------------
// per-cell constants from theta (build_auxiliary_states!, reference aux...jl:6-52, and the Arrhenius closures)
// ------------------------------------------------------------------------------------------------------------------
// INIT = false: only the numbers derived from theta are (re)computed -- the Jacobian pools, the factors, the LDS copies of the radial tables and the particle registers are
// left alone (the parameter-sensitivity phase evaluates residuals with a perturbed theta row between two solves of the integrator's factorisation: dfn_sens.h)
template <class M, bool INIT = true>
PL_DEV void cell_setup(CellLDS<M>& S, LaneRegs& R, const Tables* __restrict__ tb, const double* __restrict__ th) {
  PL_MODEL(M);
  const int lane = lane_id();
  if (lane == 0 && wave_id() == 0) {
    CellConst& c = S.cc;
    const int* ix = tb->thidx;
    const double lp = th[ix[K_l_p]], ls = th[ix[K_l_s]], ln = th[ix[K_l_n]];
    c.h[0] = lp / NP; c.h[1] = ls / NS; c.h[2] = ln / NN;
    const double efp = th[ix[K_eps_fp]], efn = th[ix[K_eps_fn]];
    const double esp = 1.0 - (efp + th[ix[K_eps_p]]);       // active_material, aux...jl:537-545
    const double esn = 1.0 - (efn + th[ix[K_eps_n]]);
    c.eps[0] = 1.0 - (efp + esp);                           // build_ϵ!, aux...jl:92-105
    c.eps[1] = th[ix[K_eps_s]];
    c.eps[2] = 1.0 - (efn + esn);
    c.bf[0] = pow(c.eps[0], th[ix[K_brugg_p]]);
    c.bf[1] = pow(c.eps[1], th[ix[K_brugg_s]]);
    c.bf[2] = pow(c.eps[2], th[ix[K_brugg_n]]);
    if (M::CHEM == PLH_CHEM_LCO_LIC6) {                     // D_eff_linear, custom_functions.jl:59-69
      c.Dc[0] = th[ix[K_D_p]] * c.bf[0]; c.Dc[1] = th[ix[K_D_s]] * c.bf[1]; c.Dc[2] = th[ix[K_D_n]] * c.bf[2];
    } else { c.Dc[0] = c.Dc[1] = c.Dc[2] = 0.0; }           // NMC, LGM50: D_eff(c_e[, T]) per control volume (node pass)
    c.De = M::CHEM == PLH_CHEM_LGM50 ? th[ix[K_D_e]] : 0.0;
    const double Rp_p = th[ix[K_Rp_p]], Rp_n = th[ix[K_Rp_n]];
    c.a_p = 3 * esp / Rp_p; c.a_n = 3 * esn / Rp_n;         // build_a!, aux...jl:124-139
    c.sig_p = th[ix[K_sig_p]] * esp; c.sig_n = th[ix[K_sig_n]] * esn;
    const double T0 = th[ix[K_T0]];
    c.ce0 = th[ix[K_c_e0]];
    c.T0 = T0; c.iso_ref = (T0 == TREF);
    double arr_kp = 1.0, arr_kn = 1.0, arr_dp = 1.0, arr_dn = 1.0;   // temperature_switch, custom_functions.jl:1,16-31,44-57
    if (!c.iso_ref && !M::THERMAL) {   // with temperature = true the Arrhenius factors are evaluated per node from T(x)
      const double dT = 1.0 / T0 - 1.0 / TREF;
      arr_kp = exp(-(th[ix[K_Ea_k_p]] / RGAS) * dT); arr_kn = exp(-(th[ix[K_Ea_k_n]] / RGAS) * dT);
      arr_dp = exp(-(th[ix[K_Ea_D_sp]] / RGAS) * dT); arr_dn = exp(-(th[ix[K_Ea_D_sn]] / RGAS) * dT);
    }
    c.kp = th[ix[K_k_p]] * arr_kp; c.kn = th[ix[K_k_n]] * arr_kn;
    c.cmaxp = th[ix[K_c_max_p]]; c.cmaxn = th[ix[K_c_max_n]];
    c.kap_p = th[ix[K_D_sp]] * arr_dp / (Rp_p * Rp_p); c.kap_n = th[ix[K_D_sn]] * arr_dn / (Rp_n * Rp_n);
    c.bj_p = -tb->BJ / Rp_p; c.bj_n = -(NR_EQ ? tb->BJ : tb->BJ_N) / Rp_n;
    c.fRT = 0.5 * FAR / (RGAS * T0);
    c.tplus = th[ix[K_tplus]];
    c.Kfac = 2 * RGAS * (1 - c.tplus) * 1.0 / FAR;          // nu = 1: thermodynamic_factor_linear, custom_functions.jl:177
    c.thmin_p = th[ix[K_th_min_p]]; c.thmax_p = th[ix[K_th_max_p]];
    c.thmin_n = th[ix[K_th_min_n]]; c.thmax_n = th[ix[K_th_max_n]];
    const double qa = esp * lp * c.cmaxp * (c.thmin_p - c.thmax_p);
    const double qb = esn * ln * c.cmaxn * (c.thmax_n - c.thmin_n);
    c.I1C = (FAR / 3600.0) * (qa < qb ? qa : qb);           // calc_I1C, aux...jl:632-647
    c.JI0 = c.I1C * c.h[0] / c.sig_p;                       // d(Phi_s row of first p node)/dI
    c.JI29 = -c.I1C * c.h[2] / c.sig_n;                     // d(Phi_s row of last n node)/dI
    for (int q = 0; q < 3; q++) { c.rh[q] = 1.0 / c.h[q]; c.reps[q] = 1.0 / c.eps[q]; }
    c.rd_ps = 1.0 / (c.h[0] / 2 + c.h[1] / 2); c.rd_sn = 1.0 / (c.h[1] / 2 + c.h[2] / 2);
    c.beta_ps = (c.h[0] / 2) / (c.h[1] / 2 + c.h[0] / 2); c.beta_sn = (c.h[1] / 2) / (c.h[2] / 2 + c.h[1] / 2);
    c.rsg_p = 1.0 / c.sig_p; c.rsg_n = 1.0 / c.sig_n; c.rcm_p = 1.0 / c.cmaxp; c.rcm_n = 1.0 / c.cmaxn;
    c.Dh_ps = hmean(c.beta_ps, c.Dc[0], c.Dc[1]); c.Dh_sn = hmean(c.beta_sn, c.Dc[1], c.Dc[2]);   // D_eff_linear: constant edge means
    S.tb = tb;
    if constexpr (!M::THERMAL) {
      for (int q = 0; q < 3; q++) {
        const bool pq = q == 0;
        S.nt.sec[q] = {c.h[q], c.eps[q], c.bf[q], c.Dc[q], c.rh[q], c.reps[q], pq ? c.a_p : c.a_n, pq ? c.cmaxp : c.cmaxn, pq ? c.kp : c.kn, pq ? c.rsg_p : c.rsg_n, pq ? c.rcm_p : c.rcm_n, 0.0};
        S.nt.edge[q] = {0.5, c.rh[q], c.Dc[q], 0.0};
      }
      S.nt.edge[3] = {c.beta_ps, c.rd_ps, c.Dh_ps, 0.0}; S.nt.edge[4] = {c.beta_sn, c.rd_sn, c.Dh_sn, 0.0};
    }
    c.R_SEI = c.rkag = c.Mrho = c.i0F = c.wexp = c.Uref = 0.0;
    c.EaKp = c.EaKn = c.EaDp = c.EaDn = 0.0; c.Tamb = 0.0;
    {   // quadratic / polynomial particle models (aux...jl:212-248, residuals.jl:108-127, 237-258); D_s_eff = D_s * Arrhenius factor
      const double Dsp = th[ix[K_D_sp]] * arr_dp, Dsn = th[ix[K_D_sn]] * arr_dn;
      const double den = M::SD == 2 ? 35.0 : 5.0;
      c.csj[0] = -Rp_p / (Dsp * den); c.csj[1] = -Rp_n / (Dsn * den);
      c.csq[0] = M::SD == 2 ? (Rp_p / (Dsp * 35.0)) * (8.0 * Dsp) : 0.0; c.csq[1] = M::SD == 2 ? (Rp_n / (Dsn * 35.0)) * (8.0 * Dsn) : 0.0;
      c.csr[0] = -3.0 / Rp_p; c.csr[1] = -3.0 / Rp_n;
      c.qj[0] = -(45.0 / 2.0) / (Rp_p * Rp_p); c.qj[1] = -(45.0 / 2.0) / (Rp_n * Rp_n);
    }
    c.lam[0] = c.lam[1] = 1.0; c.mhc_k0[0] = c.mhc_k0[1] = 0.0; c.rce0 = 1.0 / c.ce0;
    if constexpr (M::RXN == 1) {
      c.lam[0] = th[ix[K_lam_MHC_p]]; c.lam[1] = th[ix[K_lam_MHC_n]];
      for (int q = 0; q < 2; q++) { const double l = c.lam[q], a1 = 1.0 + sqrt(l); c.mhc_k0[q] = (q == 0 ? c.kp : c.kn) / ((1.0 - erf((l - sqrt(a1)) / (2.0 * sqrt(l)))) / 2.0); }
    }
    if constexpr (M::THERMAL) {
      c.iso_ref = 0;                                        // the temperature_switch always takes the exp branch (custom_functions.jl:1)
      c.Tamb = th[ix[K_T_amb]];
      c.EaKp = th[ix[K_Ea_k_p]] / RGAS; c.EaKn = th[ix[K_Ea_k_n]] / RGAS; c.EaDp = th[ix[K_Ea_D_sp]] / RGAS; c.EaDn = th[ix[K_Ea_D_sn]] / RGAS;
      for (int q = 0; q < 3; q++) c.r2h[q] = 1.0 / (2.0 * c.h[q]);
      c.qps_r = 2.0 / (3 * c.h[0] + c.h[1]); c.qps_l = 2.0 / (c.h[0] + 3 * c.h[1]);
      c.qsn_r = 2.0 / (3 * c.h[1] + c.h[2]); c.qsn_l = 2.0 / (c.h[1] + 3 * c.h[2]);
    }
    if constexpr (M::SEI) {
      c.R_SEI = th[ix[K_R_SEI]]; c.rkag = 1.0 / th[ix[K_k_n_aging]]; c.Mrho = th[ix[K_M_n]] / th[ix[K_rho_n]];
      c.i0F = th[ix[K_i_0_jside]] / FAR; c.wexp = th[ix[K_w]]; c.Uref = th[ix[K_Uref_s]];
      // residuals_SOH!, residuals.jl:278-297: rhs_SOH = F a_n/(3600 I1C) * trapz(x l_n, [extrap_0(j_s[1:3]); j_s; extrap_0(j_s[end:-1:end-2])])
      // (extrapolate_section / extrap_x_0, external.jl:469-523) -- linear in j_s, so it is a fixed weight vector
      double x[NN + 2], wq[NN + 2], e[3];
      x[0] = 0.0; x[NN + 1] = 1.0;
      for (int k = 0; k < NN; k++) x[k + 1] = 1.0 / (2 * NN) + k * ((1.0 - 1.0 / NN) / (NN - 1));
      for (int m = 0; m < NN + 2; m++) wq[m] = 0.5 * ((m > 0 ? x[m] - x[m - 1] : 0.0) + (m < NN + 1 ? x[m + 1] - x[m] : 0.0));
      for (int u = 0; u < 3; u++) {       // extrapolation to 0 of the parabola through (x1..x3, unit vector u)
        const double y0 = u == 0, y1 = u == 1, y2 = u == 2, x0 = x[1], x1 = x[2], x2 = x[3];
        const double q = (y2 - y0 - (x2 - x0) / (x1 - x0) * (y1 - y0)) / (x2 * x2 - x0 * x0 - (x1 * x1 - x0 * x0) / (x1 - x0) * (x2 - x0));
        e[u] = y0 - q * x0 * x0 - (y1 - y0 - q * (x1 * x1 - x0 * x0)) / (x1 - x0) * x0;
      }
      const double C = FAR * c.a_n / (3600.0 * c.I1C) * ln;
      for (int k = 0; k < NN; k++)
        S.sei.sohw[k] = C * (wq[k + 1] + (k < 3 ? wq[0] * e[k] : 0.0) + (k > NN - 4 ? wq[NN + 1] * e[NN - 1 - k] : 0.0));
      if constexpr (INIT) S.sei.cjf = 0.0;
    }
  }
  if constexpr (INIT) {
  if constexpr (M::SD == 0) { if (wave_id() == M::NWAVES - 1) {
    if constexpr (M::PHI_GLOBAL_ALL) {                      // M only: V, W, LAM are read from the tables at a factorisation (iso_factor)
      for (int k = lane; k < NR * NR; k += WAVE) { S.Mr[k] = tb->Mp()[k]; S.Ainv[0][k] = 0.0; S.Ainv[1][k] = 0.0; if constexpr (!NR_EQ) S.Mr[S.mr_el(1) + k] = tb->Mp(1)[k]; }
    } else {
    for (int k = lane; k < NR * NR; k += WAVE) { S.Mr[k] = tb->Mp()[k]; S.Mr[S.OFF_VR + k] = tb->Vp()[k]; S.Mr[S.OFF_WR + k] = tb->Wp()[k]; if constexpr (!M::THERMAL) { S.Ainv[0][k] = 0.0; S.Ainv[1][k] = 0.0; } }
    if constexpr (!M::THERMAL) { if (lane < NR) S.Mr[S.OFF_LAMR + lane] = tb->LAMp()[lane]; }
    if constexpr (!NR_EQ) {                                 // the anode's block (the tables are already zero-padded to the common stride: plh_model_create)
      for (int k = lane; k < NR * NR; k += WAVE) { S.Mr[S.mr_el(1) + k] = tb->Mp(1)[k]; S.Mr[S.mr_el(1) + S.OFF_VR + k] = tb->Vp(1)[k]; S.Mr[S.mr_el(1) + S.OFF_WR + k] = tb->Wp(1)[k]; }
      if constexpr (!M::THERMAL) { if (lane < NR) S.Mr[S.mr_el(1) + S.OFF_LAMR + lane] = tb->LAMp(1)[lane]; }
    } }
    } }
  for (int k = 0; k < LR_PASS; k++) { R.wreg[k] = 0.0; R.rcp[k] = 0.0; }
  // the LDS state vectors start at zero, padding included: the padding of M::VPAD must be (and then stays) zero, and the branch-free history sums of the integrator multiply
  // the orders that are not in use by 0.0 -- which must not meet the NaN a previous workgroup may have left behind
  for (int k = (int)threadIdx.x; k < M::NPAD; k += WAVE * M::NWAVES) {
    for (int j = 0; j < M::PHI_LDS; j++) S.phi[j][k] = 0.0;
    S.yy[k] = 0.0; S.yp[k] = 0.0; S.delta[k] = 0.0;
  }
  }
  PL_XSYNC();
  if constexpr (M::THERMAL) thermal_setup<M, INIT>(S, tb, th);
}

// initial_guess!, reference src/states_definition.jl:80-121
template <class M>
PL_DEV void cell_initial_guess(CellLDS<M>& S, double* Y, double SOC) {
  PL_MODEL(M);
  const int lane = lane_id();
  const CellConst& c = S.cc;
  const double csp = c.cmaxp * (SOC * (c.thmax_p - c.thmin_p) + c.thmin_p);
  const double csn = c.cmaxn * (SOC * (c.thmax_n - c.thmin_n) + c.thmin_n);
  double Up, Un, d;
  if (M::CHEM == PLH_CHEM_LCO_LIC6) { ocv_lco(csp / c.cmaxp, c.T0, c.iso_ref, Up, d); ocv_lic6(csn / c.cmaxn, c.T0, c.iso_ref, Un, d); }
  else if (M::CHEM == PLH_CHEM_LGM50) { ocv_nmc_lgm50(csp / c.cmaxp, Up, d); ocv_lic6_lgm50(csn / c.cmaxn, Un, d); }
  else { ocv_nmc(csp / c.cmaxp, Up, d); ocv_lic6_nmc(csn / c.cmaxn, Un, d); }
  _Pragma("unroll") for (int k__ = 0; k__ < NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wave_id()); vok<M>(k__, lane, wave_id())) {
    double v = 0.0;
    if (n < O_CS) v = c.ce0;
    else if (n < O_CS + (M::SD == 0 ? NP * NRP : NP)) v = csp;
    else if (n < N_CECS) v = csn;
    else if (M::THERMAL && n < M::O_T + NT) v = c.T0;
    else if (n >= O_PS && n < O_PS + NP) v = Up;
    else if (n >= O_PS + NP && n < O_PS + NJ) v = Un;
    else if (M::SEI && n == O_SOH) v = 1.0;               // film = 0, SOH = 1, j_s = 0 (states_definition.jl:80-121)
    Y[n] = v;
  }
  PL_XSYNC();
}

// ------------------------------------------------------------------------------------------------------------------
// node pass shared by the residual and the Jacobian partials.  Lane i < 30 owns control volume i and edge i (i|i+1).
// ------------------------------------------------------------------------------------------------------------------
template <bool WANT_RES, bool WANT_JAC, class M>
PL_DEV void iso_node_pass(CellLDS<M>& S, const double* Y, const double* YP, double* Fo, int mode, double value) {
  PL_MODEL(M);
  if constexpr (M::W2) { if (wave_id() != 0) return; }       // two waves per cell: the finite-volume rows belong to wave 0
  const int lane = lane_id_pred();
  const CellConst& c = S.cc;
  const int i = lane < NE ? lane : NE - 1;
  const bool act = lane < NE;
  const int sc = sec_of(i);
  const bool elec = sc != 1;
  const int jx = sc == 0 ? i : i - NS;                 // index into j / Phi_s / particles
  // ---- every LDS operand of this pass is loaded here, up front (one latency instead of one stall per use) ----
  const double ce = Y[O_CE + i], pe = Y[O_PE + i];
  const int el = sc == 2 ? 1 : 0;                      // electrode index of the per-electrode constants
  const double jv_l = Y[O_J + jx], ps_l = Y[O_PS + jx], yI = Y[O_I];
  // surface concentration: the last radial node (Fickian FDM) or c_avg + csj j (+ csq Q) (quadratic / polynomial, build_c_s_star!, aux...jl:193-248)
  double cs_l, cavg_l = 0.0, q_l = 0.0;
  if constexpr (M::SD == 0) cs_l = Y[O_CS + cs_surf(jx)];
  else {
    cavg_l = Y[O_CS + jx]; if (M::SD == 2) q_l = Y[O_Q + jx];
    cs_l = cavg_l + c.csj[el] * jv_l + c.csq[el] * q_l;
  }
  const double ypce = WANT_RES ? YP[O_CE + i] : 0.0;
  // SEI (anode nodes): side-reaction flux j_s, film thickness, film resistance R_film = R_SEI + film/k_n_aging (aux...jl:272-300)
  const int ks = (M::SEI && sc == 2) ? i - (NP + NS) : 0;
  double js = 0.0, film = 0.0, ypfilm = 0.0;
  if constexpr (M::SEI) { js = Y[O_JS + ks]; film = Y[O_FILM + ks]; if (WANT_RES) ypfilm = YP[O_FILM + ks]; if (sc != 2) { js = 0.0; film = 0.0; } }
  const double cRSEI = c.R_SEI, crkag = c.rkag, cMrho = c.Mrho, ci0F = c.i0F, cwexp = c.wexp, cUref = c.Uref;
  const double Rfilm = cRSEI + film * crkag;
  const double cT0 = c.T0, cKfac = c.Kfac, ctplus = c.tplus, cfRT = c.fRT, cI1C = c.I1C;
  const int ciso = c.iso_ref;
  // the constants of this lane's section and edge: one table row each, read with a per-lane address (NodeTab)
  const typename NodeTab<true>::Sec& qs = S.nt.sec[sc];
  const int ek = i == NP - 1 ? 3 : (i == NP + NS - 1 ? 4 : sc);
  const typename NodeTab<true>::Edge& qe = S.nt.edge[ek];
  const double h = qs.h, epsc = qs.eps, bfc = qs.bf;
  // divisions by per-cell constants are multiplications by reciprocals formed once in cell_setup (an fp64 division costs ~80 cycles of
  // dependent latency on gfx950; the results differ from the reference's by one rounding, far below the 1e-12 residual parity bar)
  const double rh = qs.rh, reps = qs.reps;
  const double a = qs.a, cmax = qs.cmax, kk = qs.kk, rsg = qs.rsg, rcm = qs.rcm, dcs = qs.Dc;
  const double beta = qe.beta, rdist = qe.rdist, Dh_edge = qe.Dh;
 double K, dK;
  if (M::CHEM == PLH_CHEM_LGM50) keff_lgm50(ce, K, dK); else keff(ce, cT0, K, dK);
  K *= bfc; dK *= bfc;
  double nu = 1.0, dnu = 0.0;                          // thermodynamic factor of this control volume (custom_functions.jl:177-203)
  if constexpr (M::TF == 1) {
    const double x = ce * 1e-3, sx = sqrt(x), tfac = 0.982 * (1.0 - 0.0052 * (cT0 - 293.0));
    nu = 0.601 - 0.24 * sx + tfac * x * sx;
    dnu = (-0.12 / sx + 1.5 * tfac * sx) * 1e-3;
  }
  double D, dD;
  if (M::CHEM == PLH_CHEM_LCO_LIC6) { D = dcs; dD = 0.0; }
  else if (M::CHEM == PLH_CHEM_LGM50) { deff_lgm50(ce, c.De, D, dD); D *= bfc; dD *= bfc; }
  else { deff_nmc(ce, cT0, D, dD); D *= bfc; dD *= bfc; }
  const double ce_n = shift_down1(ce), pe_n = shift_down1(pe), K_n = shift_down1(K), dK_n = shift_down1(dK), D_n = shift_down1(D);
  const double dD_n = (WANT_JAC && M::CHEM != PLH_CHEM_LCO_LIC6) ? shift_down1(dD) : 0.0;
  // edge i : geometry (numerical_tools.jl:106-215)
  const bool edge = i < NE - 1;
  const double rdenK = pl_rcp(beta * K_n + (1 - beta) * K), Kh = K * K_n * rdenK;
  double rdenD = 0.0, Dh;
  if (M::CHEM == PLH_CHEM_LCO_LIC6) Dh = Dh_edge;                                                        // constant D: the edge means are constants too (no division)
  else { rdenD = pl_rcp(beta * D_n + (1 - beta) * D); Dh = D * D_n * rdenD; }
  const double denC = beta * ce_n + (1 - beta) * ce;
  const double rcb = denC * pl_rcp(ce * ce_n);             // 1 / (harmonic mean of c_e at the edge)
  const double Tb = cT0;                                   // harmonic mean of equal temperatures
  const double dc = (ce_n - ce) * rdist;
  const double w = Kh * rdist;
  const double g = Kh * Tb * dc * rcb;
  // Phi_e-row edge flux.  With nu = 1 the concentration part is a flux too (E = w dPhi + Kfac g); with nu(c_e) the reference weights BOTH edges of row i
  // with the row's own nu_i (K[i] (g_i - g_{i-1}), residuals.jl:626-645), so g is differenced separately
  const double gE = edge ? g : 0.0;
  double E = edge ? (M::TF == 1 ? w * (pe - pe_n) : w * (pe - pe_n) + cKfac * g) : 0.0;
  double Nf = edge ? Dh * dc : 0.0;                        // c_e-row edge flux
  const double E_p = shift_up1(E), Nf_p = shift_up1(Nf);
  const double g_p = (M::TF == 1 || M::REFORD) ? shift_up1(gE) : 0.0;
  // PLH_PREC_F64_REFORDER: the matrix-form rows need the left neighbour's states and the left edge's coefficients
  [[maybe_unused]] double ce_pv = 0.0, pe_pv = 0.0, w_m = 0.0, dcoef = 0.0, dcoef_m = 0.0;
  if constexpr (M::REFORD) { ce_pv = shift_up1(ce); pe_pv = shift_up1(pe); w_m = shift_up1(w); dcoef = Dh * rdist; dcoef_m = shift_up1(dcoef); }
  const double Em = E_p, Nm = Nf_p, gm = g_p;             // (shift_up1 hands lane 0 a zero: the left edge of the first control volume carries no flux)
  // electrode quantities
  const double jv = elec ? jv_l : 0.0;
  const double ps = elec ? ps_l : 0.0;
  const double cs = elec ? cs_l : 1.0;
  double U = 0, dU = 0;
  if (M::CHEM == PLH_CHEM_LCO_LIC6) {
    if (sc == 0) ocv_lco(cs * rcm, cT0, ciso, U, dU);
    else if (sc == 2) ocv_lic6(cs * rcm, cT0, ciso, U, dU);
  } else if (M::CHEM == PLH_CHEM_LGM50) {
    if (sc == 0) ocv_nmc_lgm50(cs * rcm, U, dU);
    else if (sc == 2) ocv_lic6_lgm50(cs * rcm, U, dU);
  } else {
    if (sc == 0) ocv_nmc(cs * rcm, U, dU);
    else if (sc == 2) ocv_lic6_nmc(cs * rcm, U, dU);
  }
  const double jt = jv + js;                             // j_total (aux...jl:160-178) drives the c_e / Phi_e / Phi_s sources
  const double eta = (M::SEI && sc == 2) ? ps - pe - U - FAR * jv * Rfilm : ps - pe - U;
  const double arg = ce * cs * (cmax - cs);
  double sq, inv_sq;                                     // sqrt(arg) and its reciprocal (Jacobian partials) from one v_rsq_f64; both 0 where the reference's sqrt_ReLU clamps
  pl_sqrt_rsqrt(arg > 0.0 ? arg : 1.0, sq, inv_sq);
  if (!(arg > 0.0)) { sq = 0.0; inv_sq = 0.0; }
  const double xx = cfRT * eta;
  double sh = 0.0, chh = 0.0;
  // reaction rate j_calc and its partials with respect to (c_e, c_s*, eta): Butler-Volmer (rxn_BV) or Marcus-Hush-Chidsey (rxn_MHC, custom_functions.jl:241-298)
  double jc = 0.0, jc_ce = 0.0, jc_cs = 0.0, jc_eta = 0.0;
  if constexpr (M::RXN == 0) sinh_cosh(xx, sh, chh);
  else {
    const double lam = c.lam[el], sl = sqrt(lam), a1 = 1.0 + sl, k0 = c.mhc_k0[el], ce0 = c.ce0, rce0 = c.rce0;
    const double th_i = cs * rcm, ratio = ce * rce0 / th_i;                      // c_e_hat / theta_i
    const bool lin = ratio > 1e-4;                                               // log_ReLU(.; minval = 1e-4)
    const double ef = 2.0 * cfRT * eta + log(lin ? ratio : 1e-4);               // eta_f = F eta/(R T) + log(...)
    const double sr = sqrt(a1 + ef * ef), zz = (lam - sr) / (2.0 * sl);
    const double co = k0 * (1.0 - erf(zz));                                      // coeff_rd_ox
    const double dco = k0 * 1.1283791670955126 * exp(-zz * zz) * ef / (2.0 * sl * sr);   // d coeff / d eta_f  (2/sqrt(pi) e^{-z^2} * (eta_f / sr) / (2 sqrt(lambda)))
    const double em = exp(-ef), fo = 1.0 / (1.0 + em), fr = 1.0 - fo;            // 1/(1 + e^{-eta_f}), 1/(1 + e^{+eta_f})
    const double brk = fo * ce0 * cs - fr * ce * cmax;
    const double sarg = (1.0 - th_i) * rce0, sq2 = sqrt(sarg > 0.0 ? sarg : 0.0);   // sqrt_ReLU((1 - c_s*/c_max)/c_e0)
    const double dfo = fo * fr;                                                    // d fo / d eta_f = - d fr / d eta_f
    jc = co * brk * sq2;
    const double djc_ef = (dco * brk + co * dfo * (ce0 * cs + ce * cmax)) * sq2;
    const double def_ce = lin ? 1.0 / ce : 0.0, def_cs = lin ? -1.0 / cs : 0.0;
    jc_eta = djc_ef * 2.0 * cfRT;
    jc_ce = djc_ef * def_ce + co * (-fr * cmax) * sq2;
    jc_cs = djc_ef * def_cs + co * (fo * ce0) * sq2 + (sq2 > 0.0 ? co * brk * (-rcm * rce0) / (2.0 * sq2) : 0.0);
  }
  const double ps_p = shift_up1(ps), ps_n = shift_down1(ps);
  const bool first = (i == 0) || (i == NP + NS), last = (i == NP - 1) || (i == NE - 1);
  if (WANT_RES) {
    if (act) {
#ifdef PL_TEST_BREAK_NODE_PASS      /* tests/test_gpu_parity.py::test_selftest_catches_a_broken_plain_kernel: a deliberately wrong coefficient (1e-3 relative) in the c_e source */
      const double src = elec ? (1 - ctplus) * nu * a * jt * 1.001 : 0.0;
#else
      const double src = elec ? (1 - ctplus) * nu * a * jt : 0.0;
#endif
      if constexpr (!M::REFORD) {
      Fo[O_CE + i] = ((Nf - Nm) * rh + src) * reps - ypce;                 // residuals_c_e!, residuals.jl:6-106
      const double dE = M::TF == 1 ? (E - Em) + cKfac * nu * (gE - gm) : E - Em;
      Fo[O_PE + i] = (i < NE - 1) ? (dE - (elec ? h * FAR * a * jt : 0.0)) : pe;      // residuals_Φ_e!, residuals.jl:554-654
      } else {
        // PLH_PREC_F64_REFORDER: the reference's matrix form, A x - f with every product A_ik x_k rounded before the row sum, columns in ascending order
        // (A_tot * c_e: residuals.jl:30-104; A_tot * Phi_e .- f with f = -K (g_i - g_{i-1}) + dx F a j: residuals.jl:577-648)
        const double dL = i > 0 ? dcoef_m : 0.0, dU_ = edge ? dcoef : 0.0;          // D^_{i-1} / dist_{i-1}, D^_i / dist_i
        const double accC = (pl_rounded(dL * ce_pv) + pl_rounded(-(dL + dU_) * ce)) + pl_rounded(dU_ * ce_n);
        Fo[O_CE + i] = (accC * rh + src) * reps - ypce;
        const double wL = i > 0 ? w_m : 0.0, wU = edge ? w : 0.0;
        const double accP = (pl_rounded(-wL * pe_pv) + pl_rounded((wL + wU) * pe)) + pl_rounded(-wU * pe_n);
        const double fE = pl_rounded(-(M::TF == 1 ? cKfac * nu : cKfac) * (gE - gm)) + (elec ? h * FAR * a * jt : 0.0);
        Fo[O_PE + i] = (i < NE - 1) ? accP - fE : pe;
      }
      if (elec) {
        Fo[O_J + jx] = (M::RXN == 0 ? 2.0 * kk * sq * sh : jc) - jv;                    // residuals_j!, residuals.jl:491-517
        if constexpr (M::SD != 0) {                                                    // residuals_c_s_avg! (quadratic / polynomial), residuals_Q!
          Fo[O_CS + jx] = c.csr[el] * jv - YP[O_CS + jx];
          if (M::SD == 2) Fo[O_Q + jx] = -(el == 0 ? c.kap_p : c.kap_n) * q_l + c.qj[el] * jv - YP[O_Q + jx];
        }
        double lap = first ? (-ps + ps_n) : (last ? (ps_p - ps) : (ps_p - 2 * ps + ps_n));
        double f = h * h * a * FAR * jt;
        const double Idens = yI * cI1C;
        if (i == 0) f += -Idens * h;
        if (i == NE - 1) f += Idens * h;
        if constexpr (!M::REFORD) Fo[O_PS + jx] = lap - f * rsg;                       // residuals_Φ_s!, residuals.jl:656-703
        else Fo[O_PS + jx] = phi_s_row_reford(i, first, last, ps_p, ps, ps_n, f * rsg);
      }
    }
    if constexpr (M::SEI) {
      const double Idens = yI * cI1C;
      double calc = 0.0;                                                               // residuals_j_s!, residuals.jl:519-552
      if (Idens > 0.0) calc = -(ci0F * pl_pow(pl_div(Idens, cI1C), cwexp)) * pl_exp(-cfRT * (ps - pe - cUref - FAR * jt * Rfilm));
      if (act && sc == 2) {
        Fo[O_JS + ks] = js - calc;
        Fo[O_FILM + ks] = -js * cMrho - ypfilm;                                        // residuals_film!, residuals.jl:260-276
      }
      const double soh = wave_sum((act && sc == 2) ? S.sei.sohw[ks] * js : 0.0);       // residuals_SOH!, residuals.jl:278-297
      if (lane == 0) Fo[O_SOH] = soh - YP[O_SOH];
    }
    if (lane == 0) {                                                                   // scalar_residual!, scalar_residual.jl:167-172
      const double Vc = Y[O_PS] - Y[O_PS + NJ - 1];
      Fo[O_I] = mode == PLH_MODE_I ? yI - value : (mode == PLH_MODE_V ? Vc - value : (mode == PLH_MODE_P ? yI * cI1C * Vc - value   // method_P
                                                   : (mode == PLH_MODE_RES ? -value                                                  // method_res = 0: `value` is f - x (closure_input)
                                                   : Y[O_PS + NP] - Y[O_PE + NP + NS] - value)));                                   // method_η_p
    }
  }
  if (WANT_JAC) {
    if (lane == 0) { S.ctrlJ[0] = yI * cI1C; S.ctrlJ[1] = (Y[O_PS] - Y[O_PS + NJ - 1]) * cI1C; }   // scalar_jacobian! of method_P
    // edge derivatives
    const double dKh_a = dK * beta * K_n * K_n * (rdenK * rdenK), dKh_b = dK_n * (1 - beta) * K * K * (rdenK * rdenK);
    const double rdenC = pl_rcp(denC);
    const double dcb_a = beta * ce_n * ce_n * (rdenC * rdenC), dcb_b = (1 - beta) * ce * ce * (rdenC * rdenC);
    const double Tq = Tb * rdist;
    const double dg_a = Tq * (dKh_a * (ce_n - ce) * rcb - Kh * rcb - Kh * (ce_n - ce) * dcb_a * (rcb * rcb));
    const double dg_b = Tq * (dKh_b * (ce_n - ce) * rcb + Kh * rcb - Kh * (ce_n - ce) * dcb_b * (rcb * rcb));
    double Ea = edge ? (pe - pe_n) * dKh_a * rdist + (M::TF == 1 ? 0.0 : cKfac * dg_a) : 0.0;
    double Eb = edge ? (pe - pe_n) * dKh_b * rdist + (M::TF == 1 ? 0.0 : cKfac * dg_b) : 0.0;
    const double Ga = (M::TF == 1 && edge) ? dg_a : 0.0, Gb = (M::TF == 1 && edge) ? dg_b : 0.0;     // d g / d c_e (left, right) of this edge
    const double Ga_p = M::TF == 1 ? shift_up1(Ga) : 0.0, Gb_p = M::TF == 1 ? shift_up1(Gb) : 0.0;
    double we = edge ? w : 0.0;
    // N = Dh (c_{i+1} - c_i)/dist ; Dh = harmonic mean of D_i, D_{i+1} (dD/dc = 0 for D_eff_linear)
    const double dDh_a = dD * beta * D_n * D_n * (rdenD * rdenD), dDh_b = dD_n * (1 - beta) * D * D * (rdenD * rdenD);
    double Na = edge ? (dDh_a * (ce_n - ce) - Dh) * rdist : 0.0, Nb = edge ? (dDh_b * (ce_n - ce) + Dh) * rdist : 0.0;
    const double Ea_p = shift_up1(Ea), Eb_p = shift_up1(Eb), we_p = shift_up1(we), Na_p = shift_up1(Na), Nb_p = shift_up1(Nb);
    if (act) {
      const double rhe = rh * reps;
      S.ceL[i] = i > 0 ? -Na_p * rhe : 0.0;
      S.ceD[i] = (Na - (i > 0 ? Nb_p : 0.0)) * rhe + ((M::TF == 1 && elec) ? (1 - ctplus) * dnu * a * jt * reps : 0.0);
      S.ceU[i] = Nb * rhe;
      S.ceJ[i] = elec ? (1 - ctplus) * nu * a * reps : 0.0;
      if (i < NE - 1) {
        S.peL[i] = i > 0 ? -we_p : 0.0; S.peD[i] = (i > 0 ? we_p : 0.0) + we; S.peU[i] = -we;
        S.pcL[i] = i > 0 ? -Ea_p : 0.0; S.pcD[i] = Ea - (i > 0 ? Eb_p : 0.0); S.pcU[i] = Eb;
        if constexpr (M::TF == 1) {                        // + Kfac nu_i (g_i - g_{i-1}) with nu_i = nu(c_e,i)
          const double kn = cKfac * nu;
          S.pcL[i] += i > 0 ? -kn * Ga_p : 0.0; S.pcD[i] += kn * (Ga - (i > 0 ? Gb_p : 0.0)) + cKfac * dnu * (gE - gm); S.pcU[i] += kn * Gb;
        }
        S.peJ[i] = elec ? -h * FAR * a : 0.0;
      } else {
        S.peL[i] = 0; S.peD[i] = 1.0; S.peU[i] = 0; S.pcL[i] = 0; S.pcD[i] = 0; S.pcU[i] = 0; S.peJ[i] = 0;
      }
      if (elec) {
        const double ch = chh;
        if constexpr (M::RXN == 0) {
          S.gce[jx] = kk * sh * cs * (cmax - cs) * inv_sq;
          S.gcs[jx] = 2.0 * kk * (sh * ce * (cmax - 2 * cs) * 0.5 * inv_sq + sq * ch * cfRT * (-dU * rcm));
          S.gps[jx] = 2.0 * kk * sq * ch * cfRT;
        } else {                                           // eta = Phi_s - Phi_e - U(c_s*/c_max)
          S.gce[jx] = jc_ce;
          S.gcs[jx] = jc_cs + jc_eta * (-dU * rcm);
          S.gps[jx] = jc_eta;
        }
        S.gpe[jx] = -S.gps[jx];
        S.psJ[jx] = -h * h * a * FAR * rsg;
        if constexpr (M::SEI) {
          if (sc == 2) {
            S.sei.jjJ[ks] = -1.0 - S.gps[jx] * FAR * Rfilm;  // d(j row)/dj: eta_n carries -F j R_film
            S.sei.jjF[ks] = -S.gps[jx] * FAR * jv * crkag;   // d(j row)/d film
            const double Idens = yI * cI1C;
            if (Idens > 0.0) {
              const double Cr = pl_div(Idens, cI1C);
              const double Ex = pl_exp(-cfRT * (ps - pe - cUref - FAR * jt * Rfilm));
              double pwC, pwCm1; pl_pow_pair(Cr, cwexp, pwC, pwCm1);
              const double aAE = cfRT * ci0F * pwC * Ex;
              S.sei.jsPS[ks] = -aAE; S.sei.jsPE[ks] = aAE;
              S.sei.jsJ[ks] = aAE * FAR * Rfilm; S.sei.jsJS[ks] = 1.0 + aAE * FAR * Rfilm;
              S.sei.jsF[ks] = aAE * FAR * jt * crkag;
              S.sei.jsI[ks] = cwexp * ci0F * pwCm1 * Ex;
            } else {
              S.sei.jsPS[ks] = 0.0; S.sei.jsPE[ks] = 0.0; S.sei.jsJ[ks] = 0.0; S.sei.jsJS[ks] = 1.0; S.sei.jsF[ks] = 0.0; S.sei.jsI[ks] = 0.0;
            }
          }
        }
      }
    }
  }
}

// c_s rows (residuals_c_s_avg!, Fickian FDM, residuals.jl:128-180): lane -> (particle = pass*CS_G + lane/NR, row = lane%NR)   (default grid: 4 passes of 6 particles)
// ROW layout (PL_CSDPP): lane -> (particle = pass * 4 + lane / 16, row = lane % 16), the vector operand through rowb_fmac.  Same sums in the same order as the LDS form below.
template <class M>
PL_DEV void iso_cs_rows_rowb(CellLDS<M>& S, const double* Y, const double* YP, double* Fo) {
  PL_MODEL(M);
  const int lane = lane_id();
  const CellConst& c = S.cc;
  const int q = lane >> 4, r = lane & 15, rc = r < NR ? r : NR - 1;      // (lanes beyond N_r of a row read valid rows and do not store)
  double Mrow[NR];
  for (int k = 0; k < NR; k++) Mrow[k] = S.Mr[rc * NR + k];
  [[maybe_unused]] double MrowN[NR_EQ ? 1 : NR];
  if constexpr (!NR_EQ) for (int k = 0; k < NR; k++) MrowN[k] = S.Mr[S.mr_el(1) + rc * NR + k];
  int pp[CSD_PASS]; double acc[CSD_PASS], cv[CSD_PASS], jv[CSD_PASS], ypv[CSD_PASS];
#pragma unroll
  for (int pass = 0; pass < CSD_PASS; pass++) {
    const int p = pass * CSD_G + q; pp[pass] = p < NJ ? p : NJ - 1; acc[pass] = 0.0;
    const int rk = rc < nr_of(pp[pass]) ? rc : nr_of(pp[pass]) - 1;       // (a lane beyond the particle's own rows: a finite entry, times the operator's zero padding)
    cv[pass] = Y[O_CS + cs_off(pp[pass]) + rk];
    jv[pass] = Y[O_J + pp[pass]];
    ypv[pass] = YP[O_CS + cs_off(pp[pass]) + rk];
  }
  csd_settle(cv);
  static_for<0, NR>([&](auto kc) {
    constexpr int k = decltype(kc)::value;
#pragma unroll
    for (int pass = 0; pass < CSD_PASS; pass++) {
      if constexpr (NR_EQ) rowb_fmac<k>(acc[pass], cv[pass], Mrow[k]);
      else rowb_fmac<k>(acc[pass], cv[pass], pp[pass] < NP ? Mrow[k] : MrowN[k]);
    }
  });
  const double kap_p = c.kap_p, kap_n = c.kap_n, bj_p = c.bj_p, bj_n = c.bj_n;
#pragma unroll
  for (int pass = 0; pass < CSD_PASS; pass++) {
    const int p = pass * CSD_G + q;
    double rhs = (p < NP ? kap_p : kap_n) * acc[pass];
    if (r == nr_of(pp[pass]) - 1) rhs += (p < NP ? bj_p : bj_n) * jv[pass];
    if (p < NJ && r < nr_of(pp[pass])) Fo[O_CS + cs_off(p) + r] = rhs - ypv[pass];
  }
}

template <class M>
PL_DEV void iso_cs_rows(CellLDS<M>& S, const LaneRegs& R, const double* Y, const double* YP, double* Fo) {
  PL_MODEL(M);
  if constexpr (PL_CSDPP<M>) { iso_cs_rows_rowb(S, Y, YP, Fo); return; }
  if constexpr (M::W2) { if (wave_id() != 1) return; }       // two waves per cell: the particle rows belong to wave 1
  const int lane = lane_id();
  const CellConst& c = S.cc;
  const int r = lane % NR, g = lane < CS_LANES ? lane / NR : CS_G - 1;
  double Mrow[NR];
  for (int k = 0; k < NR; k++) Mrow[k] = S.Mr[r * NR + k];
  [[maybe_unused]] double MrowN[NR_EQ ? 1 : NR];           // (N_r_p != N_r_n: the anode's operator row; padded rows and columns are zero)
  if constexpr (!NR_EQ) for (int k = 0; k < NR; k++) MrowN[k] = S.Mr[S.mr_el(1) + r * NR + k];
  // the passes are independent: CS_PASS accumulation chains side by side (particle index clamped so that every lane computes)
  int pp[CS_PASS]; double acc[CS_PASS];
#pragma unroll
  for (int pass = 0; pass < CS_PASS; pass++) { const int p = pass * CS_G + g; pp[pass] = p < NJ ? p : NJ - 1; acc[pass] = 0.0; }
  // issue every LDS load first (one latency), then the arithmetic
  double v[CS_PASS][NR], jv[CS_PASS], ypv[CS_PASS];
#pragma unroll
  for (int pass = 0; pass < CS_PASS; pass++) {
#pragma unroll
    for (int k = 0; k < NR; k++) v[pass][k] = Y[O_CS + cs_off(pp[pass]) + k];     // (k beyond the particle's own rows: a neighbour's entries, times the operator's zero padding)
    jv[pass] = Y[O_J + pp[pass]];
    ypv[pass] = YP[O_CS + cs_off(pp[pass]) + r];
  }
  const double kap_p = c.kap_p, kap_n = c.kap_n, bj_p = c.bj_p, bj_n = c.bj_n;
#pragma unroll
  for (int k = 0; k < NR; k++) {
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      if constexpr (NR_EQ) acc[pass] += Mrow[k] * v[pass][k];
      else acc[pass] += (pp[pass] < NP ? Mrow[k] : MrowN[k]) * v[pass][k];
    }
  }
#pragma unroll
  for (int pass = 0; pass < CS_PASS; pass++) {
    const int p = pass * CS_G + g;
    double rhs = (p < NP ? kap_p : kap_n) * acc[pass];
    if (r == nr_of(p) - 1) rhs += (p < NP ? bj_p : bj_n) * jv[pass];
    if (lane < CS_LANES && p < NJ && r < nr_of(p)) Fo[O_CS + cs_off(p) + r] = rhs - ypv[pass];
  }
}

// ---- generic entry points: dispatch to the isothermal or the thermal implementation ----
template <bool WANT_RES, bool WANT_JAC, class M>
PL_DEV void cell_node_pass(CellLDS<M>& S, const double* Y, const double* YP, double* Fo, int mode, double value) {
  if constexpr (M::THERMAL) thermal_node_pass<WANT_RES, WANT_JAC>(S, Y, YP, Fo, mode, value);
  else iso_node_pass<WANT_RES, WANT_JAC>(S, Y, YP, Fo, mode, value);
}
template <bool WANT_JAC = false, class M>
PL_DEV void cell_cs_rows(CellLDS<M>& S, const LaneRegs& R, const double* Y, const double* YP, double* Fo) {
  if constexpr (M::THERMAL) thermal_cs_rows<WANT_JAC>(S, S.tb, Y, YP, Fo);
  else if constexpr (M::SD == 0) iso_cs_rows(S, R, Y, YP, Fo);        // (quadratic / polynomial: the particle rows are node-local, written by the node pass)
}
// full residual F(Y, YP) -> Fo (all three are LDS vectors)
template <class M>
PL_DEV void cell_residual(CellLDS<M>& S, const LaneRegs& R, const double* Y, const double* YP, double* Fo, int mode, double value) {
  PL_TICE(2);
  cell_node_pass<true, false>(S, Y, YP, Fo, mode, value);
  if constexpr (M::THERMAL) PL_SYNC();                      // the particle rows read the per-node D_s(T) written by the node pass
  PL_TOCE(S, 2, 5);
  cell_cs_rows<false>(S, R, Y, YP, Fo);
  PL_SYNC();
  PL_TOCE(S, 2, 6);                                                // (two waves per cell: every wave has written its own rows of Fo; whoever reads across waves synchronises first)
}
// residual + Jacobian partials in one pass (the Newton-matrix refresh of the corrector)
template <class M>
PL_DEV void cell_res_jac(CellLDS<M>& S, const LaneRegs& R, const double* Y, const double* YP, double* Fo, int mode, double value) {
  { PL_TICD(); cell_node_pass<true, true>(S, Y, YP, Fo, mode, value); PL_TOCD(S, 0); }
  if constexpr (M::THERMAL) PL_SYNC();
  { PL_TICD(); cell_cs_rows<true>(S, R, Y, YP, Fo); PL_TOCD(S, 1); }
  PL_SYNC();
}

// ------------------------------------------------------------------------------------------------------------------
// linear algebra: J = dF/dY + cj dF/dYP  ->  particle resolvent + j elimination + block-Thomas (3x3) + border
// ------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void inv3(const double* A, double* B) {
  const double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  const double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  const double id = pl_rcp(det);
  B[0] = c00 * id; B[1] = (A[2] * A[7] - A[1] * A[8]) * id; B[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  B[3] = c01 * id; B[4] = (A[0] * A[8] - A[2] * A[6]) * id; B[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  B[6] = c02 * id; B[7] = (A[1] * A[6] - A[0] * A[7]) * id; B[8] = (A[0] * A[4] - A[1] * A[3]) * id;
}

// node block D_i (3x3 over [c_e, Phi_e, Phi_s]) after eliminating c_s and j ; alg_only: c_e is not an unknown
template <class M>
__device__ __forceinline__ void node_block(const CellLDS<M>& S, int i, double cj, bool alg_only, double* D) {
  PL_MODEL(M);
  const int sc = sec_of(i);
  const bool elec = sc != 1;
  const int jx = sc == 0 ? i : i - NS;
  if (alg_only) { D[0] = 1.0; D[1] = 0.0; D[2] = 0.0; D[3] = 0.0; }
  else { D[0] = S.ceD[i] - cj; D[1] = 0.0; D[2] = 0.0; D[3] = S.pcD[i]; }
  D[4] = S.peD[i]; D[5] = 0.0; D[6] = 0.0; D[7] = 0.0; D[8] = 1.0;
  if (elec) {
    const bool first = (i == 0) || (i == NP + NS), last = (i == NP - 1) || (i == NE - 1);
    D[8] = (first || last) ? -1.0 : -2.0;
    const double gc = S.nphi[0][i], ge = S.nphi[1][i], gs = S.nphi[2][i];      // phi (already zero in column c_e when alg_only)
    const double fs = alg_only ? 0.0 : S.ceJ[i], fp = S.peJ[i], fq = S.psJ[jx];
    D[0] -= fs * gc; D[1] -= fs * ge; D[2] -= fs * gs;
    D[3] -= fp * gc; D[4] -= fp * ge; D[5] -= fp * gs;
    D[6] -= fq * gc; D[7] -= fq * ge; D[8] -= fq * gs;
  }
}

// r - a.b and r + a.b for short dot products, written so that every term is ONE fused multiply-add onto the running value (r - (a0 b0 + a1 b1 + a2 b2) is a multiply, two
// fmas and a subtraction: four dependent instructions where three do) -- the inner statement of every sweep stage
#define PL_NMS3(r, a0, b0, a1, b1, a2, b2) ((((r) - (a0) * (b0)) - (a1) * (b1)) - (a2) * (b2))
#define PL_PMS3(r, a0, b0, a1, b1, a2, b2) ((((r) + (a0) * (b0)) + (a1) * (b1)) + (a2) * (b2))
#define PL_NMS4(r, a0, b0, a1, b1, a2, b2, a3, b3) (((((r) - (a0) * (b0)) - (a1) * (b1)) - (a2) * (b2)) - (a3) * (b3))

// ---- twisted (two-ended) block-Thomas in a mirrored lane layout ----
// Nodes 0..14 are eliminated forwards and live in lanes 0..14; nodes 29..15 are eliminated backwards and live in lanes 32..46 (node 29 in
// lane 32), so BOTH halves run the same "take the value of the lane below" recurrence (one DPP wave_shr:1 per value and stage) and the
// chain is 14 stages instead of 29.  Node 15 closes the system: it takes node 14's value through a v_readlane broadcast.  The
// back-substitution runs the other way with wave_shl:1; lane 15 is a ghost that re-publishes node 15's solution for node 14.
// Every lane re-evaluates its recurrence at every stage (idempotent once its predecessor is final), so there are no per-stage selects.
constexpr int TW_MID = NE / 2, TW_BASE = 32;
constexpr int TW_FWD = NE - TW_MID;       // forward stages + 1: the backward-eliminated half holds NE - TW_MID nodes (the closing one included; one more than the other half when NE is odd)
__device__ __forceinline__ int tw_node(int lane) { return lane < TW_MID ? lane : ((lane >= TW_BASE && lane < TW_BASE + (NE - TW_MID)) ? NE - 1 - (lane - TW_BASE) : -1); }
__host__ __device__ constexpr int tw_lane(int node) { return node < TW_MID ? node : TW_BASE + (NE - 1 - node); }
__device__ __forceinline__ double l22_of(int n) { return (n > 0 && sec_of(n) != 1 && sec_of(n - 1) == sec_of(n)) ? 1.0 : 0.0; }   // Phi_s row n x Phi_s[n-1]
__device__ __forceinline__ double u22_of(int n) { return (n < NE - 1 && sec_of(n) != 1 && sec_of(n + 1) == sec_of(n)) ? 1.0 : 0.0; }

// forward / backward substitution with the factors in S.LD / S.Dinv / S.LDmid: the lane of node n (tw_lane) holds node n's right-hand side
// on entry and node n's solution on exit.
template <class M>
__device__ __forceinline__ void thomas_sweeps(const CellLDS<M>& S, bool alg_only, double& r0, double& r1, double& r2) {
  PL_MODEL(M);
  const int lane = lane_id_pred();
  const int nd = tw_node(lane);
  const bool act = nd >= 0, top = lane < TW_MID;
  const int i = act ? nd : 0;
  // every load is unconditional (i is a valid node in every lane) and masked afterwards: a guarded load `act ? S.x[i] : 0` compiles to one
  // exec-masked branch per element, which serialises the loads of this prologue
  // (r06: no masks on the factors.  An idle lane reads node 0's: the head of the forward chain, whose L D'^-1 is ZERO -- with its right-hand side zeroed below its
  //  recurrence stays at zero whatever its neighbours hold --, and the closing block multiplies a broadcast that is masked instead of its nine entries: 48 selects less)
  double C[9], Di[9], G[9], Lm[9];
  for (int k = 0; k < 9; k++) { C[k] = S.LD[k][i]; Di[k] = S.Dinv[k][i]; Lm[k] = S.LDmid[k]; }
  {   // back-substitution block: top x_n = z_n - Dinv U_n x_{n+1}; bottom x_n = z_n - Dinv L_n x_{n-1}; node TW_MID is closed (G = 0)
    const bool z = !act || nd == TW_MID;
    const double ceu = S.ceU[i], cel = S.ceL[i], pcu = S.pcU[i], pcl = S.pcL[i], peu = S.peU[i], pel = S.peL[i];
    const double v00 = (alg_only || z) ? 0.0 : (top ? ceu : cel), v10 = (alg_only || z) ? 0.0 : (top ? pcu : pcl);
    const double v11 = z ? 0.0 : (top ? peu : pel);
    const double v22 = z ? 0.0 : (top ? u22_of(i) : l22_of(i));
    for (int rr = 0; rr < 3; rr++) {
      G[rr * 3 + 0] = Di[rr * 3 + 0] * v00 + Di[rr * 3 + 1] * v10;
      G[rr * 3 + 1] = Di[rr * 3 + 1] * v11;
      G[rr * 3 + 2] = Di[rr * 3 + 2] * v22;
    }
  }
  if (!act) { r0 = 0.0; r1 = 0.0; r2 = 0.0; }
  // Both substitutions are first-order linear recurrences along a chain of lanes, y_n = r_n - C_n y_{n-1}: a DEPENDENT chain of DPP shift + three FMAs per stage, and at one wave
  // per SIMD its latency, not its instruction count, is what a solve waits for (121 cycles per stage measured against 60 of issue).  One level of recursive doubling halves the chain:
  //     y_n = (r_n - C_n r_{n-1}) + (C_n C_{n-1}) y_{n-2}
  // -- the odd and the even nodes of a half then advance together, two lanes apart (row_shr:2 / row_shl:2: each half of the twisted layout sits inside one 16-lane DPP row), in
  // ceil(n/2) stages.  The products C_n C_{n-1} and the shifted right-hand sides cost one parallel pre-pass (27 + 9 FMAs); they depend on the factors only, but there is no LDS left to
  // keep them in.  (Grids whose halves do not fit a DPP row keep the one-lane recurrence.)
#ifdef PL_EXP_NO_STRIDE2      /* (experiment build: the one-lane recurrence, for same-box A/B runs) */
  constexpr bool STRIDE2 = false;
#else
  constexpr bool STRIDE2 = TW_FWD <= 15 && TW_MID <= 15;
#endif
  double y0, y1, y2;
  if constexpr (STRIDE2) {
    double P[9];
    {
      double Cs[9];
      for (int k = 0; k < 9; k++) Cs[k] = shift_up1(C[k]);                     // C of the previous node of the chain
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) P[a * 3 + b] = C[a * 3] * Cs[b] + C[a * 3 + 1] * Cs[3 + b] + C[a * 3 + 2] * Cs[6 + b];
      const double s0 = shift_up1(r0), s1 = shift_up1(r1), s2 = shift_up1(r2);
      r0 = PL_NMS3(r0, C[0], s0, C[1], s1, C[2], s2); r1 = PL_NMS3(r1, C[3], s0, C[4], s1, C[5], s2); r2 = PL_NMS3(r2, C[6], s0, C[7], s1, C[8], s2);
    }
    y0 = r0; y1 = r1; y2 = r2;
    constexpr int NST2 = (TW_FWD - 1 + 1) / 2;                               // the last node of the longer half is TW_FWD - 1 steps from its head
    _Pragma("unroll") for (int it = 0; it < NST2; it++) {
      const double p0 = row_up2(y0), p1 = row_up2(y1), p2 = row_up2(y2);
      y0 = PL_PMS3(r0, P[0], p0, P[1], p1, P[2], p2); y1 = PL_PMS3(r1, P[3], p0, P[4], p1, P[5], p2); y2 = PL_PMS3(r2, P[6], p0, P[7], p1, P[8], p2);
    }
  } else {
  y0 = r0; y1 = r1; y2 = r2;
  // the stage loops are fully unrolled for the models without aging (no loop bookkeeping between the DPP shifts: +2.7 % on C4, +0.9 % on C2); with SEI, whose integrate
  // kernel is already out of registers, that costs 1.8 %, so it keeps the loop unrolled by two
#define PL_FWD_STAGE { const double p0 = shift_up1(y0), p1 = shift_up1(y1), p2 = shift_up1(y2); \
    y0 = PL_NMS3(r0, C[0], p0, C[1], p1, C[2], p2); y1 = PL_NMS3(r1, C[3], p0, C[4], p1, C[5], p2); y2 = PL_NMS3(r2, C[6], p0, C[7], p1, C[8], p2); }
  if constexpr (M::SEI) { _Pragma("unroll 2") for (int it = 1; it < TW_FWD; it++) PL_FWD_STAGE }
  else { _Pragma("unroll") for (int it = 1; it < TW_FWD; it++) PL_FWD_STAGE }
#undef PL_FWD_STAGE
  }
  {   // closing node: y_mid -= (L_mid Dinv_{mid-1}) y_{mid-1}
    const bool mid = nd == TW_MID;
    const double b0 = lane_bcast(y0, TW_MID - 1), b1 = lane_bcast(y1, TW_MID - 1), b2 = lane_bcast(y2, TW_MID - 1);
    const double m0 = mid ? b0 : 0.0, m1 = mid ? b1 : 0.0, m2 = mid ? b2 : 0.0;
    y0 = PL_NMS3(y0, Lm[0], m0, Lm[1], m1, Lm[2], m2); y1 = PL_NMS3(y1, Lm[3], m0, Lm[4], m1, Lm[5], m2); y2 = PL_NMS3(y2, Lm[6], m0, Lm[7], m1, Lm[8], m2);
  }
  double z0 = Di[0] * y0 + Di[1] * y1 + Di[2] * y2, z1 = Di[3] * y0 + Di[4] * y1 + Di[5] * y2, z2 = Di[6] * y0 + Di[7] * y1 + Di[8] * y2;
  {   // ghost of the closing node in lane TW_MID (its G is zero: it just holds x_mid for lane TW_MID-1)
    const double g0 = lane_bcast(z0, tw_lane(TW_MID)), g1 = lane_bcast(z1, tw_lane(TW_MID)), g2 = lane_bcast(z2, tw_lane(TW_MID));
    if (lane == TW_MID) { z0 = g0; z1 = g1; z2 = g2; }
  }
  double x0, x1, x2;
  if constexpr (STRIDE2) {                                                    // x_n = (z_n - G_n z_{n+1}) + (G_n G_{n+1}) x_{n+2}  ("n+1" = the next lane of the chain)
    double Q[9];
    {
      double Gs[9];
      for (int k = 0; k < 9; k++) Gs[k] = shift_down1(G[k]);
      for (int a = 0; a < 3; a++) for (int b = 0; b < 3; b++) Q[a * 3 + b] = G[a * 3] * Gs[b] + G[a * 3 + 1] * Gs[3 + b] + G[a * 3 + 2] * Gs[6 + b];
      const double s0 = shift_down1(z0), s1 = shift_down1(z1), s2 = shift_down1(z2);
      z0 = PL_NMS3(z0, G[0], s0, G[1], s1, G[2], s2); z1 = PL_NMS3(z1, G[3], s0, G[4], s1, G[5], s2); z2 = PL_NMS3(z2, G[6], s0, G[7], s1, G[8], s2);
    }
    x0 = z0; x1 = z1; x2 = z2;
    constexpr int NST2 = (TW_MID + 1) / 2;                                    // lane 0 is TW_MID steps from the ghost of the closing node
    _Pragma("unroll") for (int it = 0; it < NST2; it++) {
      const double q0 = row_down2(x0), q1 = row_down2(x1), q2 = row_down2(x2);
      x0 = PL_PMS3(z0, Q[0], q0, Q[1], q1, Q[2], q2); x1 = PL_PMS3(z1, Q[3], q0, Q[4], q1, Q[5], q2); x2 = PL_PMS3(z2, Q[6], q0, Q[7], q1, Q[8], q2);
    }
  } else {
  x0 = z0; x1 = z1; x2 = z2;
#define PL_BWD_STAGE { const double q0 = shift_down1(x0), q1 = shift_down1(x1), q2 = shift_down1(x2); \
    x0 = PL_NMS3(z0, G[0], q0, G[1], q1, G[2], q2); x1 = PL_NMS3(z1, G[3], q0, G[4], q1, G[5], q2); x2 = PL_NMS3(z2, G[6], q0, G[7], q1, G[8], q2); }
  if constexpr (M::SEI) { _Pragma("unroll 2") for (int it = 0; it < TW_MID; it++) PL_BWD_STAGE }
  else { _Pragma("unroll") for (int it = 0; it < TW_MID; it++) PL_BWD_STAGE }
#undef PL_BWD_STAGE
  }
  r0 = x0; r1 = x1; r2 = x2;
}

// factor the Newton matrix at the Jacobian partials currently in S (cell_node_pass<.,true> must have run).
// mode selects the control row; alg_only = the 71x71 algebraic block of the consistent-initialisation Newton.
template <class M>
PL_DEV void iso_factor(CellLDS<M>& S, LaneRegs& R, const Tables* __restrict__ tb, double cj, int mode, bool alg_only) {
  PL_MODEL(M);
  const int lane = lane_id();
  const CellConst& c = S.cc;
  PL_TICD();
  // 1. particle resolvent rows: (kappa M - cj I)^-1 = V diag(1/(kappa lam - cj)) W
  if constexpr (M::SD != 0) {
    if (lane < 2) { S.rcjf[lane][0] = alg_only ? 0.0 : -1.0 / cj; S.rcjf[lane][1] = alg_only ? 0.0 : 1.0 / (-(lane == 0 ? c.kap_p : c.kap_n) - cj); }   // -1/cj, 1/(-kappa - cj)
  } else
  if (!alg_only && (!M::W2 || wave_id() == 1)) {          // (two waves per cell: the resolvents are wave 1's, next to wave 0's Jacobian node pass)
    const int r = lane % NR;
    // the 2 N_r reciprocals 1/(kappa lam_m - cj) are formed by 2 N_r lanes in parallel and passed through S.w9 (free outside the solves)
    if constexpr (M::PHI_GLOBAL_ALL) { if (lane < 2 * NR) S.w9[lane] = pl_rcp((lane < NR ? c.kap_p : c.kap_n) * tb->LAMp(lane < NR ? 0 : 1)[r] - cj); }
    else if constexpr (NR_EQ) { if (lane < 2 * NR) S.w9[lane] = pl_rcp((lane < NR ? c.kap_p : c.kap_n) * S.Mr[S.OFF_LAMR + r] - cj); }
    else { if (lane < 2 * NR) S.w9[lane] = pl_rcp((lane < NR ? c.kap_p : c.kap_n) * S.Mr[(lane < NR ? 0 : S.MR_BLK) + S.OFF_LAMR + r] - cj); }      // (padded modes: lam = 0, V = W = 0)
    PL_SYNC();
    // entry (row, k) of electrode el = sum_m V[row][m] w_el[m] W[m][k], m ascending; lanes 0..31 build the cathode's resolvent, 32..63 the anode's, RS_KG lanes per row with
    // RS_KW columns each (N_r = 10: 3 lanes x 4 columns)
    constexpr int RS_KG = 32 / NR, RS_KW = (NR + RS_KG - 1) / RS_KG;
    const int el = lane >> 5, q = lane & 31, row = q / RS_KG < NR ? q / RS_KG : NR - 1, k0 = (q % RS_KG) * RS_KW;
    const bool act = q < NR * RS_KG;
    double acc[RS_KW];
    for (int kk = 0; kk < RS_KW; kk++) acc[kk] = 0.0;
    _Pragma("unroll 2") for (int m = 0; m < NR; m++) {            // (fully unrolled, the 60 operands of the sums are all live at once: 132 B/lane of scratch in the integrate kernel)
      if constexpr (M::PHI_GLOBAL_ALL) {                        // (the same sums in the same order; operands from the model tables: L2 hits, once per Jacobian refresh)
        const double* __restrict__ Vt = tb->Vp(el); const double* __restrict__ Wt = tb->Wp(el);
        const double f = Vt[row * NR + m] * S.w9[el * NR + m];
        for (int kk = 0; kk < RS_KW; kk++) acc[kk] += f * Wt[m * NR + (k0 + kk < NR ? k0 + kk : NR - 1)];
      } else if constexpr (NR_EQ) {
        const double f = S.Mr[S.OFF_VR + row * NR + m] * S.w9[el * NR + m];
        for (int kk = 0; kk < RS_KW; kk++) acc[kk] += f * S.Mr[S.OFF_WR + m * NR + (k0 + kk < NR ? k0 + kk : NR - 1)];
      } else {
        const double f = S.Mr[el * S.MR_BLK + S.OFF_VR + row * NR + m] * S.w9[el * NR + m];
        for (int kk = 0; kk < RS_KW; kk++) acc[kk] += f * S.Mr[el * S.MR_BLK + S.OFF_WR + m * NR + (k0 + kk < NR ? k0 + kk : NR - 1)];
      }
    }
    if (act) for (int kk = 0; kk < RS_KW; kk++) if (k0 + kk < NR) {
      S.Ainv[el][row * NR + k0 + kk] = PL_F32(acc[kk]);
      if constexpr (NR_EQ) { if (row == NR - 1 && k0 + kk == NR - 1) S.sig[el] = acc[kk]; }
      else { const int nl = (el ? NRN : NRP) - 1; if (row == nl && k0 + kk == nl) S.sig[el] = acc[kk]; }
    }
  }
  PL_XSYNC();
  if constexpr (M::W2) { if (wave_id() != 0) return; }    // two waves per cell: the node-local elimination, the block factorisation and the border are wave 0's
  PL_TOCD(S, 2);
  // 2. node-local elimination.  Without SEI the local unknown is j (pivot d = -1 - gcs sigma bj after the particle Schur complement);
  //    with SEI the anode nodes eliminate u = (j, j_s, film) through the inverse W of their 3x3 local block.  Both cases reduce to
  //    D[r][c] -= t_r phi_c with t = (ceJ, peJ, psJ) (j and j_s enter the node rows only through j_total) and phi = omega . A_ux,
  //    omega = W[j,:] + W[j_s,:]; the j_s rows also depend on I, which adds -t_r omega_{j_s} d(j_s row)/dI to the column of I.
  if (lane < NE) {
    const int i = lane, sc = sec_of(i);
    double ph0 = 0.0, ph1 = 0.0, ph2 = 0.0, cI0 = 0.0, cI1 = 0.0, cI2 = 0.0;
    if (sc != 1) {
      const int jx = sc == 0 ? i : i - NS;
      const double bj = sc == 0 ? c.bj_p : c.bj_n;
      double schur;
      if constexpr (M::SD == 0) schur = alg_only ? 0.0 : S.gcs[jx] * S.sig[sc == 0 ? 0 : 1] * bj;
      else {   // d c_s*/d j with c_avg (and Q) eliminated:  -cj dc + csr dj = b_c ,  (-kappa - cj) dQ + qj dj = b_Q ,  c_s* = c + csj j + csq Q
        const int el = sc == 0 ? 0 : 1;
        const double resp = alg_only ? c.csj[el] : c.csj[el] + c.csr[el] * (1.0 / cj) - c.csq[el] * c.qj[el] * (1.0 / (-(el == 0 ? c.kap_p : c.kap_n) - cj));
        S.resp[jx] = resp; schur = -S.gcs[jx] * resp;
      }
      if (i == 0) cI2 = c.JI0;
      if (i == NE - 1) cI2 = c.JI29;
      bool local3 = false;
      if constexpr (M::SEI) {
        if (sc == 2) {
          const int k = i - (NP + NS);
          double A[9], Wm[9];
          A[0] = S.sei.jjJ[k] - schur; A[1] = 0.0; A[2] = alg_only ? 0.0 : S.sei.jjF[k];
          A[3] = S.sei.jsJ[k]; A[4] = S.sei.jsJS[k]; A[5] = alg_only ? 0.0 : S.sei.jsF[k];
          A[6] = 0.0; A[7] = alg_only ? 0.0 : -c.Mrho; A[8] = alg_only ? 1.0 : -cj;
          inv3(A, Wm);
          for (int q = 0; q < 9; q++) S.sei.Wl[q][k] = Wm[q];
          const double w0 = Wm[0] + Wm[3], w1 = Wm[1] + Wm[4];
          ph0 = alg_only ? 0.0 : w0 * S.gce[jx];
          ph1 = w0 * S.gpe[jx] + w1 * S.sei.jsPE[k];
          ph2 = w0 * S.gps[jx] + w1 * S.sei.jsPS[k];
          const double qI = w1 * S.sei.jsI[k];
          cI0 -= (alg_only ? 0.0 : S.ceJ[i]) * qI; cI1 -= S.peJ[i] * qI; cI2 -= S.psJ[jx] * qI;
          local3 = true;
        }
      }
      if (!local3) {
        const double d = -1.0 - schur;
        const double rd = pl_rcp(d);
        S.dj[jx] = rd;                                     // the reciprocal pivot: the solves multiply
        ph0 = alg_only ? 0.0 : S.gce[jx] * rd; ph1 = S.gpe[jx] * rd; ph2 = S.gps[jx] * rd;
      }
    }
    S.nphi[0][i] = ph0; S.nphi[1][i] = ph1; S.nphi[2][i] = ph2;
    S.colI[0][i] = cI0; S.colI[1][i] = cI1; S.colI[2][i] = cI2;
  }
  if constexpr (M::SEI) { if (lane == 0) S.sei.cjf = cj; }
  PL_SYNC();
  PL_TOCD(S, 3);
  // 3. twisted block-Thomas factorisation (lane layout of thomas_sweeps): top nodes D'_n = D_n - L_n D'^-1_{n-1} U_{n-1}, bottom nodes
  //    D'_n = D_n - U_n D'^-1_{n+1} L_{n+1}; with the mirrored layout both read "a P b" with P = the factor of the lane below.
  {
    const int nd = tw_node(lane);
    const bool act = nd >= 0, top = lane < TW_MID;
    const int i = act ? nd : 0;
    double D[9], Dinv[9], LDm[9], Dn[9];
    node_block(S, i, cj, alg_only, D);
    // a = left block (L_n for top, U_n for bottom), b = right block (U_{n-1} for top, L_{n+1} for bottom); zero at the two chain heads
    const bool head = !act || nd == 0 || nd == NE - 1;
    const int nb = top ? (i > 0 ? i - 1 : 0) : (i < NE - 1 ? i + 1 : NE - 1);
    // unconditional loads, masked afterwards (see thomas_sweeps)
    const double ceLi = S.ceL[i], ceUi = S.ceU[i], pcLi = S.pcL[i], pcUi = S.pcU[i], peLi = S.peL[i], peUi = S.peU[i];
    const double ceLn = S.ceL[nb], ceUn = S.ceU[nb], pcLn = S.pcL[nb], pcUn = S.pcU[nb], peLn = S.peL[nb], peUn = S.peU[nb];
    const double a00 = (alg_only || head) ? 0.0 : (top ? ceLi : ceUi), a10 = (alg_only || head) ? 0.0 : (top ? pcLi : pcUi);
    const double a11 = head ? 0.0 : (top ? peLi : peUi), a22 = head ? 0.0 : (top ? l22_of(i) : u22_of(i));
    const double b00 = (alg_only || head) ? 0.0 : (top ? ceUn : ceLn), b10 = (alg_only || head) ? 0.0 : (top ? pcUn : pcLn);
    const double b11 = head ? 0.0 : (top ? peUn : peLn), b22 = a22;
    for (int k = 0; k < 9; k++) { LDm[k] = 0.0; Dn[k] = D[k]; }
    inv3(D, Dinv);
#pragma unroll 1
    for (int it = 1; it < TW_FWD; it++) {
      double P[9];
      for (int k = 0; k < 9; k++) P[k] = shift_up1(Dinv[k]);
      for (int k = 0; k < 3; k++) {
        LDm[k] = a00 * P[k];
        LDm[3 + k] = a10 * P[k] + a11 * P[3 + k];
        LDm[6 + k] = a22 * P[6 + k];
      }
      for (int rr = 0; rr < 3; rr++) {
        Dn[rr * 3 + 0] = (D[rr * 3 + 0] - LDm[rr * 3 + 0] * b00) - LDm[rr * 3 + 1] * b10;
        Dn[rr * 3 + 1] = D[rr * 3 + 1] - LDm[rr * 3 + 1] * b11;
        Dn[rr * 3 + 2] = D[rr * 3 + 2] - LDm[rr * 3 + 2] * b22;
      }
      inv3(Dn, Dinv);
    }
    {   // closing node TW_MID: D'' = D'_mid - L_mid D'^-1_{mid-1} U_{mid-1}
      double P[9], L2[9], Dm[9], Dmi[9];
      for (int k = 0; k < 9; k++) P[k] = lane_bcast(Dinv[k], TW_MID - 1);
      const double c00 = alg_only ? 0.0 : S.ceL[TW_MID], c10 = alg_only ? 0.0 : S.pcL[TW_MID], c11 = S.peL[TW_MID], c22 = l22_of(TW_MID);
      const double e00 = alg_only ? 0.0 : S.ceU[TW_MID - 1], e10 = alg_only ? 0.0 : S.pcU[TW_MID - 1], e11 = S.peU[TW_MID - 1], e22 = c22;
      for (int k = 0; k < 3; k++) { L2[k] = c00 * P[k]; L2[3 + k] = c10 * P[k] + c11 * P[3 + k]; L2[6 + k] = c22 * P[6 + k]; }
      for (int rr = 0; rr < 3; rr++) {
        Dm[rr * 3 + 0] = (Dn[rr * 3 + 0] - L2[rr * 3 + 0] * e00) - L2[rr * 3 + 1] * e10;
        Dm[rr * 3 + 1] = Dn[rr * 3 + 1] - L2[rr * 3 + 1] * e11;
        Dm[rr * 3 + 2] = Dn[rr * 3 + 2] - L2[rr * 3 + 2] * e22;
      }
      inv3(Dm, Dmi);
      if (nd == TW_MID) for (int k = 0; k < 9; k++) { Dinv[k] = Dmi[k]; S.LDmid[k] = PL_F32(L2[k]); }
    }
    if (act) for (int k = 0; k < 9; k++) { S.Dinv[k][i] = PL_F32(Dinv[k]); S.LD[k][i] = PL_F32(LDm[k]); }
  }
  PL_SYNC();
  PL_TOCD(S, 4);
  // 4. border vector for modes whose control row is not "I = value":  x2 = T^-1 (column of I)
  if (mode != PLH_MODE_I) {
    const int nd = tw_node(lane);
    const int il = nd >= 0 ? nd : 0;
    double r0 = nd >= 0 ? S.colI[0][il] : 0.0, r1 = nd >= 0 ? S.colI[1][il] : 0.0, r2 = nd >= 0 ? S.colI[2][il] : 0.0;
    thomas_sweeps(S, alg_only, r0, r1, r2);
    if (nd >= 0) { S.x2[0][nd] = r0; S.x2[1][nd] = r1; S.x2[2][nd] = r2; }
    // border pivot d - v.x2 of the control row (v, d): V: Phi_s[1] - Phi_s[end]; P: I I1C (same) with d = V I1C; eta_p: Phi_s.n[1] - Phi_e.n[1]
    const double vx = (mode == PLH_MODE_ETA_P) ? lane_bcast(r2, tw_lane(NP + NS)) - lane_bcast(r1, tw_lane(NP + NS)) : lane_bcast(r2, tw_lane(0)) - lane_bcast(r2, tw_lane(NE - 1));
    if (lane == 0) S.bord = (mode == PLH_MODE_P) ? S.ctrlJ[1] - S.ctrlJ[0] * vx : -vx;
    PL_SYNC();
  }
  PL_TOCD(S, 5);
}

// solve J x = b in place (b is an LDS vector of NST entries).  alg_only: only rows/cols NDIFF.. are touched.
template <class M>
PL_DEV void iso_solve(CellLDS<M>& S, LaneRegs& R, double* b, int mode, bool alg_only) {
  PL_MODEL(M);
  const int lane = lane_id();
  const CellConst& c = S.cc;
  const int r = lane % NR, g = lane / NR;
  PL_TICE(2);
  // a. particle partial solutions  w = A^-1 b_cs : four independent accumulation chains (pass = particles pass*6 .. pass*6+5)
  if constexpr (PL_CSDPP<M>) {
    if (!alg_only) {       // ROW layout: one particle per 16-lane DPP row, the right-hand side through rowb_fmac (same sums, same order)
      const int q = lane >> 4, rr = lane & 15, rc = rr < NR ? rr : NR - 1;
      double AP[NR], AN[NR];
#pragma unroll
      for (int k = 0; k < NR; k++) { AP[k] = S.Ainv[0][rc * NR + k]; AN[k] = S.Ainv[1][rc * NR + k]; }
      int pp[CSD_PASS]; double w[CSD_PASS], bc[CSD_PASS];
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) {
        const int p = pass * CSD_G + q; pp[pass] = p < NJ ? p : NJ - 1; w[pass] = 0.0;
        const int rk = rc < nr_of(pp[pass]) ? rc : nr_of(pp[pass]) - 1;
        bc[pass] = b[O_CS + cs_off(pp[pass]) + rk];
      }
      csd_settle(bc);
      static_for<0, NR>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
#pragma unroll
        for (int pass = 0; pass < CSD_PASS; pass++)      // (a pass that lies entirely in one electrode takes that electrode's resolvent without a select: known per pass after unrolling)
          rowb_fmac<k>(w[pass], bc[pass], (pass + 1) * CSD_G <= NP ? AP[k] : (pass * CSD_G >= NP ? AN[k] : (pp[pass] < NP ? AP[k] : AN[k])));
      });
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) {
        const int p = pass * CSD_G + q;
        if (p < NJ && rr == nr_of(pp[pass]) - 1) S.w9[p] = w[pass];
        R.wreg[pass] = w[pass];
      }
    }
  } else
  if constexpr (M::SD == 0)
  if (!alg_only && (!M::W2 || wave_id() == 1)) {
    const int gg = lane < CS_LANES ? g : CS_G - 1;
    int pp[CS_PASS]; double w[CS_PASS];
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) { const int p = pass * CS_G + gg; pp[pass] = p < NJ ? p : NJ - 1; w[pass] = 0.0; }
    double AP[NR], AN[NR], bv[CS_PASS][NR];
#pragma unroll
    for (int k = 0; k < NR; k++) { AP[k] = S.Ainv[0][r * NR + k]; AN[k] = S.Ainv[1][r * NR + k]; }
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
#pragma unroll
      for (int k = 0; k < NR; k++) bv[pass][k] = b[O_CS + cs_off(pp[pass]) + k];
    }
#pragma unroll
    for (int k = 0; k < NR; k++) {
      if constexpr (GRID_DEFAULT) {
        w[0] += AP[k] * bv[0][k];                                   // pass 0: particles 0..5 (cathode)
        w[1] += (pp[1] < NP ? AP[k] : AN[k]) * bv[1][k];            // pass 1: particles 6..11 (mixed)
        w[2] += AN[k] * bv[2][k];                                   // passes 2,3: anode
        w[3] += AN[k] * bv[3][k];
      } else {
        // a pass that lies entirely in one electrode takes that electrode's resolvent without a select (known per pass after unrolling)
#pragma unroll
        for (int pass = 0; pass < CS_PASS; pass++)
          w[pass] += ((pass + 1) * CS_G <= NP ? AP[k] : (pass * CS_G >= NP ? AN[k] : (pp[pass] < NP ? AP[k] : AN[k]))) * bv[pass][k];
      }
    }
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p = pass * CS_G + gg;
      if (lane < CS_LANES && p < NJ && r == nr_of(p) - 1) S.w9[p] = w[pass];
      R.wreg[pass] = w[pass];
    }
  }
  PL_XSYNC();
  PL_TOCE(S, 2, 0);
  if (!M::W2 || wave_id() == 0) {                       // b .. e: wave 0 (two waves per cell)
  // b. fold c_s and j elimination into the node right-hand sides
  double bjp = 0.0, bjs = 0.0, bfl = 0.0, m0 = 0.0, m1 = 0.0, m2 = 0.0;
  [[maybe_unused]] double bca = 0.0, bq = 0.0;       // right-hand sides of the c_avg / Q rows of this node (quadratic / polynomial particles)
  int jx = 0; bool elec = false, sei_node = false;
  const int nd = tw_node(lane_id_pred());       // node of this lane in the twisted layout of thomas_sweeps (-1: idle)
  if constexpr (M::SD == 0 && !M::SEI) {       // (measured: +1.5 % on the isothermal kernels, -2 % with SEI, whose integrate kernel is already spilling: selected per model)
    // every LDS operand of this phase is loaded unconditionally, with indices clamped into range for the lanes / nodes that do not use it, and selected afterwards:
    // loads under the nested `if`s (node lane? electrode node? current mode?) were three dependent LDS round trips
    const int i = nd >= 0 ? nd : 0, sc = sec_of(i);
    elec = nd >= 0 && sc != 1; jx = sc == 0 ? i : (sc == 2 ? i - NS : 0);
    const double l_ce = b[O_CE + i], l_pe = b[O_PE + i], l_j = b[O_J + jx], l_ps = b[O_PS + jx], l_bI = b[O_I];
    const double l_gcs = S.gcs[jx], l_w9 = S.w9[jx], l_dj = S.dj[jx], l_ceJ = S.ceJ[i], l_peJ = S.peJ[i], l_psJ = S.psJ[jx];
    const double l_c0 = S.colI[0][i], l_c1 = S.colI[1][i], l_c2 = S.colI[2][i];
    if (nd >= 0) {
      double r0 = alg_only ? 0.0 : l_ce, r1 = l_pe, r2 = 0.0;
      if (elec) {
        bjp = l_j - (alg_only ? 0.0 : l_gcs * l_w9);
        r2 = l_ps;
        const double beta = bjp * l_dj;       // omega . b_u : what the eliminated local unknown j feeds back into the node rows
        if (!alg_only) r0 -= l_ceJ * beta;
        r1 -= l_peJ * beta; r2 -= l_psJ * beta;
        if (mode == PLH_MODE_I) {             // control row: 1 * x_I = b_I
          r0 -= l_c0 * l_bI; r1 -= l_c1 * l_bI; r2 -= l_c2 * l_bI;
        }
      }
      m0 = r0; m1 = r1; m2 = r2;
    }
  } else
  if (nd >= 0) {
    const int i = nd, sc = sec_of(i);
    elec = sc != 1; jx = sc == 0 ? i : i - NS;
    sei_node = M::SEI && sc == 2;
    double r0 = alg_only ? 0.0 : b[O_CE + i], r1 = b[O_PE + i], r2 = 0.0;
    if (elec) {
      if constexpr (M::SD == 0) bjp = b[O_J + jx] - (alg_only ? 0.0 : S.gcs[jx] * S.w9[jx]);
      else {   // particular part of c_s*: dc = b_c (-1/cj), dQ = b_Q / (-kappa - cj)
        const int el = sc == 0 ? 0 : 1;
        bca = alg_only ? 0.0 : b[O_CS + jx]; bq = (alg_only || M::SD != 2) ? 0.0 : b[O_Q + jx];
        bjp = b[O_J + jx] - S.gcs[jx] * (bca * S.rcjf[el][0] + c.csq[el] * bq * S.rcjf[el][1]);
      }
      r2 = b[O_PS + jx];
      double beta;                          // omega . b_u : what the eliminated local unknowns feed back into the node rows
      bool local3 = false;
      if constexpr (M::SEI) {
        if (sc == 2) {
          const int k = i - (NP + NS);
          bjs = b[O_JS + k]; bfl = alg_only ? 0.0 : b[O_FILM + k];
          double Wm[9]; for (int q = 0; q < 9; q++) Wm[q] = S.sei.Wl[q][k];
          beta = (Wm[0] + Wm[3]) * bjp + (Wm[1] + Wm[4]) * bjs + (Wm[2] + Wm[5]) * bfl;
          local3 = true;
        }
      }
      if (!local3) beta = bjp * S.dj[jx];
      if (!alg_only) r0 -= S.ceJ[i] * beta;
      r1 -= S.peJ[i] * beta; r2 -= S.psJ[jx] * beta;
      if (mode == PLH_MODE_I) {             // control row: 1 * x_I = b_I
        const double bI = b[O_I];
        r0 -= S.colI[0][i] * bI; r1 -= S.colI[1][i] * bI; r2 -= S.colI[2][i] * bI;
      }
    }
    m0 = r0; m1 = r1; m2 = r2;
  }
  PL_TOCE(S, 2, 1);
  // c. block-Thomas forward / backward substitution (systolic, registers + DPP)
  thomas_sweeps(S, alg_only, m0, m1, m2);
  PL_TOCE(S, 2, 2);
  double mx[3] = {m0, m1, m2};
  // d. border: control row couples Phi_s[first] - Phi_s[last] (voltage mode)
  double xI;
  if (mode == PLH_MODE_I) xI = b[O_I];
  else {
    double vm = (mode == PLH_MODE_ETA_P) ? lane_bcast(m2, tw_lane(NP + NS)) - lane_bcast(m1, tw_lane(NP + NS)) : lane_bcast(m2, tw_lane(0)) - lane_bcast(m2, tw_lane(NE - 1));
    if (mode == PLH_MODE_P) vm *= S.ctrlJ[0];
    xI = pl_div(b[O_I] - vm, S.bord);
    if (nd >= 0) { mx[0] -= xI * S.x2[0][nd]; mx[1] -= xI * S.x2[1][nd]; mx[2] -= xI * S.x2[2][nd]; }
  }
  PL_SYNC();
  // e. back-substitute the node-local unknowns (j; with SEI also j_s and film), write node unknowns
  double djs = 0.0;
  if (nd >= 0) {
    const int i = nd;
    if (!alg_only) b[O_CE + i] = mx[0];
    b[O_PE + i] = mx[1];
    if (elec) {
      b[O_PS + jx] = mx[2];
      const double gc = alg_only ? 0.0 : S.gce[jx];
      const double v0 = bjp - gc * mx[0] - S.gpe[jx] * mx[1] - S.gps[jx] * mx[2];
      bool local3 = false;
      if constexpr (M::SEI) {
        if (sei_node) {
          const int k = i - (NP + NS);
          double Wm[9]; for (int q = 0; q < 9; q++) Wm[q] = S.sei.Wl[q][k];
          const double v1 = bjs - S.sei.jsPE[k] * mx[1] - S.sei.jsPS[k] * mx[2] - S.sei.jsI[k] * xI;
          b[O_J + jx] = Wm[0] * v0 + Wm[1] * v1 + Wm[2] * bfl;
          djs = Wm[3] * v0 + Wm[4] * v1 + Wm[5] * bfl;
          b[O_JS + k] = djs;
          if (!alg_only) b[O_FILM + k] = Wm[6] * v0 + Wm[7] * v1 + Wm[8] * bfl;
          djs *= S.sei.sohw[k];
          local3 = true;
        }
      }
      if (!local3) b[O_J + jx] = v0 * S.dj[jx];
      if constexpr (M::SD != 0) {
        if (!alg_only) {
          const int el = sec_of(i) == 0 ? 0 : 1;
          const double dj = v0 * S.dj[jx];
          b[O_CS + jx] = (bca - c.csr[el] * dj) * S.rcjf[el][0];                       // dc = (csr dj - b_c)/cj
          if (M::SD == 2) b[O_Q + jx] = (bq - c.qj[el] * dj) * S.rcjf[el][1];        // dQ = (b_Q - qj dj)/(-kappa - cj)
        }
      }
    }
  }
  if constexpr (M::SEI) {                   // SOH row: sum_k sohw_k dj_s,k - cj dSOH = b_SOH (decoupled from everything else)
    const double sd = wave_sum(djs);
    if (lane == 0 && !alg_only) b[O_SOH] = (sd - b[O_SOH]) / S.sei.cjf;
  }
  if (lane == 0) b[O_I] = xI;
  }
  PL_XSYNC();
  PL_TOCE(S, 2, 3);
  // f. particles:  dc = w - (A^-1 e_last) * bj * dj
  if constexpr (PL_CSDPP<M>) {
    if (!alg_only) {       // ROW layout
      const int q = lane >> 4, rr = lane & 15, rc = rr < NR ? rr : NR - 1;
      const double aP = S.Ainv[0][rc * NR + NRP - 1] * c.bj_p, aN = S.Ainv[1][rc * NR + NRN - 1] * c.bj_n;
      double dj[CSD_PASS];
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) { const int p = pass * CSD_G + q; dj[pass] = b[O_J + (p < NJ ? p : NJ - 1)]; }
#pragma unroll
      for (int pass = 0; pass < CSD_PASS; pass++) {
        const int p = pass * CSD_G + q;
        const double v = R.wreg[pass] - (p < NP ? aP : aN) * dj[pass];
        if (p < NJ && rr < nr_of(p < NJ ? p : NJ - 1)) b[O_CS + cs_off(p) + rr] = v;
      }
    }
  } else
  if constexpr (M::SD == 0)
  if (!alg_only && (!M::W2 || wave_id() == 1)) {
    // unconditional loads with clamped indices first (one LDS latency), guarded stores last: a load inside `if (lane < ..)` is one exec-masked round trip per pass
    const int gg = lane < CS_LANES ? g : CS_G - 1;
    const double aP = S.Ainv[0][r * NR + NRP - 1] * c.bj_p, aN = S.Ainv[1][r * NR + NRN - 1] * c.bj_n;
    double dj[CS_PASS];
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) { const int p = pass * CS_G + gg; dj[pass] = b[O_J + (p < NJ ? p : NJ - 1)]; }
#pragma unroll
    for (int pass = 0; pass < CS_PASS; pass++) {
      const int p = pass * CS_G + gg;
      const double v = R.wreg[pass] - (p < NP ? aP : aN) * dj[pass];
      if (lane < CS_LANES && p < NJ && r < nr_of(p)) b[O_CS + cs_off(p) + r] = v;
    }
  }
  PL_SYNC();
  PL_TOCE(S, 2, 4);
}


// one entry of the full Jacobian in CSC order from its decode word (see build_csc_codes in petlion_hip.hip)
//   word = type<<24 | a<<16 | b<<8 | c
enum JT { JT_CE_L = 1, JT_CE_D, JT_CE_U, JT_CE_J, JT_CS_CS, JT_CS_J, JT_J_CE, JT_J_CS, JT_J_J, JT_J_PE, JT_J_PS,
          JT_PE_CL, JT_PE_CD, JT_PE_CU, JT_PE_L, JT_PE_D, JT_PE_U, JT_PE_J, JT_PS_L, JT_PS_D, JT_PS_U, JT_PS_J, JT_PS_I,
          JT_CTRL_P1, JT_CTRL_M1,
          JT_CTRL_PA, JT_CTRL_PB, JT_CTRL_PI, JT_CSA_D, JT_Q_Q, JT_Q_J, JT_J_Q, JT_CE_JS, JT_PE_JS, JT_PS_JS, JT_J_F, JT_F_JS, JT_F_F, JT_SOH_JS, JT_SOH_SOH, JT_JS_PS, JT_JS_PE, JT_JS_J, JT_JS_JS, JT_JS_F, JT_JS_I };
template <class M>
PL_DEV double iso_jac_entry(const CellLDS<M>& S, const Tables* __restrict__ tb, unsigned w, double cj) {
  PL_MODEL(M);
  const int t = w >> 24, a = (w >> 16) & 255, bb = (w >> 8) & 255, cc = w & 255;
  const CellConst& c = S.cc;
  switch (t) {
    case JT_CE_L: return S.ceL[a];
    case JT_CE_D: return S.ceD[a] - cj;
    case JT_CE_U: return S.ceU[a];
    case JT_CE_J: return S.ceJ[a];
    case JT_CS_CS: if constexpr (NR_EQ) return (a < NP ? c.kap_p : c.kap_n) * tb->Mp()[bb * NR + cc] - (bb == cc ? cj : 0.0);
                   else return (a < NP ? c.kap_p : c.kap_n) * tb->Mp(a < NP ? 0 : 1)[bb * NR + cc] - (bb == cc ? cj : 0.0);
    case JT_CS_J: if constexpr (M::SD != 0) return c.csr[a < NP ? 0 : 1]; else return a < NP ? c.bj_p : c.bj_n;
    case JT_CSA_D: return -cj;
    case JT_Q_Q: return -(a < NP ? c.kap_p : c.kap_n) - cj;
    case JT_Q_J: return c.qj[a < NP ? 0 : 1];
    case JT_J_Q: return S.gcs[a] * c.csq[a < NP ? 0 : 1];
    case JT_J_CE: return S.gce[a];
    case JT_J_CS: return S.gcs[a];
    case JT_J_J: if constexpr (M::SEI) { if (a >= NP) return S.sei.jjJ[a - NP]; }
                 if constexpr (M::SD != 0) return -1.0 + S.gcs[a] * c.csj[a < NP ? 0 : 1];
                 return -1.0;
    case JT_J_PE: return S.gpe[a];
    case JT_J_PS: return S.gps[a];
    case JT_PE_CL: return S.pcL[a];
    case JT_PE_CD: return S.pcD[a];
    case JT_PE_CU: return S.pcU[a];
    case JT_PE_L: return S.peL[a];
    case JT_PE_D: return S.peD[a];
    case JT_PE_U: return S.peU[a];
    case JT_PE_J: return S.peJ[a];
    case JT_PS_L: return 1.0;
    case JT_PS_D: return bb ? -1.0 : -2.0;
    case JT_PS_U: return 1.0;
    case JT_PS_J: return S.psJ[a];
    case JT_PS_I: return a == 0 ? c.JI0 : c.JI29;
    case JT_CTRL_P1: return 1.0;
    case JT_CTRL_M1: return -1.0;
    case JT_CTRL_PA: return S.ctrlJ[0];
    case JT_CTRL_PB: return -S.ctrlJ[0];
    case JT_CTRL_PI: return S.ctrlJ[1];
    case JT_CE_JS: return S.ceJ[a];         // j and j_s enter the node rows through j_total: same coefficients
    case JT_PE_JS: return S.peJ[a];
    case JT_PS_JS: return S.psJ[a];
  }
  if constexpr (M::SEI) {
    switch (t) {
      case JT_J_F: return S.sei.jjF[a];
      case JT_F_JS: return -c.Mrho;
      case JT_F_F: return -cj;
      case JT_SOH_JS: return S.sei.sohw[a];
      case JT_SOH_SOH: return -cj;
      case JT_JS_PS: return S.sei.jsPS[a];
      case JT_JS_PE: return S.sei.jsPE[a];
      case JT_JS_J: return S.sei.jsJ[a];
      case JT_JS_JS: return S.sei.jsJS[a];
      case JT_JS_F: return S.sei.jsF[a];
      case JT_JS_I: return S.sei.jsI[a];
    }
  }
  return 0.0;
}

template <class M>
PL_DEV void cell_factor(CellLDS<M>& S, LaneRegs& R, const Tables* __restrict__ tb, double cj, int mode, bool alg_only) {
  if constexpr (M::THERMAL) thermal_factor(S, R, tb, cj, mode, alg_only);
  else iso_factor(S, R, tb, cj, mode, alg_only);
}
template <class M>
PL_DEV void cell_solve(CellLDS<M>& S, LaneRegs& R, double* b, int mode, bool alg_only) {
  if constexpr (M::THERMAL) thermal_solve(S, R, S.tb, b, mode, alg_only);
  else iso_solve(S, R, b, mode, alg_only);
}
// FROZEN = the entry as it went into the last factorisation (thermal: the per-node kappa(T) and the T column of the particle rows are rebuilt from what
// thermal_factor kept; every other pool entry is only written by Jacobian passes anyway)
template <bool FROZEN = false, class M>
PL_DEV double jac_entry(const CellLDS<M>& S, const Tables* __restrict__ tb, unsigned w, double cj) {
  if constexpr (M::THERMAL) return thermal_jac_entry<FROZEN>(S, tb, w, cj);
  else return iso_jac_entry(S, tb, w, cj);
}

// General control row: a closure input of the state whose derivative programs are given (plh_run.n_dcol > 0; reference differentiate_residual_func,
// scalar_residual.jl:276-416 puts d(method - f)/dY into the last row of the Newton matrix).  The row can touch any column, so it is handled as a BORDER of the
// structured solve in current mode (whose control row is "x_I = b_I"):  with W = J_I^-1 e_I (the image of the I column, W[O_I] = 1; one extra solve per factorisation,
// kept in HBM) and x0 the mode-I solve of the right-hand side with b_I = 0,   x_I = (b_I - g.x0) / (g.W),   x = x0 + x_I W.
// Entry k of the row lives in lane k (value gv, column gcol).  Built by gen_factor (dfn_integrate.h).
struct GenRow {
  const plh_run* run = nullptr;   // nullptr: the run has no general control row
  double* W = nullptr;            // this cell's [NST] slice of IntegrateArgs.genW
  double gv = 0.0, bord = 1.0;    // lane k: value of entry k ; border pivot g . W
  int gcol = 0, ng = 0;           // lane k: column of entry k ; number of entries (<= 64)
  __device__ __forceinline__ bool on() const { return run != nullptr; }
  __device__ __forceinline__ int col(int k) const { return lane_bcast_i(gcol, k); }
  // g . b (wave-uniform; b an LDS vector)
  __device__ __forceinline__ double dot(const double* b, int first_col) const {
    double s = 0.0;
    for (int k = 0; k < ng; k++) { const int c = col(k); const double v = lane_bcast(gv, k); if (c >= first_col) s += v * b[c]; }
    return s;
  }
};
template <class M>
PL_DEV void gen_solve(CellLDS<M>& S, LaneRegs& R, double* b, bool alg_only, const GenRow& g) {
  PL_MODEL(M);
  const int lane = lane_id(), wv = wave_id();
  PL_XSYNC();                                            // (two waves per cell: the control row of b is wave 0's, and nothing has synchronised since it was written)
  const double rho = b[O_I];
  PL_XSYNC();
  if (lane == 0 && wv == 0) b[O_I] = 0.0;
  PL_XSYNC();
  cell_solve(S, R, b, PLH_MODE_I, alg_only);
  PL_XSYNC();
  const double xI = (rho - g.dot(b, alg_only ? NDIFF : 0)) / g.bord;
  PL_XSYNC();
  _Pragma("unroll") for (int k__ = 0; k__ < NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wv); vokg<M>(k__, lane, wv) && (!alg_only || n >= NDIFF)) b[n] += xI * g.W[n];
  PL_XSYNC();
}

// J x = b with `nref` steps of iterative refinement  x += J^-1 (b - J x)  against the matrix of the last factorisation (cjf = its cj): the parity mode
// plh_opts.refine.  The structured elimination (cell_solve) and a sparse LU (KLU in the reference, the oracle's LU) order their operations differently,
// so their solutions of the ill-conditioned Newton systems (cond ~1e15, badly row-scaled) differ by 1e-12 .. 3e-9 (tools/solve_accuracy.py); one
// refinement step with the fp64 residual makes both solutions agree with the exact one to ~1e-13 and with each other.  The product b - J x walks the
// Jacobian in row-major order through the same decode words as the CSC export (plh_jacobian), so it is generic over the model variants.
// b: LDS vector (in: right-hand side, out: solution); bsave: a free LDS vector.  alg_only: rows/columns >= NDIFF of the consistent-initialisation Newton
// (the dT twin row has no exported entries: its residual is taken as zero, i.e. the row is not refined).
template <class M>
PL_DEV void cell_solve_refined(CellLDS<M>& S, LaneRegs& R, const Tables* __restrict__ tb, double* b, double* bsave, double cjf, int mode, bool alg_only, int nref, const GenRow* g = nullptr) {
  PL_MODEL(M);
  const int lane = lane_id();
  const bool twin = mode == PL_MODE_DT_TWIN;
  const int jm = g ? PLH_MODE_I : (twin ? PLH_MODE_DT : mode);       // (general control row: rows other than the last as exported for current mode, the last from g)
  const int* __restrict__ ptr = tb->csr_ptr[jm]; const unsigned* __restrict__ code = tb->csr_code[jm]; const unsigned short* __restrict__ col = tb->csr_col[jm];
  const double cj = alg_only ? 0.0 : cjf;
  double xr[NTRIP];
  const int wv = wave_id();
  _Pragma("unroll") for (int k__ = 0; k__ < NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wv); vokg<M>(k__, lane, wv)) bsave[n] = b[n];
  PL_XSYNC();
  for (int it = 0; it <= nref; it++) {
    if (it > 0) {
      double rr[NTRIP];
      PL_XSYNC();                                          // the solution of the previous pass is complete in b (both waves)
      const double gdot = g ? g->dot(b, alg_only ? NDIFF : 0) : 0.0;
      _Pragma("unroll") for (int k__ = 0; k__ < NTRIP; k__++) {
        const int n = vrow<M>(k__, lane, wv);
        xr[k__] = 0.0; rr[k__] = 0.0;
        if (vokg<M>(k__, lane, wv) && (!alg_only || n >= NDIFF)) {
          xr[k__] = b[n];
          double s = 0.0;
          for (int k = ptr[n]; k < ptr[n + 1]; k++) { const int c = col[k]; if (!alg_only || c >= NDIFF) s += jac_entry<true>(S, tb, code[k], cj) * b[c]; }
          if (g && n == O_I) s = gdot;
          rr[k__] = (twin && n == O_I) ? 0.0 : bsave[n] - s;
        }
      }
      PL_XSYNC();
      _Pragma("unroll") for (int k__ = 0; k__ < NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wv); vokg<M>(k__, lane, wv) && (!alg_only || n >= NDIFF)) b[n] = rr[k__];
      PL_XSYNC();
    }
    if (g) gen_solve(S, R, b, alg_only, *g); else cell_solve(S, R, b, mode, alg_only);
    if (it > 0) {
      _Pragma("unroll") for (int k__ = 0; k__ < NTRIP; k__++) if (const int n = vrow<M>(k__, lane, wv); vokg<M>(k__, lane, wv) && (!alg_only || n >= NDIFF)) b[n] += xr[k__];
      PL_SYNC();
    }
  }
}

}  // namespace pl

#include "dfn_thermal.h"

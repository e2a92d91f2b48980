// plh_host.h -- what the host side of the C ABI (petlion_hip.hip) and the per-variant kernel translation units (variant_tu.hip) share.
//
// Every model variant (chemistry x aging x temperature x factor precision) is ONE template instantiation ModelT<...> of the device source and is
// compiled in its own translation unit (-DPL_VARIANT=<id>): the variants build in parallel, each with the optimisation level that suits its kernels,
// and link into the one library libpetlion_hip.so.  The host side reaches a variant only through its VariantOps table.
#pragma once
#include <hip/hip_runtime.h>
#include "dfn_cell.h"

// id, chemistry, SEI aging, temperature, precision (false / true / 2 = PLH_PREC_F64 / PLH_PREC_MIXED: fp32 storage of the Newton-matrix factors / PLH_PREC_F64_REFORDER), solid diffusion, thermodynamic factor, reaction kinetics,
// two waves per cell
#define PL_VARIANT_LIST(X)                                                                         \
  X(0, PLH_CHEM_LCO_LIC6, false, false, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)          \
  X(1, PLH_CHEM_NMC_LIC6, false, false, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)          \
  X(2, PLH_CHEM_LCO_LIC6, true, false, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)           \
  X(3, PLH_CHEM_NMC_LIC6, true, false, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)           \
  X(4, PLH_CHEM_LCO_LIC6, false, true, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)           \
  X(5, PLH_CHEM_LCO_LIC6, false, false, true, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)           \
  X(6, PLH_CHEM_NMC_LIC6, true, false, true, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)            \
  X(7, PLH_CHEM_LCO_LIC6, false, true, true, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)            \
  X(8, PLH_CHEM_LCO_LIC6, false, false, false, PLH_SD_QUADRATIC, PLH_TF_LINEAR, PLH_RXN_BV, 0)        \
  X(9, PLH_CHEM_LCO_LIC6, false, false, false, PLH_SD_POLYNOMIAL, PLH_TF_LINEAR, PLH_RXN_BV, 0)       \
  X(10, PLH_CHEM_LCO_LIC6, false, false, false, PLH_SD_FICKIAN, PLH_TF_NONLINEAR, PLH_RXN_BV, 0)      \
  X(11, PLH_CHEM_LCO_LIC6, false, false, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_MHC, 0)       \
  X(12, PLH_CHEM_LGM50, false, false, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)            \
  X(13, PLH_CHEM_LCO_LIC6, false, false, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 1)          \
  X(14, PLH_CHEM_LGM50, false, true, false, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)             \
  X(15, PLH_CHEM_LCO_LIC6, false, false, 2, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)              \
  X(16, PLH_CHEM_LCO_LIC6, false, true, 2, PLH_SD_FICKIAN, PLH_TF_LINEAR, PLH_RXN_BV, 0)
constexpr int PL_N_VARIANTS = 17;

struct IntegrateArgs {
  const pl::Tables* tb; int n_cells; const double* theta; const double* SOC0; const double* Y_init; const double* t_init; int n_runs; const plh_run* runs; plh_opts opts;
  plh_outputs out; double* scratch;   // scratch: [n_cells][2][NST]
  double* genW;                       // [n_cells][NST] or nullptr: border vector of the general control row (closures with derivative programs)
  pl::SensArgs sens;                  // forward parameter sensitivities (dfn_sens.h); n_sens = 0: none
  double* phig;                       // [n_cells][4][NPAD] or nullptr: BDF history orders 2 .. 5 of the variants that keep them in global memory (ModelT::PHI_GLOBAL)
};

struct SectionInfo { const char* name; int start, len; };

struct VariantOps {
  int id, chem, sei, thermal, mixed, sd, tf, rxn, w2;       // (mixed: plh_model_desc.precision, PLH_PREC_*)
  int N, Nd;
  int grid[7];                                                                       // N_p, N_s, N_n, N_r_p, N_a, N_z, N_r_n this table was compiled for
  const double *rad_M[2], *rad_LAM[2], *rad_V[2], *rad_W[2]; double rad_BJ[2];       // radial operator tables of N_r_p / N_r_n (radial_tables.h), N_r x N_r packed
  size_t lds_bytes;                                                                  // sizeof(CellLDS<M>): LDS per cell (= per workgroup)
  int fsave_doubles;                                                                 // doubles per cell a sensitivity step needs to park the integrator's factorisation (dfn_sens.h, sens_factor_copy)
  int phig_doubles;                                                                  // doubles of global-memory BDF history per cell (0: the history is LDS / register resident)
  unsigned (*classify)(const pl::Tables& tb, int mode, int r, int c);              // decode word of the structural Jacobian entry (r, c), 0 if structurally zero
  int (*sections)(SectionInfo* out);
  void (*initial_guess)(hipStream_t st, const pl::Tables* tb, int n, const double* theta, const double* SOC, double* Y);
  void (*residual)(hipStream_t st, const pl::Tables* tb, int n, const double* theta, const double* Y, const double* YP, int mode, double value, double* F, int row0, int nrows);
  void (*jacobian)(hipStream_t st, const pl::Tables* tb, int n, const double* theta, const double* Y, const double* YP, double cj, int mode, double* nz, const int* sel, int nsel);
  void (*linear_solve)(hipStream_t st, const pl::Tables* tb, int n, const double* theta, const double* Y, const double* YP, double cj, int mode, double* b, int nref);
  void (*init_consistent)(hipStream_t st, const pl::Tables* tb, int n, const double* theta, int mode, double value, double reltol_init, double* Y, double* YP, int* status,
                          int* iters, int nref);
  void (*integrate)(hipStream_t st, const IntegrateArgs& a, int features);    // features: pl::GenFlag bits (dfn_integrate.h)
};

// one definition per variant translation unit.  Weak: an experiment build may link a subset of the variants (tools/), plh_model_create then refuses the
// missing ones with PLH_E_UNSUPPORTED; the product build links all of them (tests/test_capi_symbols.py checks that every variant can be created)
#define PL_DECLARE_OPS(ID, CHEM, SEI, TH, MIX, SD, TF, RXN, W2) const VariantOps* plh_variant_ops_##ID() __attribute__((weak));
PL_VARIANT_LIST(PL_DECLARE_OPS)
#undef PL_DECLARE_OPS

// A library of variants compiled for ANOTHER discretisation (variant_tu.hip with -DPL_NP=.. etc. and -DPL_GRID_LIBRARY; petlion.jl_amd/grids.py) exports these two
// C symbols; plh_register_grid_library() loads it and consults its tables in plh_model_create.
extern "C" const VariantOps* plh_grid_variant_ops(int id);     // nullptr: variant not built into this grid library
extern "C" void plh_grid_dims(int* grid7);
// what a grid library was compiled against: bump PLH_HOST_ABI whenever VariantOps / IntegrateArgs / Tables change, so that a stale cached library is refused, not misread
constexpr int PLH_HOST_ABI = 9;
extern "C" void plh_grid_abi(int* abi, int* sizeof_ops, int* sizeof_args, int* sizeof_tables);

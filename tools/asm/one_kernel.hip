// ISA inspection: ONE kernel instantiation of one model variant (seconds to a minute instead of the three minutes of a whole variant translation unit).
//   hipcc --offload-arch=gfx950 -std=c++17 -O3 -DPL_ASM_MARKS -DPL_ONE_CHEM=0 -DPL_ONE_SEI=false -DPL_ONE_TH=true -DPL_ONE_KERNEL=2 [-DPL_DEV=...] -S --cuda-device-only
//         -Rpass-analysis=kernel-resource-usage -I petlion.jl_amd/csrc tools/asm/one_kernel.hip -o /tmp/one.s          (tools/asm/isa.py drives it and summarises the marked phases)
#include "radial_tables.h"
#include "petlion_kernels.h"
#ifndef PL_ONE_FEATURES
#define PL_ONE_FEATURES 0
#endif
using M1 = pl::ModelT<PL_ONE_CHEM, PL_ONE_SEI, PL_ONE_TH>;
#if PL_ONE_KERNEL == 0
template __global__ void pl::k_residual<M1>(const pl::Tables*, int, const double*, const double*, const double*, int, double, double*, int, int);
#elif PL_ONE_KERNEL == 1
template __global__ void pl::k_linear_solve<M1>(const pl::Tables*, int, const double*, const double*, const double*, double, int, double*, int);
#else
template __global__ void pl::k_integrate<M1, PL_ONE_FEATURES>(IntegrateArgs);
#endif

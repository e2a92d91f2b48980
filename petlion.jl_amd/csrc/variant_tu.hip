// variant_tu.hip -- one model variant's kernels: compile with -DPL_VARIANT=<id> (ids and template arguments: PL_VARIANT_LIST in plh_host.h).
// All variants link into libpetlion_hip.so next to the host side of the C ABI (petlion_hip.hip).
#include "petlion_kernels.h"

#ifndef PL_VARIANT
#error "compile with -DPL_VARIANT=<id>"
#endif

#define PL_DEFINE_OPS(ID, CHEM, SEI, TH, MIX) \
  const VariantOps* plh_variant_ops_##ID() { return pl::OpsOf<pl::ModelT<CHEM, SEI, TH, MIX>>::table(ID); }

#if PL_VARIANT == 0
PL_DEFINE_OPS(0, PLH_CHEM_LCO_LIC6, false, false, false)
#elif PL_VARIANT == 1
PL_DEFINE_OPS(1, PLH_CHEM_NMC_LIC6, false, false, false)
#elif PL_VARIANT == 2
PL_DEFINE_OPS(2, PLH_CHEM_LCO_LIC6, true, false, false)
#elif PL_VARIANT == 3
PL_DEFINE_OPS(3, PLH_CHEM_NMC_LIC6, true, false, false)
#elif PL_VARIANT == 4
PL_DEFINE_OPS(4, PLH_CHEM_LCO_LIC6, false, true, false)
#elif PL_VARIANT == 5
PL_DEFINE_OPS(5, PLH_CHEM_LCO_LIC6, false, false, true)
#elif PL_VARIANT == 6
PL_DEFINE_OPS(6, PLH_CHEM_NMC_LIC6, true, false, true)
#elif PL_VARIANT == 7
PL_DEFINE_OPS(7, PLH_CHEM_LCO_LIC6, false, true, true)
#elif PL_VARIANT == -1   /* every variant in one translation unit (the test-only wave-emulator build) */
PL_VARIANT_LIST(PL_DEFINE_OPS)
#else
#error "unknown PL_VARIANT"
#endif

// variant_tu.hip -- one model variant's kernels: compile with -DPL_VARIANT=<id> (ids and template arguments: PL_VARIANT_LIST in plh_host.h).
// All variants link into libpetlion_hip.so next to the host side of the C ABI (petlion_hip.hip).
#include "petlion_kernels.h"

#ifndef PL_VARIANT
#error "compile with -DPL_VARIANT=<id>"
#endif

#define PL_DEFINE_OPS(ID, CHEM, SEI, TH, MIX, SD, TF, RXN, W2) \
  const VariantOps* plh_variant_ops_##ID() { return pl::OpsOf<pl::ModelT<CHEM, SEI, TH, MIX, SD, TF, RXN, W2>>::table(ID); }

// the variant selected by -DPL_VARIANT (or all of them for PL_VARIANT == -1: the test-only wave-emulator build is one translation unit)
#define PL_V(ID, CHEM, SEI, TH, MIX, SD, TF, RXN, W2) template <> struct pl::VariantSel<ID> { using M = pl::ModelT<CHEM, SEI, TH, MIX, SD, TF, RXN, W2>; };
namespace pl { template <int ID> struct VariantSel; }
PL_VARIANT_LIST(PL_V)
#undef PL_V

#if PL_VARIANT == -1
PL_VARIANT_LIST(PL_DEFINE_OPS)
#else
#define PL_ONE_(ID) const VariantOps* plh_variant_ops_##ID() { return pl::OpsOf<pl::VariantSel<ID>::M>::table(ID); }
#define PL_ONE(ID) PL_ONE_(ID)
PL_ONE(PL_VARIANT)
#endif

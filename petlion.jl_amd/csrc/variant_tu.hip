// variant_tu.hip -- one model variant's kernels: compile with -DPL_VARIANT=<id> (ids and template arguments: PL_VARIANT_LIST in plh_host.h).
// All variants link into libpetlion_hip.so next to the host side of the C ABI (petlion_hip.hip).
#include "radial_tables.h"
#include "petlion_kernels.h"

#if !defined(PL_VARIANT) && !defined(PL_GRID_GLUE)
#error "compile with -DPL_VARIANT=<id>"
#endif

#ifdef PL_GRID_GLUE
// Glue object of a library of variants compiled for ANOTHER discretisation (petlion.jl_amd/grids.py): the variant objects of that library are this file compiled with
// -DPL_VARIANT=<id> -DPL_NP=.. -DPL_NS=.. -DPL_NN=.. -DPL_NR=.. -DPL_NA=.. -DPL_NZ=.. (and -Dpl=<per-grid namespace>, hidden visibility), this object adds the two C
// entry points plh_register_grid_library() looks for.  Variants that were not built into the library resolve to null (weak declarations in plh_host.h).
extern "C" __attribute__((visibility("default"))) const VariantOps* plh_grid_variant_ops(int id) {
  switch (id) {
#define PL_OPS_CASE(ID, CHEM, SEI, TH, MIX, SD, TF, RXN, W2) case ID: return plh_variant_ops_##ID ? plh_variant_ops_##ID() : nullptr;
    PL_VARIANT_LIST(PL_OPS_CASE)
#undef PL_OPS_CASE
  }
  return nullptr;
}
extern "C" __attribute__((visibility("default"))) void plh_grid_abi(int* abi, int* so, int* sa, int* st) {
  *abi = PLH_HOST_ABI; *so = (int)sizeof(VariantOps); *sa = (int)sizeof(IntegrateArgs); *st = (int)sizeof(pl::Tables);
}
#ifdef PL_CLOSURE_HEADER
// a closure library (petlion.jl_amd/closure_lib.py): the digest of the protocol whose closures are compiled in (plh_model_attach_closure_library)
extern "C" __attribute__((visibility("default"))) unsigned long long plh_closure_library_digest() { return pl::PL_CLOSURE_DIGEST; }
#endif
extern "C" __attribute__((visibility("default"))) void plh_grid_dims(int* g) { g[0] = pl::NP; g[1] = pl::NS; g[2] = pl::NN; g[3] = pl::NRP; g[4] = pl::NA; g[5] = pl::NZ; g[6] = pl::NRN; }
#else

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
#endif  // PL_GRID_GLUE

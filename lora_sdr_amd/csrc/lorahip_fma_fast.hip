// The opt-in CONTRACTED build of the tuned batch kernels, SF6-10 (lorahip_set_variant(ctx, LORAHIP_VARIANT_FMA)): lorahip_fast.hip
// compiled once more with every complex multiply as one packed multiply + one packed FMA (LORAHIP_FMA, lorahip_device.h). It exists to
// MEASURE what the reference's bit-exact operation graph costs (profiles/r04): its bins differ from the CPU
// build's in the last place or two, so nothing selects it by default and level 3 has no access to it.
//
// The whole translation unit lives in its own namespace (the kernels' mangled names must differ from the exact build's).
#define LORAHIP_FMA 1
#define lorahip lorahip_fma
#include "lorahip_fast.hip"
#undef lorahip

namespace lorahip { hipError_t ensureDynamicLds(const void *kernel, size_t bytes, unsigned long long &doneMask); }
namespace lorahip_fma {
// the one host helper the kernels' launchers need from the rest of the library (shared with lorahip_fma_wide.hip)
hipError_t ensureDynamicLds(const void *kernel, size_t bytes, unsigned long long &doneMask) { return ::lorahip::ensureDynamicLds(kernel, bytes, doneMask); }
}

extern "C" int lorahip_fma_fast_launch(const int sf, const void *args, const void *tables, void *stream)
{
    return int(lorahip_fma::launchFast(sf, 0, *static_cast<const lorahip_fma::DetectArgs *>(args), *static_cast<const lorahip_fma::FastTables *>(tables),
                                       static_cast<hipStream_t>(stream)));
}

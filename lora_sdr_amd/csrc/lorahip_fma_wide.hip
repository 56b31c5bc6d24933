// The opt-in CONTRACTED build of the tuned batch kernels, SF11 / SF12: lorahip_wide.hip's detectWide compiled once more with every
// complex multiply as one packed multiply + one packed FMA. See lorahip_fma_fast.hip.
#define LORAHIP_FMA 1
#define lorahip lorahip_fma
#include "lorahip_wide.hip"
#undef lorahip

extern "C" int lorahip_fma_wide_launch(const int sf, const void *args, const void *tables, void *stream)
{
    return int(lorahip_fma::launchWide(sf, 0, *static_cast<const lorahip_fma::DetectArgs *>(args), *static_cast<const lorahip_fma::FastTables *>(tables),
                                       static_cast<hipStream_t>(stream)));
}

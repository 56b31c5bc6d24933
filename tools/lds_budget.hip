// LDS per workgroup of the SF11 / SF12 streaming kernels and what it allows per compute unit (host program; no GPU needed):
//   hipcc --offload-arch=gfx950 -O1 -std=c++17 -ffp-contract=off -Ilora_sdr_amd/csrc -Iinclude -x hip tools/lds_budget.hip -o /tmp/lds_budget && /tmp/lds_budget
#include "lorahip_wide.hip"
#include "lorahip_streamkernel.h"
#include <cstdio>
using namespace lorahip;
template <class C> static void wide(const char *name)
{
    const size_t smem = size_t(C::TWN + C::XW) * sizeof(float2) + 4 * sizeof(RedRec) + 2 * sizeof(float2) + 12 * sizeof(int) + FineDims<C::LOG2N>::BYTES;
    printf("%-14s threads %4d  LDS per workgroup %6zu B = exchange %6zu + stage twiddles %6zu + split fine-tune tables %6zu + rest %zu -> %d workgroups per CU by LDS (160 KiB) = %d wavefronts per SIMD\n",
           name, C::T, smem, size_t(C::XW) * sizeof(float2), size_t(C::TWN) * sizeof(float2), size_t(FineDims<C::LOG2N>::BYTES),
           4 * sizeof(RedRec) + 2 * sizeof(float2) + 12 * sizeof(int), int(163840 / smem), int(163840 / smem) * (C::T / 64) / 4);
}
int main()
{
    wide<StreamWide11>("StreamWide11");
    wide<StreamWide12>("StreamWide12");
    return 0;
}
namespace lorahip { hipError_t ensureDynamicLds(const void *, size_t, unsigned long long &) { return hipSuccess; } int residentWorkgroups(const void *, int, size_t) { return 0; } }

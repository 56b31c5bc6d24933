// issue rate of v_pk_fma_f32 in the forms the channeliser uses (run on the GPU): all-VGPR, SGPR src0, with op_sel / neg
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters)
{
    v2f acc[16];
    for (int i = 0; i < 16; i++) acc[i] = (v2f){in[i], in[i + 1]};
    v2f x = {in[threadIdx.x & 15], in[17]};
    v2f gv = {in[3], in[5]};
    v2f gs = {__builtin_amdgcn_readfirstlane(__float_as_int(in[3])) * 1.0f, in[5]};
    unsigned long long gsi; { const v2f *p = reinterpret_cast<const v2f *>(in); gsi = *reinterpret_cast<const unsigned long long *>(p + blockIdx.x % 3); }
    (void)gs;
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
                if (MODE == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(gv), "v"(x));
                if (MODE == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(gsi), "v"(x));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc[i]) : "v"(gv), "v"(x));
                if (MODE == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc[i]) : "s"(gsi), "v"(x));
                if (MODE == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(acc[i].x) : "v"(gv.x), "v"(x.x));
                if (MODE == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"(gsi), "v"(x));
                if (MODE >= 6) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[i]) : "s"(gsi), "v"(x));
            }
        // the channeliser's block boundary: a wait every 32 FMAs (MODE 6), plus ten s_nop (MODE 7), plus an LDS read (MODE 8)
        if (MODE == 6) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(x));
        if (MODE == 7) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0\n\ts_nop 0" : "+v"(x));
        if (MODE == 8) { unsigned addr = threadIdx.x * 8; asm volatile("s_waitcnt lgkmcnt(0)\n\tds_read_b64 %0, %1" : "+v"(x) : "v"(addr)); }
    }
    float s = 0;
    for (int i = 0; i < 16; i++) s += acc[i].x + acc[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE> void run(const char *name, float *out, float *in, int wavesPerSimd)
{
    const int iters = 400000, blocks = 256 * wavesPerSimd;       // 256 CUs x (4 waves per block = 1 per SIMD) x wavesPerSimd
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<blocks, 256>>>(out, in, 2000);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<blocks, 256>>>(out, in, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double instr = double(iters) * 64 * 4 * blocks;       // wave-instructions
    const double perSimdPerSec = instr / (256.0 * 4) / (ms * 1e-3);
    printf("%-44s %d waves/SIMD: %.3f ms, %.3f G wave-instr/s per SIMD (= %.2f cycles per instr at 2.4 GHz), %.1f TFLOP/s\n", name, wavesPerSimd, ms,
           perSimdPerSec / 1e9, 2.4e9 / perSimdPerSec, (MODE == 4 ? 2.0 : 4.0) * 64 * instr / (ms * 1e-3) / 1e12);
}
int main()
{
    float *in, *out; hipMalloc(&in, 4096); hipMalloc(&out, 256 * 8 * 256 * 4);
    float h[64]; for (int i = 0; i < 64; i++) h[i] = 1.0f / (i + 3); hipMemcpy(in, h, sizeof(h), hipMemcpyHostToDevice);
    for (int w : {4})
    {
        run<0>("v_pk_fma_f32 vgpr", out, in, w);
        run<1>("v_pk_fma_f32 sgpr src0", out, in, w);
        run<2>("v_pk_fma_f32 vgpr op_sel+neg", out, in, w);
        run<3>("v_pk_fma_f32 sgpr op_sel+neg", out, in, w);
        run<5>("v_pk_fma_f32 sgpr op_sel_hi:[0,1,1]", out, in, w);
        run<4>("v_fma_f32 vgpr", out, in, w);
        run<6>("pk_fma sgpr + s_waitcnt per 64", out, in, w);
        run<7>("pk_fma sgpr + s_waitcnt + 10 s_nop per 64", out, in, w);
        run<8>("pk_fma sgpr + s_waitcnt + ds_read per 64", out, in, w);
    }
    return 0;
}

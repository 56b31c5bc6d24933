// Device-side helpers shared by the kernel translation units.
//
// Numerics contract: every FFT kernel evaluates the SAME dataflow graph as
// the reference's kissfft for N = 2^SF -- decimation in time, radix-4 stages outermost and one
// radix-2 stage innermost for odd SF (kissfft.hh:34-51), each butterfly in kissfft's operation
// order (kissfft.hh:128-157) on kissfft's own float twiddles (kissfft.hh:17-22, uploaded from
// the host) -- in strict IEEE fp32 with NO fused multiply-add. Every TU including this header
// is compiled with -ffp-contract=off and carries the pragma below. The graph fixes the result
// bits; which lane holds which point, what travels through LDS, and the issue order inside a
// stage are free and chosen for the machine.
#pragma once
#pragma clang fp contract(off)
#include "lorahip_internal.h"

namespace lorahip {

//! (ac - bd, ad + bc): std::complex<float>::operator* for finite operands, no FMA
__device__ __forceinline__ float2 cmul(const float2 a, const float2 b)
{
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(const float2 a, const float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(const float2 a, const float2 b) { return make_float2(a.x - b.x, a.y - b.y); }

//! kf_bfly4 for one k (kissfft.hh:143-155), forward; s0..s2 are the three twiddled inputs
__device__ __forceinline__ void bfly4core(float2 &f0, float2 &f1, float2 &f2, float2 &f3,
                                          const float2 s0, const float2 s1, const float2 s2)
{
    const float2 s5 = csub(f0, s1);
    f0 = cadd(f0, s1);
    const float2 s3 = cadd(s0, s2);
    float2 s4 = csub(s0, s2);
    s4 = make_float2(s4.y, -s4.x);
    f2 = csub(f0, s3);
    f0 = cadd(f0, s3);
    f1 = cadd(s5, s4);
    f3 = csub(s5, s4);
}

__device__ __forceinline__ void bfly4(float2 &f0, float2 &f1, float2 &f2, float2 &f3,
                                      const float2 t1, const float2 t2, const float2 t3)
{
    bfly4core(f0, f1, f2, f3, cmul(f1, t1), cmul(f2, t2), cmul(f3, t3));
}

//! kf_bfly4 whose three twiddles are twiddle(0) = (1,0): x*(1,0) == x as a value for finite x
//! (only the sign of a zero can differ, which never reaches a non-zero result or |.|^2)
__device__ __forceinline__ void bfly4unit(float2 &f0, float2 &f1, float2 &f2, float2 &f3)
{
    bfly4core(f0, f1, f2, f3, f1, f2, f3);
}

//! kf_bfly2 for one k (kissfft.hh:131-133)
__device__ __forceinline__ void bfly2(float2 &f0, float2 &f1, const float2 t)
{
    const float2 v = cmul(f1, t);
    f1 = csub(f0, v);
    f0 = cadd(f0, v);
}
__device__ __forceinline__ void bfly2unit(float2 &f0, float2 &f1)
{
    const float2 v = f1;
    f1 = csub(f0, v);
    f0 = cadd(f0, v);
}

/***********************************************************************
 * Packed-pair forms for the tuned kernels. A complex value is one 64-bit VGPR pair and every
 * operation below is the SAME IEEE multiply / add as the scalar form above, two at a time:
 * v_pk_mul_f32 / v_pk_add_f32 round each half exactly like v_mul_f32 / v_add_f32, op_sel picks
 * which half of a source feeds which half of the result and neg_lo/neg_hi flip a sign (exact),
 * so no fused operation and no re-association enters. hipcc does not find the half-negated add
 * or the rotate-by(-j) forms on its own (it falls back to v_mov shuffles), hence the asm.
 **********************************************************************/
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

//! (a.x*b.x - a.y*b.y, a.y*b.x + a.x*b.y): 3 instructions
__device__ __forceinline__ v2f cmulv(const v2f a, const v2f b)
{
    v2f p, q, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));   // (a.x*b.x, a.y*b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(q) : "v"(a), "v"(b));   // (a.y*b.y, a.x*b.y)
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p), "v"(q));                   // (p.x-q.x, p.y+q.y)
    return r;
}
//! same with b conjugated: a * (b.x, -b.y) -- the up-chirp table is conj(down-chirp table)
__device__ __forceinline__ v2f cmulConjv(const v2f a, const v2f b)
{
    v2f p, q, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));   // (a.x*b.x, a.y*b.x)
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(q) : "v"(a), "v"(b));   // (a.y*b.y, a.x*b.y)
    asm("v_pk_add_f32 %0, %1, %2 neg_hi:[0,1]" : "=v"(r) : "v"(p), "v"(q));                   // (p.x+q.x, p.y-q.y)
    return r;
}
//! s5 + (s4.y, -s4.x)  and  s5 - (s4.y, -s4.x): kissfft.hh:150,153-154 without materialising the rotation
__device__ __forceinline__ v2f addRotv(const v2f s5, const v2f s4)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_hi:[0,1]" : "=v"(r) : "v"(s5), "v"(s4));
    return r;
}
__device__ __forceinline__ v2f subRotv(const v2f s5, const v2f s4)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(r) : "v"(s5), "v"(s4));
    return r;
}

//! kf_bfly4 for one k (kissfft.hh:143-155), forward; s0..s2 are the three twiddled inputs
__device__ __forceinline__ void bfly4corev(v2f &f0, v2f &f1, v2f &f2, v2f &f3, const v2f s0, const v2f s1, const v2f s2)
{
    const v2f s5 = f0 - s1;
    f0 = f0 + s1;
    const v2f s3 = s0 + s2;
    const v2f s4 = s0 - s2;
    f2 = f0 - s3;
    f0 = f0 + s3;
    f1 = addRotv(s5, s4);
    f3 = subRotv(s5, s4);
}
__device__ __forceinline__ void bfly4v(v2f &f0, v2f &f1, v2f &f2, v2f &f3, const v2f t1, const v2f t2, const v2f t3)
{
    bfly4corev(f0, f1, f2, f3, cmulv(f1, t1), cmulv(f2, t2), cmulv(f3, t3));
}
__device__ __forceinline__ void bfly4unitv(v2f &f0, v2f &f1, v2f &f2, v2f &f3) { bfly4corev(f0, f1, f2, f3, f1, f2, f3); }
__device__ __forceinline__ void bfly2unitv(v2f &f0, v2f &f1)
{
    const v2f v = f1;
    f1 = f0 - v;
    f0 = f0 + v;
}

/***********************************************************************
 * kissfft's plan for N = 2^LOG2N as compile-time constants.
 * Stage s (0 = outermost) has radix p_s and remainder m_s; input digit q_s has weight
 * fstride_s = p_0..p_{s-1} in the sample index n and weight m_s in the work-array position
 * (kf_work recursion, kissfft.hh:83-104).
 **********************************************************************/
template <int LOG2N> struct Plan
{
    static constexpr int N = 1 << LOG2N;
    static constexpr int R4 = LOG2N / 2;          // number of radix-4 stages
    static constexpr bool HAS_R2 = (LOG2N & 1);   // innermost radix-2 stage (m = 1)
    //! work-array position of input sample n (digit reversal of the mixed-radix index)
    __host__ __device__ static constexpr int pos(int n)
    {
        int p = 0, m = N;
        for (int s = 0; s < R4; s++) { m >>= 2; p += (n & 3) * m; n >>= 2; }
        if (HAS_R2) p += (n & 1);
        return p;
    }
};

//! fine-tune index recurrence, one step (LoRaDemod.cpp:160-162):
//!   _fineTuneIndex -= _finefreqError * _fineSteps   (int -= float: float subtract, truncate)
__device__ __forceinline__ int fineStep(const int idx, const float d /* = err*128 */, const int M)
{
    int n = (int)((float)idx - d);
    if (n < 0) n += M;
    else if (n >= M) n -= M;
    return n;
}

//! arg-max combine with the reference's tie-break: strict '>' scanning upwards keeps the
//! LOWEST index among equal maxima (LoRaDetector.hpp:43)
__device__ __forceinline__ void argmaxCombine(float &v, int &i, const float ov, const int oi)
{
    const bool take = (ov > v) | ((ov == v) & (oi < i));   // branch-free: two selects
    v = take ? ov : v;
    i = take ? oi : i;
}

/***********************************************************************
 * detect() tail for one window, executed by one lane (LoRaDetector.hpp:50-61)
 **********************************************************************/
template <class CPX>
__device__ __forceinline__ void detectTail(const DetectArgs &a, const unsigned w, const int maxIndex,
                                           const float maxValue, const double total,
                                           const CPX leftBin, const CPX rightBin)
{
    const float noise = sqrtf((float)(total - (double)maxValue));
    const float fundamental = sqrtf(maxValue);
    // log10 evaluated in double and rounded once: within an ulp of a correctly rounded log10f
    const float powerAvg = 20 * (float)log10((double)noise) - a.powerScale;
    const float power = 20 * (float)log10((double)fundamental) - a.powerScale;
    // std::abs(complex<float>) = hypotf; the double form is its correctly rounded value
    const float left = (float)sqrt((double)leftBin.x * (double)leftBin.x + (double)leftBin.y * (double)leftBin.y);
    const float right = (float)sqrt((double)rightBin.x * (double)rightBin.x + (double)rightBin.y * (double)rightBin.y);
    const double demon = (2.0 * (double)fundamental) - (double)right - (double)left;
    float fIndex = 0.0f;
    if (demon != 0.0) fIndex = (float)(0.5 * (double)(right - left) / demon);
    a.sym[w] = (unsigned short)maxIndex;
    a.power[w] = power;
    a.powerAvg[w] = powerAvg;
    a.fIndex[w] = fIndex;
}

} // namespace lorahip

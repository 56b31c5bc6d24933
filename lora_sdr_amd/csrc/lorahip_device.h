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

// Wavefronts that share a SIMD are issued OLDEST FIRST: of two co-resident wavefronts of the streaming kernels (256 registers each)
// the older runs at nearly its solo speed and the other takes what is left -- workgroups of one launch that do the same work finish
// 25-30 % apart (tools/wg_timeline.py: 1.5 ms against 2.0-2.3 ms per SF11 channel) -- and a launch of long-running wavefronts ends
// in a tail where the late ones run alone, latency-bound, on half-empty SIMDs. rotatePriority(), once per work() call / window set,
// hands the SIMD's priority round on a clock counter its wavefronts see alike (s_setprio): they advance at the same rate and
// finish together. Measured (profiles/r04/s22_*, s23_*): level-3 kernels +3-9 % at SF7-10, +6 % at SF12, +27 % at SF11 together
// with the persistent grid (lorahip_wide.hip); the half period (2^16 / 2^18 clocks: 27 / 110 us) does not matter within the
// noise, 2^14 is too short.
// Only wavefronts that are NOT REPLACED when they finish take part (the last resident set of a grid, every wavefront of a persistent
// one): earlier workgroups hold a priority above them, oldest first as the hardware has it -- one of two finishes early and its slot
// goes to the next workgroup, which is what a grid of 1.5 resident sets wants (LORAHIP_PRIO_HOLD 0: everybody rotates, the A/B).
#ifndef LORAHIP_PRIO_ALTERNATE
#define LORAHIP_PRIO_ALTERNATE 18       // streaming kernels: log2 of the period per wavefront in shader clocks; 0: off (the A/B)
#endif
#ifndef LORAHIP_PRIO_HOLD
#define LORAHIP_PRIO_HOLD 1
#endif
// The batch kernels are persistent too (a wavefront walks window sets first, first + waveCount, ...: a static share each), two to
// three wavefronts per SIMD: the same rotation once per set, +2-3 % at SF7-11 and +5 % at SF12 in both shapes
// (profiles/r04/s25_ab_batch_priority.txt; tools/wave_timeline.py: without it the three slots of a SIMD end a 226 us SF7 launch after
// 122 / 172 / 219 us), another +1.5-2.5 % at SF7 / 10 / 12 with the whole order rotating (s31_ab_batch_priority_full_order.txt).
#ifndef LORAHIP_PRIO_BATCH
#define LORAHIP_PRIO_BATCH 16           // batch kernels: log2 of the period per wavefront in shader clocks (launches last 0.2-0.4 ms); 0: off
#endif
__device__ __forceinline__ int wavefrontSlot()
{
    return int(__builtin_amdgcn_s_getreg((3 << 11) | 4));      // HW_ID.wave_id: the wavefront's slot on its SIMD
}
//! a wavefront whose slot goes to another workgroup when it finishes: ahead of the ones that rotate (oldest first among its like)
template <int K>
__device__ __forceinline__ void holdPriority(const bool hold)
{
    if (K > 0 && hold) __builtin_amdgcn_s_setprio(3);
}
template <int WPS, int K>
__device__ __forceinline__ void rotatePriority(const int slot)
{
    if (K > 0 && WPS >= 2)
    {
        unsigned p = unsigned(__builtin_amdgcn_s_memtime() >> K);
        unsigned me = unsigned(slot);
        if (WPS == 2) { p &= 1u; me &= 1u; }
        else if (WPS == 4) { p &= 3u; me &= 3u; }
        else { p &= 1023u; p -= WPS * ((p * (2048u / WPS + 1u)) >> 11); me = me >= unsigned(WPS) ? me - unsigned(WPS) : me; }   // p mod 3 (exact below 1024)
        // the whole ORDER rotates, not only who is first: with three wavefronts "one high, two equal" leaves the older of the two
        // ahead whenever the first one waits, and the slots still finish 15 % apart (tools/wave_timeline.py)
        const unsigned rank = me >= p ? me - p : me + unsigned(WPS) - p;        // 0: first this turn
        if (WPS == 2)
        {
            if (rank == 0) __builtin_amdgcn_s_setprio(2);                       // (3 is holdPriority's)
            else __builtin_amdgcn_s_setprio(0);
        }
        else if (rank == 0) __builtin_amdgcn_s_setprio(3);
        else if (rank == 1) __builtin_amdgcn_s_setprio(2);
        else if (rank == 2) __builtin_amdgcn_s_setprio(1);
        else __builtin_amdgcn_s_setprio(0);
    }
}


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

#ifdef LORAHIP_FMA
/* The opt-in CONTRACTED build of the batch kernels (lorahip_fma_*.hip; lorahip_set_variant(ctx, LORAHIP_VARIANT_FMA)): every complex
 * multiply is one packed multiply and one packed FMA -- a.x*b.x - (a.y*b.y) with the inner product rounded once and the outer one not
 * at all -- instead of two multiplies and an add. NOT the reference's operation graph: bins differ from the CPU build's in the last
 * place or two. It exists to measure what the bit-exact graph costs (profiles/r04); nothing selects it by default and level 3
 * refuses it. */
__device__ __forceinline__ v2f cmulv(const v2f a, const v2f b)
{
    v2f q, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(q) : "v"(a), "v"(b));                                      // (a.y*b.y, a.x*b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(q));     // (a.x*b.x - q.x, a.y*b.x + q.y)
    return r;
}
__device__ __forceinline__ v2f cmulConjv(const v2f a, const v2f b)
{
    v2f q, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(q) : "v"(a), "v"(b));                                      // (a.y*b.y, a.x*b.y)
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_hi:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(q));     // (a.x*b.x + q.x, a.y*b.x - q.y)
    return r;
}
#else
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
#endif
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
// Issue order matters to the assembler, not to the arithmetic: the compiler cannot see inside an asm statement, assumes the
// worst (a gfx940 dst_sel forwarding hazard) whenever one asm result is consumed by the very next instruction, and pads with
// an s_nop -- which costs about two thirds of a packed operation (tools/pkfma_rate.hip). The helpers below therefore
// interleave independent products so that no asm result is used by its immediate successor. Same operations, same bits.
__device__ __forceinline__ v2f mulLoV(const v2f a, const v2f b)     // (a.x*b.x, a.y*b.x)
{
    v2f p;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(p) : "v"(a), "v"(b));
    return p;
}
__device__ __forceinline__ v2f mulHiV(const v2f a, const v2f b)     // (a.y*b.y, a.x*b.y)
{
    v2f q;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,1]" : "=v"(q) : "v"(a), "v"(b));
    return q;
}
__device__ __forceinline__ v2f subAddV(const v2f p, const v2f q)    // (p.x-q.x, p.y+q.y)
{
    v2f r;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1]" : "=v"(r) : "v"(p), "v"(q));
    return r;
}
//! (a.x*b.x - q.x, a.y*b.x + q.y) for q = mulHiV(a, b): the second half of the contracted complex product (LORAHIP_FMA builds)
__device__ __forceinline__ v2f fmaLoV(const v2f a, const v2f b, const v2f q)
{
    v2f r;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1] neg_lo:[0,0,1]" : "=v"(r) : "v"(a), "v"(b), "v"(q));
    return r;
}
#ifdef LORAHIP_FMA
__device__ __forceinline__ void bfly4v(v2f &f0, v2f &f1, v2f &f2, v2f &f3, const v2f t1, const v2f t2, const v2f t3)
{
    const v2f q1 = mulHiV(f1, t1), q2 = mulHiV(f2, t2), q3 = mulHiV(f3, t3);
    const v2f s0 = fmaLoV(f1, t1, q1), s1 = fmaLoV(f2, t2, q2), s2 = fmaLoV(f3, t3, q3);
    bfly4corev(f0, f1, f2, f3, s0, s1, s2);
}
template <int CNT>
__device__ __forceinline__ void dechirpMany(v2f *x, const v2f *c, const v2f f)
{
    static_assert(CNT % 4 == 0, "four values per round");
#pragma unroll
    for (int i = 0; i < CNT; i += 4)
    {
        v2f q[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = mulHiV(x[i + j], c[i + j]);
#pragma unroll
        for (int j = 0; j < 4; j++) y[j] = fmaLoV(x[i + j], c[i + j], q[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) q[j] = mulHiV(y[j], f);
#pragma unroll
        for (int j = 0; j < 4; j++) x[i + j] = fmaLoV(y[j], f, q[j]);
    }
}
#else
__device__ __forceinline__ void bfly4v(v2f &f0, v2f &f1, v2f &f2, v2f &f3, const v2f t1, const v2f t2, const v2f t3)
{
    const v2f p1 = mulLoV(f1, t1), q1 = mulHiV(f1, t1), p2 = mulLoV(f2, t2), q2 = mulHiV(f2, t2), p3 = mulLoV(f3, t3), q3 = mulHiV(f3, t3);
    const v2f s0 = subAddV(p1, q1), s1 = subAddV(p2, q2), s2 = subAddV(p3, q3);
    bfly4corev(f0, f1, f2, f3, s0, s1, s2);
}
//! x[i] = (x[i] * c[i]) * f for CNT values, four at a time with the products interleaved (see above)
template <int CNT>
__device__ __forceinline__ void dechirpMany(v2f *x, const v2f *c, const v2f f)
{
    static_assert(CNT % 4 == 0, "four values per round");
#pragma unroll
    for (int i = 0; i < CNT; i += 4)
    {
        v2f p[4], q[4], y[4];
#pragma unroll
        for (int j = 0; j < 4; j++) { p[j] = mulLoV(x[i + j], c[i + j]); q[j] = mulHiV(x[i + j], c[i + j]); }
#pragma unroll
        for (int j = 0; j < 4; j++) y[j] = subAddV(p[j], q[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) { p[j] = mulLoV(y[j], f); q[j] = mulHiV(y[j], f); }
#pragma unroll
        for (int j = 0; j < 4; j++) x[i + j] = subAddV(p[j], q[j]);
    }
}
#endif
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
 * group reductions without LDS: max / min of an unsigned over the aligned group of T lanes this lane
 * belongs to, result in every lane. DPP row permutations up to 16 lanes, gfx950's v_permlane16/32_swap
 * above. |X|^2 values are non-negative non-NaN floats, whose bit patterns order like unsigned integers,
 * so the arg-max is: group max of the value, then group min of the index among the lanes that hold it
 * (= the reference's lowest-index tie-break, LoRaDetector.hpp:43).
 **********************************************************************/
typedef unsigned v2u __attribute__((ext_vector_type(2)));
template <int T, bool MAX>
__device__ __forceinline__ unsigned groupReduceU(unsigned v)
{
#define LORAHIP_RED_STEP(O) { const unsigned o_ = (O); v = MAX ? (v > o_ ? v : o_) : (v < o_ ? v : o_); }
    if (T >= 2) LORAHIP_RED_STEP(__builtin_amdgcn_mov_dpp(v, 0xB1, 0xf, 0xf, false))     // quad_perm [1,0,3,2]
    if (T >= 4) LORAHIP_RED_STEP(__builtin_amdgcn_mov_dpp(v, 0x4E, 0xf, 0xf, false))     // quad_perm [2,3,0,1]
    if (T >= 8) LORAHIP_RED_STEP(__builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, false))    // row_half_mirror
    if (T >= 16) LORAHIP_RED_STEP(__builtin_amdgcn_mov_dpp(v, 0x140, 0xf, 0xf, false))   // row_mirror
    if (T >= 32) { const v2u r = __builtin_amdgcn_permlane16_swap(v, v, false, false); v = r.x; LORAHIP_RED_STEP(r.y) }
    if (T >= 64) { const v2u r = __builtin_amdgcn_permlane32_swap(v, v, false, false); v = r.x; LORAHIP_RED_STEP(r.y) }
#undef LORAHIP_RED_STEP
    return v;
}
//! sum of a double over the aligned group of T lanes, result in every lane -- the same pairing tree (strides 1, 2, 4, ...) on all
//! lanes, and every step adds the two partners' partial sums, which is commutative: all lanes end with the same bits. DPP row
//! permutations / permlane swaps of the two halves, no LDS crossbar (the `__shfl_xor` of a double is two ds_bpermute per step).
template <int T>
__device__ __forceinline__ double groupSumF64(double v)
{
#define LORAHIP_SUM_DPP(CTRL) { const unsigned lo_ = __builtin_amdgcn_mov_dpp((unsigned)__double2loint(v), CTRL, 0xf, 0xf, false), \
                                               hi_ = __builtin_amdgcn_mov_dpp((unsigned)__double2hiint(v), CTRL, 0xf, 0xf, false); \
                                v += __hiloint2double((int)hi_, (int)lo_); }
    if (T >= 2) LORAHIP_SUM_DPP(0xB1)       // quad_perm [1,0,3,2]
    if (T >= 4) LORAHIP_SUM_DPP(0x4E)       // quad_perm [2,3,0,1]
    if (T >= 8) LORAHIP_SUM_DPP(0x141)      // row_half_mirror: the other quad of the 8 holds one value in all its lanes
    if (T >= 16) LORAHIP_SUM_DPP(0x140)     // row_mirror
#undef LORAHIP_SUM_DPP
    if (T >= 32)
    {
        const v2u l = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        const v2u h = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        v = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    if (T >= 64)
    {
        const v2u l = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
        const v2u h = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
        v = __hiloint2double((int)h.x, (int)l.x) + __hiloint2double((int)h.y, (int)l.y);
    }
    return v;
}

//! sum of a float over the aligned group of T lanes, result in every lane (the same pairing tree as groupSumF64, half the moves)
template <int T>
__device__ __forceinline__ float groupSumF32(float v)
{
#define LORAHIP_SUM_DPP32(CTRL) v += __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(v), CTRL, 0xf, 0xf, false));
    if (T >= 2) LORAHIP_SUM_DPP32(0xB1)
    if (T >= 4) LORAHIP_SUM_DPP32(0x4E)
    if (T >= 8) LORAHIP_SUM_DPP32(0x141)
    if (T >= 16) LORAHIP_SUM_DPP32(0x140)
#undef LORAHIP_SUM_DPP32
    if (T >= 32) { const v2u r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r.x) + __uint_as_float(r.y); }
    if (T >= 64) { const v2u r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false); v = __uint_as_float(r.x) + __uint_as_float(r.y); }
    return v;
}

template <int T>
__device__ __forceinline__ void groupArgmax(float &bestV, int &bestI)
{
    const unsigned mine = __float_as_uint(bestV);
    const unsigned gm = groupReduceU<T, true>(mine);
    const unsigned cand = mine == gm ? (unsigned)bestI : 0x7fffffffu;
    bestI = (int)groupReduceU<T, false>(cand);
    bestV = __uint_as_float(gm);
}

/***********************************************************************
 * detect() tail for one window, executed by one lane (LoRaDetector.hpp:50-61)
 *
 * The reference evaluates log10f / hypotf in glibc; OCML's fp64 log10 (double-double inside, ~200
 * instructions) is far more than a float result needs. log10d / hypotd below are accurate to ~1e-14 /
 * ~1e-13 relative, i.e. they round to the same float as the exact value except when that value lies
 * within 2^-20 ulp of a float rounding boundary -- the same quality class as glibc's own float routines
 * (tests/: |power - ref| <= 2e-5 dB, |fIndex - ref| <= 2e-6).
 **********************************************************************/
//! log10 of a non-negative double (a widened float): x = m*2^e, m in [sqrt(1/2), sqrt 2),
//! ln m = 2 atanh f, f = (m-1)/(m+1), |f| <= 0.1716; series to f^15 (next term 3e-14 relative)
__device__ __forceinline__ double log10d(const double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lo = m < 0.70710678118654752440;
    m = lo ? m + m : m;
    e = lo ? e - 1 : e;
    const double f = (m - 1.0) / (m + 1.0);
    const double s = f * f;
    double p = 1.0 / 15.0;
    p = __builtin_fma(p, s, 1.0 / 13.0);
    p = __builtin_fma(p, s, 1.0 / 11.0);
    p = __builtin_fma(p, s, 1.0 / 9.0);
    p = __builtin_fma(p, s, 1.0 / 7.0);
    p = __builtin_fma(p, s, 1.0 / 5.0);
    p = __builtin_fma(p, s, 1.0 / 3.0);
    p = __builtin_fma(p, s, 1.0);
    const double lnm10 = (f * p) * 0.86858896380650365530;          // 2/ln(10)
    double r = __builtin_fma((double)e, 0.30102999566398119521, lnm10);
    r = x == 0.0 ? -__builtin_inf() : r;                                // log10(0) = -inf
    r = (x < 0.0 || x != x) ? __builtin_nan("") : r;                    // negative (rounded total - max) or NaN
    r = x == __builtin_inf() ? x : r;
    return r;
}
//! sqrt(a*a + b*b) of two floats in fp64: products exact, one rounding in the sum, rsq + one Newton step
__device__ __forceinline__ double hypotd(const float a, const float b)
{
    const double s = __builtin_fma((double)a, (double)a, (double)b * (double)b);
    const double r = __builtin_amdgcn_rsq(s);
    const double y0 = s * r;
    const double y1 = __builtin_fma(__builtin_fma(-y0, y0, s), 0.5 * r, y0);
    const double y2 = __builtin_fma(__builtin_fma(-y1, y1, s), 0.5 * r, y1);
    return (s == 0.0 || s != s || s == __builtin_inf()) ? s : y2;
}

//! LoRaDetector.hpp:50-61 for one window: power, powerAvg, fIndex from the peak, the fp64 total and the peak's neighbours
template <class CPX>
__device__ __forceinline__ void tailValues(const float powerScale, const float maxValue, const double total,
                                           const CPX leftBin, const CPX rightBin,
                                           float &power, float &powerAvg, float &fIndex)
{
    const float noise = sqrtf((float)(total - (double)maxValue));
    const float fundamental = sqrtf(maxValue);
    powerAvg = 20 * (float)log10d((double)noise) - powerScale;
    power = 20 * (float)log10d((double)fundamental) - powerScale;
    // std::abs(complex<float>) = hypotf
    const float left = (float)hypotd(leftBin.x, leftBin.y);
    const float right = (float)hypotd(rightBin.x, rightBin.y);
    const double demon = (2.0 * (double)fundamental) - (double)right - (double)left;
    fIndex = 0.0f;
    if (demon != 0.0) fIndex = (float)(0.5 * (double)(right - left) / demon);
}

//! The same values for a caller whose lanes hold the window's results REPLICATED (the streaming kernels: every lane of a
//! channel's group runs the frame machine): even and odd lanes each evaluate one of the two logarithms and one of the two
//! hypotenuses -- the same routines on the same operands, hence the same bits -- and swap with their neighbour (DPP quad_perm
//! [1,0,3,2]), instead of every lane evaluating all four.
template <class CPX>
__device__ __forceinline__ void tailValuesPaired(const float powerScale, const float maxValue, const double total,
                                                 const CPX leftBin, const CPX rightBin, const int lane,
                                                 float &power, float &powerAvg, float &fIndex)
{
    const bool odd = lane & 1;
    const float noise = sqrtf((float)(total - (double)maxValue));
    const float fundamental = sqrtf(maxValue);
    const float lgMine = (float)log10d((double)(odd ? noise : fundamental));
    const float hyMine = (float)hypotd(odd ? rightBin.x : leftBin.x, odd ? rightBin.y : leftBin.y);
    const float lgOther = __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(lgMine), 0xB1, 0xf, 0xf, false));
    const float hyOther = __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(hyMine), 0xB1, 0xf, 0xf, false));
    powerAvg = 20 * (odd ? lgMine : lgOther) - powerScale;
    power = 20 * (odd ? lgOther : lgMine) - powerScale;
    const float left = odd ? hyOther : hyMine, right = odd ? hyMine : hyOther;
    const double demon = (2.0 * (double)fundamental) - (double)right - (double)left;
    fIndex = 0.0f;
    if (demon != 0.0) fIndex = (float)(0.5 * (double)(right - left) / demon);
}

/*! LoRaDetector.hpp:36-48 over one lane's CNT bins, bin(j) in ascending bin order: as CH independent chains over consecutive
 * quarters. Each chain is the reference's own scan (`if (mag2 > maxValue)` from 0: a NaN is never taken, the first maximum wins);
 * the chains are merged in order with the same strict comparison -- the same winner, with a dependency depth of CNT/CH + 2
 * instead of CNT -- and the fp64 total is the sum of the chains' partial sums (like the cross-lane trees, a different association
 * of the same addends, each of them exact). Returns the winner's j. */
template <int CNT, int CHAINS, class BIN>
__device__ __forceinline__ int laneScan(BIN bin, float &bestV, double &tot)
{
    constexpr int CH = CNT >= 2 * CHAINS ? CHAINS : 1, PER = CNT / CH;
    float cv[CH];
    int cj[CH];
    double ct[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { cv[c] = 0.0f; cj[c] = 0; ct[c] = 0.0; }
#pragma unroll
    for (int k = 0; k < PER; k++)
#pragma unroll
        for (int c = 0; c < CH; c++)
        {
            const int j = c * PER + k;
            const auto b = bin(j);
#ifdef LORAHIP_FMA
            const float mag2 = __builtin_fmaf(b.x, b.x, b.y * b.y);
#else
            const float mag2 = b.x * b.x + b.y * b.y;
#endif
            ct[c] += (double)mag2;
            if (mag2 > cv[c]) { cv[c] = mag2; cj[c] = j; }
        }
#pragma unroll
    for (int w = 1; w < CH; w <<= 1)
#pragma unroll
        for (int c = 0; c + w < CH; c += 2 * w)
        {
            if (cv[c + w] > cv[c]) { cv[c] = cv[c + w]; cj[c] = cj[c + w]; }
            ct[c] += ct[c + w];
        }
    bestV = cv[0];
    tot = ct[0];
    return cj[0];
}

/*! laneScan with the total in fp32: the arg-max part is the reference's scan unchanged (same comparisons, same winner); the total is
 * only good for the streaming kernels' QUICK squelch estimate (squelchQuickF below), never for a value that leaves the kernel. */
#ifndef LORAHIP_QUICK_SCAN_CHAINS
#define LORAHIP_QUICK_SCAN_CHAINS 1
#endif
template <int CNT, class BIN>
__device__ __forceinline__ int laneScanQuick(BIN bin, float &bestV, float &totF)
{
    constexpr int CH = LORAHIP_QUICK_SCAN_CHAINS, PER = CNT / CH;
    float cv[CH], ct[CH];
    int cj[CH];
#pragma unroll
    for (int c = 0; c < CH; c++) { cv[c] = 0.0f; ct[c] = 0.0f; cj[c] = 0; }
#pragma unroll
    for (int k = 0; k < PER; k++)
#pragma unroll
        for (int c = 0; c < CH; c++)
        {
            const int j = c * PER + k;                      // chain c walks bins [c PER, (c + 1) PER) in ascending order
            const auto b = bin(j);
#ifdef LORAHIP_FMA
            const float mag2 = __builtin_fmaf(b.x, b.x, b.y * b.y);
#else
            const float mag2 = b.x * b.x + b.y * b.y;
#endif
            ct[c] += mag2;
            if (mag2 > cv[c]) { cv[c] = mag2; cj[c] = j; }
        }
#pragma unroll
    for (int w = 1; w < CH; w <<= 1)
#pragma unroll
        for (int c = 0; c + w < CH; c += 2 * w)
        {
            if (cv[c + w] > cv[c]) { cv[c] = cv[c + w]; cj[c] = cj[c + w]; }      // strict: the lower chain (lower bins) keeps a tie
            ct[c] += ct[c + w];
        }
    bestV = cv[0];
    totF = ct[0];
    return cj[0];
}

//! fIndex alone, for the same replicated lanes: the fIndex operations of tailValuesPaired (hence its bits) without the two
//! logarithms -- what FRAMESYNC consumes of an unsquelched window when no trace is kept (LoRaDemod.cpp:217-221)
template <class CPX>
__device__ __forceinline__ float fIndexPaired(const float maxValue, const CPX leftBin, const CPX rightBin, const int lane)
{
    const bool odd = lane & 1;
    const float fundamental = sqrtf(maxValue);
    const float hyMine = (float)hypotd(odd ? rightBin.x : leftBin.x, odd ? rightBin.y : leftBin.y);
    const float hyOther = __uint_as_float(__builtin_amdgcn_mov_dpp(__float_as_uint(hyMine), 0xB1, 0xf, 0xf, false));
    const float left = odd ? hyOther : hyMine, right = odd ? hyMine : hyOther;
    const double demon = (2.0 * (double)fundamental) - (double)right - (double)left;
    float fIndex = 0.0f;
    if (demon != 0.0) fIndex = (float)(0.5 * (double)(right - left) / demon);
    return fIndex;
}

/*! The squelch decision alone (LoRaDemod.cpp:173-174: `snr = power - powerAvg; squelched = snr < thresh`) without the tail: in
 * DATASYMBOLS nothing else of detect()'s float outputs is consumed (:286-306; fIndex only feeds the label), so the streaming
 * kernels skip the two logarithms, the two hypotenuses and the neighbour fetch there. snr is estimated as
 * 10 log10(maxValue / float(total - maxValue)) with the hardware log2 (error < 1e-4 dB against the exact float chain, whose own
 * roundings are ~3e-5 dB); `sure` is false within 0.01 dB of the threshold or for degenerate windows (zero peak, zero or
 * negative noise) -- the caller then evaluates the exact chain, so the decision is always the exact one. */
__device__ __forceinline__ bool squelchQuick(const float maxValue, const double total, const float thresh, bool &sure)
{
    const float noise2 = (float)(total - (double)maxValue);
    const float snrApprox = 3.01029995664f * (__builtin_amdgcn_logf(maxValue) - __builtin_amdgcn_logf(noise2));   // 10 log10 = 3.0103 log2
    sure = maxValue > 0.0f && noise2 > 0.0f && __builtin_fabsf(snrApprox - thresh) > 0.01f && snrApprox == snrApprox &&
           __builtin_fabsf(snrApprox) < 1e30f;
    return snrApprox < thresh;
}

/*! squelchQuick on a total that was accumulated in fp32 (laneScanQuick + groupSumF32: at most CNT + log2 T roundings, relative error
 * below `relErr`). noise2 = total - maxValue cancels when the peak dominates, so its relative error is relErr * (1 + maxValue /
 * noise2): the band around the threshold inside which the caller must evaluate the exact fp64 chain widens by exactly that (in dB:
 * 4.343 * relative error, doubled for slack) -- the decision taken outside the band is the exact one for any threshold the block
 * can be given. */
__device__ __forceinline__ bool squelchQuickF(const float maxValue, const float totalF, const float thresh, const float relErr, bool &sure)
{
    const float noise2 = totalF - maxValue;
    const float l2 = __builtin_amdgcn_logf(maxValue) - __builtin_amdgcn_logf(noise2);                              // log2(maxValue / noise2)
    const float snrApprox = 3.01029995664f * l2;                                                                    // 10 log10 = 3.0103 log2
    const float ratio = __builtin_amdgcn_exp2f(l2);                  // maxValue / noise2, to the accuracy a band width needs (one instruction, no division)
    const float band = 0.01f + 8.7f * relErr * (1.0f + ratio);
    sure = maxValue > 0.0f && noise2 > 0.0f && __builtin_fabsf(snrApprox - thresh) > band && snrApprox == snrApprox &&
           __builtin_fabsf(snrApprox) < 1e30f && band == band && band < 1e30f;
    return snrApprox < thresh;
}

template <class CPX>
__device__ __forceinline__ void detectTail(const DetectArgs &a, const unsigned w, const int maxIndex,
                                           const float maxValue, const double total,
                                           const CPX leftBin, const CPX rightBin)
{
    float power, powerAvg, fIndex;
    tailValues(a.powerScale, maxValue, total, leftBin, rightBin, power, powerAvg, fIndex);
    a.sym[w] = (unsigned short)maxIndex;
    a.power[w] = power;
    a.powerAvg[w] = powerAvg;
    a.fIndex[w] = fIndex;
}

} // namespace lorahip

// FFT building blocks shared by the tuned kernels (lorahip_fast.hip: a window inside one wavefront;
// lorahip_wide.hip: a window across the wavefronts of a workgroup). Numerics contract: lorahip_device.h.
#pragma once
#include "lorahip_device.h"
#include "lorahip_fine.h"

namespace lorahip {

/***********************************************************************
 * stage-major twiddle table: for every radix-4 stage with remainder m = 2^b,
 * [q-1][k] = kissfft twiddle(k * (N/(4m)) * q), k < m. Same VALUES as kissfft's table,
 * re-indexed so that lanes with consecutive k read consecutive LDS words.
 **********************************************************************/
__host__ __device__ constexpr int twStageOffset(const int log2n, const int b)
{
    int off = 0;
    for (int bb = (log2n & 1); bb < b; bb += 2) off += 3 << bb;
    return off;
}

//! reverse the radix-4 digits of an even-width bit string
__host__ __device__ constexpr int rev4(int x, const int bits)
{
    int r = 0;
    for (int i = 0; i < bits; i += 2) { r = (r << 2) | (x & 3); x >>= 2; }
    return r;
}

/***********************************************************************
 * one phase over one register group v[0..G): stages at bits [LO, HI)
 *   TWL  : LDS stage-major table (stages below the last phase)
 *   twR  : register twiddles of the last phase (slot order = stage, kl, q)
 *   klow : position bits below LO of this group (0 in phase 0)
 **********************************************************************/
template <int LOG2N, int LO, int HI, bool LAST>
__device__ __forceinline__ void runPhase(v2f (&v)[1 << (HI - LO)], const int klow,
                                         const v2f *__restrict__ TWL, const v2f *twR)
{
    constexpr int G = 1 << (HI - LO);
    constexpr bool R2 = (LO == 0) && (LOG2N & 1);
    if (R2)
    {
        // innermost radix-2 stage, m = 1: twiddle(0) = (1,0)
#pragma unroll
        for (int i = 0; i < G / 2; i++) bfly2unitv(v[2 * i], v[2 * i + 1]);
    }
    int slot = 0;
#pragma unroll
    for (int b = LO + (R2 ? 1 : 0); b < HI; b += 2)
    {
        const int sh = b - LO;                       // bit position of q inside the group index
        const int nkl = 1 << sh;                     // distinct k inside the group
#pragma unroll
        for (int kl = 0; kl < nkl; kl++)
        {
            v2f t1, t2, t3;
            const bool unit = (LO == 0) && (kl == 0);
            if (!unit)
            {
                if (LAST)
                {
                    t1 = twR[slot]; t2 = twR[slot + 1]; t3 = twR[slot + 2];
                }
                else
                {
                    const int k = klow + (kl << LO);
                    const int base = twStageOffset(LOG2N, b) + k;
                    t1 = TWL[base]; t2 = TWL[base + (1 << b)]; t3 = TWL[base + (2 << b)];
                }
            }
            slot += 3;
#pragma unroll
            for (int hi = 0; hi < (G >> (sh + 2)); hi++)
            {
                const int e0 = kl + (hi << (sh + 2));
                if (unit) bfly4unitv(v[e0], v[e0 + nkl], v[e0 + 2 * nkl], v[e0 + 3 * nkl]);
                else bfly4v(v[e0], v[e0 + nkl], v[e0 + 2 * nkl], v[e0 + 3 * nkl], t1, t2, t3);
            }
        }
    }
}

//! number of register twiddles (v2f) of the last phase for one group
template <int LOG2N, int LO, int HI>
__host__ __device__ constexpr int lastPhaseSlots()
{
    int s = 0;
    for (int b = LO; b < HI; b += 2) s += 3 << (b - LO);
    return s;
}

//! select element (e, g) with flat index idx = e*NG + g out of v[g][e] for a runtime idx: cndmask tree
template <int NG, int G>
__device__ __forceinline__ v2f selectReg(const v2f (&v)[NG][G], const int idx)
{
    constexpr int CNT = NG * G;
    v2f cur[CNT / 2];
    {
        const bool hi = idx & 1;
#pragma unroll
        for (int i = 0; i < CNT / 2; i++)
        {
            const v2f lo2 = v[(2 * i) % NG][(2 * i) / NG], hi2 = v[(2 * i + 1) % NG][(2 * i + 1) / NG];
            cur[i].x = hi ? hi2.x : lo2.x;
            cur[i].y = hi ? hi2.y : lo2.y;
        }
    }
#pragma unroll
    for (int w = CNT / 4, bit = 1; w >= 1; w >>= 1, bit++)
    {
        const bool hi = (idx >> bit) & 1;
#pragma unroll
        for (int i = 0; i < w; i++)
        {
            cur[i].x = hi ? cur[2 * i + 1].x : cur[2 * i].x;
            cur[i].y = hi ? cur[2 * i + 1].y : cur[2 * i].y;
        }
    }
    return cur[0];
}

//! v[idx] for a runtime per-lane idx: cndmask tree over a flat register array
template <int CNT>
__device__ __forceinline__ v2f selectFlat(const v2f (&v)[CNT], const int idx)
{
    v2f cur[CNT / 2];
    {
        const bool hi = idx & 1;
#pragma unroll
        for (int i = 0; i < CNT / 2; i++)
        {
            cur[i].x = hi ? v[2 * i + 1].x : v[2 * i].x;
            cur[i].y = hi ? v[2 * i + 1].y : v[2 * i].y;
        }
    }
#pragma unroll
    for (int w = CNT / 4, bit = 1; w >= 1; w >>= 1, bit++)
    {
        const bool hi = (idx >> bit) & 1;
#pragma unroll
        for (int i = 0; i < w; i++)
        {
            cur[i].x = hi ? cur[2 * i + 1].x : cur[2 * i].x;
            cur[i].y = hi ? cur[2 * i + 1].y : cur[2 * i].y;
        }
    }
    return cur[0];
}

/*! The fine-tune index recurrence (LoRaDemod.cpp:160-162), idx_{i+1} = f(idx_i), over T*P consecutive samples,
 * evaluated by a group of T lanes (T <= 64, aligned, inside one wavefront) and EXACTLY: lane t walks the P consecutive
 * samples [P*t, P*t+P) with the true step function from a guessed start; the guesses are then checked against the
 * predecessors' end values, corrected by the prefix sum of the mismatches (exact wherever f commutes with a shift, i.e.
 * away from wraps and float-exponent boundaries) and re-walked until every lane's start equals its predecessor's end.
 * The first lane with a wrong start is always repaired exactly, so T rounds bound the loop; 1-2 are typical. Every
 * lane of the wavefront must call it (groups with d == 0 converge at once). Writes the index of sample P*t+i to
 * sIdx[i*T + t] (lane-major: conflict-free) and returns the index after the T*P steps in every lane of the group. */
template <int T, int P, int M>
__device__ __forceinline__ int fineChainGroup(const int idx0, const float d, const int t, int *sIdx)
{
    const int first = fineStep(idx0, d, M);
    int c = first - idx0;                               // nominal step, wrap removed
    if (c > M / 2) c -= M;
    else if (c < -M / 2) c += M;
    int g = (idx0 + c * (P * t)) & (M - 1);            // |c*P*t| < 2^30; M is a power of two
    int loc[P];
    int e = 0;
    for (int round = 0; round <= T; round++)
    {
        int idx = g;
#pragma unroll
        for (int i = 0; i < P; i++) { loc[i] = idx; idx = fineStep(idx, d, M); }
        e = idx;
        const int prevE = __shfl_up(e, 1, T);
        int delta = t == 0 ? ((idx0 - g) & (M - 1)) : ((prevE - g) & (M - 1));
        if (!__any(delta != 0)) break;
        // inclusive prefix sum of the mismatches over the group's lanes (mod M)
#pragma unroll
        for (int off = 1; off < T; off <<= 1)
        {
            const int o = __shfl_up(delta, off, T);
            if (t >= off) delta += o;
        }
        g = (g + delta) & (M - 1);
    }
#pragma unroll
    for (int i = 0; i < P; i++) sIdx[i * T + t] = loc[i];
    return __shfl(e, T - 1, T);
}

#define MAKE2(X, Y) (v2f{(X), (Y)})

/***********************************************************************
 * The fine-tune multiplier without the table gather (lorahip_fine.h): split tables in LDS, closed-form indices.
 **********************************************************************/
//! LDS copies of the split tables (addresses fixed at compile time); split == false: gather from the table in HBM instead
struct FineLds { const double2 *A, *B; bool split; };

// A/B switches of the two steps VERDICT r5 item 7 asked to be BUILT and measured (profiles/r06/s*_ab_fine_*; both keep every bit):
//   LORAHIP_FINEB_SKEW   the B table in LDS skewed by one entry per 16 (entry i at i + i / 16): the 16 lanes of a ds_read_b128 pass whose
//                        B index runs in an EVEN step no longer fall on the same bank quads; two more address instructions per sample
//   LORAHIP_FINE_POW2    waves whose windows ALL have a negative or integer step (modulus M = 2^m): the closed-form indices by add + and
//                        instead of the add / subtract / min of the general modulus
//! position of B-table entry i in its LDS copy
__host__ __device__ constexpr unsigned fineBSlot(const unsigned i)
{
#ifdef LORAHIP_FINEB_SKEW
    return i + (i >> 4);
#else
    return i;
#endif
}
//! entries of the two split tables for N = 2^LOG2N
template <int LOG2N> struct FineDims
{
    static constexpr int LOG2M = LOG2N + 7, LH = fineSplitLog2H(LOG2N);
    static constexpr int NA = 1 << (LOG2M - LH), NB = 1 << LH;
    static constexpr int NB_LDS = int(fineBSlot(unsigned(NB - 1))) + 1;      // entries of the LDS copy of B (skewed: one gap per 16)
    static constexpr size_t BYTES = size_t(NA + NB_LDS) * sizeof(double2);
};

//! workgroup copy of the split tables into LDS (call before a __syncthreads())
template <int LOG2N>
__device__ __forceinline__ FineLds fineLoadLds(double2 *dst, const double2 *gA, const double2 *gB, const int tid, const int nThreads)
{
    typedef FineDims<LOG2N> D;
    FineLds f;
    f.A = dst; f.B = dst + D::NA; f.split = gA != nullptr;
    if (!f.split) return f;
    for (int i = tid; i < D::NA; i += nThreads) dst[i] = gA[i];
    for (int i = tid; i < D::NB; i += nThreads) dst[D::NA + fineBSlot(unsigned(i))] = gB[i];
    return f;
}

//! _fineTuneTable[y] from the split tables: one fp64 complex product, rounded to float (checked entry by entry on the host)
template <int LH>
__device__ __forceinline__ v2f fineEval(const unsigned y, const FineLds &s)
{
    const double2 a = s.A[y >> LH], b = s.B[fineBSlot(y & ((1u << LH) - 1u))];
    const double re = __builtin_fma(a.x, b.x, -(a.y * b.y));
    const double im = __builtin_fma(a.x, b.y, a.y * b.x);
    return v2f{(float)re, (float)im};
}

//! x[i] = (x[i] * chirp(i)) * _fineTuneTable[y[i]] for CNT samples (LoRaDemod.cpp:159), four at a time: the eight LDS reads of
//! a group (or its four table reads, where the split tables are not in use) are issued back to back, then the four fp64
//! products and the two complex multiplies per sample -- straight-line code, so the reads of one group overlap the arithmetic
//! of the previous one and at most four table entries are live. `chirp(i)` yields the chirp-table entry of sample i with the
//! window's conjugation applied; lanes with keep == false leave x untouched (windows fed already dechirped).
typedef double d2v __attribute__((ext_vector_type(2)));

//! LDS byte address of a pointer into shared memory
__device__ __forceinline__ unsigned ldsByteAddress(const void *p)
{
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

//! (i << 4) + base in one instruction (the compiler's own form of a 16-byte table index costs three)
__device__ __forceinline__ unsigned entryAddress16(const unsigned i, const unsigned base)
{
    unsigned r;
    asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(r) : "v"(i), "s"(base));
    return r;
}

//! the two split-table reads of one sample, issued without a wait (the caller waits with finePairWait)
template <int LH>
__device__ __forceinline__ void fineIssue(d2v &a, d2v &b, const unsigned y, const unsigned baseA, const unsigned baseB)
{
    asm volatile("ds_read_b128 %0, %1" : "=v"(a) : "v"(entryAddress16(y >> LH, baseA)));
    asm volatile("ds_read_b128 %0, %1" : "=v"(b) : "v"(entryAddress16(fineBSlot(y & ((1u << LH) - 1u)), baseB)));
}

//! CONJ: chirp(i) is the DOWN-chirp table's entry and the window wants the up-chirp table, its conjugate (LoRaDemod.cpp:103): the
//! conjugation rides on the multiply's sign modifiers (cmulConjv: the same products, the same roundings) instead of costing a
//! packed multiply by (1, -1) per sample
template <bool SELECT, bool CONJ, class CHIRP>
__device__ __forceinline__ void fineApply(v2f &x, CHIRP chirp, const int i, const d2v a, const d2v b, const bool keep)
{
    const double re = __builtin_fma(a.x, b.x, -(a.y * b.y));
    const double im = __builtin_fma(a.x, b.y, a.y * b.x);
    const v2f xc = CONJ ? cmulConjv(x, chirp(i)) : cmulv(x, chirp(i));
    const v2f v = cmulv(xc, v2f{(float)re, (float)im});
    x = (!SELECT || keep) ? v : x;
}

/*! (x0 * c0) * f0 and (x1 * c1) * f1 -- twelve packed operations -- as ONE asm statement with the two samples interleaved. The
 * compiler cannot see inside an asm statement and pads an s_nop after every one whose result is consumed by the next instruction;
 * written out like this there is no such boundary inside the pair (and by the compiler's own hazard rule for packed results --
 * a wait state when the very next instruction reads them -- none is needed: no result is read by its immediate successor except
 * the mulHi products, which that rule exempts). Same operations, same operands, same roundings as cmulv / cmulConjv. */
#ifdef LORAHIP_FMA
template <bool CONJ>
__device__ __forceinline__ void cmulPair(v2f &x0, v2f &x1, const v2f c0, const v2f c1, const v2f f0, const v2f f1)
{
    // the contracted build (lorahip_device.h): eight packed operations instead of twelve
    x0 = cmulv(CONJ ? cmulConjv(x0, c0) : cmulv(x0, c0), f0);
    x1 = cmulv(CONJ ? cmulConjv(x1, c1) : cmulv(x1, c1), f1);
}
#else
template <bool CONJ>
__device__ __forceinline__ void cmulPair(v2f &x0, v2f &x1, const v2f c0, const v2f c1, const v2f f0, const v2f f1)
{
    v2f p0, q0, p1, q1, r0, r1;
    if (CONJ)
        asm("v_pk_mul_f32 %0, %6, %8 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %6, %8 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
            "v_pk_mul_f32 %2, %7, %9 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %3, %7, %9 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
            "v_pk_add_f32 %4, %0, %1 neg_hi:[0,1]\n\t"
            "v_pk_add_f32 %5, %2, %3 neg_hi:[0,1]\n\t"
            "v_pk_mul_f32 %0, %4, %10 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %4, %10 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
            "v_pk_mul_f32 %2, %5, %11 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %3, %5, %11 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
            "v_pk_add_f32 %4, %0, %1 neg_lo:[0,1]\n\t"
            "v_pk_add_f32 %5, %2, %3 neg_lo:[0,1]"
            : "=&v"(p0), "=&v"(q0), "=&v"(p1), "=&v"(q1), "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1), "v"(c0), "v"(c1), "v"(f0), "v"(f1));
    else
        asm("v_pk_mul_f32 %0, %6, %8 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %6, %8 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
            "v_pk_mul_f32 %2, %7, %9 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %3, %7, %9 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
            "v_pk_add_f32 %4, %0, %1 neg_lo:[0,1]\n\t"
            "v_pk_add_f32 %5, %2, %3 neg_lo:[0,1]\n\t"
            "v_pk_mul_f32 %0, %4, %10 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %1, %4, %10 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
            "v_pk_mul_f32 %2, %5, %11 op_sel:[0,0] op_sel_hi:[1,0]\n\t"
            "v_pk_mul_f32 %3, %5, %11 op_sel:[1,1] op_sel_hi:[0,1]\n\t"
            "v_pk_add_f32 %4, %0, %1 neg_lo:[0,1]\n\t"
            "v_pk_add_f32 %5, %2, %3 neg_lo:[0,1]"
            : "=&v"(p0), "=&v"(q0), "=&v"(p1), "=&v"(q1), "=&v"(r0), "=&v"(r1) : "v"(x0), "v"(x1), "v"(c0), "v"(c1), "v"(f0), "v"(f1));
    x0 = r0; x1 = r1;
}
#endif

#ifndef LORAHIP_FINE_GROUP
#define LORAHIP_FINE_GROUP 2            // samples per pipeline step of dechirpFineSplit (A/B: 4 keeps twice the reads in flight)
#endif
//! wait until at most LATER of this wave's LDS reads are outstanding (in-order return): the 2 G values named here have arrived
template <int LATER, int G>
__device__ __forceinline__ void fineGroupWait(d2v (&a)[G], d2v (&b)[G])
{
    if constexpr (G == 2) asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]) : "n"(LATER));
    else asm volatile("s_waitcnt lgkmcnt(%8)" : "+v"(a[0]), "+v"(b[0]), "+v"(a[1]), "+v"(b[1]), "+v"(a[2]), "+v"(b[2]), "+v"(a[3]), "+v"(b[3]) : "n"(LATER));
}

//! the split-table path of dechirpFine as a software pipeline over groups of G samples: the 2 G LDS reads of the next group are in
//! flight while this group's fp64 products and complex multiplies issue. The reads and the waits are written out (inline asm)
//! because the compiler otherwise sinks every read to its use and waits for it at once.
template <int LH, int CNT, bool SELECT, bool CONJ, class CHIRP>
__device__ __forceinline__ void dechirpFineSplit(v2f *x, CHIRP chirp, const unsigned *y, const FineLds &s, const bool keep)
{
    constexpr int G = LORAHIP_FINE_GROUP;
    static_assert(G == 2 || G == 4, "pairs or quads");
    static_assert(CNT % G == 0, "whole groups");
    const unsigned baseA = __builtin_amdgcn_readfirstlane(ldsByteAddress(s.A)), baseB = __builtin_amdgcn_readfirstlane(ldsByteAddress(s.B));
    d2v a[2][G], b[2][G];
#pragma unroll
    for (int j = 0; j < G; j++) fineIssue<LH>(a[0][j], b[0][j], y[j], baseA, baseB);
#pragma unroll
    for (int i = 0; i < CNT; i += G)
    {
        const int cur = (i / G) & 1;
        if (i + G < CNT)
        {
#pragma unroll
            for (int j = 0; j < G; j++) fineIssue<LH>(a[cur ^ 1][j], b[cur ^ 1][j], y[i + G + j], baseA, baseB);
            fineGroupWait<2 * G, G>(a[cur], b[cur]);
        }
        else fineGroupWait<0, G>(a[cur], b[cur]);
#ifndef LORAHIP_NO_FUSED_CMUL
        if constexpr (!SELECT)
        {
#pragma unroll
            for (int j = 0; j < G; j += 2)
            {
                const d2v a0 = a[cur][j], b0 = b[cur][j], a1 = a[cur][j + 1], b1 = b[cur][j + 1];
                const v2f f0 = v2f{(float)__builtin_fma(a0.x, b0.x, -(a0.y * b0.y)), (float)__builtin_fma(a0.x, b0.y, a0.y * b0.x)};
                const v2f f1 = v2f{(float)__builtin_fma(a1.x, b1.x, -(a1.y * b1.y)), (float)__builtin_fma(a1.x, b1.y, a1.y * b1.x)};
                cmulPair<CONJ>(x[i + j], x[i + j + 1], chirp(i + j), chirp(i + j + 1), f0, f1);
            }
        }
        else
#endif
        {
#pragma unroll
            for (int j = 0; j < G; j++) fineApply<SELECT, CONJ>(x[i + j], chirp, i + j, a[cur][j], b[cur][j], keep);
        }
    }
}

template <int LH, int CNT, bool CONJ = false, class CHIRP>
__device__ __forceinline__ void dechirpFine(v2f *x, CHIRP chirp, const unsigned *y, const FineLds &s, const v2f *__restrict__ gFine, const bool keep)
{
    static_assert(CNT % 4 == 0, "four values per round");
    if (s.split)                                                // uniform over the launch
    {
        // no window of this wave fed already dechirped (the rule): no per-sample select
        if (__all(keep)) dechirpFineSplit<LH, CNT, false, CONJ>(x, chirp, y, s, keep);
        else dechirpFineSplit<LH, CNT, true, CONJ>(x, chirp, y, s, keep);
    }
    else
    {
#pragma unroll
        for (int i = 0; i < CNT; i += 4)
        {
            v2f f[4];
#pragma unroll
            for (int j = 0; j < 4; j++) f[j] = gFine[y[i + j]];
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                const v2f xc = CONJ ? cmulConjv(x[i + j], chirp(i + j)) : cmulv(x[i + j], chirp(i + j));
                const v2f v = cmulv(xc, f[j]);
                x[i + j] = keep ? v : x[i + j];
            }
        }
    }
}

//! closed-form fine-tune indices of a lane's own samples n = VEC*t + u + VEC*T*r (lorahip_fine.h); returns the largest one
template <int LOG2N, int VEC, int T, int R>
__device__ __forceinline__ unsigned fineLaneIndices(const int idx0, const FinePlan &p, const int t, unsigned (&y)[R][VEC])
{
    constexpr int LOG2M = LOG2N + 7;
#ifdef LORAHIP_FINE_POW2
    if (__all(p.mod == (1u << LOG2M) && !p.sat))
    {
        // every window of the wave steps by an integer or downwards: the modulus is M = 2^m and the wrap is a mask (never the value M)
        constexpr unsigned MM = (1u << LOG2M) - 1u;
        const unsigned Qp = (unsigned(VEC * T) * p.q) & MM;
#pragma unroll
        for (int u = 0; u < VEC; u++) y[0][u] = (unsigned(idx0) + __umul24(unsigned(VEC * t + u), p.q)) & MM;
        unsigned Qv = Qp;
#pragma unroll
        for (int w = 1; w < R; w <<= 1)
        {
#pragma unroll
            for (int r = 0; r < w && r + w < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++) y[r + w][u] = (y[r][u] + Qv) & MM;
            Qv = (Qv + Qv) & MM;
        }
        return 0u;
    }
#endif
    const unsigned Q = fineReduce(unsigned(VEC * T) * p.q, p, LOG2M);          // VEC*T*q < N*(M+1) < 2^32
    // y[r] = y[0] + r Q (mod M'): by doubling (Q, 2Q, 4Q, ...) instead of one long chain -- log2(R) dependent steps, not R
    unsigned ymax = 0;
#pragma unroll
    for (int u = 0; u < VEC; u++) y[0][u] = fineReduce(unsigned(idx0) + __umul24(unsigned(VEC * t + u), p.q), p, LOG2M);
    unsigned Qw = Q;
#pragma unroll
    for (int w = 1; w < R; w <<= 1)
    {
#pragma unroll
        for (int r = 0; r < w && r + w < R; r++)
#pragma unroll
            for (int u = 0; u < VEC; u++) y[r + w][u] = fineAdvance(y[r][u], Qw, p);
        Qw = fineAdvance(Qw, Qw, p);
    }
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int u = 0; u < VEC; u++) ymax = y[r][u] > ymax ? y[r][u] : ymax;
    if (__any(p.sat))
    {
        // windows whose index walks down by one per sample and stays at 0 (lorahip_fine.h): the modular form is right until it
        // wraps, i.e. for the samples n <= idx0
#pragma unroll
        for (int u = 0; u < VEC; u++)
#pragma unroll
            for (int r = 0; r < R; r++)
                if (p.sat && VEC * t + u + VEC * T * r > idx0) y[r][u] = 0;
        if (p.sat) ymax = 0;                                                    // never the value M
    }
    return ymax;
}

struct TailRec { unsigned w[64]; int idx[64]; float val[64]; double tot[64]; v2f l[64]; v2f r[64]; };

} // namespace lorahip

// Tuned CDNA4 kernels of the demod hot path: register-resident FFT phases, LDS exchanges.
//
// Layout idea. The reference's FFT is decimation in time: its innermost
// (first executed) stages combine samples that are FAR apart in the window (n, n+N/2, n+N/4 ..),
// its outermost stage combines neighbours. A thread therefore loads the window the way HBM
// likes it -- VEC consecutive samples (8 or 16 B) per lane, the T lanes of a window covering
// 64..512 contiguous bytes, repeated at a stride of VEC*T samples -- and finds that the R =
// N/(VEC*T) values it got per column are exactly one group of the first log2(R) stage bits:
// phase 0 runs entirely in its registers. One transposition through LDS (8 B written + 8 B
// read per sample, rows padded by one element so both sides are bank-conflict free) regroups
// the window for the remaining stages; with 3 phases there are two such exchanges. T <= 64
// keeps a window inside one wavefront, so exchanges need no barrier; several windows share a
// wave (64/T). The |X|^2 scan, the lowest-index arg-max and the fp64 total are reduced with
// wave shuffles; the log/sqrt tail is deferred and executed once per 64 windows with all 64
// lanes busy.
//
// Work-array algebra (B = log2 N, phase j owns position bits [b_j, b_{j+1})):
//   sample n            -> position pos(n) = mixed-radix digit reversal (Plan<>::pos)
//   phase 0, lane t     : n = VEC*t + u + VEC*T*a;  pos low bits  = rev(a)  = e0  (in registers)
//                                                    pos high bits = rev(VEC*t+u)
//   phase j>=1          : ci = t + T*g;  klow = ci mod 2^b_j;  high = ci >> b_j
//                         element e  <->  pos = klow + 2^b_j * e + 2^b_{j+1} * high
//   radix-4 stage at bit b (remainder m = 2^b): butterflies over position bits [b, b+2),
//   twiddle(k*fstride*q) with k = pos mod m, fstride = N/(4m)   (kissfft.hh:137-157)
#include "lorahip_device.h"

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

std::vector<cf32> buildStageTwiddles(const int sf, const std::vector<cf32> &tw)
{
    std::vector<cf32> out;
    for (int b = (sf & 1); b + 2 <= sf; b += 2)
    {
        const int m = 1 << b, fs = 1 << (sf - 2 - b);
        for (int q = 1; q <= 3; q++)
            for (int k = 0; k < m; k++) out.push_back(tw[size_t(k) * fs * q]);
    }
    return out;
}

/***********************************************************************
 * compile-time configuration of one kernel instance
 **********************************************************************/
template <int LOG2N_, int LOG2T_, int VEC_, int NPH_, int PB1_, int PB2_, int WAVES_PER_SIMD_>
struct FastCfg
{
    static constexpr int LOG2N = LOG2N_, N = 1 << LOG2N_;
    static constexpr int LOG2T = LOG2T_, T = 1 << LOG2T_;     // lanes per window (<= 64)
    static constexpr int VEC = VEC_;                            // consecutive samples per lane per load
    static constexpr int P = N / T;                             // points per lane
    static constexpr int R = P / VEC;                           // phase-0 group size
    static constexpr int NPH = NPH_;
    static constexpr int WPW = 64 / T;                          // windows per wave iteration
    static constexpr int WAVES_PER_SIMD = WAVES_PER_SIMD_;
    static constexpr bool HAS_R2 = (LOG2N_ & 1);
    __host__ __device__ static constexpr int bound(const int j)
    {
        return j <= 0 ? 0 : (j == 1 ? PB1_ : (j == 2 ? (NPH_ == 2 ? LOG2N_ : PB2_) : LOG2N_));
    }
    static_assert((1 << PB1_) == R, "phase 0 must cover exactly the bits a lane loads");
    static_assert(((LOG2N_ - PB1_) & 1) == 0, "the low sample digits must be whole radix-4 digits");
    static_assert(T <= 64 && T >= 2, "a window lives inside one wavefront");
    //! LDS elements (float2) one window needs for its exchanges: rows of G_j (+1 pad) elements
    __host__ __device__ static constexpr int exchElems()
    {
        int best = 0;
        for (int j = 0; j + 1 < NPH_; j++)
        {
            const int g = 1 << (bound(j + 1) - bound(j));
            const int e = (N / g) * (g + 1);
            best = e > best ? e : best;
        }
        return best;
    }
    //! twiddle entries staged in LDS: all stages below the last phase
    static constexpr int TW_LDS = twStageOffset(LOG2N_, bound(NPH_ - 1));
};

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
__device__ __forceinline__ void runPhase(float2 (&v)[1 << (HI - LO)], const int klow,
                                         const float2 *__restrict__ TWL, const float2 *twR)
{
    constexpr int G = 1 << (HI - LO);
    constexpr bool R2 = (LO == 0) && (LOG2N & 1);
    if (R2)
    {
        // innermost radix-2 stage, m = 1: twiddle(0) = (1,0)
#pragma unroll
        for (int i = 0; i < G / 2; i++) bfly2unit(v[2 * i], v[2 * i + 1]);
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
            float2 t1, t2, t3;
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
                if (unit) bfly4unit(v[e0], v[e0 + nkl], v[e0 + 2 * nkl], v[e0 + 3 * nkl]);
                else bfly4(v[e0], v[e0 + nkl], v[e0 + 2 * nkl], v[e0 + 3 * nkl], t1, t2, t3);
            }
        }
    }
}

//! number of register twiddles (float2) of the last phase for one group
template <int LOG2N, int LO, int HI>
__host__ __device__ constexpr int lastPhaseSlots()
{
    int s = 0;
    for (int b = LO; b < HI; b += 2) s += 3 << (b - LO);
    return s;
}

//! select v[idx] for a runtime idx in [0, CNT) with a cndmask tree (CNT a power of two)
template <int CNT>
__device__ __forceinline__ float2 selectReg(const float2 (&v)[CNT], const int idx)
{
    float2 cur[CNT];
#pragma unroll
    for (int i = 0; i < CNT; i++) cur[i] = v[i];
#pragma unroll
    for (int w = CNT / 2, bit = 0; w >= 1; w >>= 1, bit++)
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

/***********************************************************************
 * the kernel: one wave = WPW windows per iteration, persistent over window sets
 **********************************************************************/
template <class C>
__global__ void __launch_bounds__(256, C::WAVES_PER_SIMD)
detectFast(const DetectArgs a, const FastTables ft, const unsigned nSets)
{
    constexpr int N = C::N, T = C::T, VEC = C::VEC, P = C::P, R = C::R, NPH = C::NPH, WPW = C::WPW;
    constexpr int LOG2N = C::LOG2N, LOG2T = C::LOG2T;
    constexpr int B1 = C::bound(1), B2 = C::bound(2);
    constexpr int BL = C::bound(NPH - 1);                 // first bit of the last phase
    constexpr int GL = 1 << (LOG2N - BL);                 // last-phase group size
    constexpr int NGL = P / GL;                           // last-phase groups per lane
    constexpr int SLOTS = lastPhaseSlots<LOG2N, BL, LOG2N>();
    constexpr int EXCH = C::exchElems();                  // float2 per window
    constexpr int WS = EXCH + 4;                          // window stride in LDS (float2), de-phases windows
    constexpr int M = N * LORAHIP_FINE_STEPS;
    constexpr int WAVES = 4;

    extern __shared__ __attribute__((aligned(16))) char smemRaw[];
    float2 *sTw = reinterpret_cast<float2 *>(smemRaw);                                   // [TW_LDS]
    float2 *sX = sTw + ((C::TW_LDS + 1) & ~1);                                           // [WAVES][WPW][WS]
    char *sTail = reinterpret_cast<char *>(sX + WAVES * WPW * WS);                       // [WAVES] tail records

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wsub = lane >> LOG2T;                       // window inside the wave iteration
    const int t = lane & (T - 1);
    float2 *X = sX + (wave * WPW + wsub) * WS;            // this window's exchange region

    // tail records of this wave (one slot per lane)
    struct TailRec { unsigned w[64]; int idx[64]; float val[64]; double tot[64]; float2 l[64]; float2 r[64]; };
    TailRec &tr = reinterpret_cast<TailRec *>(sTail)[wave];

    // ---- one-time set-up -------------------------------------------------------------
    tr.w[lane] = 0xffffffffu;                              // empty tail slots
    for (int i = threadIdx.x; i < C::TW_LDS; i += blockDim.x) sTw[i] = ft.twStage[i];

    // register twiddles of the last phase: group g has klow = (t + T*g) mod 2^BL ... for the last
    // phase BL bits are all below -> klow = ci
    float2 twR[NGL][SLOTS];
#pragma unroll
    for (int g = 0; g < NGL; g++)
    {
        const int ci = t + T * g;
        int slot = 0;
#pragma unroll
        for (int b = BL; b < LOG2N; b += 2)
#pragma unroll
            for (int kl = 0; kl < (1 << (b - BL)); kl++)
            {
                const int k = ci + (kl << BL);
                const int base = twStageOffset(LOG2N, b) + k;
                twR[g][slot] = ft.twStage[base];
                twR[g][slot + 1] = ft.twStage[base + (1 << b)];
                twR[g][slot + 2] = ft.twStage[base + (2 << b)];
                slot += 3;
            }
    }

    // chirp table values of this lane's sample positions (down table; up = conj, LoRaDemod.cpp:103-104)
    float2 ch[R][VEC];
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int u = 0; u < VEC; u++) ch[r][u] = a.down[VEC * t + u + VEC * T * r];
    __syncthreads();

    const unsigned waveId = blockIdx.x * WAVES + wave;
    const unsigned waveCount = gridDim.x * WAVES;
    int pending = 0;                                       // tail records waiting in tr

    for (unsigned set = waveId; set < nSets; set += waveCount)
    {
        const unsigned w = set * WPW + wsub;
        const bool active = w < a.nWindows;
        const unsigned wc = active ? w : a.nWindows - 1;  // clamp: inactive lanes redo the last window, results dropped
        const int sel = a.chirpSel ? a.chirpSel[wc] : a.chirpSelAll;
        const int idx0 = a.fineIdx0 ? a.fineIdx0[wc] : 0;
        const float err = a.fineErr ? a.fineErr[wc] : 0.0f;
        const float2 *in = a.iq + (a.offsets ? a.offsets[wc] : (long long)wc * a.stride);
        const bool dechirp = sel != LORAHIP_CHIRP_NONE;
        const float d = err * (float)LORAHIP_FINE_STEPS;
        const bool moving = dechirp && d != 0.0f;

        // ---- load (coalesced: VEC*8 bytes per lane, T lanes contiguous) ---------------------
        float2 x[R][VEC];
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            const float2 *p = in + VEC * t + VEC * T * r;
            if (VEC == 2)
            {
                const float4 q = *reinterpret_cast<const float4 *>(p);
                x[r][0] = make_float2(q.x, q.y);
                x[r][VEC - 1] = make_float2(q.z, q.w);
            }
            else x[r][0] = *p;
        }

        // ---- fine-tune index chain for windows whose index moves (rare path) -----------------
        int *sIdx = reinterpret_cast<int *>(X);            // aliases the exchange region (free until phase 0 ends)
        if (__any(moving))
        {
            if (moving && t == 0)
            {
                int idx = idx0;
                for (int i = 0; i < N; i++) { sIdx[i] = idx; idx = fineStep(idx, d, M); }
                if (a.fineIdxOut && active) a.fineIdxOut[w] = idx;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (!moving && t == 0 && active && a.fineIdxOut) a.fineIdxOut[w] = idx0;

        // ---- dechirp: (samp * chirp) * fine   (LoRaDemod.cpp:159) ---------------------------
        const float2 fconst = a.fine[idx0];
        const float sgn = sel == LORAHIP_CHIRP_UP ? -1.0f : 1.0f;   // up table = conj(down table)
        if (__any(moving))
        {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++)
                {
                    const float2 c = make_float2(ch[r][u].x, sgn * ch[r][u].y);
                    const float2 f = moving ? a.fine[sIdx[VEC * t + u + VEC * T * r]] : fconst;
                    const float2 y = cmul(cmul(x[r][u], c), f);
                    x[r][u] = dechirp ? y : x[r][u];
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        else
        {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++)
                {
                    const float2 c = make_float2(ch[r][u].x, sgn * ch[r][u].y);
                    const float2 y = cmul(cmul(x[r][u], c), fconst);
                    x[r][u] = dechirp ? y : x[r][u];
                }
        }
        if (a.decOut && active)
        {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++) a.decOut[(size_t)w * N + VEC * t + u + VEC * T * r] = x[r][u];
        }

        // ---- phase 0: bits [0, B1) in registers, one group per u -----------------------------
        // register r holds sample index high part a = r; its work-array position low bits are rev(a)
        float2 v0[VEC][R];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < VEC; u++) v0[u][Plan<LOG2N>::pos(VEC * T * r) & (R - 1)] = x[r][u];
#pragma unroll
        for (int u = 0; u < VEC; u++) runPhase<LOG2N, 0, B1, false>(v0[u], 0, sTw, nullptr);

        // ---- exchange 0: rows = n_low = VEC*t+u (writer order), R (+1 pad) columns ------------
#pragma unroll
        for (int u = 0; u < VEC; u++)
#pragma unroll
            for (int e = 0; e < R; e++) X[(VEC * t + u) * (R + 1) + e] = v0[u][e];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        float2 vl[NGL][GL];                                // last-phase registers
        if (NPH == 2)
        {
            // phase 1 = last: group g has klow = ci = t + T*g (< R), elements e <-> hp = e, row = rev4(hp)
#pragma unroll
            for (int g = 0; g < NGL; g++)
#pragma unroll
                for (int e = 0; e < GL; e++) vl[g][e] = X[rev4(e, LOG2N - B1) * (R + 1) + (t + T * g)];
        }
        else
        {
            // phase 1 (middle): bits [B1, B2); ci = t + T*g; klow = ci mod R; high = ci >> B1
            constexpr int G1 = 1 << (B2 - B1);
            constexpr int NG1 = P / G1;
            constexpr int HB = LOG2N - B2;                 // bits of `high`
            float2 v1[NG1][G1];
#pragma unroll
            for (int g = 0; g < NG1; g++)
            {
                const int ci = t + T * g;
                const int klow = ci & (R - 1), high = ci >> B1;
                // hp = e + G1*high; row = rev4(hp) = rev4(e) << HB | rev4(high)
                const int rhigh = rev4(high, HB);
#pragma unroll
                for (int e = 0; e < G1; e++) v1[g][e] = X[((rev4(e, B2 - B1) << HB) | rhigh) * (R + 1) + klow];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < NG1; g++) runPhase<LOG2N, B1, B2, false>(v1[g], (t + T * g) & (R - 1), sTw, nullptr);
            // exchange 1: rows = ci (writer order), G1 (+1) columns
#pragma unroll
            for (int g = 0; g < NG1; g++)
#pragma unroll
                for (int e = 0; e < G1; e++) X[(t + T * g) * (G1 + 1) + e] = v1[g][e];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // phase 2 = last: ci' = t + T*g < 2^B2; element e2 <-> writer row (ci' mod R) + R*e2, column ci' >> B1
#pragma unroll
            for (int g = 0; g < NGL; g++)
            {
                const int ci = t + T * g;
#pragma unroll
                for (int e = 0; e < GL; e++) vl[g][e] = X[((ci & (R - 1)) + R * e) * (G1 + 1) + (ci >> B1)];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < NGL; g++) runPhase<LOG2N, BL, LOG2N, true>(vl[g], 0, nullptr, twR[g]);

        // ---- scan (LoRaDetector.hpp:36-48): bin = ci + 2^BL * e, ascending in (e, g) -----------
        float bestV = 0.0f;
        int bestI = 0;
        double tot = 0.0;
#pragma unroll
        for (int e = 0; e < GL; e++)
#pragma unroll
            for (int g = 0; g < NGL; g++)
            {
                const float2 bin = vl[g][e];
                const int i = (t + T * g) + (e << BL);
                if (a.fftOut && active) a.fftOut[(size_t)w * N + i] = bin;
                const float mag2 = bin.x * bin.x + bin.y * bin.y;
                tot += (double)mag2;
                if (mag2 > bestV) { bestV = mag2; bestI = i; }
            }
        if (!(bestV > 0.0f)) bestI = 0;
#pragma unroll
        for (int off = T / 2; off > 0; off >>= 1)
        {
            const float ov = __shfl_xor(bestV, off, 64);
            const int oi = __shfl_xor(bestI, off, 64);
            const double ot = __shfl_xor(tot, off, 64);
            argmaxCombine(bestV, bestI, ov, oi);
            tot += ot;
        }
        // every lane of the window now holds the window's (bestV, bestI); the xor tree adds the same
        // fp64 partials in the same pairing on all lanes, so tot is identical on all of them too

        // ---- neighbours of the peak for fIndex (LoRaDetector.hpp:56-57) ------------------------
        const int bl = (bestI + N - 1) & (N - 1), br = (bestI + 1) & (N - 1);
        const int cil = bl & ((1 << BL) - 1), cir = br & ((1 << BL) - 1);
        const bool ownL = (cil & (T - 1)) == t;
        const int req = ownL ? ((bl >> BL) * NGL + (cil >> LOG2T)) : ((br >> BL) * NGL + (cir >> LOG2T));
        float2 flat[P];
#pragma unroll
        for (int e = 0; e < GL; e++)
#pragma unroll
            for (int g = 0; g < NGL; g++) flat[e * NGL + g] = vl[g][e];
        const float2 mine = selectReg<P>(flat, req);
        const int base = lane & ~(T - 1);
        const float2 leftBin = make_float2(__shfl(mine.x, base + (cil & (T - 1)), 64), __shfl(mine.y, base + (cil & (T - 1)), 64));
        const float2 rightBin = make_float2(__shfl(mine.x, base + (cir & (T - 1)), 64), __shfl(mine.y, base + (cir & (T - 1)), 64));

        // ---- defer the log/sqrt tail: one record per window, flushed 64 at a time ---------------
        if (t == 0 && active)
        {
            const int s = pending + wsub;
            tr.w[s] = w; tr.idx[s] = bestI; tr.val[s] = bestV; tr.tot[s] = tot; tr.l[s] = leftBin; tr.r[s] = rightBin;
        }
        pending += WPW;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (pending == 64)
        {
            const int sl = lane;
            const unsigned ww = tr.w[sl];
            // slots of inactive windows were never written this round: mark by w >= nWindows
            if (ww < a.nWindows) detectTail(a, ww, tr.idx[sl], tr.val[sl], tr.tot[sl], tr.l[sl], tr.r[sl]);
            pending = 0;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            tr.w[lane] = 0xffffffffu;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    // flush what is left
    if (pending > 0)
    {
        const unsigned ww = lane < pending ? tr.w[lane] : 0xffffffffu;
        if (ww < a.nWindows) detectTail(a, ww, tr.idx[lane], tr.val[lane], tr.tot[lane], tr.l[lane], tr.r[lane]);
    }
}

/***********************************************************************
 * launch
 **********************************************************************/
template <class C>
static hipError_t launchCfg(const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    constexpr int WAVES = 4;
    constexpr int WS = C::exchElems() + 4;
    const size_t smem = size_t((C::TW_LDS + 1) & ~1) * sizeof(float2) + size_t(WAVES) * C::WPW * WS * sizeof(float2)
                      + size_t(WAVES) * (64 * (4 + 4 + 4 + 8 + 8 + 8));
    static bool attrSet = false;
    if (!attrSet)
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(detectFast<C>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, int(smem));
        if (e != hipSuccess) return e;
        attrSet = true;
    }
    const unsigned nSets = (a.nWindows + C::WPW - 1) / C::WPW;
    // persistent: as many blocks as stay resident, never more than there are sets of work
    const unsigned resident = unsigned(ft.nBlocksHint > 0 ? ft.nBlocksHint : 256) * unsigned(C::WAVES_PER_SIMD);
    unsigned grid = (nSets + WAVES - 1) / WAVES;
    if (grid > resident) grid = resident;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(detectFast<C>, dim3(grid), dim3(WAVES * 64), smem, stream, a, ft, nSets);
    return hipGetLastError();
}

//                LOG2N T   VEC NPH PB1 PB2 waves/SIMD
typedef FastCfg<7,  3,  2,  2,  3,  7,  2> Cfg7;      // 8 lanes x 16 points : [R2,4] X [4,4]
typedef FastCfg<8,  3,  2,  2,  4,  8,  2> Cfg8;      // 8 lanes x 32 points : [4,4] X [4,4]
typedef FastCfg<9,  5,  2,  3,  3,  7,  2> Cfg9;      // 32 lanes x 16 points: [R2,4] X [4,4] X [4]
typedef FastCfg<10, 5,  2,  3,  4,  8,  2> Cfg10;     // 32 lanes x 32 points: [4,4] X [4,4] X [4]

bool fastAvailable(const int sf) { return sf >= 7 && sf <= 10; }

hipError_t launchFast(const int sf, const int variant, const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    (void)variant;
    switch (sf)
    {
    case 7: return launchCfg<Cfg7>(a, ft, stream);
    case 8: return launchCfg<Cfg8>(a, ft, stream);
    case 9: return launchCfg<Cfg9>(a, ft, stream);
    case 10: return launchCfg<Cfg10>(a, ft, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace lorahip

// Tuned CDNA4 kernels for the long windows (SF11, SF12): one window spread over the wavefronts of a
// 256-thread workgroup, 16 points per lane.
//
// Same algebra as lorahip_fast.hip (three register phases of kissfft's DIT graph, two transpositions
// through LDS), but T = N/16 lanes per window is 128 (SF11: two windows per workgroup) or 256 (SF12: one),
// so the exchanges are fenced with workgroup barriers and the |X|^2 arg-max / fp64 total is finished across
// wavefronts through LDS. Barrier budget per window set: 4 (after each exchange write, and before the
// region is overwritten by the next exchange); the cross-wave reduction shares the last one and the
// read-back of the peak's neighbours is deferred behind the first barrier of the NEXT set, as is the
// sqrt/log tail (queued in LDS, executed 64 windows at a time by wave 0 with all lanes busy).
//
//   phase 0  bits [0,B1)      lane t holds samples n = VEC*t+u + (N/R)*r, r < R = 16/VEC     (registers)
//   exch 0   row per n_low = VEC*t+u (R elements, padded/rotated so writes and reads tile the banks)
//   phase 1  bits [B1,B1+4)   lane t: klow = t mod R, high = t / R, element e = position bits [B1,B1+4)
//   exch 1   natural position order, 8 elements of padding per 2^(B1+4) block
//   phase 2  bits [B1+4,LOG2N) lane t holds bins t + T*e                                       (registers)
#include "lorahip_fft.h"
#include "lorahip_framemachine.h"
#include "lorahip_residentproto.h"

#ifndef SCAN_CHAINS_WIDE
#define SCAN_CHAINS_WIDE 1
#endif
#ifndef STREAM_WIDE_PREFETCH
#define STREAM_WIDE_PREFETCH 0  // 1: demodStreamWide requests the window at off + N while this one is transformed. Measured 2-3 % SLOWER (profiles/r04/s10_*): kept as the A/B
#endif
namespace lorahip {

template <int LOG2N_, int VEC_, int MINW_, int X0ROT_, int X0PAD_, int X0S_, int X0D_, bool CH_LDS_, bool TW_ALL_LDS_, bool PREFETCH_ = true, bool NT_ = false, int WPB_ = 0,
          bool INPLACE_ = false, int X0S2_ = 0, int X0D2_ = 0>
struct WideCfg
{
    static constexpr int LOG2N = LOG2N_, N = 1 << LOG2N_;
    static constexpr int P = 16;                               // points per lane
    static constexpr int LOG2T = LOG2N_ - 4, T = 1 << LOG2T;    // lanes per window
    static constexpr int VEC = VEC_, R = P / VEC_;
    static constexpr int B1 = (VEC_ == 1 ? 4 : 3), B2 = B1 + 4;
    static constexpr int WPB = WPB_ ? WPB_ : 256 / T;           // windows per workgroup iteration
    static constexpr int BLOCK = WPB * T;                       // threads per workgroup (128 or 256)
    static constexpr int BPC = MINW_ * 256 / BLOCK;             // workgroups per CU at MINW waves per SIMD
    static constexpr int WPWIN = T / 64;                        // wavefronts per window
    static constexpr int MINW = MINW_;
    static constexpr bool CH_LDS = CH_LDS_, TW_ALL_LDS = TW_ALL_LDS_;
    static constexpr bool NT = NT_;                             // non-temporal hint on the IQ loads
    static constexpr bool INPLACE = INPLACE_;                   // the middle phase writes its results back where it read its inputs:
                                                                // no barrier between the two, the last phase reads the exchange-0 layout
    static constexpr bool PREFETCH = PREFETCH_;                 // next set's samples in registers during this set (else: loaded at the top, hidden by co-resident workgroups)
    static constexpr int HB = LOG2N_ - B2;                      // = 4: position bits of the last phase
    static constexpr int NL = VEC_ * T, LOG2NL = LOG2N_ - B1;   // rows of exchange 0
    static_assert(B2 + 4 == LOG2N_, "VEC 1 <-> SF12, VEC 2 <-> SF11");
    static_assert(T >= 64 && T <= 256, "a window is 1..4 wavefronts");
    static constexpr int RS0 = R + X0PAD_;
    __host__ __device__ static constexpr int x0off(const int nlow)
    {
        const int rot = ((nlow >> X0ROT_) | (nlow << (LOG2NL - X0ROT_))) & (NL - 1);
        return rot * RS0 + ((nlow >> X0S_) & 1) * X0D_ + ((nlow >> X0S2_) & 1) * X0D2_;
    }
    static constexpr int X0ELEMS = NL * RS0 + X0D_ + X0D2_;
    static constexpr int X1 = 16 * R + 8;                       // one block of `high` in exchange 1
    static constexpr int X1ELEMS = INPLACE_ ? 0 : 16 * X1;
    static constexpr int XE = X0ELEMS > X1ELEMS ? X0ELEMS : X1ELEMS;
    static constexpr int XW = ((XE * 2 > N ? XE : (N + 1) / 2) + 1) & ~1;   // v2f per window; also holds N ints
    static constexpr int TW_LDS = twStageOffset(LOG2N_, TW_ALL_LDS_ ? LOG2N_ : B2);
    static constexpr int TWN = (TW_LDS + 1) & ~1;
    static constexpr int CH_ELEMS = CH_LDS_ ? N : 0;
};

struct RedRec { float v; int i; double tot; };

//! where fineChainGroup<64,16> (1024-sample chunks) leaves the index of sample n
__device__ __forceinline__ int chainSlot(const int n) { return (n & ~1023) + (n & 15) * 64 + ((n >> 4) & 63); }

/*! The fine-tune index recurrence of a window that spans WPWIN wavefronts (fineChainGroup's scheme with the hand-over
 * between wavefronts through LDS): every lane walks its 16 consecutive samples from a guessed start, the ends are
 * compared with the successors' starts, the guesses repaired by the prefix sum of the mismatches, until all agree.
 * Called by every thread of the workgroup (workgroup barriers inside); t = lane inside the window, wwin = wavefront
 * inside the window, sC = 12 ints of LDS scratch per window. Leaves the index of sample n at sIdx[chainSlot(n)] and
 * returns the index after the N steps (to every thread of the window). */
template <int T, int M>
__device__ __forceinline__ int fineChainBlock(const int idx0, const float d, const int t, const int wwin, int *sIdx, int *sC)
{
    constexpr int WPWIN = T / 64;
    const int lane = t & 63;
    const int first = fineStep(idx0, d, M);
    int c = first - idx0;
    if (c > M / 2) c -= M;
    else if (c < -M / 2) c += M;
    int g = (idx0 + c * (16 * t)) & (M - 1);
    int loc[16];
    int e = 0;
    for (int round = 0; round <= T; round++)
    {
        int idx = g;
#pragma unroll
        for (int i = 0; i < 16; i++) { loc[i] = idx; idx = fineStep(idx, d, M); }
        e = idx;
        if (lane == 63) sC[wwin] = e;                       // hand-over to the next wavefront
        __syncthreads();
        const int up = __shfl_up(e, 1, 64);
        const int prevE = lane > 0 ? up : (wwin == 0 ? idx0 : sC[wwin - 1]);
        int delta = (prevE - g) & (M - 1);
        if (!__syncthreads_or(delta != 0)) break;
        // inclusive prefix sum of the mismatches over the window's lanes (mod M)
#pragma unroll
        for (int off = 1; off < 64; off <<= 1)
        {
            const int o = __shfl_up(delta, off, 64);
            if (lane >= off) delta += o;
        }
        if (lane == 63) sC[4 + wwin] = delta;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < WPWIN - 1; k++) if (k < wwin) delta += sC[4 + k];
        g = (g + delta) & (M - 1);
    }
#pragma unroll
    for (int i = 0; i < 16; i++) sIdx[wwin * 1024 + i * 64 + lane] = loc[i];
    if (t == T - 1) sC[8] = e;
    __syncthreads();
    return sC[8];
}

template <class C>
struct WideSmem
{
    //! everything but the split fine-tune tables (16-byte multiple: the tables follow)
    static constexpr size_t base()
    {
        return (size_t(C::TWN + C::CH_ELEMS + C::WPB * C::XW) * sizeof(float2) + 4 * sizeof(RedRec) + size_t(C::WPB) * 2 * sizeof(float2) + sizeof(TailRec) + size_t(C::WPB) * 12 * sizeof(int) + 15) & ~size_t(15);
    }
    static constexpr size_t bytes(const bool withFine) { return base() + (withFine ? FineDims<C::LOG2N>::BYTES : 0); }
};

template <class C, bool DBG, bool UNI>
__global__ void __launch_bounds__(C::BLOCK, C::MINW)
detectWide(const DetectArgs a, const FastTables ft, const unsigned nSets)
{
    constexpr int N = C::N, T = C::T, VEC = C::VEC, R = C::R, WPB = C::WPB, WPWIN = C::WPWIN;
    constexpr int LOG2N = C::LOG2N, LOG2T = C::LOG2T, B1 = C::B1, B2 = C::B2, HB = C::HB;
    constexpr int SLOTS = lastPhaseSlots<LOG2N, B2, LOG2N>();
    constexpr int M = N * LORAHIP_FINE_STEPS;

    extern __shared__ __attribute__((aligned(16))) char smemRaw[];
    v2f *sTw = reinterpret_cast<v2f *>(smemRaw);                         // [TWN]
    v2f *sCh = sTw + C::TWN;                                             // [CH_ELEMS]
    v2f *sX = sCh + C::CH_ELEMS;                                         // [WPB][XW]
    RedRec *sRed = reinterpret_cast<RedRec *>(sX + WPB * C::XW);         // [4]: one per wavefront
    v2f *sNb = reinterpret_cast<v2f *>(sRed + 4);                        // [WPB][2]: bins left/right of the peak
    TailRec &tr = *reinterpret_cast<TailRec *>(sNb + WPB * 2);
    int *sChain = reinterpret_cast<int *>(&tr + 1) + (threadIdx.x >> LOG2T) * 12;    // fineChainBlock scratch of this window
    double2 *sFine = reinterpret_cast<double2 *>(smemRaw + WideSmem<C>::base());     // split fine-tune tables (non-UNI kernels)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wsub = tid >> LOG2T;                        // window inside the workgroup iteration
    const int t = tid & (T - 1);
    v2f *X = sX + wsub * C::XW;

    const v2f *gIq = reinterpret_cast<const v2f *>(a.iq), *gDown = reinterpret_cast<const v2f *>(a.down);
    const v2f *gFine = reinterpret_cast<const v2f *>(a.fine);
    v2f *gDec = reinterpret_cast<v2f *>(a.decOut), *gFft = reinterpret_cast<v2f *>(a.fftOut);

    // ---- one-time set-up -------------------------------------------------------------
    if (tid < 64) tr.w[tid] = 0xffffffffu;                // empty tail slots
    for (int i = tid; i < C::TW_LDS; i += C::BLOCK) sTw[i] = reinterpret_cast<const v2f *>(ft.twStage)[i];

    // register twiddles of the last phase (klow = t there)
    v2f twR[C::TW_ALL_LDS ? 1 : SLOTS];
    if (!C::TW_ALL_LDS)
    {
        int slot = 0;
#pragma unroll
        for (int b = B2; b < LOG2N; b += 2)
#pragma unroll
            for (int kl = 0; kl < (1 << (b - B2)); kl++)
            {
                const int k = t + (kl << B2);
                const int base = twStageOffset(LOG2N, b) + k;
                twR[C::TW_ALL_LDS ? 0 : slot] = reinterpret_cast<const v2f *>(ft.twStage)[base];
                twR[C::TW_ALL_LDS ? 0 : slot + 1] = reinterpret_cast<const v2f *>(ft.twStage)[base + (1 << b)];
                twR[C::TW_ALL_LDS ? 0 : slot + 2] = reinterpret_cast<const v2f *>(ft.twStage)[base + (2 << b)];
                slot += 3;
            }
    }

    FineLds fl;
    fl.A = nullptr; fl.B = nullptr; fl.split = false;
    if (!UNI) fl = fineLoadLds<LOG2N>(sFine, a.fineA, a.fineB, tid, C::BLOCK);

    // chirp table values of this lane's sample positions: _upChirpTable = conj(_downChirpTable) (LoRaDemod.cpp:103-104)
    const bool perWindowSel = !UNI && a.chirpSel != nullptr;
    const float s0 = (!perWindowSel && a.chirpSelAll == LORAHIP_CHIRP_UP) ? -1.0f : 1.0f;
    v2f ch[C::CH_LDS ? 1 : R][C::CH_LDS ? 1 : VEC];
    if (C::CH_LDS)
    {
        for (int i = tid; i < N; i += C::BLOCK)
        {
            const v2f c = gDown[i];
            sCh[i] = MAKE2(c.x, s0 * c.y);
        }
    }
    else
    {
#pragma unroll
        for (int r = 0; r < (C::CH_LDS ? 0 : R); r++)
#pragma unroll
            for (int u = 0; u < VEC; u++)
            {
                const v2f c = gDown[VEC * t + u + VEC * T * r];
                ch[C::CH_LDS ? 0 : r][C::CH_LDS ? 0 : u] = MAKE2(c.x, s0 * c.y);
            }
    }
    __syncthreads();

    // coalesced window load: VEC*8 bytes per lane, a wavefront covers 512 or 1024 contiguous bytes, R rows
    v2f xn[R][VEC];
    auto issueLoads = [&](const unsigned set_)
    {
        const unsigned w_ = set_ * WPB + wsub;
        const unsigned wc_ = w_ < a.nWindows ? w_ : a.nWindows - 1;
        const v2f *in_ = gIq + (a.offsets ? a.offsets[wc_] : (long long)wc_ * a.stride);
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            const v2f *p = in_ + VEC * t + VEC * T * r;
            if (VEC == 2)
            {
                const v4f q = C::NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p)) : *reinterpret_cast<const v4f *>(p);
                xn[r][0] = MAKE2(q.x, q.y);
                xn[r][VEC - 1] = MAKE2(q.z, q.w);
            }
            else xn[r][0] = C::NT ? __builtin_nontemporal_load(p) : *p;
        }
    };
    if (C::PREFETCH) issueLoads(blockIdx.x);
    const v2f fconst0 = gFine[0];

    // result of the previous set, waiting for its neighbours / tail record
    bool havePrev = false, pActive = false;
    unsigned pW = 0, cnt = 0;
    int pI = 0;
    float pV = 0.0f;
    double pTot = 0.0;

    auto recordPrev = [&]()
    {
        if (havePrev && pActive && t == 0)
        {
            const int s = (cnt + wsub) & 63;
            tr.w[s] = pW; tr.idx[s] = pI; tr.val[s] = pV; tr.tot[s] = pTot;
            tr.l[s] = sNb[wsub * 2]; tr.r[s] = sNb[wsub * 2 + 1];
        }
    };
    auto flushIfFull = [&]()
    {
        if (havePrev)
        {
            cnt += WPB;
            if ((cnt & 63) == 0 && wave == 0)
            {
                const unsigned ww = tr.w[lane];
                if (ww < a.nWindows) detectTail(a, ww, tr.idx[lane], tr.val[lane], tr.tot[lane], tr.l[lane], tr.r[lane]);
                tr.w[lane] = 0xffffffffu;
            }
        }
    };

    const int prioSlot = wavefrontSlot();
    for (unsigned set = blockIdx.x; set < nSets; set += gridDim.x)
    {
        rotatePriority<C::MINW, LORAHIP_PRIO_BATCH>(prioSlot);
        const unsigned w = set * WPB + wsub;
        const bool active = w < a.nWindows;
        const unsigned wc = active ? w : a.nWindows - 1;  // an inactive half redoes the last window, results dropped
        const int sel = perWindowSel ? a.chirpSel[wc] : a.chirpSelAll;
        const int idx0 = a.fineIdx0 ? a.fineIdx0[wc] : 0;
        const float err = (!UNI && a.fineErr) ? a.fineErr[wc] : 0.0f;
        const bool dechirp = sel != LORAHIP_CHIRP_NONE;
        const float d = err * (float)LORAHIP_FINE_STEPS;
        const bool moving = dechirp && d != 0.0f;

        v2f x[R][VEC];
        if (!C::PREFETCH) issueLoads(set);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < VEC; u++) x[r][u] = xn[r][u];

        // ---- fine-tune indices of this lane's samples for windows whose index moves (LoRaDemod.cpp:160-162): closed form
        // (lorahip_fine.h); a workgroup that holds a window where the form does not apply walks the exact chain instead
        int *sIdx = reinterpret_cast<int *>(X);             // aliases the exchange region (free until phase 0 ends)
        bool anyMoving = false, usedChain = false;
        unsigned yv[R][VEC];
        if constexpr (!UNI)
        {
            // one window per workgroup (the default shapes): `moving` is the same in every thread, no vote needed
            if constexpr (WPB == 1) anyMoving = moving;
            else anyMoving = __syncthreads_or(moving);
            if (anyMoving)
            {
                const FinePlan pl = finePlan(moving ? d : 0.0f, M);
                const unsigned ymax = fineLaneIndices<LOG2N, VEC, T, R>(idx0, pl, t, yv);
                int idxEnd = fineEndIndex(idx0, pl, LOG2N, LOG2N + 7);
                // the closed form can only reach the value M under the modulus M + 1 (a positive non-integer step that is not the
                // walk-down-and-stay case, lorahip_fine.h): every other window of a one-window workgroup skips the vote and its barrier
                if (WPB == 1 && (pl.mod == (unsigned)M || pl.sat)) usedChain = !pl.regular;
                else usedChain = __syncthreads_or(!pl.regular || ymax == (unsigned)M);
                if (usedChain)
                {
                    idxEnd = fineChainBlock<T, M>(idx0, moving ? d : 0.0f, t, t >> 6, sIdx, sChain);
#pragma unroll
                    for (int r = 0; r < R; r++)
#pragma unroll
                        for (int u = 0; u < VEC; u++) yv[r][u] = (unsigned)sIdx[chainSlot(VEC * t + u + VEC * T * r)];
                }
                if (moving && t == 0 && a.fineIdxOut && active) a.fineIdxOut[w] = idxEnd;
            }
        }
        if (!moving && t == 0 && active && a.fineIdxOut) a.fineIdxOut[w] = idx0;

        // ---- dechirp: (samp * chirp) * fine   (LoRaDemod.cpp:159) ---------------------------
        const v2f fconst = a.fineIdx0 ? gFine[idx0] : fconst0;
        v2f cw[R][VEC];
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            if (C::CH_LDS)
            {
                const v2f *p = sCh + VEC * t + VEC * T * r;
                if (VEC == 2)
                {
                    const v4f q = *reinterpret_cast<const v4f *>(p);
                    cw[r][0] = MAKE2(q.x, q.y);
                    cw[r][VEC - 1] = MAKE2(q.z, q.w);
                }
                else cw[r][0] = *p;
            }
            else
            {
#pragma unroll
                for (int u = 0; u < VEC; u++) cw[r][u] = ch[C::CH_LDS ? 0 : r][C::CH_LDS ? 0 : u];
            }
        }
        if (UNI || (!perWindowSel && !anyMoving))
        {
            if (a.chirpSelAll != LORAHIP_CHIRP_NONE)
            {
                dechirpMany<R * VEC>(&x[0][0], &cw[0][0], fconst);
            }
        }
        else
        {
            const float sgn = (perWindowSel && sel == LORAHIP_CHIRP_UP) ? -1.0f : 1.0f;
            const v2f *cwf = &cw[0][0];
            const v2f sgn2 = MAKE2(1.0f, sgn);                       // one packed multiply: (re, +-im), both exact
            const auto chirpOf = [&](const int i) { return cwf[i] * sgn2; };
            if (anyMoving)
            {
                // yv = idx0 in the windows that do not move; a launch-uniform selection is already in the values (s0): no sign multiply
                const auto chirpRaw = [&](const int i) { return cwf[i]; };
                if (perWindowSel) dechirpFine<fineSplitLog2H(LOG2N), R * VEC>(&x[0][0], chirpOf, &yv[0][0], fl, gFine, dechirp);
                else dechirpFine<fineSplitLog2H(LOG2N), R * VEC>(&x[0][0], chirpRaw, &yv[0][0], fl, gFine, dechirp);
            }
            else
            {
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int u = 0; u < VEC; u++)
                    {
                        const v2f y = cmulv(cmulv(x[r][u], chirpOf(r * VEC + u)), fconst);
                        x[r][u] = dechirp ? y : x[r][u];
                    }
            }
            if (usedChain) __syncthreads();               // sIdx is about to be overwritten by exchange 0
        }
        if (DBG && a.decOut && active)
        {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++) gDec[(size_t)w * N + VEC * t + u + VEC * T * r] = x[r][u];
        }

        // ---- phase 0: bits [0, B1) in registers, one group per u -----------------------------
        v2f v0[VEC][R];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < VEC; u++) v0[u][Plan<LOG2N>::pos(VEC * T * r) & (R - 1)] = x[r][u];
        // next set's samples go in flight now (past the end: re-read the last set, harmless and branch-free)
        if (C::PREFETCH) issueLoads(set + gridDim.x < nSets ? set + gridDim.x : nSets - 1);
#pragma unroll
        for (int u = 0; u < VEC; u++) runPhase<LOG2N, 0, B1, false>(v0[u], 0, sTw, nullptr);

        // ---- exchange 0 ----------------------------------------------------------------------
#pragma unroll
        for (int u = 0; u < VEC; u++)
        {
            v2f *row = X + C::x0off(VEC * t + u);
#pragma unroll
            for (int e = 0; e < R; e++) row[e] = v0[u][e];
        }
        __syncthreads();                                                                      // B1
        recordPrev();                                   // previous set: its neighbours are visible now

        // phase 1 (middle): bits [B1, B2); klow = t mod R, high = t / R
        v2f v1[16];
        {
            const int klow = t & (R - 1), high = t >> B1;
            const int rhigh = rev4(high, HB);
#pragma unroll
            for (int e = 0; e < 16; e++) v1[e] = X[C::x0off((rev4(e, 4) << HB) | rhigh) + klow];
        }
        if (!C::INPLACE)
        {
            __syncthreads();                                                                  // B2
            flushIfFull();                              // wave 0, once per 64 windows
        }
        runPhase<LOG2N, B1, B2, false>(v1, t & (R - 1), sTw, nullptr);
        v2f vl[16];
        if (C::INPLACE)
        {
            // write-back in place: only this lane touches these 16 words between B1 and B3
            const int klow = t & (R - 1), rhigh = rev4(t >> B1, HB);
#pragma unroll
            for (int e = 0; e < 16; e++) X[C::x0off((rev4(e, 4) << HB) | rhigh) + klow] = v1[e];
            __syncthreads();                                                                  // B3
            flushIfFull();
            // last phase: lane t holds positions t + T*e2; t = klow + R*e_mid, e2 = high
            const int rmid = rev4(t >> B1, 4) << HB;
#pragma unroll
            for (int e = 0; e < 16; e++) vl[e] = X[C::x0off(rmid | rev4(e, HB)) + klow];
        }
        else
        {
            // exchange 1: position klow + R*e + 16R*high, 8 elements of padding per `high`
            v2f *base = X + (t >> B1) * C::X1 + (t & (R - 1));
#pragma unroll
            for (int e = 0; e < 16; e++) base[e * R] = v1[e];
            __syncthreads();                                                                  // B3
            // phase 2 = last: lane t holds positions t + T*e
#pragma unroll
            for (int e = 0; e < 16; e++) vl[e] = X[e * C::X1 + t];
        }
        if (C::TW_ALL_LDS) runPhase<LOG2N, B2, LOG2N, false>(vl, t, sTw, nullptr);
        else runPhase<LOG2N, B2, LOG2N, true>(vl, 0, nullptr, twR);

        // ---- scan (LoRaDetector.hpp:36-48): bin = t + T*e, ascending in e ----------------------
        if (DBG && a.fftOut && active)
        {
#pragma unroll
            for (int e = 0; e < 16; e++) gFft[(size_t)w * N + t + (e << LOG2T)] = vl[e];
        }
        float bestV;
        double tot;
        const int bestE = laneScan<16, SCAN_CHAINS_WIDE>([&](const int e) { return vl[e]; }, bestV, tot);
        int bestI = t + (bestE << LOG2T);
        if (!(bestV > 0.0f)) bestI = 0;
        groupArgmax<64>(bestV, bestI);
        tot = groupSumF64<64>(tot);
        if (lane == 0) { sRed[wave].v = bestV; sRed[wave].i = bestI; sRed[wave].tot = tot; }
        __syncthreads();                                                                      // B4
        // every lane combines the window's wavefronts in the same order: identical results on all of them
        {
            const RedRec r0 = sRed[wsub * WPWIN];
            bestV = r0.v; bestI = r0.i; tot = r0.tot;
#pragma unroll
            for (int k = 1; k < WPWIN; k++)
            {
                const RedRec rk = sRed[wsub * WPWIN + k];
                argmaxCombine(bestV, bestI, rk.v, rk.i);
                tot += rk.tot;
            }
        }

        // ---- neighbours of the peak for fIndex (LoRaDetector.hpp:56-57): the owners post them. Only the
        // wavefront(s) that hold bin k-1 / k+1 walk the register-select tree.
        {
            // (a tree of scalar branches on the wave-uniform bin number instead of the per-lane select tree measured 1 % SLOWER:
            // profiles/r03/s6_ab_scalar_pick_negative.txt)
            const int bl = (bestI + N - 1) & (N - 1), br = (bestI + 1) & (N - 1);
            const bool ownL = (bl & (T - 1)) == t, ownR = (br & (T - 1)) == t;
            if (__any(ownL | ownR))
            {
                const v2f mine = selectFlat<16>(vl, ownL ? (bl >> LOG2T) : (br >> LOG2T));
                if (ownL) sNb[wsub * 2] = mine;
                if (ownR) sNb[wsub * 2 + 1] = mine;
            }
        }
        havePrev = true; pActive = active; pW = w; pI = bestI; pV = bestV; pTot = tot;
    }

    // ---- drain: last set's record, then whatever is queued --------------------------------------
    __syncthreads();
    recordPrev();
    __syncthreads();
    if (wave == 0)
    {
        const unsigned ww = tr.w[lane];
        if (ww < a.nWindows) detectTail(a, ww, tr.idx[lane], tr.val[lane], tr.tot[lane], tr.l[lane], tr.r[lane]);
    }
}

/***********************************************************************
 * launch
 **********************************************************************/
template <class C, bool DBG, bool UNI>
static hipError_t launchOneWide(const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    const size_t smem = WideSmem<C>::bytes(!UNI);
    static unsigned long long attrDone = 0;
    {
        const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(detectWide<C, DBG, UNI>), smem, attrDone);
        if (e != hipSuccess) return e;
    }
    const unsigned nSets = (a.nWindows + C::WPB - 1) / C::WPB;
    // persistent: as many workgroups as stay resident, never more than there are sets of work
    unsigned perCu = unsigned((160u * 1024u) / smem);
    if (perCu > unsigned(C::BPC)) perCu = unsigned(C::BPC);
    if (perCu < 1) perCu = 1;
    unsigned grid = unsigned(ft.nBlocksHint > 0 ? ft.nBlocksHint : 256) * perCu;
    if (grid > nSets) grid = nSets;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL((detectWide<C, DBG, UNI>), dim3(grid), dim3(C::BLOCK), smem, stream, a, ft, nSets);
    return hipGetLastError();
}

template <class C>
static hipError_t launchCfgWide(const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    const bool uni = a.chirpSel == nullptr && a.fineErr == nullptr;
    if (a.decOut || a.fftOut) return launchOneWide<C, true, false>(a, ft, stream);
    return uni ? launchOneWide<C, false, true>(a, ft, stream) : launchOneWide<C, false, false>(a, ft, stream);
}

/***********************************************************************
 * configurations: geometry per SF x layout (the plain padded exchange-0 layout, or the swizzled one the in-place middle
 * phase needs -- both found with tools/lds_conflicts.py), plus an option mask.
 **********************************************************************/
template <int SF, bool INPLACE> struct WGeo;
//                                                  VEC  X0: ROT PAD S  D   S2 D2
template <> struct WGeo<11, false> { enum { VEC = 2, ROT = 0, PAD = 1, S = 4, D = 1,  S2 = 0, D2 = 0 }; };    // 128 lanes x 16 pts: [R2,4] X [4,4] X [4,4]
template <> struct WGeo<11, true>  { enum { VEC = 2, ROT = 3, PAD = 1, S = 1, D = 4,  S2 = 2, D2 = 8 }; };
template <> struct WGeo<12, false> { enum { VEC = 1, ROT = 0, PAD = 1, S = 0, D = 0,  S2 = 0, D2 = 0 }; };    // 256 lanes x 16 pts: [4,4] X [4,4] X [4,4]
template <> struct WGeo<12, true>  { enum { VEC = 1, ROT = 0, PAD = 1, S = 6, D = 16, S2 = 7, D2 = 16 }; };

enum : unsigned
{
    WW2 = 1u << 0, WW4 = 1u << 1,       // waves per SIMD the register budget is set for (default 3)
    WCH_LDS = 1u << 2,                   // chirp values from an LDS copy of the table (default: registers)
    WTW_LDS = 1u << 3,                   // last-phase twiddles from the LDS table (default: registers)
    WPF_NONE = 1u << 4,                  // no register prefetch of the next window
    WNT = 1u << 5,                       // non-temporal IQ loads
    WONE = 1u << 6,                      // one window per workgroup (SF11: 128 threads)
    WINPLACE = 1u << 7                   // in-place middle phase (3 barriers per window instead of 4)
};
template <int SF, unsigned O>
using Wide = WideCfg<SF, WGeo<SF, (O & WINPLACE) != 0>::VEC, (O & WW2) ? 2 : (O & WW4) ? 4 : 3,
                     WGeo<SF, (O & WINPLACE) != 0>::ROT, WGeo<SF, (O & WINPLACE) != 0>::PAD, WGeo<SF, (O & WINPLACE) != 0>::S,
                     WGeo<SF, (O & WINPLACE) != 0>::D, (O & WCH_LDS) != 0, (O & WTW_LDS) != 0, !(O & WPF_NONE), (O & WNT) != 0,
                     (O & WONE) ? 1 : 0, (O & WINPLACE) != 0, WGeo<SF, (O & WINPLACE) != 0>::S2, WGeo<SF, (O & WINPLACE) != 0>::D2>;

// streaming demodulator configurations (demodStreamWide below): one channel per workgroup, in-place middle phase
typedef Wide<11, WW2 | WPF_NONE | WONE | WINPLACE> StreamWide11;
typedef Wide<12, WW2 | WPF_NONE | WINPLACE> StreamWide12;
#ifdef LORAHIP_FMA
#define LORAHIP_NO_STREAM_WIDE 1        // the contracted build holds batch kernels only: level 3 stays on the reference's operation graph
#endif

bool wideAvailable(const int sf) { return sf == 11 || sf == 12; }

//! host-side check of a configuration's exchange-0 layout: every (row, element) has its own word inside the region
template <class C>
static bool layoutOk()
{
    std::vector<char> used(size_t(C::XW), 0);
    for (int n = 0; n < C::NL; n++)
        for (int e = 0; e < C::R; e++)
        {
            const int a = C::x0off(n) + e;
            if (a < 0 || a >= C::XW || used[size_t(a)]) return false;
            used[size_t(a)] = 1;
        }
    return true;
}

bool wideLayoutsOk();

/***********************************************************************
 * selectable variants (lorahip_set_variant): 0 = the measured best per SF (profiles/r01/s8_variants.txt)
 **********************************************************************/
typedef hipError_t (*WideLaunch)(const DetectArgs &, const FastTables &, hipStream_t);
struct WideVariant { int sf, variant; WideLaunch launch; bool (*layoutOk)(); };
#define V(SF, N, OPTS) { SF, N, &launchCfgWide<Wide<SF, (OPTS)>>, &layoutOk<Wide<SF, (OPTS)>> }
// ships: the default (0) and one alternative per SF (10: the plain exchange-0 layout, four barriers per window, register
// prefetch); the rest of the round-1 A/B set only with -DLORAHIP_ALL_VARIANTS (profiles/r01/s8_variants.txt)
//! defaults by call shape, like lorahip_fast.hip's: per-window settings (moving fine-tune index) at two waves per SIMD
template <class UNI_CFG, class MOVING_CFG>
static hipError_t launchWideByShape(const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    // the debug ports (dec / fft outputs, 3x the traffic: not occupancy-bound) at the two-waves-per-SIMD register budget too: at three
    // they kept 408-508 B of scratch (tools/kernel_resources.py, profiles/r04)
    if (a.decOut || a.fftOut) return launchOneWide<MOVING_CFG, true, false>(a, ft, stream);
    const bool uni = a.chirpSel == nullptr && a.fineErr == nullptr;
    return uni ? launchOneWide<UNI_CFG, false, true>(a, ft, stream) : launchOneWide<MOVING_CFG, false, false>(a, ft, stream);
}
static bool defaultLayoutsOk()
{
    return layoutOk<Wide<11, WPF_NONE | WNT | WONE | WINPLACE>>() && layoutOk<Wide<11, WW2 | WNT | WONE | WINPLACE>>() &&
           layoutOk<Wide<12, WNT | WINPLACE>>() && layoutOk<Wide<12, WW2 | WNT | WINPLACE>>();
}
static const WideVariant kWideVariants[] = {
    { 11, 0, &launchWideByShape<Wide<11, WPF_NONE | WNT | WONE | WINPLACE>, Wide<11, WW2 | WNT | WONE | WINPLACE>>, &defaultLayoutsOk },   // default
#ifndef LORAHIP_FMA      // (the contracted build carries the defaults only)
    { 11, 10, &launchWideByShape<Wide<11, 0>, Wide<11, WW2>>, &layoutOk<Wide<11, 0>> },
#endif
    { 12, 0, &launchWideByShape<Wide<12, WNT | WINPLACE>, Wide<12, WW2 | WNT | WINPLACE>>, &defaultLayoutsOk },                            // default
#ifndef LORAHIP_FMA
    { 12, 10, &launchWideByShape<Wide<12, 0>, Wide<12, WW2>>, &layoutOk<Wide<12, 0>> },
#endif
#ifdef LORAHIP_ALL_VARIANTS
    V(11, 2, WW2), V(11, 3, WW2 | WCH_LDS), V(11, 4, WW2 | WTW_LDS), V(11, 5, WW2 | WCH_LDS | WTW_LDS), V(11, 6, WW4 | WPF_NONE),
    V(11, 7, WPF_NONE), V(11, 8, WNT), V(11, 9, WPF_NONE | WNT), V(11, 11, WPF_NONE | WNT | WONE), V(11, 12, WNT | WONE),
    V(11, 13, WPF_NONE | WNT | WONE | WINPLACE), V(11, 14, WPF_NONE | WNT | WINPLACE),
    V(12, 2, WW2), V(12, 3, WW2 | WCH_LDS), V(12, 4, WW2 | WTW_LDS), V(12, 5, WW2 | WCH_LDS | WTW_LDS), V(12, 6, WW4 | WPF_NONE),
    V(12, 7, WPF_NONE), V(12, 8, WNT), V(12, 9, WPF_NONE | WNT), V(12, 13, WPF_NONE | WNT | WINPLACE), V(12, 14, WNT | WINPLACE),
    // round 2: the defaults at the 256-register budget of two waves per SIMD
    V(11, 30, WW2 | WPF_NONE | WNT | WONE | WINPLACE), V(11, 31, WW2 | WNT | WINPLACE), V(11, 29, WW2 | WNT | WONE | WINPLACE),
    V(12, 30, WW2 | WNT | WINPLACE), V(12, 31, WW2 | WPF_NONE | WNT | WINPLACE),
    // round 4: the two-waves-per-SIMD kernels with the chirp values from an LDS copy (32 registers back: no scratch)
    V(11, 32, WW2 | WNT | WONE | WINPLACE | WCH_LDS), V(12, 32, WW2 | WNT | WINPLACE | WCH_LDS), V(12, 33, WW2 | WPF_NONE | WNT | WINPLACE | WCH_LDS),
#endif
};
#undef V

bool wideLayoutsOk()
{
    for (const WideVariant &v : kWideVariants) if (!v.layoutOk()) return false;
    return layoutOk<StreamWide11>() && layoutOk<StreamWide12>();
}

hipError_t launchWide(const int sf, const int variant, const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    const WideVariant *def = nullptr;
    for (const WideVariant &v : kWideVariants)
    {
        if (v.sf != sf) continue;
        if (v.variant == variant) return v.launch(a, ft, stream);
        if (v.variant == 0) def = &v;
    }
    return def ? def->launch(a, ft, stream) : hipErrorInvalidValue;     // unknown numbers run the default
}

#ifndef LORAHIP_NO_STREAM_WIDE
/***********************************************************************
 * Streaming demodulator for the long windows: a workgroup OWNS a channel (T = 128 / 256 lanes) and walks its stream
 * window after window -- the level-3 twin of lorahip_stream.hip, same frame machine (lorahip_framemachine.h), the
 * in-place three-phase FFT of detectWide. Five workgroup barriers per window (two more while the fine-tune index moves).
 **********************************************************************/
#ifdef LORAHIP_WG_TIMELINE     // profiling build (tools/wg_timeline.py): when and where every workgroup of the last streaming launch ran
__device__ unsigned long long gWgTimeline[16384][4];
__device__ unsigned gWgWaveHwId[16384][4];
extern "C" int lorahip_debug_wg_timeline(void *out, const size_t bytes)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gWgTimeline), bytes < sizeof(gWgTimeline) ? bytes : sizeof(gWgTimeline)) == hipSuccess ? 0 : -1;
}
extern "C" int lorahip_debug_wg_waves(void *out, const size_t bytes)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gWgWaveHwId), bytes < sizeof(gWgWaveHwId) ? bytes : sizeof(gWgWaveHwId)) == hipSuccess ? 0 : -1;
}
#endif

//! the channel of a demodStreamWide workgroup as residentPackOwn sees it: one channel, packed by the 64 lanes of wavefront 0
struct ResPackWide { static constexpr int WPW = 1, T = 64, LOG2T = 6; };

//! RES: the resident receiver (see demodStream in lorahip_streamkernel.h, lorahip_residentproto.h): the launch stays, the steps arrive as
//! messages; a workgroup moves through a step as ONE (its wavefronts share a window), wavefront 0 waits for the message, packs the
//! channel's packets and signals, and reports
template <class C, bool PERSIST, bool RES = false>
__global__ void __launch_bounds__(C::T, 2)
demodStreamWide(const StreamArgs s)
{
    static_assert(!(RES && PERSIST), "the resident receiver walks its channels itself");
#ifdef LORAHIP_WG_TIMELINE
    const unsigned long long tl0 = wall_clock64();
#endif
    static_assert(C::INPLACE && C::WPB == 1 && !C::CH_LDS && !C::TW_ALL_LDS, "stream configs: one channel per workgroup, in-place middle phase");
    constexpr int N = C::N, T = C::T, VEC = C::VEC, R = C::R, WPWIN = C::WPWIN;
    constexpr int LOG2N = C::LOG2N, LOG2T = C::LOG2T, B1 = C::B1, B2 = C::B2, HB = C::HB;
    constexpr int SLOTS = lastPhaseSlots<LOG2N, B2, LOG2N>();
    constexpr int M = N * LORAHIP_FINE_STEPS;

    extern __shared__ __attribute__((aligned(16))) char smemRaw[];
    v2f *sTw = reinterpret_cast<v2f *>(smemRaw);                         // [TWN]
    v2f *X = sTw + C::TWN;                                               // [XW]
    RedRec *sRed = reinterpret_cast<RedRec *>(X + C::XW);                // [WPWIN]
    v2f *sNb = reinterpret_cast<v2f *>(sRed + 4);                        // [2]
    int *sChain = reinterpret_cast<int *>(sNb + 2);                      // [12] fineChainBlock scratch
    double2 *sFine = reinterpret_cast<double2 *>(sChain + 12);           // split fine-tune tables (lorahip_fine.h)
    static_assert(((size_t(C::TWN + C::XW) * sizeof(v2f) + 4 * sizeof(RedRec) + 2 * sizeof(v2f) + 12 * sizeof(int)) & 15) == 0, "the split tables are read with ds_read_b128");

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const v2f *gIq = reinterpret_cast<const v2f *>(s.iq), *gFine = reinterpret_cast<const v2f *>(s.fine);
    for (int i = t; i < C::TW_LDS; i += T) sTw[i] = reinterpret_cast<const v2f *>(s.twStage)[i];
    v2f twR[SLOTS];
    {
        int slot = 0;
#pragma unroll
        for (int b = B2; b < LOG2N; b += 2)
#pragma unroll
            for (int kl = 0; kl < (1 << (b - B2)); kl++)
            {
                const int base = twStageOffset(LOG2N, b) + t + (kl << B2);
                twR[slot] = reinterpret_cast<const v2f *>(s.twStage)[base];
                twR[slot + 1] = reinterpret_cast<const v2f *>(s.twStage)[base + (1 << b)];
                twR[slot + 2] = reinterpret_cast<const v2f *>(s.twStage)[base + (2 << b)];
                slot += 3;
            }
    }
    v2f ch[R][VEC];                                        // down-chirp values of this lane's sample positions
#pragma unroll
    for (int r = 0; r < R; r++)
#pragma unroll
        for (int u = 0; u < VEC; u++) ch[r][u] = reinterpret_cast<const v2f *>(s.down)[VEC * t + u + VEC * T * r];
    const FineLds fl = fineLoadLds<LOG2N>(sFine, s.fineA, s.fineB, t, T);
    ResLds *sR = reinterpret_cast<ResLds *>(reinterpret_cast<char *>(sFine) + FineDims<LOG2N>::BYTES);       // RES only (the launcher adds the bytes)
    if constexpr (RES)
    {
        if (t == 0) { sR->carry = s.carry; sR->carryCap = s.carryCap; }
        if (t < RES_RING) { sR->calls[t] = 0; sR->arrive[t] = 0; sR->more[t] = 0; sR->msgSeq[t] = 0u; }
        // the census: the host rings the first step only when every workgroup is on the device
        if (t == 0 && atomicAdd(&s.res->arrived, 1u) + 1u == gridDim.x) sysStore(&s.resHost->arrivedAll, 1u);
    }
    __syncthreads();

    // One channel per workgroup: everything the frame machine touches is WORKGROUP-UNIFORM. Saying so (v_readfirstlane where a value
    // comes out of the vector unit: the peak, the squelch decision, fIndex) lets the whole machine -- state, 64-bit positions, record
    // pointers and counters, the fine-tune plan -- live in scalar registers and run on the scalar unit instead of being replicated
    // in every lane's vector registers.
    const auto uniI = [](const int v) { return __builtin_amdgcn_readfirstlane(v); };
    const auto uniF = [](const float v) { return __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); };
    // persistent grid (s.maxBlocks workgroups at most), a workgroup takes one channel after the other: see demodStream
    // RES: one turn of the outer loop per receiver step (otherwise exactly one turn)
    unsigned step = 0;
    unsigned long long resNValid = 0;
    unsigned resCalls = 0;
    bool resMore = false;
    // (ONE loop over the workgroup's channels, step after step -- the wait for the next step's message sits at the end of the last channel's
    // turn, not in a loop around this one: the loop nest the compiler sees is the persistent instance's)
    const auto nextStep = [&]() -> bool
    {
        if (wave == 0)
        {
            ResMsgR m0;
            const bool ok = residentWait(s, step + 1u, m0, sR, true);
            if (lane == 0) sR->go = ok ? 1 : 0;
        }
        __syncthreads();
        if (!__builtin_amdgcn_readfirstlane(sR->go)) return false;  // (the quit message, the abort flag or the watchdog: the whole workgroup leaves)
        // (of the message only n_valid stays in registers across the windows; every wavefront from LDS, wavefront 0 too: one path)
        resNValid = uni64(sR->msg[(step + 1u) & 3u].nValid);
        step++;
        resCalls = 0; resMore = false;
        return true;
    };
    if constexpr (RES) { if (!nextStep()) return; }
    unsigned c = blockIdx.x;                                // (the grid never exceeds the channel count)
    do
    {
    StreamState st;
    if constexpr (RES)
    {
        // (what this workgroup stored a step ago: read at agent scope -- past the L1 and the scalar cache, which may hold an older copy --
        // and made scalar again)
        const unsigned long long *q = reinterpret_cast<const unsigned long long *>(&s.state[c]);
        unsigned long long w[5];
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = uni64(agentLoad(q + i));
        static_assert(sizeof(StreamState) == 40, "five 64-bit words");
        __builtin_memcpy(&st, w, sizeof(st));
    }
    else st = s.state[c];
    if (s.flags & 1) { st.pos = 0; st.callCount = 0; }                        // a new run: every stream from its first sample
    if (s.flags & 2) { st.state = ST_FRAMESYNC; st.downTable = 0; }           // activate() (LoRaDemod.cpp:139-143)
    const long long base = s.uniformLen >= 0 ? (long long)c * s.uniformStride : s.base[c];
    const long long len = RES ? (long long)resNValid : (s.uniformLen >= 0 ? s.uniformLen : s.len[c]);         // (RES: what the step's message says)
    StreamOut o;
    o.init(s, c);
    if constexpr (RES)
    {
        // the record arrays of this step's set (StreamArgs::resRecStride)
        const size_t setOff = size_t(step & 3u) * size_t(s.resRecStride);
        o.symOut = reinterpret_cast<short *>(reinterpret_cast<char *>(o.symOut) + setOff);
        o.pktOut = reinterpret_cast<StreamPacket *>(reinterpret_cast<char *>(o.pktOut) + setOff);
        if (o.sigOut) o.sigOut = reinterpret_cast<StreamSignal *>(reinterpret_cast<char *>(o.sigOut) + setOff);
    }
    o.carryIn(s, st, c, t, T, !RES);

    // one window: LoRaDemod.cpp:157-166 + LoRaDetector::detect; every argument is workgroup-uniform
    // What a work() call consumes of the float outputs depends on its state (see demodStream, lorahip_stream.hip): without a
    // trace the squelch decision comes from a quick estimate (squelchQuick) with the exact chain as the fallback near the
    // threshold, fIndex is evaluated only for an unsquelched FRAMESYNC window, and the neighbour fetch with its barrier is skipped
    // whenever neither is needed
    // Signals without a trace (lorahip_demod_set_signals): the DOWNCHIRP1 call of a packet takes the traced path (`full`), see demodStream
    const bool traced = s.calls != nullptr;
    const bool sig = s.sigOut != nullptr;
    // The window's samples are PREFETCHED: most calls consume N samples (both down-chirp states, every data symbol, a squelched
    // FRAMESYNC window; the second sync window sits at pos + N too), so the window at off + N is requested as soon as this one's
    // samples have left their registers for phase 0 -- the loads stay in flight across the window's barriers (which wait for LDS
    // only) and hide the HBM latency the frame machine otherwise exposes once per call. A call that consumed something else
    // (N - value while acquiring, the quarter chirp) finds another offset than the prefetched one and loads its own window.
    v2f x[R][VEC];
    long long preOff = -1;                                   // workgroup-uniform: which window x holds (or is about to hold)
    const long long endOff = base + len;
    auto loadWindow = [&](const long long off)
    {
        const v2f *win = gIq + off;                          // scalar base, per-lane offsets
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            const v2f *p = win + VEC * t + VEC * T * r;
            if (VEC == 2)
            {
                const v4f q = *reinterpret_cast<const v4f *>(p);
                x[r][0] = MAKE2(q.x, q.y);
                x[r][VEC - 1] = MAKE2(q.z, q.w);
            }
            else x[r][0] = *p;
        }
    };
    auto detect = [&](const bool all, const bool wantSq, const int wantFi, const long long off, const bool downTable, const int idx0, const float err,
                      int &value, float &power, float &powerAvg, float &fIndex, int &idxEnd, bool &squelched)
    {
        if (!STREAM_WIDE_PREFETCH || off != preOff) loadWindow(off);
        const float d = err * (float)LORAHIP_FINE_STEPS;
        const bool moving = d != 0.0f;                       // workgroup-uniform
        int *sIdx = reinterpret_cast<int *>(X);
        idxEnd = idx0;
        unsigned yv[R][VEC];
        bool usedChain = false;
        if (moving)
        {
            // closed-form indices of this lane's samples (lorahip_fine.h); the exact chain where the form does not apply
            FinePlan pl = finePlan(d, M);
            pl.q = (unsigned)uniI((int)pl.q); pl.mod = (unsigned)uniI((int)pl.mod); pl.regular = uniI(pl.regular); pl.sat = uniI(pl.sat);
            const unsigned ymax = fineLaneIndices<LOG2N, VEC, T, R>(idx0, pl, t, yv);
            idxEnd = uniI(fineEndIndex(idx0, pl, LOG2N, LOG2N + 7));
            // the closed form can only reach the value M under the modulus M + 1 (lorahip_fine.h): no vote, no barrier otherwise
            usedChain = !pl.regular || ((pl.mod != (unsigned)M && !pl.sat) && __syncthreads_or(ymax == (unsigned)M));
            if (t == 0 && nearStep(d)) atomicAdd(s.near + 1, 1u);                    // counted, not changed (lorahip_internal.h)
            if (usedChain)
            {
                idxEnd = uniI(fineChainBlock<T, M>(idx0, d, t, t >> 6, sIdx, sChain));
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int u = 0; u < VEC; u++) yv[r][u] = (unsigned)sIdx[chainSlot(VEC * t + u + VEC * T * r)];
            }
        }
        // _upChirpTable = conj(entry) (LoRaDemod.cpp:103): the table selection is workgroup-uniform here, so the conjugation rides on
        // the multiply's sign modifiers (cmulConjv / the CONJ pipeline) instead of a packed multiply by (1, -1) per sample
        const v2f *chf = &ch[0][0];
        const auto chirpRaw = [&](const int i) { return chf[i]; };
        if (moving)
        {
            if (downTable) dechirpFine<fineSplitLog2H(LOG2N), R * VEC, false>(&x[0][0], chirpRaw, &yv[0][0], fl, gFine, true);
            else dechirpFine<fineSplitLog2H(LOG2N), R * VEC, true>(&x[0][0], chirpRaw, &yv[0][0], fl, gFine, true);
        }
        else
        {
            const v2f fconst = gFine[idx0];
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++)
                    x[r][u] = cmulv(downTable ? cmulv(x[r][u], chf[r * VEC + u]) : cmulConjv(x[r][u], chf[r * VEC + u]), fconst);
        }
        if (usedChain) __syncthreads();                    // sIdx is about to be overwritten by exchange 0

        // phase 0 -> exchange 0
        v2f v0[VEC][R];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < VEC; u++) v0[u][Plan<LOG2N>::pos(VEC * T * r) & (R - 1)] = x[r][u];
        if (STREAM_WIDE_PREFETCH)
        {
            // x is free: the next window, speculatively (it must lie inside the stream: the call's own 2N are checked, the third N is not)
            if (off + 2 * N <= endOff) { preOff = off + N; loadWindow(preOff); }
            else preOff = -1;
        }
#pragma unroll
        for (int u = 0; u < VEC; u++) runPhase<LOG2N, 0, B1, false>(v0[u], 0, sTw, nullptr);
#pragma unroll
        for (int u = 0; u < VEC; u++)
        {
            v2f *row = X + C::x0off(VEC * t + u);
#pragma unroll
            for (int e = 0; e < R; e++) row[e] = v0[u][e];
        }
        __syncthreads();                                                                      // B1
        // phase 1 in place
        v2f v1[16];
        const int klow = t & (R - 1), rhigh = rev4(t >> B1, HB);
#pragma unroll
        for (int e = 0; e < 16; e++) v1[e] = X[C::x0off((rev4(e, 4) << HB) | rhigh) + klow];
        runPhase<LOG2N, B1, B2, false>(v1, klow, sTw, nullptr);
#pragma unroll
        for (int e = 0; e < 16; e++) X[C::x0off((rev4(e, 4) << HB) | rhigh) + klow] = v1[e];
        __syncthreads();                                                                      // B3
        // last phase: lane t holds positions t + T*e
        v2f vl[16];
        const int rmid = rev4(t >> B1, 4) << HB;
#pragma unroll
        for (int e = 0; e < 16; e++) vl[e] = X[C::x0off(rmid | rev4(e, HB)) + klow];
        runPhase<LOG2N, B2, LOG2N, true>(vl, 0, nullptr, twR);

        // scan (LoRaDetector.hpp:36-48). Without a trace the total only feeds the quick squelch estimate: fp32 then (laneScanQuick,
        // squelchQuickF); the fp64 total is summed where the exact chain is evaluated (below)
        float bestV;
        double tot;
        int bestE;
        if (all)
        {
            bestE = laneScan<16, SCAN_CHAINS_WIDE>([&](const int e) { return vl[e]; }, bestV, tot);
            tot = groupSumF64<64>(tot);
        }
        else
        {
            float totF;
            bestE = laneScanQuick<16>([&](const int e) { return vl[e]; }, bestV, totF);
            tot = (double)groupSumF32<64>(totF);
        }
        int bestI = t + (bestE << LOG2T);
        if (!(bestV > 0.0f)) bestI = 0;
        groupArgmax<64>(bestV, bestI);
        if (lane == 0) { sRed[wave].v = bestV; sRed[wave].i = bestI; sRed[wave].tot = tot; }
        __syncthreads();                                                                      // B4
        {
            const RedRec r0 = sRed[0];
            bestV = r0.v; bestI = r0.i; tot = r0.tot;
#pragma unroll
            for (int k = 1; k < WPWIN; k++)
            {
                const RedRec rk = sRed[k];
                argmaxCombine(bestV, bestI, rk.v, rk.i);
                tot += rk.tot;
            }
        }
        // workgroup-uniform from here on: every thread combined the same records in the same order
        bestI = uniI(bestI);
        bestV = uniF(bestV);
        tot = __hiloint2double(uniI(__double2hiint(tot)), uniI(__double2loint(tot)));
        value = bestI;
        bool needLogs = all, needFi = all;
        if (!all)
        {
            bool sure;
            squelched = squelchQuickF(bestV, (float)tot, s.thresh, float(16 + 6 + WPWIN + 2) * 0x1p-24f, sure);
            squelched = uniI(squelched) != 0;
            sure = uniI(sure) != 0;
            power = powerAvg = fIndex = 0.0f;                   // not consumed without a trace
            needLogs = wantSq && !sure;
            needFi = wantFi == 2 || (wantFi == 1 && (!sure || !squelched));   // 2: the second window of a FRAMESYNC call (:203, :217-221)
        }
        if (needLogs || needFi)
        {
            // neighbours of the peak (LoRaDetector.hpp:56-57)
            const int bl = (bestI + N - 1) & (N - 1), br = (bestI + 1) & (N - 1);
            const bool ownL = (bl & (T - 1)) == t, ownR = (br & (T - 1)) == t;
            if (__any(ownL | ownR))
            {
                const v2f mine = selectFlat<16>(vl, ownL ? (bl >> LOG2T) : (br >> LOG2T));
                if (ownL) sNb[0] = mine;
                if (ownR) sNb[1] = mine;
            }
            __syncthreads();                                                                  // B5
            if (needLogs && !all)
            {
                // the exact chain wants LoRaDetector.hpp:36-48's double total, in the traced path's association (bins still in registers)
                float bv_;
                double te;
                (void)laneScan<16, SCAN_CHAINS_WIDE>([&](const int e) { return vl[e]; }, bv_, te);
                te = groupSumF64<64>(te);
                if (lane == 0) sRed[wave].tot = te;
                __syncthreads();
                tot = sRed[0].tot;
#pragma unroll
                for (int k = 1; k < WPWIN; k++) tot += sRed[k].tot;
                tot = __hiloint2double(uniI(__double2hiint(tot)), uniI(__double2loint(tot)));
            }
            if (needLogs)
            {
                tailValuesPaired(s.powerScale, bestV, tot, sNb[0], sNb[1], lane, power, powerAvg, fIndex);
                power = uniF(power); powerAvg = uniF(powerAvg); fIndex = uniF(fIndex);
                squelched = (power - powerAvg) < s.thresh;                                   // :173-174
                squelched = uniI(squelched) != 0;
                if (wantSq && t == 0 && nearSquelch(power - powerAvg, s.thresh)) atomicAdd(s.near, 1u);
                if (!all) power = powerAvg = 0.0f;
            }
            else fIndex = uniF(fIndexPaired(bestV, sNb[0], sNb[1], lane));
        }
    };

    // A FRAMESYNC call that is sync'd and matches the first sync word looks at a SECOND window (LoRaDemod.cpp:183-206): it is taken
    // in the next pass of the loop (`pend`), so that the kernel holds ONE instance of the window code, not two -- nothing is written
    // and nothing is consumed in between, and the limits checked for the first pass cover the whole call (same scheme as demodStream).
    bool pend = false;
    int value0 = 0, fineIdxBefore0 = 0;
    float snr0 = 0.0f, fineErrBefore0 = 0.0f;
    const int slot = wavefrontSlot();
    const bool lastRound = PERSIST || RES || !LORAHIP_PRIO_HOLD || blockIdx.x >= s.lastRoundFrom;
    holdPriority<LORAHIP_PRIO_ALTERNATE>(!lastRound);
    while (pend || ((len - st.pos >= 2 * N) && o.calls < s.cap && o.nPkt < s.capPkt && o.nSig < s.capPkt))          // LoRaDemod.cpp:148
    {
        if (lastRound) rotatePriority<2, LORAHIP_PRIO_ALTERNATE>(slot);
        const bool second = pend;
        int value, idxEnd;
        float power, powerAvg, fIndex;
        const int fineIdxBefore = second ? fineIdxBefore0 : st.fineTuneIndex;
        const float fineErrBefore = second ? fineErrBefore0 : st.finefreqError;
        const bool fs = st.state == ST_FRAMESYNC;
        bool squelched;
        // window 0 of a call (:157-172), or window 1 of a parked one (:189-206: `int ft = _fineTuneIndex` starts from the committed
        // index and is not committed itself)
        detect(traced || (sig && !second && st.state == ST_DOWNCHIRP1), !second && (fs || st.state == ST_DATASYMBOLS), second ? 2 : (fs ? 1 : 0), base + st.pos + (second ? N : 0), st.downTable != 0,
               st.fineTuneIndex, st.finefreqError, value, power, powerAvg, fIndex, idxEnd, squelched);
        float snr = power - powerAvg;                                                   // :173 (squelched = snr < thresh, :174, comes from detect)
        if (!second) st.fineTuneIndex = idxEnd;                                         // :160-162
        bool syncd = !squelched && (st.prevValue + 4) / 8 == 0;                        // :183
        bool match0 = (value + 4) / 8 == (s.sync >> 4);                                // :184
        bool match1 = false, step = true;
        if (second)
        {
            match1 = (value + 4) / 8 == (s.sync & 0xf);                                // :205
            // detect() overwrote power / powerAvg / fIndex (:203); value, snr and what follows from them are window 0's
            value = value0; snr = snr0; squelched = false; syncd = true; match0 = true;
            pend = false;
        }
        else if (fs && syncd && match0)
        {
            pend = true; step = false;
            value0 = value; snr0 = snr; fineIdxBefore0 = fineIdxBefore; fineErrBefore0 = fineErrBefore;
        }
        if (step)
        {
            frameStep<N>(st, s, o, t == 0, value, power, powerAvg, snr, fIndex, squelched, syncd, match0, match1, fineIdxBefore, fineErrBefore);
            st.finefreqError = uniF(st.finefreqError);                                  // a float add runs on the vector unit: back to a scalar
        }
    }
    // (RES: the channel's packets and signals of the step into the step's rows first -- wavefront 0 wrote the records, t == 0 is the writer
    // lane; a packet's first symbols may still be in the carry row carryOut is about to overwrite)
    if constexpr (RES) { if (wave == 0) residentPackOwn<ResPackWide>(s, sR, step, c, o, true, lane, st.state == ST_DATASYMBOLS ? st.symCount : 0); }
    o.carryOut(s, st, c, t, T, true, true);             // T > 64: a barrier between the writer's last store and the other wavefronts' loads
#ifdef LORAHIP_WG_TIMELINE
    if ((t & 63) == 0 && c < 16384) gWgWaveHwId[c][(t >> 6) & 3] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
    if (t == 0)
    {
        s.state[c] = st;
        s.nCalls[c] = o.calls;
        s.nSym[c] = o.nSym;
        s.nPkt[c] = o.nPkt;
        if (s.nSig) s.nSig[c] = o.nSig;
        s.end[c] = make_int2(st.state == ST_DATASYMBOLS ? st.symCount : -1, st.callCount);
#ifdef LORAHIP_WG_TIMELINE
        if (c < 16384)
        {
            gWgTimeline[c][0] = tl0; gWgTimeline[c][1] = wall_clock64();
            gWgTimeline[c][2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);   // HW_ID, XCC_ID
            gWgTimeline[c][3] = (unsigned long long)o.calls;
        }
#endif
    }
    if constexpr (RES)
    {
        resCalls += unsigned(o.calls);
        resMore = resMore || (len - st.pos >= 2 * N);       // stopped with samples left: a record buffer was full
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // the state and the rows are in L2 / in memory before the workgroup moves on
    }
    if (PERSIST || RES) __syncthreads();                    // the next channel reuses the exchange region and the reduction records
    if constexpr (RES)
    {
        c += gridDim.x;
        if (c >= s.nChannels)
        {
            // the step's last channel: report, and on to the next step
            residentLookAhead(s, step + 1u);                // (the relay wavefronts: the next step's message into the mirrors)
            if (t == 0) residentWgDone(s, sR, step, resCalls, resMore ? 1u : 0u);
            if (!nextStep()) break;
            c = blockIdx.x;
        }
    }
    } while (RES || (PERSIST && (c += gridDim.x) < s.nChannels));    // without PERSIST / RES there is no loop at all (it would cost registers)
}


// The grid. One workgroup per channel by default: the dispatcher hands every free slot the next channel. SF11 is the exception: its
// workgroup is TWO wavefronts, four workgroups per compute unit, and when workgroups of a second round are placed as slots free up
// one by one the unit can end with its two free wavefront slots on the SAME SIMD -- where the hardware does not place a workgroup
// (tools/wg_timeline.py, profiles/r04/s19_*: a slot stayed empty for 1.1 ms with a workgroup waiting for it; 2048 channels took
// 4.1-5.6 ms from launch to launch). There the grid is PERSISTENT: the resident number of workgroups, placed once on an empty
// device, each taking channel after channel (c += gridDim.x) -- with the alternating priority (lorahip_framemachine.h) they advance
// alike, so the static split leaves no tail: 3.94-3.97 ms every launch. s.maxBlocks: 0 = this default, < 0 = never persistent,
// > 0 = at most that many workgroups (lorahip_demod_set_stream_grid: tests, A/B).
template <class C>
static hipError_t launchStreamWideCfg(const StreamArgs &args, hipStream_t stream)
{
    const size_t smem = size_t(C::TWN + C::XW) * sizeof(float2) + 4 * sizeof(RedRec) + 2 * sizeof(float2) + 12 * sizeof(int) + FineDims<C::LOG2N>::BYTES;
    static unsigned long long attrDone = 0, attrDoneP = 0;
    static PerDeviceCount resident, residentP;
    StreamArgs s = args;
    if (s.nChannels == 0) return hipSuccess;
    int cap = s.maxBlocks;
    if (cap == 0 && C::LOG2N == 11)
    {
        const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStreamWide<C, true>), smem, attrDoneP);
        if (e != hipSuccess) return e;
        cap = residentWorkgroupsCached(residentP, reinterpret_cast<const void *>(demodStreamWide<C, true>), C::T, smem);
    }
    if (cap > 0 && s.nChannels > unsigned(cap))
    {
        const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStreamWide<C, true>), smem, attrDoneP);
        if (e != hipSuccess) return e;
        s.lastRoundFrom = 0;                                    // nobody is replaced: every workgroup takes its turn at the priority
        hipLaunchKernelGGL((demodStreamWide<C, true>), dim3(unsigned(cap)), dim3(C::T), smem, stream, s);
        return hipGetLastError();
    }
    const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStreamWide<C, false>), smem, attrDone);
    if (e != hipSuccess) return e;
    s.lastRoundFrom = lastRoundFrom(s.nChannels, residentWorkgroupsCached(resident, reinterpret_cast<const void *>(demodStreamWide<C, false>), C::T, smem));
    hipLaunchKernelGGL((demodStreamWide<C, false>), dim3(s.nChannels), dim3(C::T), smem, stream, s);
    return hipGetLastError();
}

//! the resident receiver's launch at SF11 / SF12 (see launchStreamResidentCfg in lorahip_streamkernel.h): refused unless every workgroup is
//! resident at once; with more channels than that every workgroup walks several per step
template <class C>
static hipError_t launchStreamResidentWideCfg(const StreamArgs &args, hipStream_t stream, unsigned *gridOut)
{
    const size_t smem = size_t(C::TWN + C::XW) * sizeof(float2) + 4 * sizeof(RedRec) + 2 * sizeof(float2) + 12 * sizeof(int) + FineDims<C::LOG2N>::BYTES + sizeof(ResLds);
    static unsigned long long attrDone = 0;
    static PerDeviceCount resident;
    if (gridOut) *gridOut = 0;
    if (args.nChannels == 0) return hipErrorNotSupported;
    const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStreamWide<C, false, true>), smem, attrDone);
    if (e != hipSuccess) return e;
    const int res = residentWorkgroupsCached(resident, reinterpret_cast<const void *>(demodStreamWide<C, false, true>), C::T, smem);
    if (res <= 0) return hipErrorNotSupported;
    const unsigned perWg = (args.nChannels + unsigned(res) - 1) / unsigned(res);
    const unsigned grid = (args.nChannels + perWg - 1) / perWg;
    if (grid > 4095u) return hipErrorNotSupported;
    if (gridOut) *gridOut = grid;
    hipLaunchKernelGGL((demodStreamWide<C, false, true>), dim3(grid), dim3(C::T), smem, stream, args);
    return hipGetLastError();
}

hipError_t launchStreamResidentWide(const int sf, const StreamArgs &s, hipStream_t stream, unsigned *grid)
{
    return sf == 11 ? launchStreamResidentWideCfg<StreamWide11>(s, stream, grid) : launchStreamResidentWideCfg<StreamWide12>(s, stream, grid);
}

hipError_t launchStreamWide(const int sf, const StreamArgs &s, hipStream_t stream)
{
    return sf == 11 ? launchStreamWideCfg<StreamWide11>(s, stream) : launchStreamWideCfg<StreamWide12>(s, stream);
}

#endif // LORAHIP_NO_STREAM_WIDE

} // namespace lorahip

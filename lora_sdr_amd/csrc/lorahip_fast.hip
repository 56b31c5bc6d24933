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
#include "lorahip_fft.h"

namespace lorahip {

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
 *   X0ROT/X0PAD/X0S/X0D: exchange-0 LDS layout (found with tools/lds_conflicts.py): a row per low sample
 *   index n_low = VEC*t+u at element offset rotr(n_low, X0ROT)*(WPW*R+X0PAD) + ((n_low>>X0S)&1)*X0D,
 *   the WPW windows of a wave side by side inside the row (stride R), so that the writers' ds_write_b64
 *   groups and the readers' ds_read_b64 groups each tile the LDS banks exactly once.
 **********************************************************************/
template <int LOG2N_, int LOG2T_, int VEC_, int NPH_, int PB1_, int PB2_, int WAVES_PER_SIMD_,
          int X0ROT_, int X0PAD_, int X0S_, int X0D_, bool CH_LDS_, bool TW_ALL_LDS_, int PREFETCH_>
struct FastCfg
{
    static constexpr int PREFETCH = PREFETCH_;        // next window set's loads: 0 none (loaded at the top), 1 issued after the dechirp of this set, 2 at the top of this set
    static constexpr bool CH_LDS = CH_LDS_;           // chirp table read from LDS per window (else loop-invariant registers)
    static constexpr bool TW_ALL_LDS = TW_ALL_LDS_;   // last-phase twiddles from the LDS table too (else registers)
    static constexpr int LOG2N = LOG2N_, N = 1 << LOG2N_;
    static constexpr int LOG2T = LOG2T_, T = 1 << LOG2T_;     // lanes per window (<= 64)
    static constexpr int VEC = VEC_;                            // consecutive samples per lane per load
    static constexpr int P = N / T;                             // points per lane
    static constexpr int R = P / VEC;                           // phase-0 group size
    static constexpr int NPH = NPH_;
    static constexpr int WPW = 64 / T;                          // windows per wave iteration
    static constexpr int WAVES_PER_SIMD = WAVES_PER_SIMD_;
    static constexpr bool HAS_R2 = (LOG2N_ & 1);
    static constexpr int NL = VEC * T;                          // distinct n_low
    static constexpr int LOG2NL = LOG2N_ - PB1_;
    __host__ __device__ static constexpr int bound(const int j)
    {
        return j <= 0 ? 0 : (j == 1 ? PB1_ : (j == 2 ? (NPH_ == 2 ? LOG2N_ : PB2_) : LOG2N_));
    }
    static_assert((1 << PB1_) == R, "phase 0 must cover exactly the bits a lane loads");
    static_assert(((LOG2N_ - PB1_) & 1) == 0, "the low sample digits must be whole radix-4 digits");
    static_assert(T <= 64 && T >= 4, "a window lives inside one wavefront");
    // exchange 0
    static constexpr int RS0 = WPW * R + X0PAD_;
    __host__ __device__ static constexpr int x0off(const int nlow)
    {
        const int rot = ((nlow >> X0ROT_) | (nlow << (LOG2NL - X0ROT_))) & (NL - 1);
        return rot * RS0 + ((nlow >> X0S_) & 1) * X0D_;
    }
    static constexpr int X0ELEMS = NL * RS0 + X0D_;             // per wave
    // exchange 1 (3 phases): per window, element (rl, rh, col) at rh*X1 + col*R + rl
    static constexpr int G1 = 1 << (bound(2) - bound(1));
    static constexpr int X1 = G1 * R + 8;
    static constexpr int X1ELEMS = NPH_ == 3 ? WPW * (N / (G1 * R)) * X1 : 0;   // per wave
    static constexpr int XELEMS = (X0ELEMS > X1ELEMS ? X0ELEMS : X1ELEMS) + (N > X0ELEMS ? N - X0ELEMS : 0) / 2 * 0;
    //! twiddle entries staged in LDS: all stages below the last phase, or every stage
    static constexpr int TW_LDS = twStageOffset(LOG2N_, TW_ALL_LDS_ ? LOG2N_ : bound(NPH_ - 1));
    static constexpr int CH_ELEMS = CH_LDS_ ? N : 0;
};

/***********************************************************************
 * the kernel: one wave = WPW windows per iteration, persistent over window sets.
 * DBG = the optional "dec" / "fft" debug outputs are wanted (LoRaDemod.cpp:164, :154).
 * UNI = launch-uniform batch: one chirp selection for all windows and no moving fine-tune index
 *       (chirp_sel == NULL, fine_err == NULL): the steady-state shape, compiled without the rare paths.
 **********************************************************************/
template <class C, bool DBG, bool UNI>
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
    constexpr int XE = (C::X0ELEMS > C::X1ELEMS ? C::X0ELEMS : C::X1ELEMS);
    constexpr int FS = N + 8;                             // final-bin rows of the wave's windows (8 pad: 16-lane write groups tile the banks)
    constexpr int XW = (XE > WPW * FS ? XE : WPW * FS) + 2;   // v2f per wave; also holds WPW*N ints
    constexpr int M = N * LORAHIP_FINE_STEPS;
    constexpr int WAVES = 4;

    extern __shared__ __attribute__((aligned(16))) char smemRaw[];
    v2f *sTw = reinterpret_cast<v2f *>(smemRaw);                                   // [TW_LDS]
    v2f *sCh = sTw + ((C::TW_LDS + 1) & ~1);                                          // [CH_ELEMS]
    v2f *sX = sCh + C::CH_ELEMS;                                                      // [WAVES][XW]
    TailRec *sTail = reinterpret_cast<TailRec *>(sX + WAVES * XW);                       // [WAVES]

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // wave-uniform by construction: keep it in an SGPR
    const int wsub = lane >> LOG2T;                       // window inside the wave iteration
    const int t = lane & (T - 1);
    v2f *X = sX + wave * XW;                           // this wave's exchange region
    TailRec &tr = sTail[wave];

    const v2f *gIq = reinterpret_cast<const v2f *>(a.iq), *gDown = reinterpret_cast<const v2f *>(a.down);
    const v2f *gFine = reinterpret_cast<const v2f *>(a.fine);
    v2f *gDec = reinterpret_cast<v2f *>(a.decOut), *gFft = reinterpret_cast<v2f *>(a.fftOut);

    // ---- one-time set-up -------------------------------------------------------------
    tr.w[lane] = 0xffffffffu;                             // empty tail slots
    for (int i = threadIdx.x; i < C::TW_LDS; i += blockDim.x) sTw[i] = reinterpret_cast<const v2f *>(ft.twStage)[i];

    // register twiddles of the last phase: in the last phase klow = ci = t + T*g
    v2f twR[C::TW_ALL_LDS ? 1 : NGL][C::TW_ALL_LDS ? 1 : SLOTS];
#pragma unroll
    for (int g = 0; g < (C::TW_ALL_LDS ? 0 : NGL); g++)
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
                twR[g][slot] = reinterpret_cast<const v2f *>(ft.twStage)[base];
                twR[g][slot + 1] = reinterpret_cast<const v2f *>(ft.twStage)[base + (1 << b)];
                twR[g][slot + 2] = reinterpret_cast<const v2f *>(ft.twStage)[base + (2 << b)];
                slot += 3;
            }
    }

    // chirp table values of this lane's sample positions. One table serves both selections:
    // _upChirpTable = conj(_downChirpTable) entry by entry (LoRaDemod.cpp:103-104)
    const bool perWindowSel = !UNI && a.chirpSel != nullptr;
    const float s0 = (!perWindowSel && a.chirpSelAll == LORAHIP_CHIRP_UP) ? -1.0f : 1.0f;
    v2f ch[C::CH_LDS ? 1 : R][C::CH_LDS ? 1 : VEC];
    if (C::CH_LDS)
    {
        for (int i = threadIdx.x; i < N; i += blockDim.x)
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
                ch[r][u] = MAKE2(c.x, s0 * c.y);
            }
    }
    __syncthreads();

    const unsigned waveId = blockIdx.x * WAVES + wave;
    const unsigned waveCount = gridDim.x * WAVES;
    int pending = 0;                                       // tail records waiting in tr

    // coalesced window load: VEC*8 bytes per lane, the T lanes of a window contiguous, R rows
    v2f xn[R][VEC];
    auto issueLoads = [&](const unsigned set_)
    {
        const unsigned w_ = set_ * WPW + wsub;
        const unsigned wc_ = w_ < a.nWindows ? w_ : a.nWindows - 1;
        const v2f *in_ = gIq + (a.offsets ? a.offsets[wc_] : (long long)wc_ * a.stride);
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            const v2f *p = in_ + VEC * t + VEC * T * r;
            if (VEC == 2)
            {
                const v4f q = *reinterpret_cast<const v4f *>(p);
                xn[r][0] = MAKE2(q.x, q.y);
                xn[r][VEC - 1] = MAKE2(q.z, q.w);
            }
            else xn[r][0] = *p;
        }
    };
    if (C::PREFETCH && waveId < nSets) issueLoads(waveId);
    const v2f fconst0 = gFine[0];

    for (unsigned set = waveId; set < nSets; set += waveCount)
    {
        const unsigned w = set * WPW + wsub;
        const bool active = w < a.nWindows;
        const unsigned wc = active ? w : a.nWindows - 1;  // inactive lanes redo the last window, results dropped
        const int sel = perWindowSel ? a.chirpSel[wc] : a.chirpSelAll;
        const int idx0 = a.fineIdx0 ? a.fineIdx0[wc] : 0;
        const float err = (!UNI && a.fineErr) ? a.fineErr[wc] : 0.0f;
        const bool dechirp = sel != LORAHIP_CHIRP_NONE;
        const float d = err * (float)LORAHIP_FINE_STEPS;
        const bool moving = dechirp && d != 0.0f;
        const bool anyMoving = !UNI && __any(moving);

        // ---- samples of this set: loaded here, or already in flight since the previous iteration ---
        v2f x[R][VEC];
        if (!C::PREFETCH) issueLoads(set);
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < VEC; u++) x[r][u] = xn[r][u];
        if (C::PREFETCH == 2) issueLoads(set + waveCount < nSets ? set + waveCount : nSets - 1);

        // ---- fine-tune index chain for windows whose index moves (rare path) -----------------
        int *sIdx = reinterpret_cast<int *>(X) + wsub * N;   // aliases the exchange region (free until phase 0 ends)
        if (!UNI && anyMoving)
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
        // fine-tune entry of this window: constant over the launch when no per-window index is given
        const v2f fconst = a.fineIdx0 ? gFine[idx0] : fconst0;
        v2f cw[R][VEC];                                 // chirp values of this lane's samples
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
            // launch-uniform table selection, constant fine-tune entry: the hot path
            if (a.chirpSelAll != LORAHIP_CHIRP_NONE)
            {
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int u = 0; u < VEC; u++) x[r][u] = cmulv(cmulv(x[r][u], cw[r][u]), fconst);
            }
        }
        else
        {
            const float sgn = (perWindowSel && sel == LORAHIP_CHIRP_UP) ? -1.0f : 1.0f;
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++)
                {
                    const v2f c = MAKE2(cw[r][u].x, sgn * cw[r][u].y);
                    v2f f = fconst;
                    if (anyMoving && moving) f = gFine[sIdx[VEC * t + u + VEC * T * r]];
                    const v2f y = cmulv(cmulv(x[r][u], c), f);
                    x[r][u] = dechirp ? y : x[r][u];
                }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (DBG && a.decOut && active)
        {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++) gDec[(size_t)w * N + VEC * t + u + VEC * T * r] = x[r][u];
        }

        // ---- phase 0: bits [0, B1) in registers, one group per u -----------------------------
        // register r holds sample index high part a = r; its work-array position low bits are rev(a)
        v2f v0[VEC][R];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < VEC; u++) v0[u][Plan<LOG2N>::pos(VEC * T * r) & (R - 1)] = x[r][u];
        // next set's samples go in flight now and land while this set is transformed (past the end:
        // re-read the last set, harmless and branch-free)
        if (C::PREFETCH == 1) issueLoads(set + waveCount < nSets ? set + waveCount : nSets - 1);
#pragma unroll
        for (int u = 0; u < VEC; u++) runPhase<LOG2N, 0, B1, false>(v0[u], 0, sTw, nullptr);

        // ---- exchange 0: one row per n_low = VEC*t+u, the wave's windows side by side ----------
        {
            v2f *Xw = X + wsub * R;
#pragma unroll
            for (int u = 0; u < VEC; u++)
            {
                v2f *row = Xw + C::x0off(VEC * t + u);
#pragma unroll
                for (int e = 0; e < R; e++) row[e] = v0[u][e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        v2f vl[NGL][GL];                                // last-phase registers
        if (NPH == 2)
        {
            // phase 1 = last: group g has klow = ci = t + T*g (< R), element e <-> hp = e, n_low = rev4(hp)
            const v2f *Xr = X + wsub * R;
#pragma unroll
            for (int g = 0; g < NGL; g++)
#pragma unroll
                for (int e = 0; e < GL; e++) vl[g][e] = Xr[C::x0off(rev4(e, LOG2N - B1)) + (t + T * g)];
        }
        else
        {
            // phase 1 (middle): bits [B1, B2); ci = t + T*g; klow = ci mod R; high = ci >> B1
            constexpr int G1 = C::G1;
            constexpr int NG1 = P / G1;
            constexpr int HB = LOG2N - B2;                 // bits of `high`
            v2f v1[NG1][G1];
            const v2f *Xr = X + wsub * R;
#pragma unroll
            for (int g = 0; g < NG1; g++)
            {
                const int ci = t + T * g;
                const int klow = ci & (R - 1), high = ci >> B1;
                const int rhigh = rev4(high, HB);
                // hp = e + G1*high; n_low = rev4(hp) = rev4(e) << HB | rev4(high)
#pragma unroll
                for (int e = 0; e < G1; e++) v1[g][e] = Xr[C::x0off((rev4(e, B2 - B1) << HB) | rhigh) + klow];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < NG1; g++) runPhase<LOG2N, B1, B2, false>(v1[g], (t + T * g) & (R - 1), sTw, nullptr);
            // exchange 1: element (rl, rh, col) of this window at rh*X1 + col*R + rl
            v2f *X1w = X + wsub * (GL * C::X1);
#pragma unroll
            for (int g = 0; g < NG1; g++)
            {
                const int ci = t + T * g;
                v2f *base = X1w + (ci >> B1) * C::X1 + (ci & (R - 1));
#pragma unroll
                for (int e = 0; e < G1; e++) base[e * R] = v1[g][e];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // phase 2 = last: ci' = t + T*g = col*R + rl, element e2 = rh
#pragma unroll
            for (int g = 0; g < NGL; g++)
#pragma unroll
                for (int e = 0; e < GL; e++) vl[g][e] = X1w[e * C::X1 + (t + T * g)];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < NGL; g++)
        {
            if (C::TW_ALL_LDS) runPhase<LOG2N, BL, LOG2N, false>(vl[g], t + T * g, sTw, nullptr);
            else runPhase<LOG2N, BL, LOG2N, true>(vl[g], 0, nullptr, twR[C::TW_ALL_LDS ? 0 : g]);
        }

        // ---- final bins into the (now free) exchange region: the peak's neighbours are fetched from there ----
        v2f *F = X + wsub * FS;
#pragma unroll
        for (int e = 0; e < GL; e++)
#pragma unroll
            for (int g = 0; g < NGL; g++) F[(t + T * g) + (e << BL)] = vl[g][e];

        // ---- scan (LoRaDetector.hpp:36-48): bin = ci + 2^BL * e, ascending in (e, g) -----------
        float bestV = 0.0f;
        int bestJ = 0;                                     // element number e*NGL + g of the lane's best bin
        double tot = 0.0;
#pragma unroll
        for (int e = 0; e < GL; e++)
#pragma unroll
            for (int g = 0; g < NGL; g++)
            {
                const v2f bin = vl[g][e];
                if (DBG && a.fftOut && active) gFft[(size_t)w * N + (t + T * g) + (e << BL)] = bin;
                const float mag2 = bin.x * bin.x + bin.y * bin.y;
                tot += (double)mag2;
                if (mag2 > bestV) { bestV = mag2; bestJ = e * NGL + g; }
            }
        int bestI = (t + T * (bestJ & (NGL - 1))) + ((bestJ / NGL) << BL);
        if (!(bestV > 0.0f)) bestI = 0;
        groupArgmax<T>(bestV, bestI);
#pragma unroll
        for (int off = T / 2; off > 0; off >>= 1) tot += __shfl_xor(tot, off, 64);
        // every lane of the window now holds the window's (bestV, bestI); the xor tree adds the same
        // fp64 partials in the same pairing on all lanes, so tot is identical on all of them too

        // ---- defer the log/sqrt tail: one record per window, flushed 64 at a time ---------------
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (t == 0 && active)
        {
            // neighbours of the peak for fIndex (LoRaDetector.hpp:56-57)
            const int s = pending + wsub;
            tr.w[s] = w; tr.idx[s] = bestI; tr.val[s] = bestV; tr.tot[s] = tot;
            tr.l[s] = F[(bestI + N - 1) & (N - 1)]; tr.r[s] = F[(bestI + 1) & (N - 1)];
        }
        pending += WPW;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (pending == 64)
        {
            const unsigned ww = tr.w[lane];
            if (ww < a.nWindows) detectTail(a, ww, tr.idx[lane], tr.val[lane], tr.tot[lane], tr.l[lane], tr.r[lane]);
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
static size_t smemBytes()
{
    constexpr int WAVES = 4;
    constexpr int XE = (C::X0ELEMS > C::X1ELEMS ? C::X0ELEMS : C::X1ELEMS);
    constexpr int XW = (XE > C::WPW * (C::N + 8) ? XE : C::WPW * (C::N + 8)) + 2;
    return size_t(((C::TW_LDS + 1) & ~1) + C::CH_ELEMS) * sizeof(float2) + size_t(WAVES) * XW * sizeof(float2) + size_t(WAVES) * sizeof(TailRec);
}

template <class C, bool DBG, bool UNI>
static hipError_t launchOne(const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    constexpr int WAVES = 4;
    const size_t smem = smemBytes<C>();
    static bool attrSet = false;
    if (!attrSet)
    {
        const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(detectFast<C, DBG, UNI>),
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
    hipLaunchKernelGGL((detectFast<C, DBG, UNI>), dim3(grid), dim3(WAVES * 64), smem, stream, a, ft, nSets);
    return hipGetLastError();
}

template <class C>
static hipError_t launchCfg(const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    const bool uni = a.chirpSel == nullptr && a.fineErr == nullptr;
    if (a.decOut || a.fftOut) return launchOne<C, true, false>(a, ft, stream);
    return uni ? launchOne<C, false, true>(a, ft, stream) : launchOne<C, false, false>(a, ft, stream);
}

//              LOG2N T VEC NPH PB1 PB2 w/SIMD  X0: ROT PAD S  D   chLDS twLDS prefetch
typedef FastCfg<7,  3, 2,  2,  3,  7,  3,          1,  1,  0, 0,  true,  true,  false> Cfg7a;   // 8 lanes x 16 pts: [R2,4] X [4,4]
typedef FastCfg<7,  3, 2,  2,  3,  7,  3,          1,  1,  0, 0,  true,  true,  true>  Cfg7b;
typedef FastCfg<7,  3, 2,  2,  3,  7,  2,          1,  1,  0, 0,  true,  true,  true>  Cfg7c;
typedef FastCfg<7,  3, 2,  2,  3,  7,  4,          1,  1,  0, 0,  true,  true,  false> Cfg7d;
typedef FastCfg<7,  3, 2,  2,  3,  7,  2,          1,  1,  0, 0,  false, false, true>  Cfg7e;
typedef FastCfg<7,  3, 2,  2,  3,  7,  3,          1,  1,  0, 0,  true,  true,  2>     Cfg7f;   // loads issued at the top of the set
typedef FastCfg<8,  4, 1,  2,  4,  8,  3,          0,  1,  0, 0,  true,  true,  true>  Cfg8;
typedef FastCfg<8,  4, 1,  2,  4,  8,  3,          0,  1,  0, 0,  true,  true,  2>     Cfg8f;    // 16 lanes x 16 pts: [4,4] X [4,4]
typedef FastCfg<9,  5, 2,  3,  3,  7,  3,          2,  1,  1, 8,  true,  true,  true>  Cfg9;    // 32 lanes x 16 pts: [R2,4] X [4,4] X [4]
typedef FastCfg<9,  5, 2,  3,  3,  7,  3,          2,  1,  1, 8,  true,  true,  2>     Cfg9f;
typedef FastCfg<10, 6, 1,  3,  4,  8,  3,          0,  1,  0, 0,  true,  true,  true>  Cfg10;
typedef FastCfg<10, 6, 1,  3,  4,  8,  3,          0,  1,  0, 0,  true,  true,  2>     Cfg10f;   // 64 lanes x 16 pts: [4,4] X [4,4] X [4]

bool fastAvailable(const int sf) { return sf >= 7 && sf <= 10; }

hipError_t launchFast(const int sf, const int variant, const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    switch (sf)
    {
    case 7:
        switch (variant)
        {
        case 2: return launchCfg<Cfg7a>(a, ft, stream);
        case 3: return launchCfg<Cfg7c>(a, ft, stream);
        case 4: return launchCfg<Cfg7d>(a, ft, stream);
        case 5: return launchCfg<Cfg7e>(a, ft, stream);
        case 6: return launchCfg<Cfg7f>(a, ft, stream);
        default: return launchCfg<Cfg7b>(a, ft, stream);
        }
    case 8: return variant == 6 ? launchCfg<Cfg8f>(a, ft, stream) : launchCfg<Cfg8>(a, ft, stream);
    case 9: return variant == 6 ? launchCfg<Cfg9f>(a, ft, stream) : launchCfg<Cfg9>(a, ft, stream);
    case 10: return variant == 6 ? launchCfg<Cfg10f>(a, ft, stream) : launchCfg<Cfg10>(a, ft, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace lorahip

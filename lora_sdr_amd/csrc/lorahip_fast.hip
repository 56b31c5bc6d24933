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
#include "lorahip_fastcore.h"

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
 * the kernel: one wave = WPW windows per iteration, persistent over window sets.
 * DBG = the optional "dec" / "fft" debug outputs are wanted (LoRaDemod.cpp:164, :154).
 * UNI = launch-uniform batch: one chirp selection for all windows and no moving fine-tune index
 *       (chirp_sel == NULL, fine_err == NULL): the steady-state shape, compiled without the rare paths.
 **********************************************************************/
#ifdef LORAHIP_WG_TIMELINE     // profiling build (tools/wave_timeline.py): when and where every wavefront of the last batch launch ran
__device__ unsigned long long gWaveTimeline[16384][4];
extern "C" int lorahip_debug_wave_timeline(void *out, const size_t bytes)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(gWaveTimeline), bytes < sizeof(gWaveTimeline) ? bytes : sizeof(gWaveTimeline)) == hipSuccess ? 0 : -1;
}
#endif

template <class C, bool DBG, bool UNI>
__global__ void __launch_bounds__(256, C::WAVES_PER_SIMD)
detectFast(const DetectArgs a, const FastTables ft, const unsigned nSets)
{
#ifdef LORAHIP_WG_TIMELINE
    const unsigned long long tl0 = wall_clock64();
    unsigned long long tlLoop = 0;
    unsigned tlSets = 0;
#endif
    typedef FastCore<C> K;
    constexpr int N = C::N, T = C::T, VEC = C::VEC, R = C::R, WPW = C::WPW;
    constexpr int LOG2T = C::LOG2T;
    constexpr int NGL = C::NGL, GL = C::GL;
    constexpr int FS = C::FS, XW = C::XW;
    constexpr int WAVES = 4;

    extern __shared__ __attribute__((aligned(16))) char smemRaw[];
    v2f *sTw = reinterpret_cast<v2f *>(smemRaw);                                   // [TW_LDS]
    v2f *sCh = sTw + C::TWN;                                                          // [CH_ELEMS]
    v2f *sX = sCh + C::CH_ELEMS;                                                      // [WAVES][XW]
    TailRec *sTail = reinterpret_cast<TailRec *>(sX + WAVES * XW);                       // [WAVES]
    double2 *sFine = reinterpret_cast<double2 *>(sTail + WAVES);                         // split fine-tune tables (non-UNI kernels)
    static_assert(((size_t(C::TWN + C::CH_ELEMS + WAVES * XW) * sizeof(v2f) + WAVES * sizeof(TailRec)) & 15) == 0, "the split tables are read with ds_read_b128");

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

    typename K::TwR twR;
    K::loadTwR(twR, reinterpret_cast<const v2f *>(ft.twStage), t);
    typename K::TwM twM;
    K::loadTwM(twM, reinterpret_cast<const v2f *>(ft.twStage), t);
    FineLds fl;
    fl.A = nullptr; fl.B = nullptr; fl.split = false;
    if (!UNI) fl = fineLoadLds<C::LOG2N>(sFine, a.fineA, a.fineB, threadIdx.x, blockDim.x);

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

    // which window sets this wave walks: set = first, first + waveCount, ... < last
    unsigned waveId = blockIdx.x * WAVES + wave, waveCount = gridDim.x * WAVES, setEnd = nSets;
    if (C::XCD_CONTIG && gridDim.x >= 8 && (gridDim.x & 7) == 0)
    {
        // workgroups are dealt round-robin to the 8 XCDs: give each XCD one contiguous eighth of the batch
        const unsigned xcd = blockIdx.x & 7, perXcd = (nSets + 7) / 8;
        waveCount = (gridDim.x >> 3) * WAVES;
        waveId = xcd * perXcd + (blockIdx.x >> 3) * WAVES + wave;
        setEnd = (xcd + 1) * perXcd < nSets ? (xcd + 1) * perXcd : nSets;
    }
    int pending = 0;                                       // tail records waiting in tr

    v2f xn[R][VEC];
    auto issueLoads = [&](const unsigned set_)
    {
        const unsigned w_ = set_ * WPW + wsub;
        const unsigned wc_ = w_ < a.nWindows ? w_ : a.nWindows - 1;
        K::load(xn, gIq + (a.offsets ? a.offsets[wc_] : (long long)wc_ * a.stride), t);
    };
    if (C::PREFETCH && waveId < setEnd) issueLoads(waveId);
    const v2f fconst0 = gFine[0];

    const int prioSlot = wavefrontSlot();
#ifdef LORAHIP_WG_TIMELINE
    tlLoop = wall_clock64();
#endif
    for (unsigned set = waveId; set < setEnd; set += waveCount)
    {
#ifdef LORAHIP_WG_TIMELINE
        tlSets++;
#endif
        rotatePriority<C::WAVES_PER_SIMD, LORAHIP_PRIO_BATCH>(prioSlot);
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
        if (C::PREFETCH == 2) issueLoads(set + waveCount < setEnd ? set + waveCount : nSets - 1);

        // ---- fine-tune indices of this lane's samples for windows whose index moves (LoRaDemod.cpp:160-162): closed form
        // (lorahip_fine.h); a wave that holds a window where the form does not apply walks the exact chain instead
        int *sIdx = reinterpret_cast<int *>(X) + wsub * N;   // aliases the exchange region (free until phase 0 ends)
        unsigned yv[R][VEC];
        if constexpr (!UNI) if (anyMoving)
        {
            const FinePlan pl = finePlan(moving ? d : 0.0f, K::M);
            const unsigned ymax = fineLaneIndices<C::LOG2N, VEC, T, R>(idx0, pl, t, yv);
            int idxEnd = fineEndIndex(idx0, pl, C::LOG2N, C::LOG2N + 7);
            if (__any(!pl.regular || ymax == (unsigned)K::M))
            {
                idxEnd = K::fineChain(idx0, moving ? d : 0.0f, t, sIdx);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int u = 0; u < VEC; u++) yv[r][u] = (unsigned)sIdx[K::idxSlot(VEC * t + u + VEC * T * r)];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (moving && t == 0 && a.fineIdxOut && active) a.fineIdxOut[w] = idxEnd;
        }
        if (!moving && t == 0 && active && a.fineIdxOut) a.fineIdxOut[w] = idx0;

        // ---- dechirp: (samp * chirp) * fine   (LoRaDemod.cpp:159) ---------------------------
        // fine-tune entry of this window: constant over the launch when no per-window index is given
        const v2f fconst = a.fineIdx0 ? gFine[idx0] : fconst0;
        v2f cw[R][VEC];                                 // chirp values of this lane's samples
        if (C::CH_LDS) K::chirpFromLds(cw, sCh, t);
        else
        {
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++) cw[r][u] = ch[C::CH_LDS ? 0 : r][C::CH_LDS ? 0 : u];
        }
        if (UNI || (!perWindowSel && !anyMoving))
        {
            // launch-uniform table selection, constant fine-tune entry: the hot path
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
            const auto chirpRaw = [&](const int i) { return cwf[i]; };
            if (anyMoving)
            {
                // yv = idx0 in the windows that do not move. A launch-uniform table selection is already in the values (s0 above):
                // no per-sample sign multiply then
                if (perWindowSel) dechirpFine<fineSplitLog2H(C::LOG2N), R * VEC>(&x[0][0], chirpOf, &yv[0][0], fl, gFine, dechirp);
                else dechirpFine<fineSplitLog2H(C::LOG2N), R * VEC>(&x[0][0], chirpRaw, &yv[0][0], fl, gFine, dechirp);
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

        // ---- phases / exchanges; the next set's samples go in flight once phase 0's inputs are staged and land
        // while this set is transformed (past the end: re-read the last set, harmless and branch-free)
        v2f vl[NGL][GL];
        K::fft(x, X, wsub, t, sTw, twR, vl, [&]() { if (C::PREFETCH == 1) issueLoads(set + waveCount < setEnd ? set + waveCount : nSets - 1); }, &twM);

        // ---- scan (LoRaDetector.hpp:36-48); final bins into the (now free) exchange region for the neighbour fetch
        v2f *F = X + wsub * FS;
        float bestV;
        int bestI;
        double tot;
        K::template scan<true, UNI ? 4 : 1>(vl, F, (DBG && a.fftOut && active) ? gFft + (size_t)w * N : nullptr, t, bestV, bestI, tot);

        // ---- defer the log/sqrt tail: one record per window, flushed 64 at a time ---------------
        v2f leftBin, rightBin;
        K::neighbours(vl, F, bestI, lane, t, leftBin, rightBin);
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
#ifdef LORAHIP_WG_TIMELINE
    if (lane == 0 && blockIdx.x * WAVES + wave < 16384)
    {
        unsigned long long *r = gWaveTimeline[blockIdx.x * WAVES + wave];
        r[0] = tl0; r[1] = wall_clock64();
        r[2] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4) | ((unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) << 32);
        r[3] = (unsigned long long)tlSets | ((tlLoop - tl0) << 32);
    }
#endif
}

/***********************************************************************
 * launch
 **********************************************************************/
template <class C>
static size_t smemBytes(const bool withFine)
{
    constexpr int WAVES = 4;
    return size_t(C::TWN + C::CH_ELEMS) * sizeof(float2) + size_t(WAVES) * C::XW * sizeof(float2) + size_t(WAVES) * sizeof(TailRec) +
           (withFine ? FineDims<C::LOG2N>::BYTES : 0);
}

template <class C, bool DBG, bool UNI>
static hipError_t launchOne(const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    constexpr int WAVES = 4;
    const size_t smem = smemBytes<C>(!UNI);
    static unsigned long long attrDone = 0;
    {
        const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(detectFast<C, DBG, UNI>), smem, attrDone);
        if (e != hipSuccess) return e;
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

/***********************************************************************
 * configurations: the geometry of an SF is fixed (lanes per window, vector width, phases, the exchange-0 LDS layout found
 * with tools/lds_conflicts.py); what varies between the selectable variants is a set of options.
 **********************************************************************/
template <int SF> struct Geo;
//                                         LOG2T VEC NPH PB1 PB2   X0: ROT PAD S  D
template <> struct Geo<6>  { enum { LOG2T = 2, VEC = 4, NPH = 2, PB1 = 2, PB2 = 6, ROT = 2, PAD = 1, S = 0, D = 0 }; };   //  4 lanes x 16 pts: [4] X [4,4]
template <> struct Geo<7>  { enum { LOG2T = 3, VEC = 2, NPH = 2, PB1 = 3, PB2 = 7, ROT = 1, PAD = 1, S = 0, D = 0 }; };   //  8 lanes x 16 pts: [R2,4] X [4,4]
template <> struct Geo<8>  { enum { LOG2T = 4, VEC = 1, NPH = 2, PB1 = 4, PB2 = 8, ROT = 0, PAD = 1, S = 0, D = 0 }; };   // 16 lanes x 16 pts: [4,4] X [4,4]
template <> struct Geo<9>  { enum { LOG2T = 5, VEC = 2, NPH = 3, PB1 = 3, PB2 = 7, ROT = 2, PAD = 1, S = 1, D = 8 }; };   // 32 lanes x 16 pts: [R2,4] X [4,4] X [4]
template <> struct Geo<10> { enum { LOG2T = 6, VEC = 1, NPH = 3, PB1 = 4, PB2 = 8, ROT = 0, PAD = 1, S = 0, D = 0 }; };   // 64 lanes x 16 pts: [4,4] X [4,4] X [4]

enum : unsigned
{
    W2 = 1u << 0, W4 = 1u << 1,         // waves per SIMD the register budget is set for (default 3)
    CH_REG = 1u << 2,                    // chirp values of the lane's sample positions in registers (default: LDS copy of the table)
    TW_REG = 1u << 3,                    // last-phase twiddles in registers (default: LDS table)
    PF_NONE = 1u << 4, PF_EARLY = 1u << 5,   // next set's loads: none / at the top of the set (default: after the dechirp)
    NT = 1u << 6,                        // non-temporal IQ loads
    NB_SEL = 1u << 7,                    // peak's neighbours by register select (default: bins staged in LDS)
    X1_SWAP = 1u << 8,                   // exchange 1 by row swaps / DPP (SF9, SF10)
    TWM_REG = 1u << 9,                   // middle-phase twiddles in registers
    XCD = 1u << 10,                      // XCD-contiguous walk over the batch
    W1 = 1u << 11                        // one wave per SIMD: the 512-register budget (what does not fit 256 lands in AGPRs, not scratch)
};
template <int SF, unsigned O>
using Fast = FastCfg<SF, Geo<SF>::LOG2T, Geo<SF>::VEC, Geo<SF>::NPH, Geo<SF>::PB1, Geo<SF>::PB2, (O & W2) ? 2 : (O & W4) ? 4 : 3,
                     Geo<SF>::ROT, Geo<SF>::PAD, Geo<SF>::S, Geo<SF>::D, !(O & CH_REG), !(O & TW_REG), (O & PF_NONE) ? 0 : (O & PF_EARLY) ? 2 : 1,
                     (O & NT) != 0, (O & NB_SEL) != 0, (O & X1_SWAP) != 0, (O & TWM_REG) != 0, (O & XCD) != 0>;

// second SF9 geometry: 16 lanes x 32 points, two phases [R2,4,4] X [4,4] -- one LDS exchange, no exchange 1; at the 256-register budget of
// two waves per SIMD. +3.6 % on launch-uniform batches, -6 % where every window carries its own settings (spills): the default picks per call
template <unsigned O>
using Fast9b = FastCfg<9, 4, 1, 2, 5, 9, (O & W2) ? 2 : (O & W4) ? 4 : 3, 0, 1, 0, 0, !(O & CH_REG), !(O & TW_REG), (O & PF_NONE) ? 0 : (O & PF_EARLY) ? 2 : 1,
                       (O & NT) != 0, (O & NB_SEL) != 0, false, false, false>;

// SF11 inside one wavefront: 64 lanes x 32 points, phases [R2,4,4] X [4] X [4,4], both exchanges wave-local (no workgroup barrier)
template <unsigned O>
using Fast11q = FastCfg<11, 6, 1, 3, 5, 7, (O & W1) ? 1 : (O & W2) ? 2 : (O & W4) ? 4 : 3, 0, 1, 0, 0, !(O & CH_REG), !(O & TW_REG), (O & PF_NONE) ? 0 : (O & PF_EARLY) ? 2 : 1,
                        (O & NT) != 0, (O & NB_SEL) != 0, false, (O & TWM_REG) != 0, false>;

// one LDS exchange less for the long windows, as two-phase geometries of 64 points per lane (profiling variants;
// profiles/r03/README.md has their register / scratch / LDS-instruction counts and the A/B):
//   SF10: 16 lanes x 64 points, [4,4,4] X [4,4]            (4 windows per wavefront)
//   SF11: 32 lanes x 64 points, [R2,4,4] X [4,4,4]         (2 windows per wavefront)
//   SF12: 64 lanes x 64 points, [4,4,4] X [4,4,4]          (1 window per wavefront, no workgroup barrier)
template <unsigned O>
using Fast10b = FastCfg<10, 4, 1, 2, 6, 10, (O & W1) ? 1 : (O & W2) ? 2 : 3, 0, 1, 0, 0, !(O & CH_REG), !(O & TW_REG), (O & PF_NONE) ? 0 : (O & PF_EARLY) ? 2 : 1,
                        (O & NT) != 0, (O & NB_SEL) != 0, false, false, false>;
template <unsigned O>
using Fast11b = FastCfg<11, 5, 2, 2, 5, 11, (O & W1) ? 1 : (O & W2) ? 2 : 3, 0, 1, 0, 0, !(O & CH_REG), !(O & TW_REG), (O & PF_NONE) ? 0 : (O & PF_EARLY) ? 2 : 1,
                        (O & NT) != 0, (O & NB_SEL) != 0, false, false, false>;
template <unsigned O>
using Fast12b = FastCfg<12, 6, 1, 2, 6, 12, (O & W1) ? 1 : (O & W2) ? 2 : 3, 0, 1, 0, 0, !(O & CH_REG), !(O & TW_REG), (O & PF_NONE) ? 0 : (O & PF_EARLY) ? 2 : 1,
                        (O & NT) != 0, (O & NB_SEL) != 0, false, false, false>;

bool fastAvailable(const int sf) { return sf >= 6 && sf <= 10; }

//! host-side check of a configuration's exchange-0 layout: every (row, window, element) has its own word inside the
//! wave's region
template <class C>
static bool layoutOk()
{
    std::vector<char> used(size_t(C::XW), 0);
    for (int n = 0; n < C::NL; n++)
        for (int ws = 0; ws < C::WPW; ws++)
            for (int e = 0; e < C::R; e++)
            {
                const int a = C::x0off(n) + ws * C::R + e;
                if (a < 0 || a >= C::XW || used[size_t(a)]) return false;
                used[size_t(a)] = 1;
            }
    return true;
}

bool fastLayoutsOk()
{
    bool ok = layoutOk<Fast<6, 0>>() && layoutOk<Fast<7, 0>>() && layoutOk<Fast<8, 0>>() && layoutOk<Fast<9, 0>>() && layoutOk<Fast9b<0>>() && layoutOk<Fast<10, 0>>();
#ifdef LORAHIP_ALL_VARIANTS
    ok = ok && layoutOk<Fast11q<0>>() && layoutOk<Fast10b<0>>() && layoutOk<Fast11b<0>>() && layoutOk<Fast12b<0>>();
#endif
    return ok;
}

/***********************************************************************
 * selectable variants (lorahip_set_variant): 0 = the measured best per SF (profiles/r01/s8_variants.txt keeps every A/B
 * pair); the numbers of the others are stable, tests/test_gpu_parity.py runs every one of them against the CPU restatement of the reference.
 **********************************************************************/
typedef hipError_t (*FastLaunch)(const DetectArgs &, const FastTables &, hipStream_t);

/***********************************************************************
 * defaults: the option set depends on the SHAPE of the call. Launch-uniform batches (one chirp selection, no moving fine-tune
 * index: the steady state bench.py times) run the kernels tuned in round 1. Batches with per-window settings carry the
 * closed-form index arithmetic and the fp64 table product per sample (lorahip_fine.h): at three waves per SIMD (168 registers)
 * those kernels spill from SF8 up, so they get their own option sets (profiles/r02/s3_explore_moving_variants.txt:
 * SF9 0.20 -> 0.36 of the HBM roofline, SF10 0.20 -> 0.34).
 **********************************************************************/
template <class UNI_CFG, class MOVING_CFG, class DBG_CFG>
static hipError_t launchByShape(const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    if (a.decOut || a.fftOut) return launchOne<DBG_CFG, true, false>(a, ft, stream);
    const bool uni = a.chirpSel == nullptr && a.fineErr == nullptr;
    return uni ? launchOne<UNI_CFG, false, true>(a, ft, stream) : launchOne<MOVING_CFG, false, false>(a, ft, stream);
}

// (the option sets of the per-window-settings instances of SF7 / SF10 as macros: A/B builds, tools/build_variant.py -DLORAHIP_SF10_MOVING=...)
#ifndef LORAHIP_SF7_MOVING
#define LORAHIP_SF7_MOVING (CH_REG | NT)
#endif
#ifndef LORAHIP_SF10_MOVING
#define LORAHIP_SF10_MOVING (W2 | CH_REG | TW_REG | NT | X1_SWAP | TWM_REG)
#endif
struct FastVariant { int sf, variant; FastLaunch launch; };
#define V(SF, N, OPTS) { SF, N, &launchCfg<Fast<SF, (OPTS)>> }
// What ships: per SF the default (0) and ONE alternative -- variant 10, every table (chirp, twiddles) read from LDS, the option
// set the streaming demodulator's kernels run with -- beside the generic kernel (1, lorahip_kernels.hip). The losers of the
// round-1 tuning (profiles/r01/s8_variants.txt keeps every A/B pair) are compiled only with -DLORAHIP_ALL_VARIANTS
// (python -m lora_sdr_amd.build --all-variants), under their old numbers.
static const FastVariant kFastVariants[] = {
    // The debug-port instances (dec / fft outputs, 3x the traffic: not occupancy-bound) and, from SF8 up, the per-window-settings
    // instances run at the 256-register budget of two waves per SIMD: at three (168 registers) they spilled up to 296 B (SF9 / SF10
    // debug ports; tools/kernel_resources.py, profiles/r04). Every shipped instance now keeps <= 32 B of scratch.
    { 6, 0, &launchByShape<Fast<6, 0>, Fast<6, 0>, Fast<6, W2>> },                      // default: 16 windows per wave keep the LDS copies of chirp / twiddles cheap
#ifndef LORAHIP_FMA      // (the contracted build carries the defaults only)
    { 6, 10, &launchByShape<Fast<6, CH_REG | NT>, Fast<6, CH_REG | NT>, Fast<6, W2 | CH_REG | NT>> },
#endif
    { 7, 0, &launchByShape<Fast<7, CH_REG | NT>, Fast<7, LORAHIP_SF7_MOVING>, Fast<7, W2 | CH_REG | NT>> },   // default
#ifndef LORAHIP_FMA
    { 7, 10, &launchByShape<Fast<7, 0>, Fast<7, 0>, Fast<7, W2>> },
#endif
    // default SF8: uniform batches with the tables in registers at three waves per SIMD, per-window settings with them at two (no
    // prefetch: the third wave's latency hiding is worth less than the spills it costs -- session 30)
    { 8, 0, &launchByShape<Fast<8, CH_REG | TW_REG | NT>, Fast<8, W2 | CH_REG | TW_REG | NT | PF_NONE>, Fast<8, W2 | CH_REG | TW_REG | NT | PF_NONE>> },
#ifndef LORAHIP_FMA
    { 8, 10, &launchByShape<Fast<8, 0>, Fast<8, W2>, Fast<8, W2>> },
#endif
    // default SF9: uniform batches on the two-phase geometry (16 lanes x 32 points), per-window settings and the debug ports on the
    // three-phase one (32 lanes x 16 points) at two waves per SIMD
    { 9, 0, &launchByShape<Fast9b<W2 | CH_REG | TW_REG | NT | PF_NONE>, Fast<9, W2 | CH_REG | TW_REG | NT | X1_SWAP | TWM_REG>, Fast<9, W2 | CH_REG | TW_REG | NT | X1_SWAP | TWM_REG>> },
#ifndef LORAHIP_FMA
    { 9, 10, &launchByShape<Fast<9, 0>, Fast<9, W2>, Fast<9, W2>> },
#endif
    // default SF10: per-window settings at two waves per SIMD with the middle-phase twiddles in registers too
    { 10, 0, &launchByShape<Fast<10, CH_REG | TW_REG | NT | X1_SWAP>, Fast<10, LORAHIP_SF10_MOVING>, Fast<10, W2>> },   // (debug ports: every table from LDS, no scratch)
#ifndef LORAHIP_FMA
    { 10, 10, &launchByShape<Fast<10, 0>, Fast<10, W2>, Fast<10, W2>> },
#endif
#ifdef LORAHIP_ALL_VARIANTS
    V(6, 7, TW_REG), V(6, 8, NT), V(6, 11, CH_REG | NT), V(6, 12, CH_REG | TW_REG | NT), V(6, 15, CH_REG | TW_REG | NT | PF_NONE),
    V(7, 2, PF_NONE), V(7, 3, W2), V(7, 4, W4 | PF_NONE), V(7, 5, W2 | CH_REG | TW_REG), V(7, 6, PF_EARLY), V(7, 7, TW_REG),
    V(7, 8, NT), V(7, 9, TW_REG | NT), V(7, 11, CH_REG | NT), V(7, 12, W2 | CH_REG | TW_REG | NT),
    V(7, 13, CH_REG | NT | NB_SEL), V(7, 14, CH_REG | NT | XCD), V(7, 15, CH_REG | TW_REG | NT | PF_NONE), V(7, 16, CH_REG | TW_REG | NT | PF_NONE | NB_SEL),
    V(7, 17, W4 | CH_REG | NT | PF_NONE),
    V(8, 6, PF_EARLY), V(8, 7, TW_REG), V(8, 8, NT), V(8, 9, TW_REG | NT), V(8, 11, CH_REG | TW_REG | NT),
    V(8, 13, CH_REG | TW_REG | NT | NB_SEL), V(8, 15, CH_REG | TW_REG | NT | PF_NONE), V(8, 17, W4 | CH_REG | NT | PF_NONE),
    V(9, 6, PF_EARLY), V(9, 7, TW_REG), V(9, 8, NT), V(9, 9, TW_REG | NT), V(9, 11, CH_REG | TW_REG | NT),
    V(9, 12, CH_REG | TW_REG | NT | X1_SWAP), V(9, 13, CH_REG | TW_REG | NT | NB_SEL), V(9, 15, CH_REG | TW_REG | NT | X1_SWAP | TWM_REG | PF_NONE),
    V(9, 16, W2 | CH_REG | TW_REG | NT | X1_SWAP | TWM_REG),
    { 9, 20, &launchCfg<Fast9b<W2 | CH_REG | TW_REG | NT>> }, { 9, 21, &launchCfg<Fast9b<W2 | NT>> }, { 9, 22, &launchCfg<Fast9b<W2 | TW_REG | NT>> },
    { 9, 23, &launchCfg<Fast9b<W2 | CH_REG | NT>> }, { 9, 24, &launchCfg<Fast9b<W2 | CH_REG | TW_REG | NT | PF_NONE>> },
    // SF11 per wavefront (the default SF11 kernel is lorahip_wide.hip's)
    { 11, 20, &launchCfg<Fast11q<W2 | NT | PF_NONE>> }, { 11, 21, &launchCfg<Fast11q<W2 | TW_REG | NT | PF_NONE>> },
    { 11, 22, &launchCfg<Fast11q<W2 | CH_REG | TW_REG | NT | PF_NONE>> }, { 11, 23, &launchCfg<Fast11q<W2 | NT>> },
    { 11, 24, &launchCfg<Fast11q<W2 | TW_REG | TWM_REG | NT | PF_NONE>> },
    // round 3: two-phase geometries of 64 points per lane (all tables from LDS: 64 points leave no registers for them)
    { 10, 25, &launchCfg<Fast10b<W2 | NT | PF_NONE>> }, { 10, 26, &launchCfg<Fast10b<W1 | NT | PF_NONE>> },
    { 11, 25, &launchCfg<Fast11b<W2 | NT | PF_NONE>> }, { 11, 26, &launchCfg<Fast11b<W1 | NT | PF_NONE>> },
    { 12, 25, &launchCfg<Fast12b<W2 | NT | PF_NONE>> }, { 12, 26, &launchCfg<Fast12b<W1 | NT | PF_NONE>> },
    V(10, 6, PF_EARLY), V(10, 7, TW_REG), V(10, 8, NT), V(10, 9, TW_REG | NT), V(10, 11, CH_REG | TW_REG | NT),
    V(10, 12, CH_REG | TW_REG | NT | X1_SWAP), V(10, 13, CH_REG | TW_REG | NT | NB_SEL), V(10, 14, CH_REG | TW_REG | NT | NB_SEL | X1_SWAP),
    V(10, 15, CH_REG | TW_REG | NT | X1_SWAP | TWM_REG | PF_NONE), V(10, 16, W2 | CH_REG | TW_REG | NT | X1_SWAP | TWM_REG),
    // round 2: the defaults at the 256-register budget of two waves per SIMD (the per-window-settings kernels spill at three)
    V(7, 30, W2 | CH_REG | NT), V(8, 30, W2 | CH_REG | TW_REG | NT), V(9, 30, W2 | CH_REG | TW_REG | NT | X1_SWAP), V(9, 31, CH_REG | TW_REG | NT | X1_SWAP),
    V(10, 30, W2 | CH_REG | TW_REG | NT | X1_SWAP), V(10, 31, W2 | CH_REG | TW_REG | NT | X1_SWAP | PF_NONE), V(8, 31, W2 | CH_REG | TW_REG | NT | PF_NONE),
#endif
};
#undef V

hipError_t launchFast(const int sf, const int variant, const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    const FastVariant *def = nullptr;
    for (const FastVariant &v : kFastVariants)
    {
        if (v.sf != sf) continue;
        if (v.variant == variant) return v.launch(a, ft, stream);
        if (v.variant == 0) def = &v;
    }
    return def ? def->launch(a, ft, stream) : hipErrorInvalidValue;     // unknown numbers run the default
}

} // namespace lorahip

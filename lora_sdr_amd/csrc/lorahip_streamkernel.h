// The streaming demodulator kernel (demodStream) and its launcher as templates over a FastCfg: instantiated by lorahip_stream.hip
// (the geometries of a full device: 16 points per lane) and by lorahip_stream_lanes.hip (more lanes per channel, for receivers with
// fewer channels than the device holds wavefronts). What the kernel does: the head of lorahip_stream.hip.
#pragma once
#include "lorahip_fastcore.h"
#include "lorahip_framemachine.h"
#include "lorahip_residentproto.h"
#include <cstdlib>

#ifndef STREAM_STAGE_BINS
#define STREAM_STAGE_BINS 0       // untraced path: neighbours by register select (1: all bins staged in LDS whenever a channel is in FRAMESYNC; 1-1.7 % slower, profiles/r03)
#endif
#ifndef STREAM_SCAN_CHAINS
#define STREAM_SCAN_CHAINS 1
#endif
#ifndef STREAM_TWLDS
#define STREAM_TWLDS false      // last-phase twiddles from the LDS table instead of registers (A/B: frees ~30 registers)
#endif
#ifndef STREAM_TWLDS9
#define STREAM_TWLDS9 false
#endif
#ifndef LORAHIP_RES_NOCOPY
#define LORAHIP_RES_NOCOPY 1      // the resident receiver leaves the open packets' symbols in the carry rows (carryIn without the copy; the pack reads them there)
#endif
#ifndef STREAM_WPS
#define STREAM_WPS 2            // wavefronts per SIMD the register budget is set for (3 = 168 VGPRs: A/B builds, profiles/r03)
#endif
namespace lorahip {

//! PERSIST: a grid of at most s.maxBlocks workgroups, each looping over channel sets -- for launches over more channels than are
//! resident at once. The loop costs registers (96 / 112 B of scratch at SF7 / SF9 against 20 / 28), so a launch that fits the
//! device takes the instance without it (one workgroup per channel set).
//! RES: the resident receiver -- the launch stays, the steps arrive as messages (residentWait / residentStepEnd above). One workgroup
//! per channel set, all of them resident at once (the launcher checks); s.flags carries the carry bits only.
template <class C, bool PERSIST, bool RES = false>
__global__ void __launch_bounds__(256, C::WAVES_PER_SIMD)
demodStream(const StreamArgs s)
{
    static_assert(!(RES && PERSIST), "the resident receiver has one workgroup per channel set");
    typedef FastCore<C> K;
    constexpr int N = C::N, T = C::T, VEC = C::VEC, R = C::R, WPW = C::WPW;
    constexpr int LOG2T = C::LOG2T;
    constexpr int NGL = C::NGL, GL = C::GL;
    constexpr int FS = C::FS, XW = C::XW;
    constexpr int WAVES = 4;

    extern __shared__ __attribute__((aligned(16))) char smemRaw[];
    v2f *sTw = reinterpret_cast<v2f *>(smemRaw);          // [TWN]
    v2f *sCh = sTw + C::TWN;                                // [N] down-chirp table (the up-chirp is its conjugate)
    v2f *sX = sCh + N;                                      // [WAVES][XW]
    double2 *sFine = reinterpret_cast<double2 *>(sX + WAVES * XW);   // split fine-tune tables (lorahip_fine.h)
    static_assert(((size_t(C::TWN + N + WAVES * XW) * sizeof(v2f)) & 15) == 0, "the split tables are read with ds_read_b128");

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wsub = lane >> LOG2T;                        // channel inside the wavefront
    const int t = lane & (T - 1);
    v2f *X = sX + wave * XW;

    const v2f *gIq = reinterpret_cast<const v2f *>(s.iq), *gFine = reinterpret_cast<const v2f *>(s.fine);
    for (int i = threadIdx.x; i < C::TW_LDS; i += blockDim.x) sTw[i] = reinterpret_cast<const v2f *>(s.twStage)[i];
    for (int i = threadIdx.x; i < N; i += blockDim.x) sCh[i] = reinterpret_cast<const v2f *>(s.down)[i];
    typename K::TwR twR;
    K::loadTwR(twR, reinterpret_cast<const v2f *>(s.twStage), t);
    typename K::TwM twM;
    K::loadTwM(twM, reinterpret_cast<const v2f *>(s.twStage), t);
    const FineLds fl = fineLoadLds<C::LOG2N>(sFine, s.fineA, s.fineB, threadIdx.x, blockDim.x);
    typedef ResLds ResL;
    ResL *sR = reinterpret_cast<ResL *>(reinterpret_cast<char *>(sFine) + FineDims<C::LOG2N>::BYTES);        // RES only (the launcher adds the bytes)
    if (RES && threadIdx.x == 0) { sR->carry = s.carry; sR->carryCap = s.carryCap; }
    if (RES && threadIdx.x < RES_RING) { sR->calls[threadIdx.x] = 0; sR->arrive[threadIdx.x] = 0; sR->more[threadIdx.x] = 0; sR->msgSeq[threadIdx.x] = 0u; }
    if constexpr (RES)
    {
        // the census: the host rings the first step only when every workgroup is on the device (one that had to wait for a slot would wait
        // for ever: the others never leave)
        if (threadIdx.x == 0 && atomicAdd(&s.res->arrived, 1u) + 1u == gridDim.x) sysStore(&s.resHost->arrivedAll, 1u);
    }
    __syncthreads();

    // With one channel per wavefront (T = 64: SF10) everything the frame machine touches is WAVE-UNIFORM; saying so (v_readfirstlane
    // on what comes out of the vector unit) moves the machine -- state, 64-bit positions, record pointers, counters -- into scalar
    // registers and onto the scalar unit. With several channels per wavefront the state is replicated in each channel's T lanes.
    constexpr bool UNI = WPW == 1;
    const auto uniI = [](const int v) { return UNI ? __builtin_amdgcn_readfirstlane(v) : v; };
    const auto uniF = [](const float v) { return UNI ? __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))) : v; };

    // The grid is PERSISTENT: at most the resident number of workgroups (s.maxBlocks), each walking one set of WAVES * WPW channels
    // after the other -- the tables above are loaded once, and a launch over more channels than fit the device does not run a second,
    // half-empty round of workgroups. From here on the wavefronts of a workgroup are independent (no workgroup barrier below).
    const unsigned nSets = (s.nChannels + WAVES * WPW - 1) / (WAVES * WPW);
    // RES: one turn of this loop per receiver step (otherwise exactly one turn). In a step the workgroup walks its channel sets --
    // blockIdx.x, + gridDim.x, ... like a persistent grid -- with the tables it staged once; a channel's state lives in s.state between the
    // steps (40 bytes per channel and step, against 8 N per window read).
    unsigned step = 0;                                      // RES: the last receiver step taken
    unsigned long long resNValid = 0;
    int resCalls = 0, setIdx = 0;
    bool resStopped = false;
#ifdef LORAHIP_RESIDENT_STAMPS
    const bool dbgW = RES && blockIdx.x == 0 && threadIdx.x == 0;
#else
    constexpr bool dbgW = false;
#endif
    for (;;)
    {
    if constexpr (RES)
    {
        if (dbgW) s.res->dbg[step & 7u][0] = wall_clock64();
#ifdef LORAHIP_RESIDENT_STAMPS
        // (a measurement build, -DLORAHIP_RESIDENT_STAMPS + LORAHIP_RESIDENT_DEBUG=<step>: every wavefront's stamps of ONE step, each stored
        // at once. Not in the shipped kernel: the few registers it takes moved 80 more bytes per lane into scratch at SF7.)
        const bool dbgS = s.resDebug == step + 1u && lane == 0 && blockIdx.x * 4u + unsigned(wave) < 16384u;
        if (dbgS) s.res->dbgWave[blockIdx.x * 4u + unsigned(wave)][0] = wall_clock64();
#endif
        // (of the step's message only n_valid stays in registers across the windows: what packs and what reports read the rest back from
        // the workgroup's copy in LDS -- thirteen scalar values kept live took the registers of the window loop into scratch)
        {
            ResMsgR m0;
            if (!residentWait(s, step + 1u, m0, sR)) break;
            resNValid = m0.nValid;
        }
        if (dbgW) s.res->dbg[step & 7u][1] = wall_clock64();
#ifdef LORAHIP_RESIDENT_STAMPS
        if (dbgS) s.res->dbgWave[blockIdx.x * 4u + unsigned(wave)][1] = wall_clock64();
#endif
        step++;
        resCalls = 0; setIdx = 0; resStopped = false;
    }
    unsigned cset = blockIdx.x;                             // (the grid never exceeds the number of sets)
    do
    {
    // ---- this lane group's channel and its state (replicated in the T lanes) ----------------------
    const unsigned c = UNI ? (unsigned)uniI(int((cset * WAVES + wave) * WPW)) : (cset * WAVES + wave) * WPW + wsub;
    const bool mine = c < s.nChannels;
    const unsigned cc = mine ? c : 0;
    StreamState st;
    if constexpr (RES)
    {
        // (what this workgroup stored a step ago: read at agent scope, an older copy of the line may still sit in the compute unit's L1)
        const unsigned long long *q = reinterpret_cast<const unsigned long long *>(&s.state[cc]);
        unsigned long long w[5];
#pragma unroll
        for (int i = 0; i < 5; i++) w[i] = agentLoad(q + i);
        static_assert(sizeof(StreamState) == 40, "five 64-bit words");
        __builtin_memcpy(&st, w, sizeof(st));
    }
    else st = s.state[cc];
    if (s.flags & 1) { st.pos = 0; st.callCount = 0; }                        // a new run: every stream from its first sample
    if (s.flags & 2) { st.state = ST_FRAMESYNC; st.downTable = 0; }           // activate() (LoRaDemod.cpp:139-143)
    const long long base = s.uniformLen >= 0 ? (long long)cc * s.uniformStride : s.base[cc];
    const long long len = !mine ? 0 : (RES ? (long long)resNValid : (s.uniformLen >= 0 ? s.uniformLen : s.len[cc]));    // (RES: what the step's message says)
    StreamOut o;
    o.init(s, cc);
    if constexpr (RES)
    {
        // the record arrays of this step's parity (see StreamArgs::resRecStride)
        const size_t setOff = size_t(step & 3u) * size_t(s.resRecStride);
        o.symOut = reinterpret_cast<short *>(reinterpret_cast<char *>(o.symOut) + setOff);
        o.pktOut = reinterpret_cast<StreamPacket *>(reinterpret_cast<char *>(o.pktOut) + setOff);
        if (o.sigOut) o.sigOut = reinterpret_cast<StreamSignal *>(reinterpret_cast<char *>(o.sigOut) + setOff);
    }
    if (mine) o.carryIn(s, st, cc, t, T, !(RES && LORAHIP_RES_NOCOPY));                 // the packet the channel is inside: its symbols so far, from the carry rows
    if (dbgW && step > 0u && setIdx == 0) s.res->dbg[(step - 1u) & 7u][2] = wall_clock64();

    // one window: LoRaDemod.cpp:157-166 + LoRaDetector::detect. Every lane of the wavefront takes part; groups
    // whose `on` is false run on the head of the buffer and their results are ignored by the caller.
    // What of detect()'s float outputs a work() call consumes depends on the state (LoRaDemod.cpp:176-312): DATASYMBOLS the squelch
    // decision alone; FRAMESYNC the squelch decision and, for an unsquelched window, fIndex; the down-chirp and quarter-chirp
    // states only the peak's index. power / powerAvg / snr themselves only reach the labels and signals, i.e. the per-call trace.
    // So unless a trace is kept (`all`), the squelch comes from a quick estimate with the exact chain as the fallback near the
    // threshold (squelchQuick), the two logarithms are never evaluated otherwise, and the neighbours + fIndex only for lanes
    // with wantFi = 1 whose window is not squelched -- or in any case for wantFi = 2: the second window of a FRAMESYNC call, whose
    // fIndex the reference consumes without looking at that window's own snr (:203, :217-221). wantSq / wantFi are per lane group;
    // the branches are wave-uniform.
    // Signals without a trace (lorahip_demod_set_signals): the one call per packet that emits them (DOWNCHIRP1, :267-269) evaluates
    // power and snr through the exact chain (wantLogs); every decision is the same on either path.
    const bool all = s.calls != nullptr;
    const bool sig = s.sigOut != nullptr;
#ifdef LORAHIP_STREAM_TIMING
    unsigned long long tsec[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, tlast = 0;
#define TMARK(i) do { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsec[i] += now_ - tlast; tlast = now_; } while (0)
#define TMARK_NOWAIT(i) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsec[i] += now_ - tlast; tlast = now_; } while (0)
#else
#define TMARK(i)
#define TMARK_NOWAIT(i)
#endif
    // (instances with few points per lane -- the ones for receivers that leave the device partly empty, lorahip_stream_lanes.hip --
    // keep the NEXT window's samples in registers: C::PREFETCH)
    constexpr bool PF = C::PREFETCH != 0;
    v2f xp[PF ? R : 1][PF ? VEC : 1];
    long long pfOff = -1;
    auto detect = [&](const bool full, const bool on, const bool wantSq, const bool wantLogs, const int wantFi, const long long off, const bool downTable, const int idx0, const float err,
                      int &value, float &power, float &powerAvg, float &fIndex, int &idxEnd, bool &squelched)
    {
        v2f x[R][VEC];
        TMARK(5);
        if constexpr (PF)
        {
            // the window that was asked for while the call before was still computing (below), if this call reads where that one
            // guessed it would; the channels of the wave that guessed wrong load now (from lines the guess has brought closer)
            const bool hit = on && off == pfOff;
            if (!__all(hit || !on)) K::load(x, gIq + (on ? off : 0), t);
#pragma unroll
            for (int r = 0; r < R; r++)
#pragma unroll
                for (int u = 0; u < VEC; u++) x[r][u] = hit ? xp[r][u] : x[r][u];
        }
        else K::load(x, gIq + (on ? off : 0), t);
        const float d = err * (float)LORAHIP_FINE_STEPS;
        const bool moving = on && d != 0.0f;
        int *sIdx = reinterpret_cast<int *>(X) + wsub * N;
        idxEnd = idx0;
        const bool anyMoving = __any(moving);
        unsigned yv[R][VEC];
        if (anyMoving)
        {
            // closed-form indices of this lane's samples (lorahip_fine.h); a wave that holds a channel where the form does not
            // apply walks the exact chain instead
            const FinePlan pl = finePlan(moving ? d : 0.0f, K::M);
            const unsigned ymax = fineLaneIndices<C::LOG2N, VEC, T, R>(idx0, pl, t, yv);
            int e = fineEndIndex(idx0, pl, C::LOG2N, C::LOG2N + 7);
            if (__any(!pl.regular || ymax == (unsigned)K::M))
            {
                e = K::fineChain(idx0, moving ? d : 0.0f, t, sIdx);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int u = 0; u < VEC; u++) yv[r][u] = (unsigned)sIdx[K::idxSlot(VEC * t + u + VEC * T * r)];
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            if (moving) idxEnd = e;
            if (moving && t == 0 && nearStep(d)) atomicAdd(s.near + 1, 1u);          // counted, not changed (lorahip_internal.h)
        }
        TMARK_NOWAIT(0);
        TMARK(1);
        v2f cw[R][VEC];
        K::chirpFromLds(cw, sCh, t);
        const float sgn = downTable ? 1.0f : -1.0f;        // _upChirpTable = conj(entry)  LoRaDemod.cpp:103
        const v2f *cwf = &cw[0][0];
        const v2f sgn2 = MAKE2(1.0f, sgn);                       // one packed multiply: (re, +-im), both exact
        const auto chirpOf = [&](const int i) { return cwf[i] * sgn2; };
        const auto chirpRaw = [&](const int i) { return cwf[i]; };
        // When every channel of the wave uses the same table -- the up-chirp table in FRAMESYNC / DATASYMBOLS, i.e. nearly always --
        // the conjugation rides on the multiply's sign modifiers (cmulConjv / the CONJ pipeline: the same products and roundings)
        // instead of a packed multiply by (1, -1) per sample; a wave that mixes the two (a channel in its two down-chirp calls)
        // takes the general form. Idle groups (`on` false) go with whatever the others use: their results are dropped.
        const bool allUp = !__any(on && downTable), allDown = !__any(on && !downTable);
        // yv = idx0 in the channels where nothing moves
        if (anyMoving)
        {
            if (allUp) dechirpFine<fineSplitLog2H(C::LOG2N), R * VEC, true>(&x[0][0], chirpRaw, &yv[0][0], fl, gFine, true);
            else if (allDown) dechirpFine<fineSplitLog2H(C::LOG2N), R * VEC, false>(&x[0][0], chirpRaw, &yv[0][0], fl, gFine, true);
            else dechirpFine<fineSplitLog2H(C::LOG2N), R * VEC, false>(&x[0][0], chirpOf, &yv[0][0], fl, gFine, true);
        }
        else
        {
            const v2f fconst = gFine[idx0];
            if (allUp)
            {
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int u = 0; u < VEC; u++) x[r][u] = cmulv(cmulConjv(x[r][u], cwf[r * VEC + u]), fconst);
            }
            else
            {
#pragma unroll
                for (int r = 0; r < R; r++)
#pragma unroll
                    for (int u = 0; u < VEC; u++) x[r][u] = cmulv(cmulv(x[r][u], chirpOf(r * VEC + u)), fconst);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        TMARK(2);
        v2f vl[NGL][GL];
        K::fft(x, X, wsub, t, sTw, twR, vl, [&]()
        {
            if constexpr (PF)
            {
                // The chain of a channel exposes the latency of every window's load: call k + 1 reads where call k's result says.
                // Mostly that is the next window (DATASYMBOLS, the down-chirps, a quiet or an aligned FRAMESYNC call consume N; the
                // second window of a sync check is the next one too): ask for it now, with this window's samples in the FFT's
                // registers, so that it arrives while this call computes. A wrong guess (FRAMESYNC on noise: N - value) costs the
                // request, and the right window then overlaps the guessed one: its lines are already on their way.
                const bool can = on && off + 2 * N <= base + len;
                pfOff = can ? off + N : -1;
                // (unconditional, from the head of the buffer where there is nothing to ask for: behind a branch the compiler no
                // longer knows how many loads are in flight and waits for ALL of them at the next wait for an older one)
                K::load(xp, gIq + (can ? off + N : 0), t);
            }
        }, &twM);
        TMARK(3);
        v2f *F = X + wsub * FS;
        float bestV;
        int bestI;
        double tot;
        v2f l, r;
        value = 0;
        const bool staged = full || (STREAM_STAGE_BINS && __any(on && wantFi != 0));       // bins to LDS for the neighbour fetch (else: register select)
        if (full)
        {
            K::template scan<true, STREAM_SCAN_CHAINS>(vl, F, nullptr, t, bestV, bestI, tot);
            K::neighbours(vl, F, bestI, lane, t, l, r);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            tailValuesPaired(s.powerScale, bestV, tot, l, r, lane, power, powerAvg, fIndex);
            squelched = (power - powerAvg) < s.thresh;                                   // :173-174
            if (on && wantSq && t == 0 && nearSquelch(power - powerAvg, s.thresh)) atomicAdd(s.near, 1u);
        }
        else
        {
            // no fp64 total on this path: the squelch estimate takes an fp32 one (scanQuick / squelchQuickF, whose `sure` band
            // accounts for it); the exact total is summed only where the exact chain is evaluated
            float totF;
            if (staged) K::template scanQuick<true>(vl, F, t, bestV, bestI, totF);
            else K::template scanQuick<false>(vl, F, t, bestV, bestI, totF);
            TMARK(6);
            bool sure;
            squelched = squelchQuickF(bestV, totF, s.thresh, K::QUICK_REL_ERR, sure);
            power = powerAvg = fIndex = 0.0f;                                           // not consumed without a trace
            const bool exact = on && wantSq && !sure;
            const bool logs = on && wantLogs;                                          // the call of a packet that emits the signals (:267-269)
            const bool fi = on && (wantFi == 2 || (wantFi == 1 && (!sure || !squelched)));
            TMARK(7);
            if (__any(exact || logs || fi))
            {
                if (staged) K::neighbours(vl, F, bestI, lane, t, l, r);
                else K::template neighbours<true>(vl, F, bestI, lane, t, l, r);
                if (staged)
                {
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
                if (__any(exact || logs))
                {
                    // LoRaDetector.hpp:36-48's double total, in scan()'s association (the bins are still in registers)
                    {
                        float bv_;
                        (void)laneScan<GL * NGL, STREAM_SCAN_CHAINS>([&](const int j) { return vl[j % NGL][j / NGL]; }, bv_, tot);
                        tot = groupSumF64<T>(tot);
                    }
                    tailValuesPaired(s.powerScale, bestV, tot, l, r, lane, power, powerAvg, fIndex);
                    squelched = (power - powerAvg) < s.thresh;                           // the quick decision where it was sure, by construction
                    if (exact && t == 0 && nearSquelch(power - powerAvg, s.thresh)) atomicAdd(s.near, 1u);
                    power = logs ? power : 0.0f; powerAvg = logs ? powerAvg : 0.0f;     // (what the signal record takes; nobody else reads them)
                }
                else fIndex = fIndexPaired(bestV, l, r, lane);
            }
        }
        value = uniI(bestI);
        squelched = uniI(squelched) != 0;
        fIndex = uniF(fIndex); power = uniF(power); powerAvg = uniF(powerAvg);
        idxEnd = uniI(idxEnd);
        TMARK(4);
    };

#ifdef LORAHIP_STREAM_TIMING
    tlast = __builtin_amdgcn_s_memtime();
#endif
    // A FRAMESYNC call that is sync'd and matches the first sync word looks at a SECOND window (LoRaDemod.cpp:183-206). The wave's
    // channels run in lock step, so a second detect() inside the pass would be paid by all of them; instead the call is split over
    // two passes of the loop: the first evaluates window 0 and parks (`pend`), the second evaluates window 1 in that channel's
    // slot of the next pass -- while the other channels do their next calls -- and completes the frame machine step. Nothing is
    // written and nothing is consumed in between, and the limits checked for the first pass cover the whole call.
    bool pend = false;
    int value0 = 0, fineIdxBefore0 = 0;
    float snr0 = 0.0f, fineErrBefore0 = 0.0f;
    const int slot = wavefrontSlot();
    const bool lastRound = RES || PERSIST || !LORAHIP_PRIO_HOLD || blockIdx.x >= s.lastRoundFrom;
    holdPriority<LORAHIP_PRIO_ALTERNATE>(!lastRound);
    while (true)
    {
        if (lastRound) rotatePriority<2, LORAHIP_PRIO_ALTERNATE>(slot);
        const bool live = mine && (pend || ((len - st.pos >= 2 * N) && o.calls < s.cap && o.nPkt < s.capPkt && o.nSig < s.capPkt));   // LoRaDemod.cpp:148
        if (!__any(live)) break;

        // ---- this pass's window: window 0 of a call (:157-172), or window 1 of a parked one (:189-206) ----
        const bool second = pend;
        int value, idxEnd;
        float power, powerAvg, fIndex;
        const int fineIdxBefore = second ? fineIdxBefore0 : st.fineTuneIndex;
        const float fineErrBefore = second ? fineErrBefore0 : st.finefreqError;
        const long long here = base + st.pos + (second ? N : 0);
        const bool fs = st.state == ST_FRAMESYNC;
        bool squelched;
        // Signals without a trace: the one call per packet that emits them evaluates power and snr through the exact chain of the
        // untraced path (the same operations on the same operands as the traced path's: the same bits) -- for the channels that are in
        // that call, not, as the traced path would, the fp64 scan and both logarithms for every channel of the wavefront whenever one
        // of them is there (SF7: 8 channels per wavefront, 0.378 -> of the roofline with signals kept against 0.442 without, round 5)
        detect(all, live, !second && (fs || st.state == ST_DATASYMBOLS), sig && live && !second && st.state == ST_DOWNCHIRP1, second ? 2 : (fs ? 1 : 0), here, st.downTable != 0, st.fineTuneIndex,
               st.finefreqError, value, power, powerAvg, fIndex, idxEnd, squelched);
        float snr = power - powerAvg;                                                   // :173 (squelched = snr < thresh, :174, comes from detect)
        // window 0: the loop commits the member (:160-162); window 1: `int ft = _fineTuneIndex` (:191) starts from the committed
        // index and is not committed itself
        if (live && !second) st.fineTuneIndex = idxEnd;

        // (selects, not branches: the wave's channels are in different states at once, see frameStep)
        const bool syncdW = !squelched && (st.prevValue + 4) / 8 == 0;                 // :183
        const int word = (value + 4) / 8;
        // window 0 of a sync'd FRAMESYNC call that matches the first sync word: park it, its window 1 comes in the next pass
        const bool park = live && !second && fs && syncdW && word == (s.sync >> 4);    // :184
        const bool match1 = second && word == (s.sync & 0xf);                          // :205
        // detect() of window 1 overwrote power / powerAvg / fIndex (:203); value, snr and what follows from them are window 0's
        value0 = park ? value : value0; snr0 = park ? snr : snr0;
        fineIdxBefore0 = park ? fineIdxBefore : fineIdxBefore0; fineErrBefore0 = park ? fineErrBefore : fineErrBefore0;
        value = second ? value0 : value; snr = second ? snr0 : snr;
        squelched = second ? false : squelched;
        const bool syncd = second || syncdW, match0 = second || word == (s.sync >> 4);
        pend = park;
        const bool step = live && !park;

        // ---- the frame machine (:176-312) ----
        TMARK(8);
        if (step)
        {
            frameStep<N>(st, s, o, t == 0, value, power, powerAvg, snr, fIndex, squelched, syncd, match0, match1, fineIdxBefore, fineErrBefore);
            st.finefreqError = uniF(st.finefreqError);          // a float add runs on the vector unit: back to a scalar where uniform
        }
    }
#ifdef LORAHIP_STREAM_TIMING
    if (blockIdx.x == 7 && threadIdx.x == 0)
        printf("stream timing (s_memtime ticks): index math %llu, load wait %llu, chirp+dechirp %llu, fft %llu, scan+reductions %llu, squelch estimate %llu, neighbours+fIndex+exact tail %llu, "
               "uniform/epilogue of detect %llu, sync/match logic %llu, frame step+records+loop top %llu; calls %d\n",
               tsec[0], tsec[1], tsec[2], tsec[3], tsec[6], tsec[7], tsec[4], 0ull, tsec[8], tsec[5], o.calls);
#endif
    // (RES: the step's packets leave first -- a packet's first symbols may still be in the carry row carryOut is about to overwrite)
    if constexpr (RES && LORAHIP_RES_NOCOPY) residentPackOwn<C>(s, sR, step, (cset * WAVES + unsigned(wave)) * unsigned(WPW), o, mine, lane, st.state == ST_DATASYMBOLS ? st.symCount : 0);
    o.carryOut(s, st, cc, t, T, mine);
    if (mine && t == 0)
    {
        s.state[c] = st;
        s.nCalls[c] = o.calls;
        s.nSym[c] = o.nSym;
        s.nPkt[c] = o.nPkt;
        if (s.nSig) s.nSig[c] = o.nSig;
        s.end[c] = make_int2(st.state == ST_DATASYMBOLS ? st.symCount : -1, st.callCount);
    }
    if constexpr (RES)
    {
        if constexpr (!LORAHIP_RES_NOCOPY) residentPackOwn<C>(s, sR, step, (cset * WAVES + unsigned(wave)) * unsigned(WPW), o, mine, lane);
        resCalls += (mine && t == 0) ? o.calls : 0;
        resStopped = resStopped || (mine && len - st.pos >= 2 * N);        // stopped with samples left: a record buffer was full
        setIdx++;
    }
    } while ((PERSIST || RES) && (cset += gridDim.x) < nSets);      // without PERSIST / RES there is no loop at all (it would cost registers)
    if constexpr (!RES) break;
    else
    {
        if (dbgW) s.res->dbg[(step - 1u) & 7u][4] = wall_clock64();
#ifdef LORAHIP_RESIDENT_STAMPS
        const bool dbgE = s.resDebug == step && lane == 0 && blockIdx.x * 4u + unsigned(wave) < 16384u;
        if (dbgE) s.res->dbgWave[blockIdx.x * 4u + unsigned(wave)][2] = wall_clock64();
#endif
        residentLookAhead(s, step + 1u);
        residentStepEnd<C>(s, step, sR, resCalls, resStopped, lane);
        if (dbgW) s.res->dbg[(step - 1u) & 7u][5] = wall_clock64();
#ifdef LORAHIP_RESIDENT_STAMPS
        if (dbgE) s.res->dbgWave[blockIdx.x * 4u + unsigned(wave)][3] = wall_clock64();
#endif
    }
    }
}

template <class C>
static hipError_t launchStreamCfg(const StreamArgs &args, hipStream_t stream)
{
    constexpr int WAVES = 4;
    const size_t smem = size_t(C::TWN + C::N) * sizeof(float2) + size_t(WAVES) * C::XW * sizeof(float2) + FineDims<C::LOG2N>::BYTES;
    static unsigned long long attrDone = 0, attrDoneP = 0;
    static PerDeviceCount resident;
    StreamArgs s = args;
    const unsigned perBlock = WAVES * C::WPW;
    const unsigned grid = (s.nChannels + perBlock - 1) / perBlock;
    if (grid == 0) return hipSuccess;
#if defined(LORAHIP_ALL_VARIANTS) || defined(LORAHIP_STREAM_PERSIST)      // the persistent grid: profiling builds only (lorahip_demod.cpp::runStream)
    if (s.maxBlocks > 0 && grid > unsigned(s.maxBlocks))
    {
        const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStream<C, true>), smem, attrDoneP);
        if (e != hipSuccess) return e;
        s.lastRoundFrom = 0;
        hipLaunchKernelGGL((demodStream<C, true>), dim3(unsigned(s.maxBlocks)), dim3(WAVES * 64), smem, stream, s);
        return hipGetLastError();
    }
#endif
    (void)attrDoneP;
    const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStream<C, false>), smem, attrDone);
    if (e != hipSuccess) return e;
    s.lastRoundFrom = lastRoundFrom(grid, residentWorkgroupsCached(resident, reinterpret_cast<const void *>(demodStream<C, false>), WAVES * 64, smem));
    hipLaunchKernelGGL((demodStream<C, false>), dim3(grid), dim3(WAVES * 64), smem, stream, s);
    return hipGetLastError();
}

//! the resident receiver's launch: refused (hipErrorNotSupported) unless every workgroup is resident at once -- a workgroup that had to
//! wait for a slot would wait for ever, the others never leave
template <class C>
static hipError_t launchStreamResidentCfg(const StreamArgs &args, hipStream_t stream, unsigned *gridOut)
{
    constexpr int WAVES = 4;
    const size_t smem = size_t(C::TWN + C::N) * sizeof(float2) + size_t(WAVES) * C::XW * sizeof(float2) + FineDims<C::LOG2N>::BYTES + sizeof(ResLds);
    static unsigned long long attrDone = 0;
    static PerDeviceCount resident;
    const unsigned perBlock = WAVES * C::WPW;
    const unsigned nSets = (args.nChannels + perBlock - 1) / perBlock;
    if (gridOut) *gridOut = 0;
    if (nSets == 0) return hipErrorNotSupported;
    const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStream<C, false, true>), smem, attrDone);
    if (e != hipSuccess) return e;
    const int res = residentWorkgroupsCached(resident, reinterpret_cast<const void *>(demodStream<C, false, true>), WAVES * 64, smem);
    if (res <= 0) return hipErrorNotSupported;
    // more channel sets than the device holds workgroups: every workgroup walks several per step, all workgroups the same number but
    // the last ones
    const unsigned perWg = (nSets + unsigned(res) - 1) / unsigned(res);
    const unsigned grid = (nSets + perWg - 1) / perWg;
    if (grid > 4095u) return hipErrorNotSupported;
    if (gridOut) *gridOut = grid;
    hipLaunchKernelGGL((demodStream<C, false, true>), dim3(grid), dim3(WAVES * 64), smem, stream, args);
    return hipGetLastError();
}

} // namespace lorahip

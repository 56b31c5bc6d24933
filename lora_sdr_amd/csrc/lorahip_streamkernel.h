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
//! AHEAD: a channel takes TWO lane groups of the wavefront. The first evaluates the call's window as ever; the second, at the same
//! time, the window the NEXT call will read if this one consumes exactly N samples and leaves the fine-tune state where a plain
//! call leaves it (DATASYMBOLS inside a packet, a quiet or aligned FRAMESYNC call, the first down-chirp: most calls of a receiver
//! that is following frames). The frame machine then runs for the first window, looks whether the state it left is the one the
//! second window was evaluated for -- position, fine-tune index and error, table, state -- and if so runs for the second window
//! too: two work() calls of the chain in one pass. If not, the second window's results are dropped and nothing of them is kept or
//! counted. For receivers with fewer channels than the device holds wavefronts (lorahip_stream_pairs.hip): a chain's time is the
//! time of its passes, and this halves their number where the guess holds. Scheduling only: every call sees the operands it would
//! have seen alone.
template <class C, bool PERSIST, bool RES = false, bool AHEAD = false>
__global__ void __launch_bounds__(256, C::WAVES_PER_SIMD)
demodStream(const StreamArgs s)
{
    static_assert(!(RES && PERSIST), "the resident receiver has one workgroup per channel set");
    static_assert(!AHEAD || (!RES && !PERSIST && C::WPW >= 2 && C::PREFETCH == 0), "the look-ahead instances: one-launch grids, two lane groups per channel");
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
    const int wsub = lane >> LOG2T;                        // lane group (window slot) inside the wavefront
    const int t = lane & (T - 1);
    constexpr int CPW = AHEAD ? WPW / 2 : WPW;              // channels per wavefront
    const bool roleB = AHEAD ? (wsub & 1) != 0 : false;     // AHEAD: the group that evaluates the window after the call's
    const int csub = AHEAD ? wsub >> 1 : wsub;              // channel inside the wavefront
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
    const unsigned nSets = (s.nChannels + WAVES * CPW - 1) / (WAVES * CPW);
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
    const unsigned c = UNI ? (unsigned)uniI(int((cset * WAVES + wave) * WPW)) : (cset * WAVES + wave) * CPW + csub;
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
    const int tc = AHEAD ? t + (roleB ? T : 0) : t;         // lane inside the channel's lanes, TC of them
    constexpr int TC = AHEAD ? 2 * T : T;
    if (mine) o.carryIn(s, st, cc, tc, TC, !(RES && LORAHIP_RES_NOCOPY));                 // the packet the channel is inside: its symbols so far, from the carry rows
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
    // the near-threshold counters (StreamArgs::near): a window evaluated ahead is counted when -- and only if -- its call is made
    int nearAhead = 0;
    const auto noteNear = [&](const int which, const bool yes)
    {
        if (AHEAD && roleB) nearAhead |= yes ? 1 << which : 0;
        else if (yes) atomicAdd(s.near + which, 1u);
    };
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
            noteNear(1, moving && t == 0 && nearStep(d));                            // counted, not changed (lorahip_internal.h)
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
            noteNear(0, on && wantSq && t == 0 && nearSquelch(power - powerAvg, s.thresh));
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
                    noteNear(0, exact && t == 0 && nearSquelch(power - powerAvg, s.thresh));
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
    int tPasses = 0;
    const unsigned long long tBegin = tlast;
#endif
    // A FRAMESYNC call that is sync'd and matches the first sync word looks at a SECOND window (LoRaDemod.cpp:183-206). The wave's
    // channels run in lock step, so a second detect() inside the pass would be paid by all of them; instead the call is split over
    // two passes of the loop: the first evaluates window 0 and parks (`pend`), the second evaluates window 1 in that channel's
    // slot of the next pass -- while the other channels do their next calls -- and completes the frame machine step. Nothing is
    // written and nothing is consumed in between, and the limits checked for the first pass cover the whole call.
    bool pend = false;
    bool fsPlain = false;                                   // AHEAD: the channel's last FRAMESYNC call consumed N and left the fine-tune state alone
    float planErr = __uint_as_float(0x7fc00000u);           // AHEAD: the error planStepN / planMod were derived from (NaN: none yet)
    unsigned planStepN = 0u, planMod = 0u;                  //        N q mod M' and M' of lorahip_fine.h's closed form (planMod 0: the form does not apply)
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
#ifdef LORAHIP_STREAM_TIMING
        tPasses++;
#endif

        // ---- this pass's window: window 0 of a call (:157-172), or window 1 of a parked one (:189-206) ----
        const bool second = pend;
        int value, idxEnd;
        float power, powerAvg, fIndex;
        const long long here = base + st.pos + (second ? N : 0);
        const bool fs = st.state == ST_FRAMESYNC;
        bool squelched;
        // AHEAD: what the call after this one reads and is, IF this one turns out plain -- N samples on (:320 with total = N), the index
        // where this window's N steps leave it (:160-162; lorahip_fine.h's closed form, checked against the committed index below), the
        // same error and table, DATASYMBOLS / FRAMESYNC again or the second down-chirp after the first (:243). Not tried from the other
        // states (the calls around the quarter chirp move by something else, :278), nor while a sync check is parked.
        const int stateAhead = st.state == ST_DOWNCHIRP0 ? ST_DOWNCHIRP1 : st.state;
        const long long posAhead = st.pos + N;
        const float errAhead = st.finefreqError;
        const int tableAhead = st.downTable;
        int idxAhead = st.fineTuneIndex;
        bool tryAhead = false;
        if constexpr (AHEAD)
        {
            // (the index after N steps of the closed form, lorahip_fine.h: (idx + N q) mod M' -- N q mod M' kept per channel for as long as
            // the error stays, which inside a packet is the whole packet; a window where the form does not hold, or lands on M', ends on
            // another index than this and the second call is simply not made)
            if (__any(live && errAhead != planErr))
            {
                const FinePlan pa = finePlan(errAhead * (float)LORAHIP_FINE_STEPS, K::M);   // (d = 0: q = 0, the index stays)
                planErr = errAhead;
                // (0 < d < 1, `sat`: the index walks down by one per sample and stays at 0 -- marked by a modulus no sum reaches)
                planMod = pa.regular ? (pa.sat ? 0x80000000u : pa.mod) : 0u;
                planStepN = pa.sat ? 0u : fineReduce(pa.q << C::LOG2N, pa, C::LOG2N + 7);
            }
            {
                const unsigned sum = unsigned(st.fineTuneIndex) + planStepN, wrapped = sum - planMod;
                const int down = st.fineTuneIndex - N;
                idxAhead = planMod == 0x80000000u ? (down > 0 ? down : 0) : int(sum < wrapped ? sum : wrapped);
            }
            // (FRAMESYNC: only while the channel's last FRAMESYNC call was a plain one -- on noise above the threshold every call moves the
            // window, :219, and a window evaluated ahead there is work for nothing in every pass)
            tryAhead = live && !second && planMod != 0u && errAhead == planErr && ((fs && fsPlain) || st.state == ST_DATASYMBOLS || st.state == ST_DOWNCHIRP0);
            nearAhead = 0;
        }
        const bool onG = roleB ? tryAhead : live;
        const int stG = roleB ? stateAhead : st.state;
        const bool firstG = roleB || !second;                   // this group's window is window 0 of its call
        // Signals without a trace: the one call per packet that emits them evaluates power and snr through the exact chain of the
        // untraced path (the same operations on the same operands as the traced path's: the same bits) -- for the channels that are in
        // that call, not, as the traced path would, the fp64 scan and both logarithms for every channel of the wavefront whenever one
        // of them is there (SF7: 8 channels per wavefront, 0.378 -> of the roofline with signals kept against 0.442 without, round 5)
        detect(all, onG, firstG && (stG == ST_FRAMESYNC || stG == ST_DATASYMBOLS), sig && onG && firstG && stG == ST_DOWNCHIRP1, firstG ? (stG == ST_FRAMESYNC ? 1 : 0) : 2,
               roleB ? base + posAhead : here, st.downTable != 0, roleB ? idxAhead : st.fineTuneIndex,
               st.finefreqError, value, power, powerAvg, fIndex, idxEnd, squelched);
        // AHEAD: the two groups of a channel tell each other what they found (the channel's state is replicated in both)
        int valueB = 0, idxEndB = 0;
        float powerB = 0.0f, powerAvgB = 0.0f, fIndexB = 0.0f;
        bool squelchedB = false;
        if constexpr (AHEAD)
        {
            // (one word: value < N <= 2^9, the squelch bit, the index < 128 N <= 2^16)
            static_assert(C::LOG2N <= 9, "the packed exchange");
            const int mineW = value | (squelched ? 1 << 12 : 0) | (idxEnd << 13);
            const int otherW = __shfl_xor(mineW, T);
            const int aW = roleB ? otherW : mineW, bW = roleB ? mineW : otherW;
            value = aW & 0xfff; squelched = (aW & (1 << 12)) != 0; idxEnd = int(unsigned(aW) >> 13);
            valueB = bW & 0xfff; squelchedB = (bW & (1 << 12)) != 0; idxEndB = int(unsigned(bW) >> 13);
            // (the float outputs reach the frame machine from FRAMESYNC windows, the signals' call and a trace only)
            if (all || __any(live && st.state != ST_DATASYMBOLS))
            {
                const float oP = __shfl_xor(power, T), oA = __shfl_xor(powerAvg, T), oF = __shfl_xor(fIndex, T);
                powerB = roleB ? power : oP; power = roleB ? oP : power;
                powerAvgB = roleB ? powerAvg : oA; powerAvg = roleB ? oA : powerAvg;
                fIndexB = roleB ? fIndex : oF; fIndex = roleB ? oF : fIndex;
            }
            else power = powerAvg = fIndex = 0.0f;              // (what detect() leaves there on the untraced path)
        }

        // one call after its window: returns whether the frame machine stepped (false: idle, or parked for its second window)
        const auto finish = [&](const bool liveX, const bool secondX, int value, float power, const float powerAvg, const float fIndex, const int idxEnd, bool squelched) -> bool
        {
        const int fineIdxBefore = secondX ? fineIdxBefore0 : st.fineTuneIndex;
        const float fineErrBefore = secondX ? fineErrBefore0 : st.finefreqError;
        const bool fs = st.state == ST_FRAMESYNC;
        float snr = power - powerAvg;                                                   // :173 (squelched = snr < thresh, :174, comes from detect)
        // window 0: the loop commits the member (:160-162); window 1: `int ft = _fineTuneIndex` (:191) starts from the committed
        // index and is not committed itself
        if (liveX && !secondX) st.fineTuneIndex = idxEnd;

        // (selects, not branches: the wave's channels are in different states at once, see frameStep)
        const bool syncdW = !squelched && (st.prevValue + 4) / 8 == 0;                 // :183
        const int word = (value + 4) / 8;
        // window 0 of a sync'd FRAMESYNC call that matches the first sync word: park it, its window 1 comes in the next pass
        const bool park = liveX && !secondX && fs && syncdW && word == (s.sync >> 4);  // :184
        const bool match1 = secondX && word == (s.sync & 0xf);                         // :205
        // detect() of window 1 overwrote power / powerAvg / fIndex (:203); value, snr and what follows from them are window 0's
        value0 = park ? value : value0; snr0 = park ? snr : snr0;
        fineIdxBefore0 = park ? fineIdxBefore : fineIdxBefore0; fineErrBefore0 = park ? fineErrBefore : fineErrBefore0;
        value = secondX ? value0 : value; snr = secondX ? snr0 : snr;
        squelched = secondX ? false : squelched;
        const bool syncd = secondX || syncdW, match0 = secondX || word == (s.sync >> 4);
        pend = AHEAD ? (liveX ? park : pend) : park;            // (not live: nothing is parked -- but AHEAD's second call must not un-park the first)
        const bool step = liveX && !park;

        // ---- the frame machine (:176-312) ----
        TMARK(8);
        if (step)
        {
            frameStep<N>(st, s, o, tc == 0, value, power, powerAvg, snr, fIndex, squelched, syncd, match0, match1, fineIdxBefore, fineErrBefore);
            st.finefreqError = uniF(st.finefreqError);          // a float add runs on the vector unit: back to a scalar where uniform
        }
        return step;
        };
        const bool stepped = finish(live, second, value, power, powerAvg, fIndex, idxEnd, squelched);
        if constexpr (AHEAD)
        {
            if (live && !second && fs)
                fsPlain = stepped && st.state == ST_FRAMESYNC && int(st.pos) == int(posAhead) && st.fineTuneIndex == idxAhead && st.finefreqError == errAhead;
            // the call after: is the channel where the second group's window was evaluated for? (every input of that window's
            // detect() and the state its wants were derived from; then :148's conditions for a call)
            const bool live2 = tryAhead && stepped && st.state == stateAhead && int(st.pos) == int(posAhead) && st.fineTuneIndex == idxAhead && st.finefreqError == errAhead    // (positions: a call moves by less than 2^31)
                               && st.downTable == tableAhead && (len - st.pos >= 2 * N) && o.calls < s.cap && o.nPkt < s.capPkt && o.nSig < s.capPkt;
            if (__any(live2))
            {
                if (!all && __all(!live2 || st.state == ST_DATASYMBOLS))
                {
                    // every second call of the wavefront is inside a packet: the DATASYMBOLS arm alone (:160-162, :290-300, :320, :326), as
                    // frameStep takes it -- a twentieth of the machine written as selects over all five states
                    if (live2)
                    {
                        st.fineTuneIndex = idxEndB;                                                          // :160-162
                        const int symCount = st.symCount + 1;
                        const bool post = (unsigned)symCount >= s.mtu || squelchedB;                         // :291
#ifndef LORAHIP_TIMING_NO_RECORD_STORES
                        if (tc == 0) o.symOut[o.nSym] = (short)valueB;                                       // out[_symCount++] = value  :290
#endif
                        if (tc == 0 && post) { StreamPacket q; q.callIndex = st.callCount; q.len = symCount; o.pktOut[o.nPkt] = q; }   // :295-298
                        o.nSym++; o.nPkt += post ? 1 : 0; o.calls++;
                        st.finefreqError = post ? 0.0f : st.finefreqError;                                   // :299
                        st.state = post ? ST_FRAMESYNC : ST_DATASYMBOLS;                                     // :300
                        st.symCount = symCount;
                        st.prevValue = (short)valueB;                                                        // :326
                        st.pos += N;                                                                         // :320
                        st.callCount++;
                    }
                }
                else (void)finish(live2, false, valueB, powerB, powerAvgB, fIndexB, idxEndB, squelchedB);
                if (live2 && roleB && t == 0 && nearAhead != 0)
                {
                    if (nearAhead & 1) atomicAdd(s.near, 1u);
                    if (nearAhead & 2) atomicAdd(s.near + 1, 1u);
                }
            }
        }
        else (void)stepped;
        TMARK(9);
    }
#ifdef LORAHIP_STREAM_TIMING
    if (blockIdx.x == 7 && threadIdx.x == 0)
        printf("stream timing (s_memtime ticks): index math %llu, load wait %llu, chirp+dechirp %llu, fft %llu, scan+reductions %llu, squelch estimate %llu, neighbours+fIndex+exact tail %llu, "
               "uniform/epilogue of detect %llu, sync/match logic %llu, frame step(s)+records %llu, loop top %llu; calls %d in %d passes of the wavefront, %llu ticks\n",
               tsec[0], tsec[1], tsec[2], tsec[3], tsec[6], tsec[7], tsec[4], 0ull, tsec[8], tsec[9], tsec[5], o.calls, tPasses, (unsigned long long)__builtin_amdgcn_s_memtime() - tBegin);
#endif
    // (RES: the step's packets leave first -- a packet's first symbols may still be in the carry row carryOut is about to overwrite)
    if constexpr (RES && LORAHIP_RES_NOCOPY) residentPackOwn<C>(s, sR, step, (cset * WAVES + unsigned(wave)) * unsigned(WPW), o, mine, lane, st.state == ST_DATASYMBOLS ? st.symCount : 0);
    o.carryOut(s, st, cc, tc, TC, mine);
    if (mine && tc == 0)
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

template <class C, bool AHEAD = false>
static hipError_t launchStreamCfg(const StreamArgs &args, hipStream_t stream)
{
    constexpr int WAVES = 4;
    const size_t smem = size_t(C::TWN + C::N) * sizeof(float2) + size_t(WAVES) * C::XW * sizeof(float2) + FineDims<C::LOG2N>::BYTES;
    static unsigned long long attrDone = 0, attrDoneP = 0;
    static PerDeviceCount resident;
    StreamArgs s = args;
    const unsigned perBlock = WAVES * (AHEAD ? C::WPW / 2 : C::WPW);
    const unsigned grid = (s.nChannels + perBlock - 1) / perBlock;
    if (grid == 0) return hipSuccess;
#if defined(LORAHIP_ALL_VARIANTS) || defined(LORAHIP_STREAM_PERSIST)      // the persistent grid: profiling builds only (lorahip_demod.cpp::runStream)
    if (!AHEAD && s.maxBlocks > 0 && grid > unsigned(s.maxBlocks))
    {
        const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStream<C, true>), smem, attrDoneP);
        if (e != hipSuccess) return e;
        s.lastRoundFrom = 0;
        hipLaunchKernelGGL((demodStream<C, true>), dim3(unsigned(s.maxBlocks)), dim3(WAVES * 64), smem, stream, s);
        return hipGetLastError();
    }
#endif
    (void)attrDoneP;
    const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(demodStream<C, false, false, AHEAD>), smem, attrDone);
    if (e != hipSuccess) return e;
    s.lastRoundFrom = lastRoundFrom(grid, residentWorkgroupsCached(resident, reinterpret_cast<const void *>(demodStream<C, false, false, AHEAD>), WAVES * 64, smem));
    hipLaunchKernelGGL((demodStream<C, false, false, AHEAD>), dim3(grid), dim3(WAVES * 64), smem, stream, s);
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

// Front-end channeliser: wideband IQ -> K channel streams at the channel rate, written in the [channel][time]
// layout the batched demodulator reads (include/lorahip.h has the definition). The reference has no
// such block -- its example topologies chain Pothos' /comms/rotate and a decimating FIR in front of every LoRaDemod --
// so there is nothing to be bit-exact with: the tests check against a float64 restatement of the definition.
//
// Shape of the work: every output is L complex multiply-adds per channel, every input sample feeds K*L/D of them
// (64 for 8 channels, 8x decimation, 64 taps): 512 flop per 8 input bytes -- this one is bound by the fp32 VALU
// (v_pk_fma_f32, 157 TFLOP/s), not by HBM. One workgroup = one tile of 256*RM output times x 8 channels:
//   * the tile's input span is staged once into LDS, split by decimation phase (sample i = q*D + p lives at
//     [p][q]) so that the 64 lanes of a wavefront -- consecutive output times -- read consecutive addresses for
//     every tap;
//   * the taps are pre-rotated per channel on the host (g_k[j] = h[j] e^{+i theta_k j}), so the mixer costs one complex
//     multiply per OUTPUT (e^{-i theta_k n_m}) instead of one per input sample and channel; they are wave-uniform and
//     arrive through the scalar cache as SGPR operands of the packed FMAs: the inner loop is one ds_read_b64 per
//     16*RM v_pk_fma_f32;
//   * the NCO is a 64-bit phase counter (w_k * n mod 2^64): exact wrap-around, no drift, and a stream cut into
//     chunks gives the same bits as one call.
#include "lorahip_internal.h"
#include <cmath>
#include <cstdlib>
#include <new>

struct lorahip_channelizer
{
    lorahip_ctx *ctx;
    int K, L, D, HC, QP, RM, nGroups;
    size_t ldsBytes;
    float2 *dTaps;                  // [nGroups][L+1][8], tap order reversed (oldest sample first), last entry a dummy
    unsigned *dTapOff;              // [L+2]
    unsigned long long *dW;         // [nGroups*8]
    float2 *dRot;                   // [nGroups*8] phase step over 256 outputs, then [nGroups*8][256] phase over t outputs
    float2 *dHist[2];               // the HC samples before n0 (zeros before the start of the stream)
    int cur;
    unsigned long long n0;          // samples consumed since the last reset
};

namespace lorahip {

typedef float v2f __attribute__((ext_vector_type(2)));

constexpr int CHAN_THREADS = 256;
constexpr int CHAN_KG = 8;          // channels per workgroup

struct ChanArgs
{
    const float2 *chunk;
    long long nChunk;
    const float2 *hist;
    int histLen;
    long long n0;                   // absolute index of chunk[0]
    const v2f *taps;
    const unsigned long long *w;
    const unsigned *tapOff;         // [L+2] LDS byte offset of tap jr's sample relative to the lane's first one
    const v2f *laneRot;             // [K][256] e^{-2 pi i frac(w_k * t D / 2^64)}: the mixer phase t outputs on
    const v2f *step;                // [K] e^{-2 pi i frac(w_k * 256 D / 2^64)}: the mixer phase 256 outputs on
    float2 *out;
    long long outStride;
    long long mLo;                  // absolute index of the first output of this call
    long long nOut;
    int K, L, D, QP, nGroups;
    long long captureIn, captureOut;    // batch of independent captures (blockIdx.y): sample / output-row strides, 0 for a stream
};

//! sample n of the stream (absolute index): from this call's chunk, from the history kept from earlier calls, or 0
__device__ __forceinline__ float2 streamSample(const ChanArgs &a, const float2 *chunk, const long long n)
{
    const long long c = n - a.n0, h = c + a.histLen;
    const float2 *src = c >= 0 ? chunk + c : a.hist + h;
    const bool ok = c >= 0 ? c < a.nChunk : h >= 0;
    float2 v = make_float2(0.0f, 0.0f);
    if (ok) v = *src;
    return v;
}

//! acc += g * x, complex, two packed FMAs; g is wave-uniform (SGPR pair)
__device__ __forceinline__ void cmacS(v2f &acc, const v2f g, const v2f x)
{
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(g), "v"(x));                   // (g.x*x.x, g.x*x.y)
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc) : "s"(g), "v"(x));    // (-g.y*x.y, g.y*x.x)
}

//! complex product with fused multiply-adds
__device__ __forceinline__ v2f cmulF(const v2f a, const v2f b)
{
    return (v2f){fmaf(a.x, b.x, -(a.y * b.y)), fmaf(a.x, b.y, a.y * b.x)};
}

typedef float v4f __attribute__((ext_vector_type(4)));
template <int RM> struct TapRegs { typedef v2f X; };
template <> struct TapRegs<2> { typedef v4f X; };
struct TapCoef { v2f c0, c1, c2, c3, c4, c5, c6, c7; };         // 8 channel coefficients, one SGPR pair each

//! start the loads of one tap: 8 channel coefficients and the LDS offset of the NEXT tap (both wave-uniform, through the
//! scalar cache), and this lane's RM samples
template <int RM>
__device__ __forceinline__ void tapIssue(TapCoef &G, typename TapRegs<RM>::X &X, unsigned &offNext, const v2f *gp, const unsigned *op,
                                         const unsigned ldsAddr)
{
    asm volatile("s_load_dwordx2 %0, %9, 0x0\n\ts_load_dwordx2 %1, %9, 0x8\n\ts_load_dwordx2 %2, %9, 0x10\n\ts_load_dwordx2 %3, %9, 0x18\n\t"
                 "s_load_dwordx2 %4, %9, 0x20\n\ts_load_dwordx2 %5, %9, 0x28\n\ts_load_dwordx2 %6, %9, 0x30\n\ts_load_dwordx2 %7, %9, 0x38\n\t"
                 "s_load_dword %8, %10, 0x0"
                 : "=&s"(G.c0), "=&s"(G.c1), "=&s"(G.c2), "=&s"(G.c3), "=&s"(G.c4), "=&s"(G.c5), "=&s"(G.c6), "=&s"(G.c7), "=&s"(offNext)
                 : "s"(gp), "s"(op) : "memory");
    if constexpr (RM == 2) asm volatile("ds_read2st64_b64 %0, %1 offset1:4" : "=v"(X) : "v"(ldsAddr) : "memory");     // +0 and +256 samples
    else asm volatile("ds_read_b64 %0, %1" : "=v"(X) : "v"(ldsAddr) : "memory");
}
//! the loads have landed; everything passes through so that no use can be scheduled above the wait
template <int RM>
__device__ __forceinline__ void tapWait(TapCoef &G, typename TapRegs<RM>::X &X, unsigned &offNext)
{
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+s"(G.c0), "+s"(G.c1), "+s"(G.c2), "+s"(G.c3), "+s"(G.c4), "+s"(G.c5), "+s"(G.c6), "+s"(G.c7), "+v"(X), "+s"(offNext));
}
template <int RM>
__device__ __forceinline__ void tapFma(v2f (&acc)[RM][CHAN_KG], const TapCoef &G, const typename TapRegs<RM>::X &X)
{
    const v2f gk[CHAN_KG] = {G.c0, G.c1, G.c2, G.c3, G.c4, G.c5, G.c6, G.c7};
    v2f x[RM];
    if constexpr (RM == 2) { x[0] = (v2f){X[0], X[1]}; x[1] = (v2f){X[2], X[3]}; }
    else x[0] = X;
    // the two halves of a complex multiply-add use the same accumulator: all first halves, then all second halves, so
    // that no packed FMA waits for the one issued just before it
#pragma unroll
    for (int k = 0; k < CHAN_KG; k++)
#pragma unroll
        for (int r = 0; r < RM; r++)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "+v"(acc[r][k]) : "s"(gk[k]), "v"(x[r]));                  // (g.x*x.x, g.x*x.y)
#pragma unroll
    for (int k = 0; k < CHAN_KG; k++)
#pragma unroll
        for (int r = 0; r < RM; r++)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[1,0,0]" : "+v"(acc[r][k]) : "s"(gk[k]), "v"(x[r]));   // (-g.y*x.y, g.y*x.x)
}

//! e^{-2 pi i ph / 2^32}: nearest quarter turn taken out exactly, then the fp32 sine / cosine kernels on [-pi/4, pi/4]
//! (minimax polynomials, error ~1e-7); the phase bits below 2^-32 turn (1.5e-9 rad) are dropped
__device__ __forceinline__ v2f mixerPhase(const unsigned ph)
{
    const unsigned q = (ph + 0x20000000u) >> 30;                                    // quadrant 0..3 (4 wraps to 0 below)
    const float x = float(int(ph - (q << 30))) * 1.4629180792671596e-09f;           // 2 pi / 2^32
    const float z = x * x;
    const float sn = fmaf(x * z, fmaf(z, fmaf(z, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), x);
    const float cs = fmaf(z, fmaf(z, fmaf(z, fmaf(z, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), -0.5f), 1.0f);
    // angle = q * pi/2 + x; the result is (cos, -sin) of it
    const float c1 = (q & 1) ? -sn : cs, s1 = (q & 1) ? cs : sn;                    // cos/sin of (x + pi/2) = (-sin x, cos x)
    const bool neg = (q & 2) != 0;
    return (v2f){neg ? -c1 : c1, neg ? s1 : -s1};
}

//! the common tile: wholly inside this call's chunk. Uniform base + 32-bit byte offsets (no 64-bit vector arithmetic),
//! 8 loads in flight per lane and round; the LDS slot walks by a constant with one conditional wrap
__device__ __forceinline__ void tileStageInside(float2 *xs, const float2 *chunkAt, const int D, const int QP, const int TI, const int t)
{
    const char *__restrict__ bp = reinterpret_cast<const char *>(chunkAt);
    const int dq = CHAN_THREADS / D, dp = CHAN_THREADS - dq * D;
    const int step = dp * QP + dq, wrap = 1 - D * QP;           // next sample 256 on: phase + dp (mod D), slot + dq (+ 1 on wrap)
    int q = t / D, p = t - q * D;
    int idx = p * QP + q;
    for (int tt = t; tt < TI; tt += 8 * CHAN_THREADS)
    {
        float2 v[8];
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = *reinterpret_cast<const float2 *>(bp + min(unsigned(tt + u * CHAN_THREADS), unsigned(TI - 1)) * 8u);
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (tt + u * CHAN_THREADS < TI) xs[idx] = v[u];
            p += dp; idx += step;
            if (p >= D) { p -= D; idx += wrap; }
        }
    }
}
//! any tile, fetch and store back to back: 8 loads in flight per lane and round
__device__ __forceinline__ void tileStage(float2 *xs, const ChanArgs &a, const float2 *chunk, const long long tileStart, const int D, const int QP, const int TI, const int t)
{
    const int dq = CHAN_THREADS / D, dp = CHAN_THREADS - dq * D;
    for (int tt = t; tt < TI; tt += 8 * CHAN_THREADS)
    {
        float2 v[8];
        int q = tt / D, p = tt - q * D;
#pragma unroll
        for (int u = 0; u < 8; u++) v[u] = tt + u * CHAN_THREADS < TI ? streamSample(a, chunk, tileStart + tt + u * CHAN_THREADS) : make_float2(0.0f, 0.0f);
#pragma unroll
        for (int u = 0; u < 8; u++)
        {
            if (tt + u * CHAN_THREADS < TI) xs[p * QP + q] = v[u];
            p += dp; q += dq;
            if (p >= D) { p -= D; q++; }
        }
    }
}

// One workgroup = one tile of 256*RM output times x one group of 8 channels; blockIdx.x = tile * nGroups + group: the groups
// of a tile are neighbours in launch order and share the tile's input in L2.
template <int RM>
__global__ __launch_bounds__(CHAN_THREADS) void channelize(const ChanArgs a)
{
    extern __shared__ float2 xs[];
    constexpr int TM = CHAN_THREADS * RM;
    const int t = threadIdx.x;
    const int D = a.D, L = a.L, QP = a.QP;
    const int TI = (TM - 1) * D + L;
    // tiles sit on absolute multiples of TM, so an output's place in its tile -- and with it every rounding -- does not
    // depend on how the stream was cut into calls
    const int group = int(blockIdx.x % unsigned(a.nGroups));
    const long long mTile = (a.mLo / TM + (long long)(blockIdx.x / unsigned(a.nGroups))) * TM;
    const long long tileStart = (mTile + 1) * D - L;            // oldest sample of the tile's first output
    {
        const float2 *chunk = a.chunk + (size_t)blockIdx.y * a.captureIn;       // this capture's samples (blockIdx.y = 0 for a stream)
        const long long rel = tileStart - a.n0;
        if (rel >= 0 && rel + TI <= a.nChunk) tileStageInside(xs, chunk + rel, D, QP, TI, t);
        else tileStage(xs, a, chunk, tileStart, D, QP, TI, t);
    }
    __syncthreads();

    v2f acc[RM][CHAN_KG];
#pragma unroll
    for (int r = 0; r < RM; r++)
#pragma unroll
        for (int k = 0; k < CHAN_KG; k++) acc[r][k] = (v2f){0.0f, 0.0f};

    // One tap per step, software-pipelined by hand (the compiler sinks the loads next to their use): while the packed FMAs
    // of tap j run, tap j+1's LDS read and scalar coefficient load are in flight. Both return through lgkmcnt and scalar
    // loads complete out of order, so the only safe wait is lgkmcnt(0) -- placed BEFORE the next issue, one whole FMA block
    // (16*RM v_pk_fma_f32) after the loads it waits for were issued.
    const v2f *__restrict__ g = a.taps + (size_t)group * (L + 1) * CHAN_KG;
    const unsigned lds0 = unsigned(uintptr_t(xs)) + unsigned(t) * 8u;       // low half of a flat LDS address = the LDS offset
    // Nothing but the loads, the wait and the FMAs is left in the loop: the LDS offset of every tap ((j mod D)*QP + j/D
    // samples) comes from a table, one tap ahead. Both tables carry a dummy entry past the end for the last prefetch.
    const unsigned *__restrict__ op = a.tapOff;
    typename TapRegs<RM>::X xA;
    TapCoef gA;
    unsigned offA;                                          // LDS offset of the tap after the one in the A registers
    tapIssue<RM>(gA, xA, offA, g, op + 1, lds0);
    for (int jr = 0; jr < L; jr += 2)                       // L is even (the host pads an odd filter with a zero tap)
    {
        typename TapRegs<RM>::X xB;
        TapCoef gB;
        unsigned offB;
        g += 2 * CHAN_KG; op += 2;
        tapWait<RM>(gA, xA, offA);
#ifdef LORAHIP_CHAN_EXP_NOLOAD
        gB = gA; xB = xA; offB = offA;
#else
        tapIssue<RM>(gB, xB, offB, g - CHAN_KG, op, lds0 + offA);
#endif
        tapFma<RM>(acc, gA, xA);
        tapWait<RM>(gB, xB, offB);
#ifndef LORAHIP_CHAN_EXP_NOLOAD
        tapIssue<RM>(gA, xA, offA, g, op + 1, lds0 + offB);
#endif
        tapFma<RM>(acc, gB, xB);
    }
    tapWait<RM>(gA, xA, offA);                              // nothing may still be in flight when the registers are reused

    // Mixer phase of the output instant e^{-i theta_k n_m} = (phase at the tile's first output) x (phase over t*D more
    // samples): lane k of every wavefront evaluates the first factor for channel k -- one sine/cosine per lane instead of
    // eight --, the second comes from a per-channel table of 256 entries; 256 outputs on it is one more constant step.
    // Tiles sit on absolute output indices, so all of this is independent of how the stream was cut into calls.
    const int chBase = group * CHAN_KG;
    const v2f mine = mixerPhase(unsigned((a.w[chBase + (t & (CHAN_KG - 1))] * (unsigned long long)((mTile + 1) * D - 1)) >> 32));
    int mineRe = __float_as_int(mine.x), mineIm = __float_as_int(mine.y);
    asm volatile("" : "+v"(mineRe), "+v"(mineIm));             // two separate registers for the lane reads below
    const int mLoc = int(mTile - a.mLo) + t;                    // this call's output index (a call makes < 2^30 outputs)
#pragma unroll
    for (int k = 0; k < CHAN_KG; k++)
    {
        const int ch = chBase + k;                              // w, step, laneRot are padded to whole groups
        const v2f base = {__int_as_float(__builtin_amdgcn_readlane(mineRe, k)), __int_as_float(__builtin_amdgcn_readlane(mineIm, k))};
        v2f rot = cmulF(a.laneRot[ch * CHAN_THREADS + t], base);
        char *o = reinterpret_cast<char *>(a.out + (size_t)blockIdx.y * a.captureOut + (size_t)ch * a.outStride);     // uniform; + 32-bit byte offset per lane
#pragma unroll
        for (int r = 0; r < RM; r++)
        {
            if (r) rot = cmulF(rot, a.step[ch]);                // the output 256 places on: phase advanced by w*256*D
            const v2f y = cmulF(acc[r][k], rot);
            const int ml = mLoc + r * CHAN_THREADS;
            if (ch < a.K && ml >= 0 && ml < int(a.nOut)) *reinterpret_cast<float2 *>(o + unsigned(ml) * 8u) = make_float2(y.x, y.y);
        }
    }
}

//! the HC samples that precede the next call
__global__ void chanHistory(const ChanArgs a, float2 *newHist)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.histLen) newHist[i] = streamSample(a, a.chunk, a.n0 + a.nChunk - a.histLen + i);
}

static unsigned long long gLdsMask[2] = {0, 0};

//! captures == 0: the next nIn samples of THE stream (history and phase carried). captures > 0: that many independent captures of nIn
//! samples each, every one from sample 0 with zero history; the stream state is not touched.
static int chanRun(lorahip_channelizer *c, const float2 *wide, const size_t nIn, float2 *out, const size_t outStride, size_t *nOutP,
                   const size_t captures = 0, const size_t captureStride = 0)
{
    lorahip_ctx *ctx = c->ctx;
    const DeviceGuard guard(ctx->device);
    const unsigned long long D = (unsigned long long)c->D;
    const unsigned long long n0 = captures ? 0 : c->n0;
    const unsigned long long mLo = n0 / D, mHi = (n0 + nIn) / D;
    const size_t nOut = size_t(mHi - mLo);
    if (nOutP) *nOutP = nOut;
    if (nIn == 0) return LORAHIP_OK;
    if (nOut && (out == nullptr || outStride < nOut)) return LORAHIP_E_INVALID;
    if (nOut > (size_t(1) << 30)) { setLastError("channeliser: more than 2^30 outputs per channel in one call"); return LORAHIP_E_INVALID; }
    ChanArgs a;
    a.chunk = wide; a.nChunk = (long long)nIn;
    a.hist = c->dHist[c->cur]; a.histLen = captures ? 0 : c->HC;        // no history: samples before the capture read as 0
    a.n0 = (long long)n0;
    a.captureIn = (long long)captureStride; a.captureOut = (long long)(size_t(c->K) * outStride);
    a.taps = reinterpret_cast<const v2f *>(c->dTaps);
    a.w = c->dW;
    a.step = reinterpret_cast<const v2f *>(c->dRot);
    a.laneRot = a.step + size_t(c->nGroups) * CHAN_KG;
    a.tapOff = c->dTapOff;
    a.out = out; a.outStride = (long long)outStride;
    a.mLo = (long long)mLo; a.nOut = (long long)nOut;
    a.K = c->K; a.L = c->L; a.D = c->D; a.QP = c->QP; a.nGroups = c->nGroups;
    if (nOut)
    {
        const int TM = CHAN_THREADS * c->RM;
        const size_t nBlocks = size_t(c->nGroups) * ((mLo % TM + nOut + TM - 1) / TM);
        if (nBlocks > 0x7fffffffu) { setLastError("channeliser: channels x outputs of one call exceed the launch grid"); return LORAHIP_E_INVALID; }
        const dim3 grid((unsigned)nBlocks, (unsigned)(captures ? captures : 1));
        if (c->RM == 2)
        {
            LORAHIP_TRY(ensureDynamicLds(reinterpret_cast<const void *>(&channelize<2>), 160 * 1024, gLdsMask[1]));
            hipLaunchKernelGGL(channelize<2>, grid, dim3(CHAN_THREADS), c->ldsBytes, ctx->stream, a);
        }
        else
        {
            LORAHIP_TRY(ensureDynamicLds(reinterpret_cast<const void *>(&channelize<1>), 160 * 1024, gLdsMask[0]));
            hipLaunchKernelGGL(channelize<1>, grid, dim3(CHAN_THREADS), c->ldsBytes, ctx->stream, a);
        }
        LORAHIP_TRY(hipGetLastError());
    }
    if (captures) return LORAHIP_OK;
    hipLaunchKernelGGL(chanHistory, dim3((c->HC + 255) / 256), dim3(256), 0, ctx->stream, a, c->dHist[c->cur ^ 1]);
    LORAHIP_TRY(hipGetLastError());
    c->cur ^= 1;
    c->n0 += nIn;
    return LORAHIP_OK;
}

} // namespace lorahip

using namespace lorahip;

extern "C" {

uint64_t lorahip_channelizer_phase_inc(const double freq)
{
    if (!std::isfinite(freq)) return 0;
    const double frac = freq - std::floor(freq);                // [0, 1)
    return frac >= 1.0 ? 0 : uint64_t(std::ldexp(frac, 64));
}

int lorahip_channelizer_create(lorahip_channelizer **out, lorahip_ctx *ctx, const size_t n_channels, const double *freq,
                               const size_t decim, const float *taps, const size_t n_taps)
{
    if (out == nullptr) return LORAHIP_E_INVALID;
    *out = nullptr;
    if (ctx == nullptr || freq == nullptr || taps == nullptr || n_channels == 0 || n_channels > 65535u * CHAN_KG ||
        decim == 0 || decim > CHAN_THREADS || n_taps == 0 || n_taps > (1u << 16))
        return LORAHIP_E_INVALID;
    const int D = int(decim), Lu = int(n_taps);
    const int L = (Lu + 1) & ~1;         // the kernel takes taps in pairs: an odd filter gets a zero tap on the OLD end (j = Lu), where
                                         // the sample is the same however the stream is cut
    // two output times per lane when the tile fits 64 KiB, one otherwise; the LDS limit is 160 KiB per workgroup
    int RM = 2, QP = 0;
    if (const char *e = std::getenv("LORAHIP_CHAN_RM")) { if (std::atoi(e) == 1) RM = 1; }      // measurement hook
    size_t lds = 0;
    for (; RM >= 1; RM--)
    {
        QP = (CHAN_THREADS * RM + (L - 1) / D + 1) | 1;
        lds = size_t(D) * size_t(QP) * sizeof(float2);
        if (lds <= (RM == 2 ? (64u << 10) : (160u << 10))) break;
    }
    if (RM < 1) { setLastError("channeliser: decim * (256 + n_taps/decim) samples do not fit the LDS"); return LORAHIP_E_INVALID; }

    lorahip_channelizer *c = new (std::nothrow) lorahip_channelizer();
    if (c == nullptr) return LORAHIP_E_NOMEM;
    c->ctx = ctx; c->K = int(n_channels); c->L = L; c->D = D; c->HC = L - 1 + D; c->QP = QP; c->RM = RM;
    c->nGroups = int((n_channels + CHAN_KG - 1) / CHAN_KG);
    c->ldsBytes = lds; c->dTaps = nullptr; c->dTapOff = nullptr; c->dW = nullptr; c->dRot = nullptr; c->dHist[0] = c->dHist[1] = nullptr; c->cur = 0; c->n0 = 0;

    const size_t KP = size_t(c->nGroups) * CHAN_KG;
    std::vector<unsigned long long> w(KP, 0);
    std::vector<float2> g(size_t(c->nGroups) * size_t(L + 1) * CHAN_KG, make_float2(0.0f, 0.0f));
    std::vector<unsigned> off(size_t(L) + 2, 0u);
    for (int jr = 0; jr < L; jr++) off[size_t(jr)] = unsigned((jr % D) * QP + jr / D) * unsigned(sizeof(float2));
    for (size_t k = 0; k < n_channels; k++)
    {
        w[k] = lorahip_channelizer_phase_inc(freq[k]);
        for (int jr = 0; jr < L; jr++)
        {
            const int j = L - 1 - jr;
            if (j >= Lu) continue;                                  // the pad
            const double turns = std::ldexp(double((long long)(w[k] * (unsigned long long)j)), -64);     // [-0.5, 0.5)
            const double ang = 2.0 * M_PI * turns;
            g[((k / CHAN_KG) * size_t(L + 1) + size_t(jr)) * CHAN_KG + k % CHAN_KG] =
                make_float2(float(double(taps[j]) * std::cos(ang)), float(double(taps[j]) * std::sin(ang)));
        }
    }
    std::vector<float2> rot(KP * (1 + CHAN_THREADS));
    for (size_t k = 0; k < KP; k++)
        for (int t = 0; t <= CHAN_THREADS; t++)
        {
            // t = 256: the step over one row of lanes; t < 256: lane t's share
            const double ang = 2.0 * M_PI * std::ldexp(double((long long)(w[k] * (unsigned long long)(t * D))), -64);
            rot[t == CHAN_THREADS ? k : KP + k * CHAN_THREADS + size_t(t)] = make_float2(float(std::cos(ang)), float(-std::sin(ang)));
        }
    const DeviceGuard guard(ctx->device);
    const size_t histBytes = size_t(c->HC) * sizeof(float2);
    if (hipMalloc((void **)&c->dTaps, g.size() * sizeof(float2)) != hipSuccess ||
        hipMalloc((void **)&c->dTapOff, off.size() * sizeof(unsigned)) != hipSuccess ||
        hipMalloc((void **)&c->dW, w.size() * sizeof(unsigned long long)) != hipSuccess ||
        hipMalloc((void **)&c->dRot, rot.size() * sizeof(float2)) != hipSuccess ||
        hipMalloc((void **)&c->dHist[0], histBytes) != hipSuccess || hipMalloc((void **)&c->dHist[1], histBytes) != hipSuccess)
    {
        lorahip_channelizer_destroy(c);
        return LORAHIP_E_NOMEM;
    }
    hipError_t e = hipMemcpy(c->dTaps, g.data(), g.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->dTapOff, off.data(), off.size() * sizeof(unsigned), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->dW, w.data(), w.size() * sizeof(unsigned long long), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->dRot, rot.data(), rot.size() * sizeof(float2), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemset(c->dHist[0], 0, histBytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) { lorahip_channelizer_destroy(c); return hipFail(e, "channeliser table upload"); }
    *out = c;
    return LORAHIP_OK;
}

void lorahip_channelizer_destroy(lorahip_channelizer *c)
{
    if (c == nullptr) return;
    const DeviceGuard guard(c->ctx->device);
    if (c->dTaps) (void)hipFree(c->dTaps);
    if (c->dW) (void)hipFree(c->dW);
    if (c->dTapOff) (void)hipFree(c->dTapOff);
    if (c->dRot) (void)hipFree(c->dRot);
    if (c->dHist[0]) (void)hipFree(c->dHist[0]);
    if (c->dHist[1]) (void)hipFree(c->dHist[1]);
    delete c;
}

int lorahip_channelizer_reset(lorahip_channelizer *c)
{
    if (c == nullptr) return LORAHIP_E_INVALID;
    const DeviceGuard guard(c->ctx->device);
    LORAHIP_TRY(hipMemsetAsync(c->dHist[c->cur], 0, size_t(c->HC) * sizeof(float2), c->ctx->stream));
    c->n0 = 0;
    return LORAHIP_OK;
}

size_t lorahip_channelizer_out_count(const lorahip_channelizer *c, const size_t n_in)
{
    if (c == nullptr) return 0;
    const unsigned long long D = (unsigned long long)c->D;
    return size_t((c->n0 + n_in) / D - c->n0 / D);
}

int lorahip_channelizer_run(lorahip_channelizer *c, const float *wide_dev, const size_t n_in, float *out_dev,
                            const size_t out_stride, size_t *n_out)
{
    if (c == nullptr || (n_in && wide_dev == nullptr)) return LORAHIP_E_INVALID;
    return chanRun(c, reinterpret_cast<const float2 *>(wide_dev), n_in, reinterpret_cast<float2 *>(out_dev), out_stride, n_out);
}

int lorahip_channelizer_run_captures(lorahip_channelizer *c, const float *wide_dev, const size_t n_captures, const size_t capture_stride,
                                     const size_t n_in, float *out_dev, const size_t out_stride, size_t *n_out)
{
    if (c == nullptr || n_captures > 65535u || (n_captures && n_in && (wide_dev == nullptr || capture_stride < n_in))) return LORAHIP_E_INVALID;
    if (n_out) *n_out = n_in / size_t(c->D);
    if (n_captures == 0 || n_in == 0) return LORAHIP_OK;
    return chanRun(c, reinterpret_cast<const float2 *>(wide_dev), n_in, reinterpret_cast<float2 *>(out_dev), out_stride, n_out, n_captures, capture_stride);
}

} // extern "C"

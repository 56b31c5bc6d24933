// CDNA4 (gfx950) kernels of the LoRa demod hot path: dechirp -> 2^SF-point FFT -> detect.
// This TU: the generic LDS kernel (variant 1, every SF), the synthetic-IQ generator and the
// launch dispatch. The tuned register/LDS-phase kernels live in lorahip_fast.hip.
// Numerics contract and shared helpers: lorahip_device.h.
#include "lorahip_device.h"

namespace lorahip {

/***********************************************************************
 * Variant 1: generic LDS kernel. N/4 threads per window, every stage through LDS.
 * Correct for every SF; the tuned per-SF kernels below are validated against it.
 **********************************************************************/
template <int LOG2N>
__global__ void __launch_bounds__((1 << LOG2N) / 4 < 256 ? 256 : (1 << LOG2N) / 4)
detectGeneric(const DetectArgs a)
{
    typedef Plan<LOG2N> P;
    constexpr int N = P::N;
    constexpr int TW = N / 4;                        // threads per window
    constexpr int BLOCK = TW < 256 ? 256 : TW;
    constexpr int WPB = BLOCK / TW;                  // windows per block
    constexpr int WAVES_PER_WIN = TW >= 64 ? TW / 64 : 1;
    constexpr int M = N * LORAHIP_FINE_STEPS;

    __shared__ float2 sA[WPB][N];
    __shared__ int sIdx[WPB][N];
    __shared__ float sRedV[WPB][WAVES_PER_WIN];
    __shared__ int sRedI[WPB][WAVES_PER_WIN];
    __shared__ double sRedT[WPB][WAVES_PER_WIN];

    const int lw = threadIdx.x / TW;                 // local window
    const int t = threadIdx.x % TW;
    const unsigned w = blockIdx.x * WPB + lw;
    const bool active = w < a.nWindows;
    float2 *A = sA[lw];

    int sel = LORAHIP_CHIRP_NONE;
    int idx0 = 0;
    float err = 0.0f;
    const float2 *in = nullptr;
    if (active)
    {
        sel = a.chirpSel ? a.chirpSel[w] : a.chirpSelAll;
        idx0 = a.fineIdx0 ? a.fineIdx0[w] : 0;
        err = a.fineErr ? a.fineErr[w] : 0.0f;
        in = a.iq + (a.offsets ? a.offsets[w] : (long long)w * a.stride);
    }
    const bool dechirp = sel != LORAHIP_CHIRP_NONE;
    const float d = err * (float)LORAHIP_FINE_STEPS;
    const bool moving = dechirp && d != 0.0f;        // index changes from sample to sample

    // fine-tune index chain (sequential by definition; one lane walks it)
    if (moving && t == 0)
    {
        int idx = idx0;
        for (int i = 0; i < N; i++) { sIdx[lw][i] = idx; idx = fineStep(idx, d, M); }
        if (a.fineIdxOut) a.fineIdxOut[w] = idx;
    }
    if (active && !moving && t == 0 && a.fineIdxOut) a.fineIdxOut[w] = idx0;
    __syncthreads();

    // load + dechirp (LoRaDemod.cpp:158-159), scatter to the DIT work-array order
    if (active)
    {
        const float2 *chirp = sel == LORAHIP_CHIRP_DOWN ? a.down : a.up;
        const float2 fconst = a.fine[idx0];
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int n = t + j * TW;
            float2 x = in[n];
            if (dechirp)
            {
                x = cmul(x, chirp[n]);
                x = cmul(x, moving ? a.fine[sIdx[lw][n]] : fconst);
            }
            if (a.decOut) a.decOut[(size_t)w * N + n] = x;
            A[P::pos(n)] = x;
        }
    }
    __syncthreads();

    // innermost radix-2 stage: m = 1, fstride = N/2, twiddle(0)
    if (P::HAS_R2)
    {
        if (active)
        {
            const float2 t0 = a.tw[0];
#pragma unroll
            for (int j = 0; j < 2; j++)
            {
                const int b = t + j * TW;
                float2 f0 = A[2 * b], f1 = A[2 * b + 1];
                bfly2(f0, f1, t0);
                A[2 * b] = f0; A[2 * b + 1] = f1;
            }
        }
        __syncthreads();
    }

    // radix-4 stages, innermost (smallest m) first
#pragma unroll
    for (int s = P::R4 - 1; s >= 0; s--)
    {
        const int m = N >> (2 * (s + 1));            // remainder of stage s
        const int fstride = 1 << (2 * s);            // 4^s
        if (active)
        {
            const int k = t & (m - 1);
            const int base = ((t / m) * 4 * m) + k;
            float2 f0 = A[base], f1 = A[base + m], f2 = A[base + 2 * m], f3 = A[base + 3 * m];
            bfly4(f0, f1, f2, f3, a.tw[k * fstride], a.tw[k * fstride * 2], a.tw[k * fstride * 3]);
            A[base] = f0; A[base + m] = f1; A[base + 2 * m] = f2; A[base + 3 * m] = f3;
        }
        __syncthreads();
    }

    // scan (LoRaDetector.hpp:36-48): per-lane partials over 4 bins, then tree
    float bestV = 0.0f;
    int bestI = 0;
    double tot = 0.0;
    if (active)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            const int i = t + j * TW;
            const float2 b = A[i];
            if (a.fftOut) a.fftOut[(size_t)w * N + i] = b;
            const float mag2 = b.x * b.x + b.y * b.y;
            tot += (double)mag2;
            if (mag2 > bestV) { bestV = mag2; bestI = i; }
        }
        if (!(bestV > 0.0f)) bestI = 0;
    }
    constexpr int SEG = TW < 64 ? TW : 64;
#pragma unroll
    for (int off = SEG / 2; off > 0; off >>= 1)
    {
        const float ov = __shfl_xor(bestV, off, 64);
        const int oi = __shfl_xor(bestI, off, 64);
        const double ot = __shfl_xor(tot, off, 64);
        argmaxCombine(bestV, bestI, ov, oi);
        tot += ot;
    }
    if (WAVES_PER_WIN > 1)
    {
        const int wv = t / 64;
        if ((t & 63) == 0) { sRedV[lw][wv] = bestV; sRedI[lw][wv] = bestI; sRedT[lw][wv] = tot; }
        __syncthreads();
        if (t == 0)
        {
            bestV = sRedV[lw][0]; bestI = sRedI[lw][0]; tot = sRedT[lw][0];
            for (int k = 1; k < WAVES_PER_WIN; k++)
            {
                argmaxCombine(bestV, bestI, sRedV[lw][k], sRedI[lw][k]);
                tot += sRedT[lw][k];
            }
        }
    }
    if (active && t == 0)
    {
        const float2 l = A[bestI > 0 ? bestI - 1 : N - 1];
        const float2 r = A[bestI < N - 1 ? bestI + 1 : 0];
        detectTail(a, w, bestI, bestV, tot, l, r);
    }
}

template <int LOG2N>
static hipError_t launchGeneric(const DetectArgs &a, hipStream_t stream)
{
    constexpr int N = 1 << LOG2N;
    constexpr int TW = N / 4;
    constexpr int BLOCK = TW < 256 ? 256 : TW;
    constexpr int WPB = BLOCK / TW;
    const unsigned grid = (a.nWindows + WPB - 1) / WPB;
    if (grid == 0) return hipSuccess;
    hipLaunchKernelGGL(detectGeneric<LOG2N>, dim3(grid), dim3(BLOCK), 0, stream, a);
    return hipGetLastError();
}

} // namespace lorahip
// the contracted build's entry points (lorahip_fma_fast.hip / lorahip_fma_wide.hip: their own namespace, plain C hand-over)
extern "C" int lorahip_fma_fast_launch(int sf, const void *args, const void *tables, void *stream);
extern "C" int lorahip_fma_wide_launch(int sf, const void *args, const void *tables, void *stream);
namespace lorahip {

hipError_t launchDetect(const int sf, const int variant, const DetectArgs &a, const FastTables &ft, hipStream_t stream)
{
    if (variant == LORAHIP_VARIANT_FMA)
    {
        if (sf >= 6 && sf <= 10) return hipError_t(lorahip_fma_fast_launch(sf, &a, &ft, stream));
        if (sf == 11 || sf == 12) return hipError_t(lorahip_fma_wide_launch(sf, &a, &ft, stream));
        return hipErrorInvalidValue;
    }
    if (variant != 1 && fastAvailable(sf)) return launchFast(sf, variant, a, ft, stream);
    if (sf == 11 && variant >= 20 && variant <= 24) return launchFast(sf, variant, a, ft, stream);          // window-per-wavefront SF11 (variants 20+)
    if ((sf == 11 || sf == 12) && (variant == 25 || variant == 26)) return launchFast(sf, variant, a, ft, stream);   // 64 points per lane (profiling builds)
    if (variant != 1 && wideAvailable(sf)) return launchWide(sf, variant, a, ft, stream);
    switch (sf)
    {
    case 6: return launchGeneric<6>(a, stream);
    case 7: return launchGeneric<7>(a, stream);
    case 8: return launchGeneric<8>(a, stream);
    case 9: return launchGeneric<9>(a, stream);
    case 10: return launchGeneric<10>(a, stream);
    case 11: return launchGeneric<11>(a, stream);
    case 12: return launchGeneric<12>(a, stream);
    default: return hipErrorInvalidValue;
    }
}

/***********************************************************************
 * Synthetic up-chirp symbols + AWGN, generated in HBM (bench / test input).
 * Window w = ampl * exp(j*phi_i), phi following ChirpGenerator.hpp:22-47 for an up-chirp
 * with f0 = 2*pi*sym/N and ovs = 1 in closed form (per-window phase origin 0):
 *   f_i = -pi + f0 + (i+1)*2pi/N, wrapped by -2pi once it exceeds +pi
 *   phi_i = sum_{j<=i} f_j
 * evaluated in fp64 then rounded; noise: counter-based splitmix64 -> Box-Muller.
 **********************************************************************/
__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

template <int LOG2N>
__global__ void synthSymbols(float2 *iq, const unsigned short *sym, const size_t nWindows,
                             const float ampl, const float sigma, const unsigned long long seed)
{
    constexpr int N = 1 << LOG2N;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = nWindows * (size_t)N;
    for (size_t e = gid; e < total; e += (size_t)gridDim.x * blockDim.x)
    {
        const size_t w = e >> LOG2N;
        const int i = (int)(e & (N - 1));
        const int s = sym[w] & (N - 1);
        // number of samples j<=i whose instantaneous frequency has already wrapped: j+1+s > N  <=>  j >= N-s
        const double twoPiN = 6.283185307179586476925 / N;
        const double n1 = (double)(i + 1);
        double phi = n1 * (-3.14159265358979323846 + twoPiN * s) + twoPiN * 0.5 * n1 * (n1 + 1.0);
        const int wrapped = i - (N - s) + 1;     // count of wrapped samples among 0..i
        if (wrapped > 0) phi -= 6.283185307179586476925 * (double)wrapped;
        phi -= 6.283185307179586476925 * floor(phi / 6.283185307179586476925);
        double sn, cs;
        sincos(phi, &sn, &cs);
        float re = ampl * (float)cs, im = ampl * (float)sn;
        if (sigma > 0.0f)
        {
            const unsigned long long r = splitmix64(seed ^ (e * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
            const double u1 = ((double)(r >> 32) + 1.0) * (1.0 / 4294967296.0);
            const double u2 = (double)(r & 0xffffffffu) * (1.0 / 4294967296.0);
            const double rad = sqrt(-2.0 * log(u1));
            double s2, c2;
            sincos(6.283185307179586476925 * u2, &s2, &c2);
            re += sigma * (float)(rad * c2);
            im += sigma * (float)(rad * s2);
        }
        iq[e] = make_float2(re, im);
    }
}

hipError_t launchSynth(const int sf, float2 *iq, const unsigned short *sym, const size_t nWindows,
                       const float ampl, const float sigma, const unsigned long long seed, hipStream_t stream)
{
    if (nWindows == 0) return hipSuccess;
    const dim3 block(256), grid(2048);
    switch (sf)
    {
#define LORAHIP_SYNTH_CASE(SF) case SF: hipLaunchKernelGGL(synthSymbols<SF>, grid, block, 0, stream, iq, sym, nWindows, ampl, sigma, seed); break;
    LORAHIP_SYNTH_CASE(6) LORAHIP_SYNTH_CASE(7) LORAHIP_SYNTH_CASE(8) LORAHIP_SYNTH_CASE(9)
    LORAHIP_SYNTH_CASE(10) LORAHIP_SYNTH_CASE(11) LORAHIP_SYNTH_CASE(12)
#undef LORAHIP_SYNTH_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}


/***********************************************************************
 * Batched modulator: the frame of the LoRaMod block (LoRaMod.cpp:109-238) for many packets at once,
 * one lane per frame. genChirp<float> (ChirpGenerator.hpp:22-47) is a sequential float recurrence --
 * `f += fStep` with wrap, `phaseAccum +-= f`, one running accumulator through the whole frame, reduced
 * mod 2pi at the end of every chirp -- so a frame is walked by one lane exactly as the reference walks
 * it, 64 frames per wavefront; 16 samples at a time go through LDS so that the stores are 128-byte
 * rows. polar(ampl, phase) = (ampl*cosf, ampl*sinf): cos/sin are evaluated in fp64 and rounded once,
 * i.e. correctly rounded; glibc's cosf/sinf agree with that except for rare last-ulp cases (tests/).
 **********************************************************************/
__global__ void __launch_bounds__(256) modFrames(float2 *__restrict__ iq, const long long frameStride,
                                                 const unsigned short *__restrict__ syms, const unsigned nFrames,
                                                 const int nsyms, const int sync, const float ampl, const int padding, const int N)
{
    __shared__ float2 stage[4][64][17];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned frame0 = (blockIdx.x * 4 + wave) * 64;
    const unsigned frame = frame0 + lane;
    const bool mine = frame < nFrames;
    const unsigned short *mySyms = syms + (size_t)(mine ? frame : 0) * nsyms;
    const float fMin = (float)(-M_PI), fMax = (float)M_PI;                 // ovs = 1   ChirpGenerator.hpp:25-26
    const float fStep = (float)((2 * M_PI) / N);                           //           :27
    float phaseAccum = 0.0f;                                               // LoRaMod.cpp:135
    long long pos = 0;
    const int pad = padding < 1 ? 1 : padding;                             // one zero symbol is emitted before the test (LoRaMod.cpp:218-224)
    const int nChirps = 10 + 2 + 3 + nsyms + pad;
    for (int c = 0; c < nChirps; c++)
    {
        // what this chirp is (LoRaMod.cpp:141-229); the structure is the same for every frame, only f0 differs
        int NN = N;
        bool down = false, zero = false;
        float f0 = 0.0f;
        if (c < 10) {}
        else if (c == 10) f0 = (float)((2 * M_PI * ((sync >> 4) * 8)) / N);
        else if (c == 11) f0 = (float)((2 * M_PI * ((sync & 0xf) * 8)) / N);
        else if (c < 14) down = true;
        else if (c == 14) { down = true; NN = N / 4; }
        else if (c < 15 + nsyms) f0 = (float)((2 * M_PI * (int)mySyms[c - 15]) / N);
        else zero = true;
        float f = fMin + f0;                                               // ChirpGenerator.hpp:28
        for (int i0 = 0; i0 < NN; i0 += 16)
        {
#pragma unroll 4
            for (int i = 0; i < 16; i++)
            {
                float2 v = make_float2(0.0f, 0.0f);
                if (!zero)
                {
                    f += fStep;                                            // :31 / :39
                    if (f > fMax) f -= (fMax - fMin);
                    phaseAccum = down ? phaseAccum - f : phaseAccum + f;
                    double sn, cs;
                    sincos((double)phaseAccum, &sn, &cs);
                    v = make_float2(ampl * (float)cs, ampl * (float)sn);   // std::polar(ampl, phaseAccum)
                }
                stage[wave][lane][i] = v;
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll 4
            for (int r = 0; r < 16; r++)
            {
                const int fr = r * 4 + (lane >> 4), sidx = lane & 15;
                if (frame0 + fr < nFrames) iq[(size_t)(frame0 + fr) * frameStride + pos + i0 + sidx] = stage[wave][fr][sidx];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        if (!zero) phaseAccum = (float)((double)phaseAccum - floor((double)phaseAccum / (2 * M_PI)) * 2 * M_PI);   // :45
        pos += NN;
    }
}

hipError_t launchModFrames(float2 *iq, const long long frameStride, const unsigned short *syms, const size_t nFrames,
                           const int nsyms, const int sync, const float ampl, const int padding, const int sf, hipStream_t stream)
{
    if (nFrames == 0) return hipSuccess;
    const unsigned grid = unsigned((nFrames + 255) / 256);
    hipLaunchKernelGGL(modFrames, dim3(grid), dim3(256), 0, stream, iq, frameStride, syms, unsigned(nFrames), nsyms, sync, ampl, padding, 1 << sf);
    return hipGetLastError();
}

//! complex AWGN added in place: the channel of the loopback test (TestLoopback.cpp:98-99 adds a noise source to the
//! modulator output). Same counter-based generator as synthSymbols.
__global__ void addAwgn(float2 *iq, const size_t n, const float sigma, const unsigned long long seed)
{
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (size_t e = gid; e < n; e += (size_t)gridDim.x * blockDim.x)
    {
        const unsigned long long r = splitmix64(seed ^ (e * 0xD1342543DE82EF95ull + 0x632BE59BD9B4E019ull));
        const double u1 = ((double)(r >> 32) + 1.0) * (1.0 / 4294967296.0);
        const double u2 = (double)(r & 0xffffffffu) * (1.0 / 4294967296.0);
        const double rad = sqrt(-2.0 * log(u1));
        double s2, c2;
        sincos(6.283185307179586476925 * u2, &s2, &c2);
        float2 v = iq[e];
        v.x += sigma * (float)(rad * c2);
        v.y += sigma * (float)(rad * s2);
        iq[e] = v;
    }
}

hipError_t launchAwgn(float2 *iq, const size_t n, const float sigma, const unsigned long long seed, hipStream_t stream)
{
    if (n == 0 || sigma == 0.0f) return hipSuccess;
    hipLaunchKernelGGL(addAwgn, dim3(2048), dim3(256), 0, stream, iq, n, sigma, seed);
    return hipGetLastError();
}

/***********************************************************************
 * HBM read probe (measurement aid, not part of the demod path): streams `n16` 16-byte words
 * and folds them into one checksum per lane. pattern 0: lane-linear float4 (the classic copy
 * shape); pattern 1: the tuned SF7 kernel's shape -- a wave owns 8 KB, reads it as 8 row loads in
 * which each group of 8 lanes covers 128 contiguous bytes of a different 1 KB window.
 **********************************************************************/
__global__ void __launch_bounds__(256) membwProbe(const float4 *__restrict__ in, const size_t n16, const int pattern, float *out)
{
    const size_t nthreads = (size_t)gridDim.x * blockDim.x;
    const size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (pattern == 0)
    {
        for (size_t i = gid; i < n16; i += nthreads)
        {
            const float4 v = in[i];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
    }
    else
    {
        const int lane = threadIdx.x & 63;
        const size_t wave = gid >> 6, nwaves = nthreads >> 6;
        const size_t nsets = n16 / 512;                   // 8 KB per wave iteration
        for (size_t s = wave; s < nsets; s += nwaves)
        {
            const float4 *base = in + s * 512 + (size_t)(lane >> 3) * 64 + (lane & 7);
            float4 v[8];
#pragma unroll
            for (int r = 0; r < 8; r++) v[r] = base[r * 8];
#pragma unroll
            for (int r = 0; r < 8; r++) { acc.x += v[r].x; acc.y += v[r].y; acc.z += v[r].z; acc.w += v[r].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 1.2345e-30f) out[0] = acc.x;   // keep the loads alive
}

hipError_t launchMembw(const float2 *iq, const size_t nBytes, const int pattern, const int blocks, float *scratch, hipStream_t stream)
{
    hipLaunchKernelGGL(membwProbe, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float4 *>(iq), nBytes / 16, pattern, scratch);
    return hipGetLastError();
}

} // namespace lorahip

// Streaming demodulator: the whole LoRaDemod::work() loop of a channel on the device.
//
// The batch kernels (lorahip_fast.hip) need the host between two windows of the same channel, because the
// reference's frame machine decides from window k where window k+1 starts (LoRaDemod.cpp:219 `total = N - value`,
// :278 `N/4 + freqError/2`, :209 two windows at once). Here a group of T lanes OWNS a channel and walks its stream
// window after window: load -> fine-tune index chain -> dechirp -> FFT -> detect -> power/SNR/fIndex -> the
// reference's 5-state machine (LoRaDemod.cpp:176-312) in registers -> one record per work() call. Channels are the
// only parallel axis: a wavefront carries 64/T of them, a launch carries all of them, and nothing returns to the
// host until every channel has fewer than 2N samples left (or has filled its record buffer: the launch is
// resumable from the saved per-channel state).
//
// Numerics are those of the batch kernels (same FastCore, same tables); the state machine is the one of
// lorahip_demod.cpp's host path, which tests pin against the verbatim LoRaDemod.cpp.
#include "lorahip_streamkernel.h"

namespace lorahip {

// chirp table from LDS (both selections share it), last-phase twiddles in registers (+3-10 % over the LDS table, session 10)
//             LOG2N T VEC NPH PB1 PB2 w/SIMD  X0: ROT PAD S  D   chLDS twLDS prefetch
typedef FastCfg<6,  2, 4,  2,  2,  6,  STREAM_WPS,          2,  1,  0, 0,  true,  false,  0> Stream6;
typedef FastCfg<7,  3, 2,  2,  3,  7,  STREAM_WPS,          1,  1,  0, 0,  true,  STREAM_TWLDS,  0> Stream7;
typedef FastCfg<8,  4, 1,  2,  4,  8,  STREAM_WPS,          0,  1,  0, 0,  true,  STREAM_TWLDS,  0> Stream8;
typedef FastCfg<9,  5, 2,  3,  3,  7,  STREAM_WPS,          2,  1,  1, 8,  true,  STREAM_TWLDS9,  0, false, false, true> Stream9;    // 32 lanes x 16 points, three phases, exchange 1 as row swaps:
                                                                                                                 // with the per-sample fine-tune arithmetic the 32-point geometry spills (0.20 -> 0.26 of the roofline)
typedef FastCfg<10, 6, 1,  3,  4,  8,  STREAM_WPS,          0,  1,  0, 0,  true,  false,  0, false, false, true> Stream10;   // exchange 1 as register row swaps

//! the used columns of a [rows][capacity] record array packed densely (2-byte units): what goes back to the host is what a
//! run filled, not the worst-case capacity
__global__ void compactRows(unsigned short *__restrict__ dst, const unsigned short *__restrict__ src, const size_t rows,
                            const size_t srcPitch, const size_t width)
{
    const size_t total = rows * width;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
    {
        const size_t r = i / width;
        dst[i] = src[r * srcPitch + (i - r * width)];
    }
}

hipError_t launchCompactRows(void *dst, const void *src, const size_t rows, const size_t srcPitchBytes, const size_t rowBytes, hipStream_t stream)
{
    if (rows == 0 || rowBytes == 0) return hipSuccess;
    if ((srcPitchBytes | rowBytes) & 1) return hipErrorInvalidValue;
    const size_t total = rows * (rowBytes / 2);
    const unsigned grid = unsigned(total / 256 + 1 > 65536 ? 65536 : total / 256 + 1);
    hipLaunchKernelGGL(compactRows, dim3(grid), dim3(256), 0, stream, static_cast<unsigned short *>(dst), static_cast<const unsigned short *>(src),
                       rows, srcPitchBytes / 2, rowBytes / 2);
    return hipGetLastError();
}

//! nSeg independent copies of cf32 runs: segment i moves len[i] elements from src + srcOff[i] to dst + dstOff[i] (the level-3
//! debug ports: replayed windows -> the per-channel port streams)
__global__ void copySegments(float2 *__restrict__ dst, const float2 *__restrict__ src, const long long *__restrict__ srcOff,
                             const long long *__restrict__ dstOff, const int *__restrict__ len, const unsigned nSeg)
{
    for (unsigned i = blockIdx.x; i < nSeg; i += gridDim.x)
    {
        const float2 *s = src + srcOff[i];
        float2 *d = dst + dstOff[i];
        const int n = len[i];
        for (int j = threadIdx.x; j < n; j += blockDim.x) d[j] = s[j];
    }
}

hipError_t launchCopySegments(float2 *dst, const float2 *src, const long long *srcOff, const long long *dstOff, const int *len, const size_t nSeg,
                              hipStream_t stream)
{
    if (nSeg == 0) return hipSuccess;
    const unsigned grid = unsigned(nSeg > 65535 ? 65535 : nSeg);
    hipLaunchKernelGGL(copySegments, dim3(grid), dim3(256), 0, stream, dst, src, srcOff, dstOff, len, unsigned(nSeg));
    return hipGetLastError();
}

/***********************************************************************
 * Packets of a streaming launch straight into the batched decoder's input layout, device to device (no host round trip):
 * channel c posted nPkt[c] packets, packet j = the next pktOut[c][j].len entries of the channel's symbol stream symOut[c][..];
 * its row in the output is rowStart[c] + j (rowStart = exclusive scan of nPkt: channels ascending, time ascending inside one;
 * scanCounts below).
 **********************************************************************/
__global__ void packCopy(const short *__restrict__ symOut, const long long *__restrict__ srcOff, const int *__restrict__ nsyms,
                         unsigned short *__restrict__ dst, const int stride)
{
    const size_t p = blockIdx.x;
    const int len = nsyms[p] < stride ? nsyms[p] : stride;
    const short *src = symOut + srcOff[p];
    for (int i = threadIdx.x; i < stride; i += blockDim.x) dst[p * stride + i] = i < len ? (unsigned short)src[i] : (unsigned short)0;
}

//! rowStart = exclusive prefix sum of nPkt (one workgroup; the channel counts are tens of thousands at most): the packets' rows are
//! numbered on the device, so that packing needs neither an upload nor a host synchronisation. (Describing the packets in the same
//! workgroup -- one launch less -- was tried and is slower by far: 65536 packets walked by 1024 lanes, profiles/r04; doing all three
//! steps in one workgroup for launches of <= 2048 packets, a receiver step of a few windows, is slower too: s43_*.)
__global__ void __launch_bounds__(1024) scanCounts(const int *__restrict__ nPkt, int *__restrict__ rowStart, const unsigned nChannels)
{
    __shared__ int sPart[1024];
    const unsigned per = (nChannels + 1023u) / 1024u, lo = threadIdx.x * per, hi = lo + per < nChannels ? lo + per : nChannels;
    int sum = 0;
    for (unsigned c = lo; c < hi; c++) sum += nPkt[c];
    sPart[threadIdx.x] = sum;
    __syncthreads();
    for (int d = 1; d < 1024; d <<= 1)
    {
        const int v = (int)threadIdx.x >= d ? sPart[threadIdx.x - d] : 0;
        __syncthreads();
        sPart[threadIdx.x] += v;
        __syncthreads();
    }
    int acc = sPart[threadIdx.x] - sum;
    for (unsigned c = lo; c < hi; c++) { rowStart[c] = acc; acc += nPkt[c]; }
}

__global__ void packDescribe(const StreamPacket *__restrict__ pktOut, const int *__restrict__ nPkt, const int *__restrict__ rowStart,
                             const unsigned nChannels, const int cap, const int capPkt, long long *__restrict__ srcOff,
                             int *__restrict__ nsyms, int *__restrict__ channel)
{
    const unsigned c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nChannels) return;
    long long off = (long long)c * cap;
    const int row0 = rowStart[c];
    for (int j = 0; j < nPkt[c]; j++)
    {
        const int len = pktOut[(size_t)c * capPkt + j].len;
        srcOff[row0 + j] = off;
        nsyms[row0 + j] = len;
        if (channel) channel[row0 + j] = (int)c;
        off += len;
    }
}

hipError_t launchPackPackets(const StreamPacket *pktOut, const int *nPkt, const short *symOut, int *rowStart, const size_t nChannels,
                             const int cap, const int capPkt, const size_t nPackets, long long *srcOff, unsigned short *symsOut, const int stride,
                             int *nsymsOut, int *channelOut, hipStream_t stream)
{
    if (nPackets == 0) return hipSuccess;
    hipLaunchKernelGGL(scanCounts, dim3(1), dim3(1024), 0, stream, nPkt, rowStart, unsigned(nChannels));
    hipLaunchKernelGGL(packDescribe, dim3(unsigned((nChannels + 255) / 256)), dim3(256), 0, stream, pktOut, nPkt, rowStart, unsigned(nChannels), cap, capPkt,
                       srcOff, nsymsOut, channelOut);
    hipLaunchKernelGGL(packCopy, dim3(unsigned(nPackets)), dim3(64), 0, stream, symOut, srcOff, nsymsOut, symsOut, stride);
    return hipGetLastError();
}

/***********************************************************************
 * What the host needs after a streaming launch, reduced on the device: one workgroup reads the per-channel state and counts
 * (a few hundred KiB in HBM) and leaves 64 bytes. The per-channel arrays cross PCIe only when an accessor asks for them.
 **********************************************************************/
__global__ void __launch_bounds__(1024) streamSummary(const StreamState *__restrict__ state, const int *__restrict__ nCalls, const int *__restrict__ nSym,
                                                      const int *__restrict__ nPkt, const int *__restrict__ nSig, const unsigned nChannels, const int cap,
                                                      const int capPkt, const unsigned *__restrict__ near, StreamSummary *__restrict__ out)
{
    long long calls = 0, packets = 0, syms = 0, signals = 0, openSyms = 0;
    int more = 0, anyOpen = 0, maxCall = 0, maxOpen = 0, fullest = 0;
    for (unsigned c = threadIdx.x; c < nChannels; c += blockDim.x)
    {
        const int n = nCalls[c], p = nPkt[c], g = nSig ? nSig[c] : 0;
        calls += n; packets += p; syms += nSym[c]; signals += g;
        more |= (n == cap || p == capPkt || (nSig && g == capPkt)) ? 1 : 0;
        fullest = n > fullest ? n : fullest;
        const StreamState st = state[c];
        maxCall = st.callCount > maxCall ? st.callCount : maxCall;
        if (st.state == ST_DATASYMBOLS) { anyOpen = 1; openSyms += st.symCount; maxOpen = st.symCount > maxOpen ? st.symCount : maxOpen; }
    }
    __shared__ long long sL[5][16];
    __shared__ int sI[5][16];
    // wavefront reductions, then the 16 wavefronts' partial results by the first lanes
    for (int d = 32; d >= 1; d >>= 1)
    {
        calls += __shfl_xor(calls, d); packets += __shfl_xor(packets, d); syms += __shfl_xor(syms, d); signals += __shfl_xor(signals, d);
        openSyms += __shfl_xor(openSyms, d);
        more |= __shfl_xor(more, d); anyOpen |= __shfl_xor(anyOpen, d);
        const int a = __shfl_xor(maxCall, d), b = __shfl_xor(maxOpen, d), f = __shfl_xor(fullest, d);
        maxCall = a > maxCall ? a : maxCall; maxOpen = b > maxOpen ? b : maxOpen; fullest = f > fullest ? f : fullest;
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
    {
        sL[0][w] = calls; sL[1][w] = packets; sL[2][w] = syms; sL[3][w] = signals; sL[4][w] = openSyms;
        sI[0][w] = more; sI[1][w] = anyOpen; sI[2][w] = maxCall; sI[3][w] = maxOpen; sI[4][w] = fullest;
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        StreamSummary r;
        r.calls = r.packets = r.syms = r.signals = r.openSyms = 0;
        r.more = r.anyOpen = r.maxCallCount = r.maxOpen = r.fullest = 0;
        for (int k = 0; k < int(blockDim.x >> 6); k++)
        {
            r.calls += sL[0][k]; r.packets += sL[1][k]; r.syms += sL[2][k]; r.signals += sL[3][k]; r.openSyms += sL[4][k];
            r.more |= sI[0][k]; r.anyOpen |= sI[1][k];
            r.maxCallCount = sI[2][k] > r.maxCallCount ? sI[2][k] : r.maxCallCount;
            r.maxOpen = sI[3][k] > r.maxOpen ? sI[3][k] : r.maxOpen;
            r.fullest = sI[4][k] > r.fullest ? sI[4][k] : r.fullest;
        }
        r.nearSquelch = near[0]; r.nearStep = near[1]; r.pad = 0;
        *out = r;
    }
}

hipError_t launchStreamSummary(const StreamState *state, const int *nCalls, const int *nSym, const int *nPkt, const int *nSig, const size_t nChannels,
                               const int cap, const int capPkt, const unsigned *near, StreamSummary *out, hipStream_t stream)
{
    hipLaunchKernelGGL(streamSummary, dim3(1), dim3(1024), 0, stream, state, nCalls, nSym, nPkt, nSig, unsigned(nChannels), cap, capPkt, near, out);
    return hipGetLastError();
}

bool streamAvailable(const int sf) { return sf >= 6 && sf <= 12; }

//! log2 of the lanes a channel gets when nobody forces a choice: the 16-points-per-lane geometry unless the launch has so few
//! channels that a wider one still fits the device's wavefront slots (two per SIMD: lorahip_stream_lanes.hip; thresholds measured,
//! profiles/r05)
static int streamLanesFor(const int sf, const unsigned nChannels)
{
    static int slots = 0;                                   // wavefronts the device holds at two per SIMD
    if (slots == 0)
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        slots = cus * 8;
    }
    const int base = sf - 4 > 6 ? 6 : sf - 4;               // SF6: 4 lanes ... SF10: 64
    int best = base;
    for (int l = base + 1; l <= 6; l++)
        if (streamLanesAvailable(sf, l) && (unsigned long long)nChannels << l <= (unsigned long long)slots * 64u) best = l;
    return best;
}

//! log2 of the lanes per channel a launch over nChannels channels runs on (the current device's size decides where nobody forces it)
int streamLanesChosen(const int sf, const unsigned nChannels, const int forced)
{
    const int base = sf <= 10 ? sf - 4 : sf - 4;            // SF6: 4 lanes ... SF10: 64; SF11 / SF12: a workgroup of 128 / 256
    if (sf < 7 || sf > 9) return base;
    const int lanes = forced > 0 ? forced : (forced < 0 ? base : streamLanesFor(sf, nChannels));
    return lanes != base && streamLanesAvailable(sf, lanes) ? lanes : base;
}

hipError_t launchStream(const int sf, const StreamArgs &s, hipStream_t stream)
{
    if (sf >= 7 && sf <= 9)
    {
        const int lanes = streamLanesChosen(sf, s.nChannels, s.lanes);
        if (lanes != sf - 4) return launchStreamLanes(sf, lanes, s, stream);
    }
    switch (sf)
    {
    case 6: return launchStreamCfg<Stream6>(s, stream);
    case 7: return launchStreamCfg<Stream7>(s, stream);
    case 8: return launchStreamCfg<Stream8>(s, stream);
    case 9: return launchStreamCfg<Stream9>(s, stream);
    case 10: return launchStreamCfg<Stream10>(s, stream);
    case 11: case 12: return launchStreamWide(sf, s, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace lorahip

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
#include "lorahip_streamcfg.h"
#include <cstddef>

namespace lorahip {

// (the geometries: lorahip_streamcfg.h)

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
 * scanDescribe below).
 **********************************************************************/
__global__ void packCopy(const short *__restrict__ symOut, const long long *__restrict__ srcOff, const int *__restrict__ nsyms,
                         unsigned short *__restrict__ dst, const int stride)
{
    const size_t p = blockIdx.x;
    const int len = nsyms[p] < stride ? nsyms[p] : stride;
    const short *src = symOut + srcOff[p];
    for (int i = threadIdx.x; i < stride; i += blockDim.x) dst[p * stride + i] = i < len ? (unsigned short)src[i] : (unsigned short)0;
}

//! The packets' rows are numbered and described on the device, so that packing needs neither an upload nor a host synchronisation:
//! rowStart = exclusive prefix sum of nPkt, and for packet j of channel c, row rowStart[c] + j, where its symbols start in the
//! channel's symbol row, how many they are, which channel. One workgroup per 1024 channels, a lane per channel, and no communication
//! between the workgroups: a workgroup first adds up the counts of all channels BEFORE its own (<= 64 KiB of reads from L2 for the last
//! of 16 at 16384 channels), scans its own 1024, and every lane then walks its channel's few packets. (Round 4: ONE workgroup walking
//! every channel for the row numbers, 20.7 us per receiver step at 16384 channels -- a third of an 8-window step's streaming kernel,
//! 56 us when it ran beside that kernel -- and a second launch, a lane per channel, for the descriptions: profiles/r05/s22_*. Now 6.5 us
//! and one launch, s29_ / s36_. With ONE workgroup the descriptions in the same launch were slower by far -- 65536 packets walked by
//! 1024 lanes, profiles/r04 -- and all three steps in one workgroup for launches of <= 2048 packets are slower too: s43_*.)
__global__ void __launch_bounds__(1024) scanDescribe(const StreamPacket *__restrict__ pktOut, const int *__restrict__ nPkt, const unsigned nChannels, const int cap,
                                                     const int capPkt, long long *__restrict__ srcOff, int *__restrict__ nsyms, int *__restrict__ channel)
{
    __shared__ int sWave[16];
    const unsigned first = blockIdx.x * 1024u, c = first + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // (1) the channels before this workgroup's
    int before = 0;
    for (unsigned i = threadIdx.x; i < first; i += 1024u) before += nPkt[i];
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
    if (lane == 0) sWave[w] = before;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < 16; k++) base += sWave[k];
    __syncthreads();
    // (2) this workgroup's own: inclusive scan inside the wavefront, the wavefronts' totals through LDS
    const int mine = c < nChannels ? nPkt[c] : 0;
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) sWave[w] = incl;
    __syncthreads();
    for (int k = 0; k < w; k++) base += sWave[k];
    // (3) the channel's packets, in the order they were posted
    const int row0 = base + incl - mine;
    long long off = (long long)c * cap;
    for (int j = 0; j < mine; j++)
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
    (void)rowStart;                                     // (the row numbers stay in registers since the two steps are one launch)
    hipLaunchKernelGGL(scanDescribe, dim3(unsigned((nChannels + 1023) / 1024)), dim3(1024), 0, stream, pktOut, nPkt, unsigned(nChannels), cap, capPkt, srcOff,
                       nsymsOut, channelOut);
    hipLaunchKernelGGL(packCopy, dim3(unsigned(nPackets)), dim3(64), 0, stream, symOut, srcOff, nsymsOut, symsOut, stride);
    return hipGetLastError();
}

/***********************************************************************
 * The signals of a streaming launch (one StreamSignal per DOWNCHIRP1 call, LoRaDemod.cpp:267-269) into dense rows -- channel, error,
 * power, snr -- in the packets' order: channels ascending, time ascending inside a channel. Same shape as scanDescribe: the row numbers
 * are an exclusive prefix sum of nSig computed on the device, a lane per channel, no upload and no host synchronisation. The rows may be
 * device memory or pinned host memory the device writes directly (a host consumer: the block's emitSignal calls).
 **********************************************************************/
__global__ void __launch_bounds__(1024) packSignals(const StreamSignal *__restrict__ sigOut, const int *__restrict__ nSig, const unsigned nChannels, const int capPkt,
                                                    int *__restrict__ channel, int *__restrict__ error, float *__restrict__ power, float *__restrict__ snr,
                                                    const unsigned firstRow, const unsigned capRows)
{
    __shared__ int sWave[16];
    const unsigned first = blockIdx.x * 1024u, c = first + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int before = 0;
    for (unsigned i = threadIdx.x; i < first; i += 1024u) before += nSig[i];
    for (int d = 32; d >= 1; d >>= 1) before += __shfl_xor(before, d);
    if (lane == 0) sWave[w] = before;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < 16; k++) base += sWave[k];
    __syncthreads();
    const int mine = c < nChannels ? nSig[c] : 0;
    int incl = mine;
    for (int d = 1; d < 64; d <<= 1) { const int o = __shfl_up(incl, d); if (lane >= d) incl += o; }
    if (lane == 63) sWave[w] = incl;
    __syncthreads();
    for (int k = 0; k < w; k++) base += sWave[k];
    const unsigned row0 = firstRow + unsigned(base + incl - mine);
    for (int j = 0; j < mine; j++)
    {
        const unsigned r = row0 + unsigned(j);
        if (r >= capRows) break;                            // (the host checked the capacity before the launch: cannot happen)
        const StreamSignal q = sigOut[(size_t)c * capPkt + j];
        if (channel) channel[r] = (int)c;
        if (error) error[r] = q.error;
        if (power) power[r] = q.power;
        if (snr) snr[r] = q.snr;
    }
}

hipError_t launchPackSignals(const StreamSignal *sigOut, const int *nSig, const size_t nChannels, const int capPkt, int *channel, int *error, float *power,
                             float *snr, const size_t firstRow, const size_t capRows, hipStream_t stream)
{
    if (nChannels == 0) return hipSuccess;
    hipLaunchKernelGGL(packSignals, dim3(unsigned((nChannels + 1023) / 1024)), dim3(1024), 0, stream, sigOut, nSig, unsigned(nChannels), capPkt, channel, error, power,
                       snr, unsigned(firstRow), unsigned(capRows));
    return hipGetLastError();
}

/***********************************************************************
 * What the host needs after a streaming launch, reduced on the device: the per-channel counts and end-of-launch words (a few hundred KiB
 * in HBM) become 72 bytes. The per-channel arrays cross PCIe only when an accessor asks for them.
 * ONE workgroup, one launch, up to 32768 channels: five dense arrays (the four counts and StreamArgs::end, which the streaming kernels
 * write beside the counts), four channels per lane in flight, a 32-bit reduction. How it got there (16384 SF7 channels, per receiver
 * step; profiles/r05/s22_*, s29_*, s36_*): one channel per lane and round, three words of every 40-byte state: 18.4 us; loads in flight,
 * 32-bit reduction: 13.7; without the state: the figure in s36_. Measured and not kept: one workgroup per 4096 channels + a second launch
 * for their records (11.2 + 4.9 us: the kernel boundary is what makes records of workgroups on other XCCs visible, their L2s are not
 * coherent inside a kernel) -- kept only beyond 32768 channels; a single launch whose last workgroup adds up behind agent-scope fences
 * (21.7 us: a fence of that scope writes the XCC's L2 back).
 **********************************************************************/
enum { SUM_CALLS = 0, SUM_PACKETS, SUM_SYMS, SUM_SIGNALS, SUM_OPENSYMS, SUM_MORE, SUM_ANYOPEN, SUM_MAXCALL, SUM_MAXOPEN, SUM_FULLEST, SUM_FIELDS };

__device__ __forceinline__ long long summaryCombine(const int f, const long long a, const long long b)
{
    return f < SUM_MORE ? a + b : (f < SUM_MAXCALL ? (a | b) : (b > a ? b : a));      // sums; flags; maxima
}

//! the workgroup's combination of v over its lanes -> every lane of wavefront 0 (others: undefined); sL: [SUM_FIELDS][16]. Inside a
//! workgroup of at most 32768 channels every field fits 32 unsigned bits (a channel counts at most 65536 calls per launch): the shuffles
//! and the LDS traffic are those of 32-bit values, the 64-bit fields are for what several workgroups add up to.
__device__ __forceinline__ unsigned summaryCombine32(const int f, const unsigned a, const unsigned b)
{
    return f < SUM_MORE ? a + b : (f < SUM_MAXCALL ? (a | b) : (b > a ? b : a));
}

__device__ __forceinline__ void summaryReduce(long long (&v)[SUM_FIELDS], unsigned (*sL)[16])
{
    unsigned u[SUM_FIELDS];
#pragma unroll
    for (int f = 0; f < SUM_FIELDS; f++) u[f] = (unsigned)v[f];
    for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
        for (int f = 0; f < SUM_FIELDS; f++) u[f] = summaryCombine32(f, u[f], (unsigned)__shfl_xor((int)u[f], d));
    const int w = threadIdx.x >> 6, nw = int(blockDim.x >> 6);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int f = 0; f < SUM_FIELDS; f++) sL[f][w] = u[f];
    __syncthreads();
    if (w == 0)
#pragma unroll
        for (int f = 0; f < SUM_FIELDS; f++)
        {
            unsigned r = sL[f][0];
            for (int k = 1; k < nw; k++) r = summaryCombine32(f, r, sL[f][k]);
            v[f] = (long long)r;
        }
}

__device__ __forceinline__ void summaryWrite(const long long (&v)[SUM_FIELDS], const unsigned *__restrict__ near, StreamSummary *__restrict__ out)
{
    StreamSummary r;
    r.calls = v[SUM_CALLS]; r.packets = v[SUM_PACKETS]; r.syms = v[SUM_SYMS]; r.signals = v[SUM_SIGNALS]; r.openSyms = v[SUM_OPENSYMS];
    r.more = int(v[SUM_MORE]); r.anyOpen = int(v[SUM_ANYOPEN]); r.maxCallCount = int(v[SUM_MAXCALL]); r.maxOpen = int(v[SUM_MAXOPEN]);
    r.fullest = int(v[SUM_FULLEST]);
    r.nearSquelch = near[0]; r.nearStep = near[1]; r.pad = 0;
    *out = r;
}

//! partial == nullptr (one workgroup): the summary itself; else workgroup b's record to partial[b][SUM_FIELDS]
template <int W>
__global__ void __launch_bounds__(1024) streamSummary(const int2 *__restrict__ end, const int *__restrict__ nCalls, const int *__restrict__ nSym,
                                                      const int *__restrict__ nPkt, const int *__restrict__ nSig, const unsigned nChannels, const int cap,
                                                      const int capPkt, const unsigned *__restrict__ near, long long *__restrict__ partial,
                                                      StreamSummary *__restrict__ out)
{
    __shared__ unsigned sL[SUM_FIELDS][16];
    long long v[SUM_FIELDS];
#pragma unroll
    for (int f = 0; f < SUM_FIELDS; f++) v[f] = 0;
    // W channels per lane and round, every load of the round issued before the first is used (one workgroup walking 16384 channels one
    // by one is sixteen dependent trips to memory)
    const unsigned stride = gridDim.x * blockDim.x;
    for (unsigned c0 = blockIdx.x * blockDim.x + threadIdx.x; c0 < nChannels; c0 += unsigned(W) * stride)
    {
        int n[W], p[W], y[W], g[W];
        int2 e[W];
#pragma unroll
        for (int j = 0; j < W; j++)
        {
            const unsigned c = c0 + unsigned(j) * stride;
            const bool ok = c < nChannels;
            const unsigned ci = ok ? c : c0;
            n[j] = nCalls[ci]; p[j] = nPkt[ci]; y[j] = nSym[ci]; g[j] = nSig ? nSig[ci] : 0;
            e[j] = end[ci];
            if (!ok) { n[j] = p[j] = y[j] = g[j] = 0; e[j] = make_int2(-1, 0); }
        }
#pragma unroll
        for (int j = 0; j < W; j++)
        {
            v[SUM_CALLS] += n[j]; v[SUM_PACKETS] += p[j]; v[SUM_SYMS] += y[j]; v[SUM_SIGNALS] += g[j];
            v[SUM_MORE] |= (n[j] == cap || p[j] == capPkt || (nSig && g[j] == capPkt)) ? 1 : 0;      // (padding lanes: zeros; cap >= 8, capPkt >= 2)
            v[SUM_FULLEST] = n[j] > v[SUM_FULLEST] ? n[j] : v[SUM_FULLEST];
            v[SUM_MAXCALL] = e[j].y > v[SUM_MAXCALL] ? e[j].y : v[SUM_MAXCALL];
            if (e[j].x >= 0) { v[SUM_ANYOPEN] = 1; v[SUM_OPENSYMS] += e[j].x; v[SUM_MAXOPEN] = e[j].x > v[SUM_MAXOPEN] ? e[j].x : v[SUM_MAXOPEN]; }
        }
    }
    summaryReduce(v, sL);
    if (threadIdx.x != 0) return;
    if (partial == nullptr) { summaryWrite(v, near, out); return; }
#pragma unroll
    for (int f = 0; f < SUM_FIELDS; f++) partial[(size_t)blockIdx.x * SUM_FIELDS + f] = v[f];
}

//! the records of the workgroups above (a launch of its own: see the head of this section) -> the summary
__global__ void __launch_bounds__(64) streamSummaryFinal(const long long *__restrict__ partial, const unsigned groups, const unsigned *__restrict__ near,
                                                         StreamSummary *__restrict__ out)
{
    long long v[SUM_FIELDS];
#pragma unroll
    for (int f = 0; f < SUM_FIELDS; f++) v[f] = 0;
    for (unsigned k = threadIdx.x; k < groups; k += 64u)
#pragma unroll
        for (int f = 0; f < SUM_FIELDS; f++) v[f] = summaryCombine(f, v[f], partial[(size_t)k * SUM_FIELDS + f]);
    for (int d = 32; d >= 1; d >>= 1)
#pragma unroll
        for (int f = 0; f < SUM_FIELDS; f++) v[f] = summaryCombine(f, v[f], __shfl_xor(v[f], d));
    if (threadIdx.x == 0) summaryWrite(v, near, out);
}

//! bytes of the scratch block launchStreamSummary needs for nChannels channels (the workgroups' partial records)
size_t streamSummaryScratchBytes(const size_t nChannels)
{
    const size_t groups = (nChannels + 4095) / 4096;
    return groups * SUM_FIELDS * sizeof(long long);
}

hipError_t launchStreamSummary(const int2 *end, const int *nCalls, const int *nSym, const int *nPkt, const int *nSig, const size_t nChannels,
                               const int cap, const int capPkt, const unsigned *near, void *scratch, StreamSummary *out, hipStream_t stream)
{
    // one launch (one workgroup, four channels per lane and round in flight) up to 32768 channels; beyond that the records of one
    // workgroup per 4096 channels and a second launch
    const unsigned groups = (scratch && nChannels > 32768) ? unsigned((nChannels + 4095) / 4096) : 1u;
    if (groups <= 1)
    {
        hipLaunchKernelGGL(streamSummary<4>, dim3(1), dim3(1024), 0, stream, end, nCalls, nSym, nPkt, nSig, unsigned(nChannels), cap, capPkt, near,
                           static_cast<long long *>(nullptr), out);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(streamSummary<4>, dim3(groups), dim3(1024), 0, stream, end, nCalls, nSym, nPkt, nSig, unsigned(nChannels), cap, capPkt, near,
                       static_cast<long long *>(scratch), out);
    hipLaunchKernelGGL(streamSummaryFinal, dim3(1), dim3(64), 0, stream, static_cast<const long long *>(scratch), groups, near, out);
    return hipGetLastError();
}

bool streamAvailable(const int sf) { return sf >= 6 && sf <= 12; }

//! log2 of the lanes a channel gets when nobody forces a choice: the 16-points-per-lane geometry unless the launch has so few
//! channels that a wider one still fits the device's wavefront slots (two per SIMD: lorahip_stream_lanes.hip; thresholds measured,
//! profiles/r05)
static int streamLanesFor(const int sf, const unsigned nChannels, const unsigned otherWaves)
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
        if (streamLanesAvailable(sf, l) && ((unsigned long long)nChannels << l) + (unsigned long long)otherWaves * 64u <= (unsigned long long)slots * 64u) best = l;
    // Where even 32 lanes per channel leave half of the slots empty, SF7 takes two such lane groups per channel, the second one a
    // window ahead (lorahip_stream_pairs.hip): measured +8 % at 512 channels, +4 % at 1024, -1 % at 2048 (profiles/r06/s37_ahead_*). The other
    // AHEAD instances never won against the lanes instance that fills the same slots and run only when asked for.
    if (sf == 7 && best == 5 && (unsigned long long)nChannels * 128u + (unsigned long long)otherWaves * 64u <= (unsigned long long)slots * 64u && streamLanesAvailable(sf, LORAHIP_LANES_AHEAD | 5))
        best = LORAHIP_LANES_AHEAD | 5;
    return best;
}

//! log2 of the lanes per channel a launch over nChannels channels runs on (the current device's size decides where nobody forces it;
//! otherWaves: wavefronts that launches running BESIDE this one put on the device -- the other parts of a mixed object, lorahip_rx.cpp:
//! slots they take are not empty, and a wider geometry pays more slot-time per call than the 16-point one, so it is chosen only into
//! slots nobody else wants)
int streamLanesChosen(const int sf, const unsigned nChannels, const int forced, const unsigned otherWaves)
{
    const int base = sf <= 10 ? sf - 4 : sf - 4;            // SF6: 4 lanes ... SF10: 64; SF11 / SF12: a workgroup of 128 / 256
    if (sf < 7 || sf > 9) return base;
    const int lanes = forced > 0 ? forced : (forced < 0 ? base : streamLanesFor(sf, nChannels, otherWaves));
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

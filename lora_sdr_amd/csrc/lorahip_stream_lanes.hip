// The streaming demodulator with MORE LANES PER CHANNEL: instances of demodStream (lorahip_streamkernel.h) for receivers with fewer
// channels than the device holds wavefronts.
//
// A channel is a chain of work() calls (LoRaDemod.cpp:219, :278: where call k + 1 reads depends on call k), walked by the T lanes
// that own it; channels are the parallel axis. The geometries of lorahip_stream.hip give a lane 16 points of the window -- the
// fewest instructions per window, the right choice when there are channels for every wavefront slot of the device (SF7: 8 lanes
// per channel, 16384 channels = two wavefronts on every SIMD). With a quarter of those channels three quarters of the SIMDs hold
// no wavefront at all, and the time of the launch is still the time of one chain at 16 points per lane (level3_scaling of round 4:
// 1.105 / 1.115 / 1.156 ms for 2048 / 4096 / 8192 SF7 channels). Here a channel gets 2x / 4x / 8x the lanes -- 8 or 4 points per
// lane -- so the same channels occupy 2x / 4x / 8x the wavefronts and every call of a chain issues a half / a quarter of the
// per-sample instructions. The price is the FFT's shape: with 4 or 8 points a lane holds ONE radix-4 butterfly per phase, so the
// window goes through three or four phases (FastCfg::NPH = 4: a second in-place middle phase on the position-indexed exchange rows)
// instead of two or three -- more LDS exchanges per window, which is why these instances lose to the 16-point ones once the device
// is full. launchStream picks by channel count (launchStreamLanes); lorahip_demod_set_stream_lanes forces a choice (tests, A/B).
// Same operation graph, same tables: every bit of every result is the same.
#include "lorahip_streamkernel.h"
#include "lorahip_streamcfg.h"

namespace lorahip {

bool streamLanesAvailable(const int sf, const int log2Lanes)
{
    if (log2Lanes > 0 && (log2Lanes & LORAHIP_LANES_AHEAD)) return streamPairsAvailable(sf, log2Lanes & ~LORAHIP_LANES_AHEAD);
    switch (sf)
    {
    case 7: return log2Lanes == 4 || log2Lanes == 5;
    case 8: return log2Lanes == 5 || log2Lanes == 6;
    case 9: return log2Lanes == 6;
    default: return false;
    }
}

hipError_t launchStreamLanes(const int sf, const int log2Lanes, const StreamArgs &s, hipStream_t stream)
{
    if (log2Lanes & LORAHIP_LANES_AHEAD) return launchStreamPairs(sf, log2Lanes & ~LORAHIP_LANES_AHEAD, s, stream);
    switch (sf * 16 + log2Lanes)
    {
    case 7 * 16 + 4: return launchStreamCfg<Stream7L4>(s, stream);
    case 7 * 16 + 5: return launchStreamCfg<Stream7L5>(s, stream);
    case 8 * 16 + 5: return launchStreamCfg<Stream8L5>(s, stream);
    case 8 * 16 + 6: return launchStreamCfg<Stream8L6>(s, stream);
    case 9 * 16 + 6: return launchStreamCfg<Stream9L6>(s, stream);
    default: return hipErrorInvalidValue;
    }
}

} // namespace lorahip

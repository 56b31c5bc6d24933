// The streaming demodulator TWO WINDOWS AHEAD OF ONE: the AHEAD instances of demodStream (lorahip_streamkernel.h) for receivers with
// fewer channels than the device holds wavefronts.
//
// A channel is a chain of work() calls -- call k + 1 reads where call k's result says (LoRaDemod.cpp:219, :278, :320) --, so a launch
// over few channels takes the time of ONE chain whatever the device could do beside it. lorahip_stream_lanes.hip shortens the calls
// (more lanes per window); that stops paying where the part of a call that does not divide by lanes -- the reductions, the squelch
// estimate, the frame machine: ~400 of ~650 instructions at SF7 with 32 lanes -- is most of it. These instances shorten the CHAIN: a
// channel takes two lane groups of a wavefront, the second evaluates the window the next call will read if this call is a plain one
// (N samples consumed, fine-tune state as a plain call leaves it), and when the frame machine finds the channel exactly there after
// the first call it makes the second call in the same pass. Inside a packet, between frames and on the first down-chirp that is the
// rule (63 of the ~80 calls per frame of the level-3 workload); a FRAMESYNC call that moves the window (N - value, :219), adds to
// the error (:221) or parks for its sync check, and the calls around the quarter chirp, are made alone, and the window evaluated ahead
// of them is dropped: nothing of it is stored or counted. Same windows, same operands, same operation graph per call: every bit of
// every result is the same (tests/test_gpu_lanes.py holds these instances to the reference call by call, like the others).
//
// The code of an instance (lorahip_demod_set_stream_lanes): LORAHIP_LANES_AHEAD | log2 of the lanes of ONE window.
#include "lorahip_streamkernel.h"
#include "lorahip_streamcfg.h"

namespace lorahip {

bool streamPairsAvailable(const int sf, const int log2WindowLanes)
{
    switch (sf)
    {
    case 7: return log2WindowLanes >= 3 && log2WindowLanes <= 5;
    case 8: return log2WindowLanes == 4 || log2WindowLanes == 5;
    case 9: return log2WindowLanes == 5;
    default: return false;
    }
}

hipError_t launchStreamPairs(const int sf, const int log2WindowLanes, const StreamArgs &s, hipStream_t stream)
{
    switch (sf * 16 + log2WindowLanes)
    {
    case 7 * 16 + 3: return launchStreamCfg<Stream7, true>(s, stream);       // 2 x  8 lanes x 16 points
    case 7 * 16 + 4: return launchStreamCfg<Stream7L4, true>(s, stream);     // 2 x 16 lanes x  8 points
    case 7 * 16 + 5: return launchStreamCfg<Stream7L5, true>(s, stream);     // 2 x 32 lanes x  4 points
    case 8 * 16 + 4: return launchStreamCfg<Stream8, true>(s, stream);       // 2 x 16 lanes x 16 points
    case 8 * 16 + 5: return launchStreamCfg<Stream8L5, true>(s, stream);     // 2 x 32 lanes x  8 points
    case 9 * 16 + 5: return launchStreamCfg<Stream9, true>(s, stream);       // 2 x 32 lanes x 16 points
    default: return hipErrorInvalidValue;
    }
}

} // namespace lorahip

// The geometries of the streaming demodulator: 16 points per lane (lorahip_stream.hip; the resident receiver's instances in
// lorahip_resident.hip are the same ones) and more lanes per channel (lorahip_stream_lanes.hip, lorahip_stream_pairs.hip)
#pragma once
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


// The next window's samples asked for one call ahead (C::PREFETCH, lorahip_streamkernel.h): built, measured, OFF. A chain exposes the
// latency of every window's load in principle, but the index arithmetic and the chirp reads already sit between the request and its
// use, and the request for a window that is then not the one read (FRAMESYNC on noise: N - value) is paid in full: 0.742 -> 0.776 ms
// for 2048 SF7 channels at 32 lanes, 0.731 -> 0.767 ms for 1024 SF8 channels at 64 (profiles/r05/s6_lanes_prefetch_negative.txt).
#ifndef STREAM_LANES_PREFETCH
#define STREAM_LANES_PREFETCH 0
#endif

// (exchange layouts from tools/lds_conflicts_lanes.py: the model's cycles over the conflict-free count, before -> after:
//  Stream7L5 3.56 -> 1.22, Stream8L5 1.78 -> 1.22, Stream8L6 1.56 -> 1.22; Stream7L4 1.33 and Stream9L6 1.11 are its optimum already)
//             LOG2N T VEC NPH PB1 PB2 w/SIMD      X0: ROT PAD S  D   chLDS twLDS prefetch               NT     NBSEL  X1SWAP TWMID  XCD    PB3 X1PAD
typedef FastCfg<7,  4, 1,  3,  3,  5,  STREAM_WPS,     0,  1,  0, 0,  true, false, STREAM_LANES_PREFETCH>                                               Stream7L4;   // 16 lanes x 8 points: [0,3) [3,5) [5,7)
typedef FastCfg<7,  5, 2,  4,  1,  3,  STREAM_WPS,     1,  1,  0, 0,  true, false, STREAM_LANES_PREFETCH, false, false, false, false, false, 5,  2> Stream7L5;   // 32 lanes x 4 points: [0,1) [1,3) [3,5) [5,7)
typedef FastCfg<8,  5, 2,  4,  2,  4,  STREAM_WPS,     1,  1,  0, 0,  true, false, STREAM_LANES_PREFETCH, false, false, false, false, false, 6,  4> Stream8L5;   // 32 lanes x 8 points: [0,2) [2,4) [4,6) [6,8)
typedef FastCfg<8,  6, 1,  4,  2,  4,  STREAM_WPS,     0,  1,  0, 0,  true, false, STREAM_LANES_PREFETCH, false, false, false, false, false, 6,  4> Stream8L6;   // 64 lanes x 4 points
typedef FastCfg<9,  6, 1,  4,  3,  5,  STREAM_WPS,     0,  1,  0, 0,  true, false, STREAM_LANES_PREFETCH, false, false, false, false, false, 7>     Stream9L6;   // 64 lanes x 8 points: [0,3) [3,5) [5,7) [7,9)

} // namespace lorahip

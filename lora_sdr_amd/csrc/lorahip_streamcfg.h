// The 16-points-per-lane geometries of the streaming demodulator (lorahip_stream.hip; the resident receiver's instances in
// lorahip_resident.hip are the same ones)
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

} // namespace lorahip

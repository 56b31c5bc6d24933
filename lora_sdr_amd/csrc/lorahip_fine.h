// The fine-tune multiplier of a window whose index moves (LoRaDemod.cpp:157-166), without the per-sample gather.
//
// The reference multiplies sample i by _fineTuneTable[_fineTuneIndex] and then steps the index,
//     _fineTuneIndex -= _finefreqError * _fineSteps;        int -= float: (int)((float)idx - d), d = err * 128
//     if (idx < 0) idx += 128 N; else if (idx >= 128 N) idx -= 128 N;
// i.e. one data-dependent 8-byte read per sample from a 128*N-entry table (128 KiB at SF7 ... 4 MiB at SF12): on the GPU a
// 128-byte L2 line per 8 useful bytes. Two facts remove the gather while keeping every bit:
//
// (1) THE TABLE VALUE. _fineTuneTable[y] = cf32(polar(1.0, acc_y)) with acc_y the double sum of (y+1) copies of the FLOAT step
//     phase = 2 pi / (128 N) (LoRaDemod.cpp:108-114). A 24-bit step times at most 2^19 fits a double exactly, so
//     acc_y = (y+1) * phase without any rounding, and exp(j acc_y) = exp(j (yh H) phase) * exp(j (yl + 1) phase) for
//     y = yh H + yl. With the two factor tables in fp64 (A: M/H entries, B: H entries, 4-24 KiB together: they live in LDS)
//     the product -- one fp64 complex multiply -- is within ~2 ulp(double) of the true value and rounds to the table's
//     float. That is not a proof for every entry, so the HOST CHECKS ALL 128*N ENTRIES at context creation with the same
//     IEEE operations (mul, fma, convert: lorahip_tables.cpp::buildFineSplit); if a single one differed, the context would
//     carry no split tables and the kernels would gather from the table in HBM as before.
//
// (2) THE INDEX. With M = 128 N and M' = M (+1 when d > 0 is not an integer) the recurrence is, for almost every d,
//     y_{n+1} = (y_n + q) mod M'      q = M' - ceil(d) (d > 0),  q = floor(-d) (d < 0)
//     -- for d > 0 the truncation towards zero of a NEGATIVE difference gives one extra count per wrap, which is exactly
//     what the modulus M + 1 does. Hence y_n = (idx0 + n q) mod M' in closed form for a lane's own samples. The form
//     fails in two situations, both detected per window: (a) float rounding of (float)y - d reaches the next integer
//     (only when frac(d) is within M 2^-25 of an integer from the wrong side: `regular` below), (b) for d > 0 the
//     sequence lands on y = ceil(d) - 1, where the reference yields 0 instead of wrapping (closed form: the value M). For
//     0 < d < 1 that landing is the rule, not the exception -- the index walks down to 0 and stays there -- and has its own
//     closed form, y_n = max(idx0 - n, 0) (`sat`).
//     Such windows take the exact index chain (fineChainGroup / fineChainBlock); the two paths are tested against each
//     other and against the serial recurrence itself (tests/test_fine_index.py on the host, tests/test_gpu_parity.py on the device).
#pragma once
#include "lorahip_internal.h"

namespace lorahip {

//! log2 of the split H: y = (y >> LH) * H + (y & (H-1)); A has M >> LH entries, B has H = 1 << LH
__host__ __device__ constexpr int fineSplitLog2H(const int sf) { return (sf + 7) / 2; }

struct FinePlan
{
    unsigned q;         // per-sample increment of the closed form
    unsigned mod;       // M or M + 1
    int regular;        // closed form valid for every start index (unless the sequence reaches the value M)
    int sat;            // 0 < d < 1: the index walks down by one per sample and STAYS at 0 ((float)0 - d truncates to 0): y_n = max(idx0 - n, 0)
};

//! classify one window's step d = _finefreqError * _fineSteps; M = 128 N = 2^m, m <= 19
__host__ __device__ inline FinePlan finePlan(const float d, const int M)
{
    FinePlan p;
    p.q = 0; p.mod = unsigned(M); p.regular = 1; p.sat = 0;
    if (d == 0.0f) return p;
    const float a = d < 0.0f ? -d : d;
    if (!(a < float(M / 2))) { p.regular = 0; return p; }            // huge or NaN: the serial chain decides
    const float af = float(int(a));                                  // floor, exact
    const float fr = a - af;                                         // exact
    if (d > 0.0f)
    {
        if (fr == 0.0f) { p.q = unsigned(M) - unsigned(af); }
        else
        {
            p.mod = unsigned(M) + 1u;
            p.q = unsigned(M) - unsigned(af);                        // M' - ceil(d)
            p.regular = fr > float(M) * 0x1p-25f;                    // (float)y - d never rounds up to the next integer
            p.sat = p.regular && af == 0.0f;                         // ceil(d) = 1: the modular form reaches the value M where the reference sticks at 0
        }
    }
    else
    {
        p.q = unsigned(af);
        p.regular = fr == 0.0f || (1.0f - fr) > float(M) * 0x1p-24f; // (float)y + |d| < 2M never rounds up to the next integer
    }
    return p;
}

//! x mod p.mod for x < 2^32 with x / M <= M (x = idx0 + n q, n <= N)
__host__ __device__ inline unsigned fineReduce(const unsigned x, const FinePlan &p, const int log2M)
{
    const unsigned M = 1u << log2M;
    const unsigned lo = x & (M - 1u), hi = x >> log2M;
    // M' = M: lo.  M' = M + 1: 2^m = -1 (mod M'), so x = lo - hi (mod M')
    int r = int(lo) - int(p.mod != M ? hi : 0u);
    r += (r >> 31) & int(p.mod);
    return unsigned(r);
}

//! y + Q mod p.mod for y, Q < p.mod
__host__ __device__ inline unsigned fineAdvance(const unsigned y, const unsigned Q, const FinePlan &p)
{
    const unsigned s = y + Q, w = s - p.mod;
    return s < w ? s : w;                                            // unsigned min: w wraps to a huge value when s < mod
}

//! the index after the window's N steps; *hitEnd unused by callers that scan the samples themselves
__host__ __device__ inline int fineEndIndex(const int idx0, const FinePlan &p, const int log2N, const int log2M)
{
    if (p.sat) return idx0 > (1 << log2N) ? idx0 - (1 << log2N) : 0;
    const unsigned y = fineReduce(unsigned(idx0) + (p.q << log2N), p, log2M);      // N q < 2^31
    return y == (1u << log2M) ? 0 : int(y);                                         // landing on M at the very end: the reference holds 0
}


} // namespace lorahip

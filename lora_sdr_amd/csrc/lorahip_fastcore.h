// The per-wavefront FFT core shared by the batch kernel (lorahip_fast.hip) and the streaming demod kernel
// (lorahip_stream.hip): configuration, register twiddles, phase 0 -> exchange(s) -> last phase, the |X|^2
// scan with its reductions, and the exact parallel evaluation of the fine-tune index recurrence.
#pragma once
#include "lorahip_fft.h"

namespace lorahip {

/***********************************************************************
 * compile-time configuration of one kernel instance
 *   X0ROT/X0PAD/X0S/X0D: exchange-0 LDS layout (found with tools/lds_conflicts.py): a row per low sample
 *   index n_low = VEC*t+u at element offset rotr(n_low, X0ROT)*(WPW*R+X0PAD) + ((n_low>>X0S)&1)*X0D,
 *   the WPW windows of a wave side by side inside the row (stride R), so that the writers' ds_write_b64
 *   groups and the readers' ds_read_b64 groups each tile the LDS banks exactly once.
 **********************************************************************/
template <int LOG2N_, int LOG2T_, int VEC_, int NPH_, int PB1_, int PB2_, int WAVES_PER_SIMD_,
          int X0ROT_, int X0PAD_, int X0S_, int X0D_, bool CH_LDS_, bool TW_ALL_LDS_, int PREFETCH_, bool NT_ = false, bool NB_SELECT_ = false, bool X1_SWAP_ = false, bool TW_MID_REG_ = false, bool XCD_CONTIG_ = false,
          int PB3_ = 0, int X1PAD_ = 8>
struct FastCfg
{
    static constexpr int PREFETCH = PREFETCH_;        // next window set's loads: 0 none (loaded at the top), 1 issued after the dechirp of this set, 2 at the top of this set
    static constexpr bool XCD_CONTIG = XCD_CONTIG_;   // workgroups of one XCD (blockIdx mod 8) walk one contiguous eighth of the batch (else: sets interleaved over all workgroups)
    static constexpr bool TW_MID_REG = TW_MID_REG_;   // middle-phase twiddles (they depend on the lane only) in registers instead of LDS reads per window
    static constexpr bool X1_SWAP = X1_SWAP_;         // exchange 1 as a 4x4 transpose between the wave's 16-lane rows and registers (v_permlane16/32_swap), no LDS
    static constexpr bool NB_SELECT = NB_SELECT_;     // peak's neighbours by register select + lane shuffle instead of staging all bins in LDS
    static constexpr bool NT = NT_;                   // non-temporal hint on the IQ loads (read once, never reused)
    static constexpr bool CH_LDS = CH_LDS_;           // chirp table read from LDS per window (else loop-invariant registers)
    static constexpr bool TW_ALL_LDS = TW_ALL_LDS_;   // last-phase twiddles from the LDS table too (else registers)
    static constexpr int LOG2N = LOG2N_, N = 1 << LOG2N_;
    static constexpr int LOG2T = LOG2T_, T = 1 << LOG2T_;     // lanes per window (<= 64)
    static constexpr int VEC = VEC_;                            // consecutive samples per lane per load
    static constexpr int P = N / T;                             // points per lane
    static constexpr int R = P / VEC;                           // phase-0 group size
    static constexpr int NPH = NPH_;
    static constexpr int WPW = 64 / T;                          // windows per wave iteration
    static constexpr int WAVES_PER_SIMD = WAVES_PER_SIMD_;
    static constexpr bool HAS_R2 = (LOG2N_ & 1);
    static constexpr int NL = VEC * T;                          // distinct n_low
    static constexpr int LOG2NL = LOG2N_ - PB1_;
    //! first stage bit of phase j (NPH_ = 2, 3 or 4 phases: bits [0, PB1), [PB1, PB2), [PB2, PB3), [.., LOG2N))
    __host__ __device__ static constexpr int bound(const int j)
    {
        return j <= 0 ? 0 : (j >= NPH_ ? LOG2N_ : (j == 1 ? PB1_ : (j == 2 ? PB2_ : PB3_)));
    }
    static_assert(NPH_ >= 2 && NPH_ <= 4, "two to four phases");
    static_assert(NPH_ < 4 || (PB3_ > PB2_ && PB3_ < LOG2N_ && ((PB3_ - PB2_) & 1) == 0), "a fourth phase needs its own boundary, whole radix-4 digits");
    static_assert((1 << PB1_) == R, "phase 0 must cover exactly the bits a lane loads");
    static_assert(((LOG2N_ - PB1_) & 1) == 0, "the low sample digits must be whole radix-4 digits");
    static_assert(T <= 64 && T >= 4, "a window lives inside one wavefront");
    // exchange 0
    static constexpr int RS0 = WPW * R + X0PAD_;
    __host__ __device__ static constexpr int x0off(const int nlow)
    {
        const int rot = ((nlow >> X0ROT_) | (nlow << (LOG2NL - X0ROT_))) & (NL - 1);
        return rot * RS0 + ((nlow >> X0S_) & 1) * X0D_;
    }
    static constexpr int X0ELEMS = NL * RS0 + X0D_;             // per wave
    // exchange 1 (3 phases): per window, element (rl, rh, col) at rh*X1 + col*R + rl
    static constexpr int G1 = 1 << (bound(2) - bound(1));
    static constexpr int X1 = G1 * R + X1PAD_;                  // row pad of the position-indexed exchange (tools/lds_conflicts_lanes.py)
    static constexpr int X1ROWS = N / (G1 * R);                 // rows of 2^PB2 positions (+8 pad) per window
    static constexpr int X1ELEMS = NPH_ >= 3 ? WPW * X1ROWS * X1 : 0;   // per wave
    static constexpr int XELEMS = (X0ELEMS > X1ELEMS ? X0ELEMS : X1ELEMS) + (N > X0ELEMS ? N - X0ELEMS : 0) / 2 * 0;
    //! twiddle entries staged in LDS: all stages below the last phase, or every stage
    static constexpr int TW_LDS = twStageOffset(LOG2N_, TW_ALL_LDS_ ? LOG2N_ : bound(NPH_ - 1));
    static constexpr int CH_ELEMS = CH_LDS_ ? N : 0;
    // derived geometry of the last phase and of the per-wave LDS region
    static constexpr int BL = bound(NPH_ - 1);                  // first bit of the last phase
    static constexpr int GL = 1 << (LOG2N_ - BL);               // last-phase group size
    static constexpr int NGL = P / GL;                          // last-phase groups per lane
    static_assert(GL <= P && (NPH_ < 3 || G1 <= P), "a phase's group must fit a lane's points");
    static constexpr int SLOTS = lastPhaseSlots<LOG2N_, BL, LOG2N_>();
    static constexpr int XE = (X0ELEMS > X1ELEMS ? X0ELEMS : X1ELEMS);
    static constexpr int FS = N + 8;                            // final-bin rows of the wave's windows (8 pad: 16-lane write groups tile the banks)
    static constexpr int XW = (XE > WPW * FS ? XE : WPW * FS) + 2;   // v2f per wave; also holds WPW*N ints
    static constexpr int TWN = (TW_LDS + 1) & ~1;
    static constexpr int MSLOTS = NPH_ == 3 ? lastPhaseSlots<LOG2N_, PB1_, PB2_>() : 1;   // register twiddles of the middle phase
};


template <class C>
struct FastCore
{
    static constexpr int N = C::N, T = C::T, VEC = C::VEC, P = C::P, R = C::R, NPH = C::NPH, WPW = C::WPW;
    static constexpr int LOG2N = C::LOG2N, LOG2T = C::LOG2T;
    static constexpr int B1 = C::bound(1), B2 = C::bound(2), BL = C::BL, GL = C::GL, NGL = C::NGL, SLOTS = C::SLOTS;
    static constexpr int M = N * LORAHIP_FINE_STEPS;
    typedef v2f TwR[C::TW_ALL_LDS ? 1 : NGL][C::TW_ALL_LDS ? 1 : SLOTS];

    //! register twiddles of the last phase: there klow = ci = t + T*g
    static __device__ __forceinline__ void loadTwR(TwR &twR, const v2f *__restrict__ twStage, const int t)
    {
#pragma unroll
        for (int g = 0; g < (C::TW_ALL_LDS ? 0 : NGL); g++)
        {
            const int ci = t + T * g;
            int slot = 0;
#pragma unroll
            for (int b = BL; b < LOG2N; b += 2)
#pragma unroll
                for (int kl = 0; kl < (1 << (b - BL)); kl++)
                {
                    const int k = ci + (kl << BL);
                    const int base = twStageOffset(LOG2N, b) + k;
                    twR[g][slot] = twStage[base];
                    twR[g][slot + 1] = twStage[base + (1 << b)];
                    twR[g][slot + 2] = twStage[base + (2 << b)];
                    slot += 3;
                }
        }
    }

    typedef v2f TwM[(C::TW_MID_REG && NPH == 3) ? P / C::G1 : 1][(C::TW_MID_REG && NPH == 3) ? C::MSLOTS : 1];
    //! register twiddles of the middle phase: klow = (t + T*g) mod R
    static __device__ __forceinline__ void loadTwM(TwM &twM, const v2f *__restrict__ twStage, const int t)
    {
        if (!(C::TW_MID_REG && NPH == 3)) return;
#pragma unroll
        for (int g = 0; g < ((C::TW_MID_REG && NPH == 3) ? P / C::G1 : 0); g++)
        {
            const int klow = (t + T * g) & (R - 1);
            int slot = 0;
#pragma unroll
            for (int b = B1; b < B2; b += 2)
#pragma unroll
                for (int kl = 0; kl < (1 << (b - B1)); kl++)
                {
                    const int base = twStageOffset(LOG2N, b) + klow + (kl << B1);
                    twM[g][slot] = twStage[base];
                    twM[g][slot + 1] = twStage[base + (1 << b)];
                    twM[g][slot + 2] = twStage[base + (2 << b)];
                    slot += 3;
                }
        }
    }

    //! coalesced window load: VEC*8 bytes per lane, the T lanes of a window contiguous, R rows
    static __device__ __forceinline__ void load(v2f (&xn)[R][VEC], const v2f *__restrict__ in_, const int t)
    {
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            const v2f *p = in_ + VEC * t + VEC * T * r;
            if (VEC >= 2)
            {
#pragma unroll
                for (int h = 0; h < VEC / 2; h++)                  // 16 bytes = two samples per load instruction
                {
                    const v4f q = C::NT ? __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p) + h) : reinterpret_cast<const v4f *>(p)[h];
                    xn[r][(2 * h) % VEC] = MAKE2(q.x, q.y);
                    xn[r][(2 * h + 1) % VEC] = MAKE2(q.z, q.w);
                }
            }
            else xn[r][0] = C::NT ? __builtin_nontemporal_load(p) : *p;
        }
    }

    //! chirp values of this lane's sample positions from the LDS copy of the table
    static __device__ __forceinline__ void chirpFromLds(v2f (&cw)[R][VEC], const v2f *sCh, const int t)
    {
#pragma unroll
        for (int r = 0; r < R; r++)
        {
            const v2f *p = sCh + VEC * t + VEC * T * r;
            if (VEC >= 2)
            {
#pragma unroll
                for (int h = 0; h < VEC / 2; h++)
                {
                    const v4f q = reinterpret_cast<const v4f *>(p)[h];
                    cw[r][(2 * h) % VEC] = MAKE2(q.x, q.y);
                    cw[r][(2 * h + 1) % VEC] = MAKE2(q.z, q.w);
                }
            }
            else cw[r][0] = *p;
        }
    }

    //! dechirped samples x -> FFT bins vl (lane t holds bins (t + T*g) + 2^BL * e). X = the wave's exchange
    //! region. `mid` is called once phase 0's inputs are staged (the batch kernel issues its prefetch there).
    template <class MID>
    static __device__ __forceinline__ void fft(const v2f (&x)[R][VEC], v2f *X, const int wsub, const int t,
                                               const v2f *sTw, const TwR &twR, v2f (&vl)[NGL][GL], MID mid, const TwM *twMp = nullptr)
    {
        // ---- phase 0: bits [0, B1) in registers, one group per u -----------------------------
        // register r holds sample index high part a = r; its work-array position low bits are rev(a)
        v2f v0[VEC][R];
#pragma unroll
        for (int r = 0; r < R; r++)
#pragma unroll
            for (int u = 0; u < VEC; u++) v0[u][Plan<LOG2N>::pos(VEC * T * r) & (R - 1)] = x[r][u];
        mid();
#pragma unroll
        for (int u = 0; u < VEC; u++) runPhase<LOG2N, 0, B1, false>(v0[u], 0, sTw, nullptr);

        // ---- exchange 0: one row per n_low = VEC*t+u, the wave's windows side by side ----------
        {
            v2f *Xw = X + wsub * R;
#pragma unroll
            for (int u = 0; u < VEC; u++)
            {
                v2f *row = Xw + C::x0off(VEC * t + u);
#pragma unroll
                for (int e = 0; e < R; e++) row[e] = v0[u][e];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();

        if (NPH == 2)
        {
            // phase 1 = last: group g has klow = ci = t + T*g (< R), element e <-> hp = e, n_low = rev4(hp)
            const v2f *Xr = X + wsub * R;
#pragma unroll
            for (int g = 0; g < NGL; g++)
#pragma unroll
                for (int e = 0; e < GL; e++) vl[g][e] = Xr[C::x0off(rev4(e, LOG2N - B1)) + (t + T * g)];
        }
        else
        {
            // phase 1 (middle): bits [B1, B2); ci = t + T*g; klow = ci mod R; high = ci >> B1
            constexpr int G1 = C::G1;
            constexpr int NG1 = P / G1;
            constexpr int HB = LOG2N - B2;                 // bits of `high`
            v2f v1[NG1][G1];
            const v2f *Xr = X + wsub * R;
#pragma unroll
            for (int g = 0; g < NG1; g++)
            {
                const int ci = t + T * g;
                const int klow = ci & (R - 1), high = ci >> B1;
                const int rhigh = rev4(high, HB);
                // hp = e + G1*high; n_low = rev4(hp) = rev4(e) << HB | rev4(high)
#pragma unroll
                for (int e = 0; e < G1; e++) v1[g][e] = Xr[C::x0off((rev4(e, B2 - B1) << HB) | rhigh) + klow];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int g = 0; g < NG1; g++)
            {
                if (C::TW_MID_REG) runPhase<LOG2N, B1, B2, true>(v1[g], 0, nullptr, (*twMp)[C::TW_MID_REG ? g : 0]);
                else runPhase<LOG2N, B1, B2, false>(v1[g], (t + T * g) & (R - 1), sTw, nullptr);
            }
            if (C::X1_SWAP)
            {
                // T = 64, one group of 16 per lane: lane (klow = t & 15, row = t >> 4) holds positions klow + 16*e + 256*row and
                // must end up with positions (t + 64*g) + 256*e2, i.e. vl[g][e2] at row r = v1[r + 4g] of row e2: for every g a
                // 4x4 transpose between the row number and the low two bits of the register number -- two butterfly stages of
                // row swaps (rows r <-> r^1, then r <-> r^2), one instruction per register pair and component.
                static_assert(!C::X1_SWAP || ((T == 64 || T == 32) && NG1 == 1 && G1 == 16 && GL == 4 && NGL == 4), "row-swap exchange: SF9 / SF10 shape only");
#pragma unroll
                for (int b = 0; b < 4; b++)
                {
                    unsigned m[4][2];
#pragma unroll
                    for (int a = 0; a < 4; a++)
                    {
                        m[a][0] = __float_as_uint(v1[0][a + 4 * b].x);
                        m[a][1] = __float_as_uint(v1[0][a + 4 * b].y);
                    }
#pragma unroll
                    for (int c = 0; c < 2; c++)
                    {
                        v2u r;
                        if (T == 64)
                        {
                            // row number = lane bits (4,5)
                            r = __builtin_amdgcn_permlane16_swap(m[0][c], m[1][c], false, false); m[0][c] = r.x; m[1][c] = r.y;
                            r = __builtin_amdgcn_permlane16_swap(m[2][c], m[3][c], false, false); m[2][c] = r.x; m[3][c] = r.y;
                            r = __builtin_amdgcn_permlane32_swap(m[0][c], m[2][c], false, false); m[0][c] = r.x; m[2][c] = r.y;
                            r = __builtin_amdgcn_permlane32_swap(m[1][c], m[3][c], false, false); m[1][c] = r.x; m[3][c] = r.y;
                        }
                        else
                        {
                            // T = 32 (two windows per wave): row number = lane bits (3,4). Bit 3 lives inside a 16-lane row: DPP row_ror:8
                            // brings lane l^8, the bank mask keeps the half that stays (two movs per register pair)
#pragma unroll
                            for (int p = 0; p < 4; p += 2)
                            {
                                const unsigned A = m[p][c], B = m[p + 1][c];
                                m[p][c] = __builtin_amdgcn_update_dpp(A, B, 0x128, 0xf, 0xC, false);       // lanes 8..15 of a row take B[l^8]
                                m[p + 1][c] = __builtin_amdgcn_update_dpp(B, A, 0x128, 0xf, 0x3, false);   // lanes 0..7 take A[l^8]
                            }
                            r = __builtin_amdgcn_permlane16_swap(m[0][c], m[2][c], false, false); m[0][c] = r.x; m[2][c] = r.y;
                            r = __builtin_amdgcn_permlane16_swap(m[1][c], m[3][c], false, false); m[1][c] = r.x; m[3][c] = r.y;
                        }
                    }
#pragma unroll
                    for (int a = 0; a < 4; a++) vl[b][a] = MAKE2(__uint_as_float(m[a][0]), __uint_as_float(m[a][1]));
                }
            }
            else
            {
            // exchange 1: the window by POSITION, rows of 2^B2 positions: element (rl, rh, col) at rh*X1 + col*R + rl
            v2f *X1w = X + wsub * (C::X1ROWS * C::X1);
#pragma unroll
            for (int g = 0; g < NG1; g++)
            {
                const int ci = t + T * g;
                v2f *base = X1w + (ci >> B1) * C::X1 + (ci & (R - 1));
#pragma unroll
                for (int e = 0; e < G1; e++) base[e * R] = v1[g][e];
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if constexpr (NPH == 4)
            {
                // phase 2 (second middle phase): bits [B2, B3), in place in the position-indexed rows -- a lane writes back exactly the
                // elements it read. ci = t + T*g: klow = ci mod 2^B2 (the column), high = ci >> B2; element e at row e + G2*high
                constexpr int B3 = C::bound(3);
                constexpr int G2 = 1 << (B3 - B2), NG2 = P / G2;
                static_assert(G2 <= P, "a phase's group must fit a lane's points");
                v2f v2[NG2][G2];
#pragma unroll
                for (int g = 0; g < NG2; g++)
                {
                    const int ci = t + T * g;
                    const v2f *col = X1w + (ci >> B2) * (G2 * C::X1) + (ci & ((1 << B2) - 1));
#pragma unroll
                    for (int e = 0; e < G2; e++) v2[g][e] = col[e * C::X1];
                }
#pragma unroll
                for (int g = 0; g < NG2; g++) runPhase<LOG2N, B2, B3, false>(v2[g], (t + T * g) & ((1 << B2) - 1), sTw, nullptr);
#pragma unroll
                for (int g = 0; g < NG2; g++)
                {
                    const int ci = t + T * g;
                    v2f *col = X1w + (ci >> B2) * (G2 * C::X1) + (ci & ((1 << B2) - 1));
#pragma unroll
                    for (int e = 0; e < G2; e++) col[e * C::X1] = v2[g][e];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            // last phase: position ci + (e << BL), ci = t + T*g < 2^BL: row = position >> B2 (three phases: BL = B2, the row is e)
#pragma unroll
            for (int g = 0; g < NGL; g++)
            {
                const int ci = t + T * g;
#pragma unroll
                for (int e = 0; e < GL; e++) vl[g][e] = X1w[((ci >> B2) + (e << (BL - B2))) * C::X1 + (ci & ((1 << B2) - 1))];
            }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int g = 0; g < NGL; g++)
        {
            if (C::TW_ALL_LDS) runPhase<LOG2N, BL, LOG2N, false>(vl[g], t + T * g, sTw, nullptr);
            else runPhase<LOG2N, BL, LOG2N, true>(vl[g], 0, nullptr, twR[C::TW_ALL_LDS ? 0 : g]);
        }
    }

    //! LoRaDetector.hpp:36-48 over the window's bins: writes them to F (the window's row of the now free exchange
    //! region, for the neighbour fetch) and to fftOut (optional debug port), and leaves the window's arg-max,
    //! its |X|^2 and the fp64 total in every lane of the window.
    template <bool STORE_F = true, int CHAINS = 1>
    static __device__ __forceinline__ void scan(const v2f (&vl)[NGL][GL], v2f *F, v2f *fftOut, const int t,
                                                float &bestV, int &bestI, double &tot)
    {
        if (!C::NB_SELECT && STORE_F)
        {
#pragma unroll
            for (int e = 0; e < GL; e++)
#pragma unroll
                for (int g = 0; g < NGL; g++) F[(t + T * g) + (e << BL)] = vl[g][e];
        }
        if (fftOut)
        {
#pragma unroll
            for (int e = 0; e < GL; e++)
#pragma unroll
                for (int g = 0; g < NGL; g++) fftOut[(t + T * g) + (e << BL)] = vl[g][e];
        }
        // element j = e*NGL + g of the lane is bin (t + T g) + (e << BL): ascending in j
        const int bestJ = laneScan<GL * NGL, CHAINS>([&](const int j) { return vl[j % NGL][j / NGL]; }, bestV, tot);
        bestI = (t + T * (bestJ & (NGL - 1))) + ((bestJ / NGL) << BL);
        if (!(bestV > 0.0f)) bestI = 0;
        groupArgmax<T>(bestV, bestI);
        tot = groupSumF64<T>(tot);
        // every lane of the window now holds the window's (bestV, bestI); the tree adds the same fp64 partials in the
        // same pairing on all lanes, so tot is identical on all of them too
    }

    //! scan() for a caller that only needs the window's arg-max and a QUICK total (fp32: the streaming kernels' squelch estimate,
    //! squelchQuickF; anything that leaves the kernel comes from scan()). Same winner as scan(): the same strict comparisons.
    static constexpr float QUICK_REL_ERR = float(GL * NGL + LOG2T + 2) * 0x1p-24f;
    template <bool STORE_F>
    static __device__ __forceinline__ void scanQuick(const v2f (&vl)[NGL][GL], v2f *F, const int t, float &bestV, int &bestI, float &totF)
    {
        if (!C::NB_SELECT && STORE_F)
        {
#pragma unroll
            for (int e = 0; e < GL; e++)
#pragma unroll
                for (int g = 0; g < NGL; g++) F[(t + T * g) + (e << BL)] = vl[g][e];
        }
        const int bestJ = laneScanQuick<GL * NGL>([&](const int j) { return vl[j % NGL][j / NGL]; }, bestV, totF);
        bestI = (t + T * (bestJ & (NGL - 1))) + ((bestJ / NGL) << BL);
        if (!(bestV > 0.0f)) bestI = 0;
        groupArgmax<T>(bestV, bestI);
        totF = groupSumF32<T>(totF);
    }

    //! bins k-1 and k+1 of the window's peak k (LoRaDetector.hpp:56-57), valid in every lane of the window
    template <bool FROM_REGS = C::NB_SELECT>
    static __device__ __forceinline__ void neighbours(const v2f (&vl)[NGL][GL], const v2f *F, const int bestI, const int lane, const int t,
                                                      v2f &leftBin, v2f &rightBin)
    {
        const int bl = (bestI + N - 1) & (N - 1), br = (bestI + 1) & (N - 1);
        if (FROM_REGS)
        {
            const int cil = bl & ((1 << BL) - 1), cir = br & ((1 << BL) - 1);
            const bool ownL = (cil & (T - 1)) == t;
            const int req = ownL ? ((bl >> BL) * NGL + (cil >> LOG2T)) : ((br >> BL) * NGL + (cir >> LOG2T));
            const v2f mine = selectReg<NGL, GL>(vl, req);
            const int base = lane & ~(T - 1);
            leftBin = MAKE2(__shfl(mine.x, base + (cil & (T - 1)), 64), __shfl(mine.y, base + (cil & (T - 1)), 64));
            rightBin = MAKE2(__shfl(mine.x, base + (cir & (T - 1)), 64), __shfl(mine.y, base + (cir & (T - 1)), 64));
        }
        else
        {
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
            leftBin = F[bl];
            rightBin = F[br];
        }
    }

    //! position of sample n's fine-tune index inside the window's sIdx array (lane-major transposed so that
    //! the chain writers are conflict-free)
    static __device__ __forceinline__ int idxSlot(const int n) { return (n & (P - 1)) * T + (n >> (LOG2N - LOG2T)); }

    //! the window's fine-tune index chain (fineChainGroup, lorahip_fft.h): idx of sample n at sIdx[idxSlot(n)]
    static __device__ __forceinline__ int fineChain(const int idx0, const float d, const int t, int *sIdx)
    {
        return fineChainGroup<T, P, M>(idx0, d, t, sIdx);
    }
};

} // namespace lorahip

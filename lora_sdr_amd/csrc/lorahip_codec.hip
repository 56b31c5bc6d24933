// Batched LoRaDecoder: symbol packets -> bytes.
//
// What the LoRaDecoder block does for one message (LoRaDecoder.cpp:196-397 on LoRaCodes.hpp): Gray-code the demodulated
// symbols with rounding to the symbol size, de-interleave diagonally into codewords (the first block always 4/8), strip the
// whitening with the two interleaved LFSRs, Hamming / parity decode, parse and check the explicit header, check the
// payload CRC. Integer and bit work on a few hundred bytes per packet.
//
// decodeGroup (the default): a GROUP of G = 8 / 16 / 32 / 64 lanes of a wavefront owns a packet (G by the row length of the
// launch), the packet's symbols, codewords and nibbles live in the group's slice of LDS, and every stage is data-parallel over
// the lanes of the group: a lane Gray-codes its symbols, builds its codewords bit by bit from the (at most eight) symbols of
// their interleaver block, whitens them from a precomputed table of the two LFSR sequences (they do not depend on the packet),
// decodes its nibbles, assembles its bytes; the CRC is a sum over the lanes of byte * x^(8 k) mod P (a 256-entry table), folded
// with lane shuffles. No per-lane arrays, no scratch. The order of the reference's decisions (what is decoded before which
// check, which coding rate applies where) is kept exactly; only the loops became lanes.
//
// decodePackets (context variant 1): one lane walks one packet statement by statement in the reference's order, working arrays
// in per-lane scratch -- the round-1 kernel, kept as the A/B checker.
#include "lorahip_internal.h"

namespace lorahip {

#define LORAHIP_DEC_MAX_SYMBOLS 520            // symbols per packet incl. rounding to a block: the one-lane checker kernel (variant 1) only
#define LORAHIP_DEC_MAX_DATA_LENGTH 4096       // setDataLength without a header: bytes per packet the tables below reach (a LoRa length field is one byte)
#define LORAHIP_DEC_MAX_CODEWORDS ((LORAHIP_DEC_MAX_SYMBOLS / 4) * 12 + 4)
#define LORAHIP_HDR_RDD 4                      // LoRaCodes.hpp:103
#define LORAHIP_N_HDR_SYMBOLS 8                // :104
#define LORAHIP_N_HDR_CODEWORDS 5              // :105

namespace {

__device__ __forceinline__ int parityOf(const unsigned v) { return __popc(v) & 1; }

//! header checksum (LoRaCodes.hpp:131-156): five parity bits over the 12 header bits w = h[0] | (h[1] & 0xf) << 8, as masks
__device__ __forceinline__ unsigned char headerChecksum(const unsigned char *h)
{
    const unsigned w = h[0] | ((unsigned)(h[1] & 0xf) << 8);
    return (unsigned char)(parityOf(w & 0xF12) | (parityOf(w & 0x725) << 1) | (parityOf(w & 0xA49) << 2) | (parityOf(w & 0x18E) << 3) |
                           (parityOf(w & 0x0F0) << 4));
}

//! one byte position of the CCITT crc, MSB first (:158-168)
__device__ __forceinline__ unsigned short crc16sx(unsigned short crc, const unsigned short poly)
{
    for (int bit = 0; bit < 8; bit++)
    {
        const bool top = (crc & 0x8000) != 0;
        crc = (unsigned short)(crc << 1);
        if (top) crc ^= poly;
    }
    return crc;
}

//! payload checksum of the sx1272 (:170-194): crc over the bytes, masked with two steps of an 8-bit LFSR (taps 0xB8)
__device__ unsigned short dataChecksum(const unsigned char *data, const int length)
{
    unsigned short res = 0;
    unsigned char v = 0xff;
    for (int i = 0; i < length; i++)
    {
        const unsigned short crc = crc16sx(res, 0x1021);
        v = (unsigned char)((v << 1) | parityOf(v & 0xB8));
        res = crc ^ data[i];
    }
    res ^= v;
    v = (unsigned char)((v << 1) | parityOf(v & 0xB8));
    res ^= (unsigned short)(v << 8);
    return res;
}

//! one byte step of a whitening register: feedback byte b0^b2^b3^b4, shifted in at the top (:255-268)
__device__ __forceinline__ unsigned long long lfsrAdvance(const unsigned long long r)
{
    const unsigned long long fb = (r ^ (r >> 16) ^ (r >> 24) ^ (r >> 32)) & 0xff;
    return (r >> 8) | (fb << 56);
}

//! de-whitening with the two interleaved registers: codeword position p (counted from bitOfs) uses register p mod 2.
//! bufferSize is a uint16_t parameter in the reference.
__device__ void whiteningLfsr(unsigned char *buffer, const unsigned short bufferSize, const int bitOfs, const int RDD)
{
    unsigned long long even = (RDD == 1) ? 0x05121100F8ECFEEFull : 0x6572D100E85C2EFFull;   // single-parity mode has its own seeds
    unsigned long long odd = (RDD == 1) ? 0xF8ECFEEFEFEFEFEFull : 0xE85C2EFFFFFFFFFFull;
    const unsigned char keep = (unsigned char)(0xff >> (4 - RDD));
    for (int p = 0; p < bitOfs; p++)
    {
        if (p & 1) odd = lfsrAdvance(odd);
        else even = lfsrAdvance(even);
    }
    for (int j = 0; j < bufferSize; j++)
    {
        if ((bitOfs + j) & 1) { buffer[j] ^= (unsigned char)(odd & keep); odd = lfsrAdvance(odd); }
        else { buffer[j] ^= (unsigned char)(even & keep); even = lfsrAdvance(even); }
    }
}

// The sx Hamming codes (:222-259, :284-312) and parity checks (:318-323, :335-343) as syndrome masks: parity bit k covers
// the codeword bits in COVERk; a syndrome naming a data bit flips it, one naming a parity bit is ignored, anything else is
// uncorrectable (8,4 only). The flip tables are packed 4 bits per syndrome value (0xF = uncorrectable).
#define LORAHIP_COVER0 0x17u
#define LORAHIP_COVER1 0x2Eu
#define LORAHIP_COVER2 0x4Bu
#define LORAHIP_COVER3 0x8Du

__device__ __forceinline__ unsigned char decodeHamming84(const unsigned char b, bool &error, bool &bad)
{
    const unsigned syn = parityOf(b & LORAHIP_COVER0) | (parityOf(b & LORAHIP_COVER1) << 1) | (parityOf(b & LORAHIP_COVER2) << 2) |
                         (parityOf(b & LORAHIP_COVER3) << 3);
    // syndrome 0..15 -> data bit to flip: 0xD:1 0x7:2 0xB:4 0xE:8; 0,1,2,4,8 -> none; the rest uncorrectable
    const unsigned long long fix = 0xF81F4FF02FF0F000ull;
    const unsigned f = (unsigned)(fix >> (4 * syn)) & 0xf;
    if (syn) error = true;
    if (f == 0xf) { bad = true; return b & 0xf; }
    return (b ^ f) & 0xf;
}

__device__ __forceinline__ unsigned char decodeHamming74(const unsigned char b, bool &error)
{
    const unsigned syn = parityOf(b & LORAHIP_COVER0 & 0x7f) | (parityOf(b & LORAHIP_COVER1 & 0x7f) << 1) | (parityOf(b & LORAHIP_COVER2 & 0x7f) << 2);
    const unsigned fix = 0x28104000u;                  // syndrome 0..7 -> flip: 5:1 7:2 3:4 6:8
    if (syn) error = true;
    return (b ^ ((fix >> (4 * syn)) & 0xf)) & 0xf;
}

__device__ __forceinline__ unsigned char checkParity54(const unsigned char b, bool &error)
{
    if (parityOf(b & 0x1F)) error = true;              // one parity bit (b4) over the data bits
    return b & 0xf;
}

__device__ __forceinline__ unsigned char checkParity64(const unsigned char b, bool &error)
{
    if (parityOf(b & LORAHIP_COVER0) | parityOf(b & LORAHIP_COVER1)) error = true;   // the first two Hamming parity bits
    return b & 0xf;
}

//! one FEC nibble of the payload at coding rate rdd
__device__ __forceinline__ unsigned char decodeNibble(const unsigned char cw, const int rdd, bool &error, bool &bad)
{
    switch (rdd)
    {
    case 0: return cw;                            // the callers mask / shift exactly like the reference
    case 1: return checkParity54(cw, error);
    case 2: return checkParity64(cw, error);
    case 3: return decodeHamming74(cw, error);
    default: return decodeHamming84(cw, error, bad);
    }
}

__device__ void diagonalDeinterleave(const unsigned short *symbols, const int numSymbols, unsigned char *codewords,
                                     const int PPM, const int RDD)                                         // :366-381
{
    for (int x = 0; x < numSymbols / (4 + RDD); x++)
    {
        const int cwOff = x * PPM, symOff = x * (4 + RDD);
        for (int k = 0; k < 4 + RDD; k++)
        {
            const unsigned sym = symbols[symOff + k];
            int i = k % PPM;
            for (int m = 0; m < PPM; m++)
            {
                codewords[cwOff + i] |= (unsigned char)(((sym >> m) & 1) << k);
                if (++i == PPM) i = 0;
            }
        }
    }
}

} // namespace

__global__ void __launch_bounds__(64) decodePackets(const DecodeArgs a)
{
    const unsigned p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= a.nPackets) return;
    const int nsyms = a.nsyms[p];
    const unsigned short *in = a.syms + (size_t)p * a.symStride;
    unsigned char *out = a.out + (size_t)p * a.outStride;
    int outLen = -1, dropped = 0;

    unsigned short symbols[LORAHIP_DEC_MAX_SYMBOLS];
    unsigned char codewords[LORAHIP_DEC_MAX_CODEWORDS];
    unsigned char bytes[LORAHIP_DEC_MAX_CODEWORDS / 2 + 8];

    const int sf = a.sf;
    const int PPM = a.ppm == 0 ? sf : a.ppm;                                                 // LoRaDecoder.cpp:201
    do
    {
        if (PPM > sf || nsyms < LORAHIP_N_HDR_SYMBOLS) break;                                // :202 (throws), :208
        const int numSymbols = ((nsyms + (4 + a.rdd) - 1) / (4 + a.rdd)) * (4 + a.rdd);      // :210
        const int numCodewords = (numSymbols / (4 + a.rdd)) * PPM;                           // :211
        if (numSymbols > LORAHIP_DEC_MAX_SYMBOLS || nsyms > a.symStride || numCodewords + 4 > LORAHIP_DEC_MAX_CODEWORDS) { outLen = -2; break; }   // larger than this build / this row supports
        int rdd = a.rdd;                                                                     // :215
        for (int i = 0; i < numSymbols; i++)                                                 // :218-222
        {
            unsigned short sym = i < nsyms ? in[i] : 0;
            sym = (unsigned short)(sym + (1 << (sf - PPM)) / 2);
            sym = (unsigned short)(sym >> (sf - PPM));
            sym = (unsigned short)(sym ^ (sym >> 1));
            symbols[i] = sym;
        }
        if (!a.interleaving)                                                                 // :264-270
        {
            for (int i = 0; i < numSymbols; i++) reinterpret_cast<unsigned short *>(out)[i] = symbols[i];
            outLen = numSymbols;
            break;
        }
        for (int i = 0; i < numCodewords + 4; i++) codewords[i] = 0;
        {                                                                                    // :225-255
            int sOfs = 0, cOfs = 0;
            if (rdd != LORAHIP_HDR_RDD)
            {
                diagonalDeinterleave(symbols, LORAHIP_N_HDR_SYMBOLS, codewords, PPM, LORAHIP_HDR_RDD);
                if (a.explicitHdr) whiteningLfsr(codewords + LORAHIP_N_HDR_CODEWORDS, (unsigned short)(PPM - LORAHIP_N_HDR_CODEWORDS), 0, LORAHIP_HDR_RDD);
                else whiteningLfsr(codewords, (unsigned short)PPM, 0, LORAHIP_HDR_RDD);
                cOfs += PPM;
                sOfs += LORAHIP_N_HDR_SYMBOLS;
                if (numSymbols - sOfs > 0)
                {
                    diagonalDeinterleave(symbols + sOfs, numSymbols - sOfs, codewords + cOfs, PPM, rdd);
                    if (a.explicitHdr) whiteningLfsr(codewords + cOfs, (unsigned short)(numCodewords - cOfs), PPM - LORAHIP_N_HDR_CODEWORDS, rdd);
                    else whiteningLfsr(codewords + cOfs, (unsigned short)(numCodewords - cOfs), PPM, rdd);
                }
            }
            else
            {
                diagonalDeinterleave(symbols, numSymbols, codewords, PPM, rdd);
                if (a.explicitHdr) whiteningLfsr(codewords + LORAHIP_N_HDR_CODEWORDS, (unsigned short)(numCodewords - LORAHIP_N_HDR_CODEWORDS), 0, rdd);
                else whiteningLfsr(codewords, (unsigned short)numCodewords, 0, rdd);
            }
        }

        bool error = false, bad = false;                                                     // :273-274
        const int nbytes = (numCodewords + 1) / 2;
        for (int i = 0; i < nbytes + 8; i++) bytes[i] = 0;
        int dOfs = 0, cOfs = 0;
        long long packetLength = 0, dataLength = 0;
        bool checkCrc = a.crcc != 0;
        if (a.explicitHdr)                                                                   // :283-303
        {
            bytes[0] = decodeHamming84(codewords[1], error, bad) & 0xf;
            bytes[0] |= (unsigned char)(decodeHamming84(codewords[0], error, bad) << 4);     // length
            bytes[1] = decodeHamming84(codewords[2], error, bad) & 0xf;                      // coding rate and crc enable
            bytes[2] = decodeHamming84(codewords[4], error, bad) & 0xf;
            bytes[2] |= (unsigned char)(decodeHamming84(codewords[3], error, bad) << 4);     // checksum
            bytes[2] ^= headerChecksum(bytes);
            if (error && a.errorCheck) { dropped = 1; break; }
            if (0 == (bytes[1] & 1)) checkCrc = false;
            rdd = (bytes[1] >> 1) & 0x7;
            if (rdd > 4) { dropped = 1; break; }
            packetLength = bytes[0];
            dataLength = packetLength + ((bytes[1] & 1) ? 5 : 3);
            cOfs = LORAHIP_N_HDR_CODEWORDS;
            dOfs = 6;
        }
        else                                                                                 // :304-311
        {
            packetLength = a.dataLength;
            dataLength = a.crcc ? packetLength + 2 : packetLength;
        }
        if (dataLength > nbytes) { dropped = 1; break; }                                     // :313
        for (; cOfs < PPM; cOfs++, dOfs++)                                                   // :315-320
        {
            if (dOfs & 1) bytes[dOfs >> 1] |= (unsigned char)(decodeHamming84(codewords[cOfs], error, bad) << 4);
            else bytes[dOfs >> 1] = decodeHamming84(codewords[cOfs], error, bad) & 0xf;
        }
        if (dOfs & 1)                                                                        // :322-339
        {
            bytes[dOfs >> 1] |= (unsigned char)(decodeNibble(codewords[cOfs++], rdd, error, bad) << 4);
            dOfs++;
        }
        dOfs >>= 1;
        if (error && a.errorCheck) { dropped = 1; break; }                                   // :342
        for (long long i = dOfs; i < dataLength; i++)                                        // :346-361
        {
            const unsigned char c0 = codewords[cOfs++], c1 = codewords[cOfs++];
            bytes[i] = decodeNibble(c0, rdd, error, bad) & 0xf;
            bytes[i] |= (unsigned char)(decodeNibble(c1, rdd, error, bad) << 4);
        }
        if (error && a.errorCheck) { dropped = 1; break; }                                   // :363
        dOfs = 0;
        if (a.explicitHdr)                                                                   // :367-379
        {
            if (bytes[1] & 1)
            {
                const unsigned short crc = dataChecksum(bytes + 3, (int)packetLength);
                const unsigned short packetCrc = (unsigned short)(bytes[3 + packetLength] | (bytes[4 + packetLength] << 8));
                if (crc != packetCrc && checkCrc) { dropped = 1; break; }
                bytes[3 + packetLength] ^= (unsigned char)crc;
                bytes[4 + packetLength] ^= (unsigned char)(crc >> 8);
            }
            if (!a.hdr) { dOfs = 3; dataLength -= 5; }
        }
        else if (checkCrc)                                                                   // :380-388
        {
            const unsigned short crc = dataChecksum(bytes, a.dataLength);
            const unsigned short packetCrc = (unsigned short)(bytes[a.dataLength] | (bytes[a.dataLength + 1] << 8));
            if (crc != packetCrc) { dropped = 1; break; }
            bytes[a.dataLength + 0] ^= (unsigned char)crc;
            bytes[a.dataLength + 1] ^= (unsigned char)(crc >> 8);
        }
        // `dataLength -= 5` on a size_t wraps for a header without the crc flag that announces fewer than 2 bytes; the
        // reference then fails to allocate its output and posts nothing
        if (dataLength < 0) break;
        for (long long i = 0; i < dataLength; i++) out[i] = bytes[dOfs + i];                 // :391-395
        outLen = (int)dataLength;
    } while (false);
    a.outLen[p] = outLen;
    a.dropped[p] = dropped;
}


/***********************************************************************
 * group-per-packet decoder
 **********************************************************************/
namespace {

#define LORAHIP_WHITEN_LEN 4104                // positions per register: the codewords that can reach the output (LORAHIP_DEC_MAX_DATA_LENGTH
                                               // bytes + crc = 8196 nibbles + the five header codewords) / 2, and the byte steps of the crc

//! the byte sequences of the two whitening registers for both seed sets, and the CRC helper tables -- none depends on the packet
struct CodecTables
{
    unsigned char white[2][2][LORAHIP_WHITEN_LEN];     // [single-parity seeds][odd register][step]: low byte before the step (:255-268)
    unsigned short xpow[LORAHIP_WHITEN_LEN];            // x^(8 k) mod (x^16 + x^12 + x^5 + 1): one byte position of crc16sx, k times
    unsigned char lfsr8[LORAHIP_WHITEN_LEN + 2];        // the 8-bit register of dataChecksum after k steps from 0xff (:170-194)
};

constexpr unsigned long long lfsrAdvanceC(const unsigned long long r)
{
    return (r >> 8) | ((((r ^ (r >> 16) ^ (r >> 24) ^ (r >> 32)) & 0xff)) << 56);
}
constexpr int parityC(unsigned v) { int p = 0; while (v) { p ^= 1; v &= v - 1; } return p; }

constexpr CodecTables makeCodecTables()
{
    CodecTables t = {};
    const unsigned long long seeds[2][2] = { { 0x6572D100E85C2EFFull, 0xE85C2EFFFFFFFFFFull }, { 0x05121100F8ECFEEFull, 0xF8ECFEEFEFEFEFEFull } };
    for (int c = 0; c < 2; c++)
        for (int o = 0; o < 2; o++)
        {
            unsigned long long r = seeds[c][o];
            for (int n = 0; n < LORAHIP_WHITEN_LEN; n++) { t.white[c][o][n] = (unsigned char)(r & 0xff); r = lfsrAdvanceC(r); }
        }
    unsigned v = 1;                                      // x^0
    for (int k = 0; k < LORAHIP_WHITEN_LEN; k++)
    {
        t.xpow[k] = (unsigned short)v;
        for (int b = 0; b < 8; b++) v = ((v << 1) ^ ((v & 0x8000) ? 0x1021 : 0)) & 0xffff;
    }
    unsigned char l = 0xff;
    for (int k = 0; k < LORAHIP_WHITEN_LEN + 2; k++) { t.lfsr8[k] = l; l = (unsigned char)((l << 1) | parityC(l & 0xB8)); }
    return t;
}

__device__ const CodecTables kCodec = makeCodecTables();

//! a * x^(8k) mod P for a byte a: the contribution of data byte a to the crc after k further byte steps
__device__ __forceinline__ unsigned crcShift(const unsigned a, const unsigned xk)
{
    unsigned acc = 0, t = xk;
#pragma unroll
    for (int b = 0; b < 8; b++)
    {
        acc ^= ((a >> b) & 1) ? t : 0u;
        t = ((t << 1) ^ ((t & 0x8000) ? 0x1021u : 0u)) & 0xffffu;
    }
    return acc;
}

template <int G> __device__ __forceinline__ unsigned groupXor(unsigned v)
{
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v ^= __shfl_xor(v, o, G);
    return v;
}
template <int G> __device__ __forceinline__ bool groupAny(const bool p)
{
    unsigned v = p ? 1u : 0u;
#pragma unroll
    for (int o = G / 2; o > 0; o >>= 1) v |= __shfl_xor(v, o, G);
    return v != 0;
}

} // namespace

/*! What of a packet can reach the output. The reference de-interleaves and de-whitens EVERY codeword of the message (LoRaDecoder.cpp:
 * 225-255), but decodes only those the announced length needs (:315-361): the first PPM of them, the nibble that completes an odd
 * byte, then two per byte up to dataLength -- at most 2 * (255 + 5) with an explicit header (the length is one byte), 2 * (setDataLength
 * + 2) without. Codewords behind that, and the symbols they are made of, change nothing (numSymbols / numCodewords enter the checks
 * as numbers only). So the group's slice of LDS is sized by what the CONFIGURATION can need, not by the length of the row: a packet of
 * any length the demodulator can produce decodes like the reference decodes it. */
struct DecodeCaps { int symCap, cwCap; };

template <int G>
__global__ void __launch_bounds__(256) decodeGroup(const DecodeArgs a, const DecodeCaps caps, const int perBlock)
{
    // per packet: symCap Gray-coded symbols (u16), then the codewords (u8), then the nibbles (u8), 16-byte aligned slices
    extern __shared__ __attribute__((aligned(16))) unsigned char smemDec[];
    const int symCap = caps.symCap, cwCap = caps.cwCap;
    const int slice = ((symCap * 2 + 2 * cwCap + 15) & ~15);
    const int g = threadIdx.x / G, t = threadIdx.x % G;
    unsigned short *sSym = reinterpret_cast<unsigned short *>(smemDec + (size_t)g * slice);
    unsigned char *sCw = smemDec + (size_t)g * slice + symCap * 2;
    unsigned char *sNib = sCw + cwCap;
    const unsigned p = blockIdx.x * perBlock + g;
    const bool have = g < perBlock && p < a.nPackets;               // (long slices: fewer packets per workgroup than groups, the rest idle)
    const unsigned pc = have ? p : 0;
    const int nsyms = have ? a.nsyms[pc] : 0;
    const unsigned short *in = a.syms + (size_t)pc * a.symStride;
    unsigned char *out = a.out + (size_t)pc * a.outStride;
    int outLen = -1, dropped = 0;

    const int sf = a.sf;
    const int PPM = a.ppm == 0 ? sf : a.ppm;                                                 // LoRaDecoder.cpp:201
    const int bs = 4 + a.rdd;                                                                // symbols per interleaver block
    const int numSymbols = ((nsyms + bs - 1) / bs) * bs;                                     // :210
    const int numCodewords = (numSymbols / bs) * PPM;                                        // :211
    // the row holds the first symStride symbols of a longer packet: whether that is enough is known once the length is (below)
    const int rowSyms = nsyms < a.symStride ? nsyms : a.symStride;
    const int symLoad = numSymbols < symCap ? numSymbols : symCap;                           // what of it this slice holds
    const int cwLoad = numCodewords + 4 < cwCap ? numCodewords + 4 : cwCap;
    // group-uniform early outs (:202 throws, :208 too short)
    bool go = have && !(PPM > sf || nsyms < LORAHIP_N_HDR_SYMBOLS);

    // ---- Gray code with rounding to the symbol size (:218-222) --------------------------------------------------------
    if (go && !a.interleaving && nsyms > a.symStride) { outLen = -2; go = false; }           // every symbol is output: the row must hold them all
    if (go)
        for (int i = t; i < (a.interleaving ? symLoad : numSymbols); i += G)
        {
            unsigned short sym = i < rowSyms ? in[i] : 0;
            sym = (unsigned short)(sym + (1 << (sf - PPM)) / 2);
            sym = (unsigned short)(sym >> (sf - PPM));
            sym = (unsigned short)(sym ^ (sym >> 1));
            if (a.interleaving) sSym[i] = sym;
            else reinterpret_cast<unsigned short *>(out)[i] = sym;                          // :264-270
        }
    if (go && !a.interleaving) { outLen = numSymbols; go = false; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __syncthreads();

    // ---- diagonal de-interleave (:366-381) + de-whitening (:225-255), one codeword per lane and round ------------------
    // the first block is always 4/8 over 8 symbols unless the whole packet is (then every block is)
    const bool hdrBlock = a.rdd != LORAHIP_HDR_RDD;
    if (go)
        for (int c = t; c < cwLoad; c += G)
        {
            unsigned cw = 0;
            if (c < numCodewords)
            {
                int x, i, symOff, R;
                if (hdrBlock && c < PPM) { x = 0; i = c; symOff = 0; R = LORAHIP_HDR_RDD; }
                else if (hdrBlock) { x = (c - PPM) / PPM; i = (c - PPM) - x * PPM; symOff = LORAHIP_N_HDR_SYMBOLS + x * bs; R = a.rdd; }
                else { x = c / PPM; i = c - x * PPM; symOff = x * bs; R = a.rdd; }
                // with a header block the payload's symbols may end inside a block of `bs`: the reference de-interleaves
                // (numSymbols - 8) / bs whole blocks and leaves the rest of the codewords zero
                // (... and a block whose symbols lie behind the slice is behind what any length can need: left zero, never read)
                const bool whole = (!hdrBlock || c < PPM || (symOff + (4 + R) <= numSymbols)) && symOff + (4 + R) <= symLoad;
                if (whole)
                    for (int k = 0; k < 4 + R; k++)
                    {
                        int m = i - (k % PPM);
                        if (m < 0) m += PPM;
                        cw |= ((sSym[symOff + k] >> m) & 1u) << k;
                    }
                // position in the whitening sequence: codeword index, minus the five header codewords that are not whitened
                // (with a header block and nothing but it -- exactly 8 symbols -- the reference skips the payload's whitening call)
                const int pos = a.explicitHdr ? c - LORAHIP_N_HDR_CODEWORDS : c;
                // (the tables end behind the last codeword any length can need: LORAHIP_DEC_MAX_DATA_LENGTH)
                if (pos >= 0 && (pos >> 1) < LORAHIP_WHITEN_LEN && !(hdrBlock && c >= PPM && numSymbols <= LORAHIP_N_HDR_SYMBOLS))
                {
                    const int Rw = (hdrBlock && c < PPM) ? LORAHIP_HDR_RDD : a.rdd;
                    const unsigned keep = 0xffu >> (4 - Rw);
                    cw ^= kCodec.white[Rw == 1 ? 1 : 0][pos & 1][pos >> 1] & keep;
                }
            }
            sCw[c] = (unsigned char)cw;
        }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __syncthreads();

    // ---- header (:283-311), evaluated by every lane of the group alike -------------------------------------------------
    bool error = false, bad = false;
    const int nbytes = (numCodewords + 1) / 2;
    long long packetLength = 0, dataLength = 0;
    bool checkCrc = a.crcc != 0;
    int rdd = a.rdd;
    unsigned char h0 = 0, h1 = 0, h2 = 0;
    if (go)
    {
        if (a.explicitHdr)
        {
            h0 = (unsigned char)((decodeHamming84(sCw[1], error, bad) & 0xf) | (decodeHamming84(sCw[0], error, bad) << 4));   // length
            h1 = decodeHamming84(sCw[2], error, bad) & 0xf;                                      // coding rate and crc enable
            h2 = (unsigned char)((decodeHamming84(sCw[4], error, bad) & 0xf) | (decodeHamming84(sCw[3], error, bad) << 4));   // checksum
            const unsigned char hb[2] = { h0, h1 };
            h2 ^= headerChecksum(hb);
            if (error && a.errorCheck) { dropped = 1; go = false; }
            if (go)
            {
                if (0 == (h1 & 1)) checkCrc = false;
                rdd = (h1 >> 1) & 0x7;
                if (rdd > 4) { dropped = 1; go = false; }
                packetLength = h0;
                dataLength = packetLength + ((h1 & 1) ? 5 : 3);
            }
        }
        else
        {
            packetLength = a.dataLength;
            dataLength = a.crcc ? packetLength + 2 : packetLength;
        }
        if (go && dataLength > nbytes) { dropped = 1; go = false; }                          // :313
    }

    // ---- nibbles (:315-361): codeword c -> nibble c (+1 after an explicit header); the first block is 4/8, the rest at the
    // coding rate the HEADER names; decoded are the first block, the nibble that completes its last byte, and what
    // dataLength bytes need -- errors anywhere else do not count, exactly as in the reference's loops
    const int c0 = a.explicitHdr ? LORAHIP_N_HDR_CODEWORDS : 0, shift = a.explicitHdr ? 1 : 0;
    long long cEnd = 0;
    if (go)
    {
        cEnd = 2 * dataLength - shift;                                                       // first codeword NOT needed by the bytes
        const int fill = PPM + (((PPM + shift) & 1) ? 1 : 0);                               // first block + the odd nibble
        if (cEnd < fill) cEnd = fill;
        // the symbols those codewords are made of: whole interleaver blocks. A packet longer than its row decodes all the same while
        // they lie inside the row; if not, the caller's rows are too short for this packet (-2: reported, never guessed)
        const long long nBlocks = (cEnd + PPM - 1) / PPM;
        const long long needSyms = hdrBlock ? LORAHIP_N_HDR_SYMBOLS + (nBlocks - 1) * bs : nBlocks * bs;
        if ((nsyms > a.symStride && needSyms > a.symStride) || needSyms > symCap || cEnd + 4 > cwCap) { outLen = -2; go = false; }
    }
    if (go)
    {
        if (a.explicitHdr) { sNib[0] = h0 & 0xf; sNib[1] = h0 >> 4; sNib[2] = h1 & 0xf; sNib[3] = 0; sNib[4] = h2 & 0xf; sNib[5] = h2 >> 4; }
        for (int c = c0 + t; c < cEnd; c += G)
        {
            const unsigned char cw = sCw[c];
            const unsigned char nib = c < PPM ? decodeHamming84(cw, error, bad) : decodeNibble(cw, rdd, error, bad);
            sNib[c + shift] = nib & 0xf;
        }
        // the bytes beyond the decoded nibbles are zero in the reference's buffer
        for (long long v = cEnd + shift + t; v < 2 * (dataLength + 1); v += G) sNib[v] = 0;
    }
    error = groupAny<G>(error);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __syncthreads();
    if (go && error && a.errorCheck) { dropped = 1; go = false; }                            // :342, :363

    // ---- checksum (:367-388): crc = sum over the bytes of byte * x^(8 (len-1-i)), then the two LFSR masks ---------------
    int dOfs = 0;
    if (go && (a.explicitHdr ? (h1 & 1) != 0 : checkCrc))
    {
        const int base = a.explicitHdr ? 3 : 0;
        const int len = a.explicitHdr ? (int)packetLength : a.dataLength;
        unsigned acc = 0;
        for (int i = t; i < len; i += G)
        {
            const unsigned byte = sNib[2 * (base + i)] | (sNib[2 * (base + i) + 1] << 4);
            // res_{i+1} = S(res_i) ^ d_i: byte i is shifted by the len-1-i byte steps that follow it
            acc ^= crcShift(byte, kCodec.xpow[len - 1 - i]);
        }
        unsigned crc = groupXor<G>(acc);
        crc ^= kCodec.lfsr8[len];                                                            // res ^= v after len steps
        crc ^= (unsigned)kCodec.lfsr8[len + 1] << 8;                                         // ... and one step later, high byte
        crc &= 0xffff;
        const unsigned packetCrc = (sNib[2 * (base + len)] | (sNib[2 * (base + len) + 1] << 4)) |
                                   ((sNib[2 * (base + len) + 2] | (sNib[2 * (base + len) + 3] << 4)) << 8);
        if (a.explicitHdr)
        {
            if (crc != packetCrc && checkCrc) { dropped = 1; go = false; }
        }
        else if (crc != packetCrc) { dropped = 1; go = false; }
        if (go && t == 0)
        {
            // bytes[len] ^= crc, bytes[len + 1] ^= crc >> 8 (:377-378, :386-387)
            const unsigned lo = (sNib[2 * (base + len)] | (sNib[2 * (base + len) + 1] << 4)) ^ (crc & 0xff);
            const unsigned hi = (sNib[2 * (base + len) + 2] | (sNib[2 * (base + len) + 3] << 4)) ^ (crc >> 8);
            sNib[2 * (base + len)] = lo & 0xf; sNib[2 * (base + len) + 1] = (lo >> 4) & 0xf;
            sNib[2 * (base + len) + 2] = hi & 0xf; sNib[2 * (base + len) + 3] = (hi >> 4) & 0xf;
        }
    }
    if (go && a.explicitHdr && !a.hdr) { dOfs = 3; dataLength -= 5; }                        // :379
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __syncthreads();
    // `dataLength -= 5` on a size_t wraps for a header without the crc flag that announces fewer than 2 bytes; the
    // reference then fails to allocate its output and posts nothing
    if (go && dataLength >= 0)
    {
        for (long long i = t; i < dataLength; i += G) out[i] = (unsigned char)(sNib[2 * (dOfs + i)] | (sNib[2 * (dOfs + i) + 1] << 4));   // :391-395
        outLen = (int)dataLength;
    }
    if (have && t == 0)
    {
        a.outLen[p] = outLen;
        a.dropped[p] = dropped;
    }
}

//! the slice a configuration needs (see DecodeCaps): the symbols / codewords that can reach the output, or the whole row if that is less
static DecodeCaps decodeCaps(const DecodeArgs &a)
{
    const int PPM = a.ppm == 0 ? a.sf : a.ppm, bs = 4 + a.rdd;
    const long long bytesMax = a.explicitHdr ? 255 + 5 : (long long)a.dataLength + 2;
    long long cEnd = 2 * bytesMax;
    if (cEnd < PPM + 1) cEnd = PPM + 1;
    const long long nBlocks = (cEnd + PPM - 1) / PPM + 1;
    const long long symNeed = LORAHIP_N_HDR_SYMBOLS + nBlocks * bs;
    const long long symRow = ((a.symStride + 7 + 8) / bs + 1) * bs + LORAHIP_N_HDR_SYMBOLS;       // the row, rounded up to whole blocks
    DecodeCaps c;
    c.symCap = int(symNeed < symRow ? symNeed : symRow);
    // codewords of those symbols (+ the 4 zero entries behind numCodewords the nibble loop may read, + nibble offsets)
    c.cwCap = int((c.symCap / bs + 2) * PPM + 16);
    return c;
}

template <int G>
static hipError_t launchDecodeGroup(const DecodeArgs &a, const DecodeCaps &caps, hipStream_t stream)
{
    const size_t slice = size_t((caps.symCap * 2 + 2 * caps.cwCap + 15) & ~15);
    const size_t ldsMax = 160 * 1024;
    unsigned perBlock = 256 / G;
    if (slice > ldsMax) return hipErrorInvalidValue;                // (lorahip_decode_packets bounds data_length: cannot happen)
    if (slice * perBlock > ldsMax) perBlock = unsigned(ldsMax / slice);
    const size_t smem = slice * perBlock;
    static unsigned long long attrDone = 0;
    {
        const hipError_t e = ensureDynamicLds(reinterpret_cast<const void *>(decodeGroup<G>), ldsMax, attrDone);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL((decodeGroup<G>), dim3((a.nPackets + perBlock - 1) / perBlock), dim3(256), smem, stream, a, caps, int(perBlock));
    return hipGetLastError();
}

hipError_t launchDecode(const DecodeArgs &a, const int variant, hipStream_t stream)
{
    if (a.nPackets == 0) return hipSuccess;
    if (variant == 1)
    {
        hipLaunchKernelGGL(decodePackets, dim3((a.nPackets + 63) / 64), dim3(64), 0, stream, a);
        return hipGetLastError();
    }
    // lanes per packet by what a packet can need: n symbols make about n * PPM / (4 + rdd) codewords
    const DecodeCaps caps = decodeCaps(a);
    if (caps.symCap <= 48) return launchDecodeGroup<8>(a, caps, stream);
    if (caps.symCap <= 96) return launchDecodeGroup<16>(a, caps, stream);
    if (caps.symCap <= 200) return launchDecodeGroup<32>(a, caps, stream);
    return launchDecodeGroup<64>(a, caps, stream);
}

// rows: the de-whitening takes its length as a uint16_t in the reference (LoRaCodes.hpp: Sx1272ComputeWhiteningLfsr), i.e. messages of
// up to 65535 codewords are what the reference itself decodes as written; 16384 symbols stay below that at every setting
int decodeMaxSymbols() { return 16384; }
int decodeMaxDataLength() { return LORAHIP_DEC_MAX_DATA_LENGTH; }

} // namespace lorahip

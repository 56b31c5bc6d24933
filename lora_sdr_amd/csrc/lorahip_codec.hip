// Batched LoRaDecoder: symbol packets -> bytes, one lane per packet.
//
// What the LoRaDecoder block does for one message (LoRaDecoder.cpp:196-397 on LoRaCodes.hpp): Gray-code the demodulated
// symbols with rounding to the symbol size, de-interleave diagonally into codewords (the first block always 4/8), strip the
// whitening with the two interleaved LFSRs, Hamming / parity decode, parse and check the explicit header, check the
// payload CRC. Integer and bit work on a few hundred bytes per packet: the parallel axis is the packet (the demodulator
// hands over thousands per launch), each lane walks its packet exactly in the reference's order. The working arrays
// (symbols, codewords, bytes) live in per-lane scratch.
#include "lorahip_internal.h"

namespace lorahip {

#define LORAHIP_DEC_MAX_SYMBOLS 520            // symbols per packet incl. rounding to a block
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

hipError_t launchDecode(const DecodeArgs &a, hipStream_t stream)
{
    if (a.nPackets == 0) return hipSuccess;
    hipLaunchKernelGGL(decodePackets, dim3((a.nPackets + 63) / 64), dim3(64), 0, stream, a);
    return hipGetLastError();
}

int decodeMaxSymbols() { return LORAHIP_DEC_MAX_SYMBOLS - 8; }

} // namespace lorahip

/* TEST INFRASTRUCTURE ONLY (oracle/) -- see lora_oracle.h for the rules.
 *
 * Plain-C restatement of the receive-side codec of LoRa-SDR: the LoRaDecoder block's work()
 * (LoRaDecoder.cpp:196-397) on the code primitives of LoRaCodes.hpp. Each block cites the lines it
 * follows. Pinned bit-exact against the verbatim blocks compiled in oracle/_ref
 * (tests/test_oracle_vs_ref.py) and against golden vectors recorded from them.
 *
 * Where the reference reads one element past a std::vector (an odd number of codewords and a data
 * length that needs the missing nibble) this restatement reads a zero.
 */
#include "lora_oracle.h"
#include <stdlib.h>
#include <string.h>

#define HEADER_RDD 4                              /* LoRaCodes.hpp:103 */
#define N_HEADER_SYMBOLS (HEADER_RDD + 4)         /* :104 */
#define N_HEADER_CODEWORDS 5                      /* :105 */

static unsigned round_up(unsigned num, unsigned factor) { return ((num + factor - 1) / factor) * factor; }  /* :111-114 */

static int parity_of(unsigned v) { return __builtin_parity(v); }

/* Header checksum (LoRaCodes.hpp:131-156): five parity bits over the 12 header bits w = h[0] | (h[1] & 0xf) << 8.
 * The reference spells each one out as a chain of xors of named bits; as masks over w, bit 4 down to bit 0: */
static uint8_t header_checksum(const uint8_t *h)
{
    static const unsigned mask[5] = { 0xF12, 0x725, 0xA49, 0x18E, 0x0F0 };
    const unsigned w = h[0] | ((unsigned)(h[1] & 0xf) << 8);
    uint8_t res = 0;
    for (int k = 0; k < 5; k++) res |= (uint8_t)(parity_of(w & mask[k]) << k);
    return res;
}

/* CRC-CCITT step for one byte position, MSB first (LoRaCodes.hpp:158-168) */
static uint16_t crc16sx(uint16_t crc, const uint16_t poly)
{
    for (int bit = 0; bit < 8; bit++) {
        const int top = crc & 0x8000;
        crc = (uint16_t)(crc << 1);
        if (top) crc ^= poly;
    }
    return crc;
}

/* payload checksum of the sx1272 (LoRaCodes.hpp:170-194): CCITT crc over the bytes, the result masked with two steps
 * of an 8-bit LFSR (taps 0xB8) that advances once per byte */
static uint16_t sx1272_data_checksum(const uint8_t *data, int length)
{
    uint16_t res = 0;
    uint8_t v = 0xff;
    for (int i = 0; i < length; i++) {
        const uint16_t crc = crc16sx(res, 0x1021);
        v = (uint8_t)((v << 1) | parity_of(v & 0xB8));
        res = crc ^ data[i];
    }
    res ^= v;
    v = (uint8_t)((v << 1) | parity_of(v & 0xB8));
    res ^= (uint16_t)(v << 8);
    return res;
}

/* De-whitening with the two interleaved 64-bit LFSRs (LoRaCodes.hpp:255-268): codeword position p (counted from
 * bitOfs) takes the low bits of register p mod 2, which then advances by one byte: feedback byte = b0^b2^b3^b4 of the
 * register (x^8 polynomial 0x1D on bytes), shifted in at the top. The length is a uint16_t in the reference. */
static uint64_t lfsr_advance(const uint64_t r)
{
    const uint64_t fb = (r ^ (r >> 16) ^ (r >> 24) ^ (r >> 32)) & 0xff;
    return (r >> 8) | (fb << 56);
}

static void whitening_lfsr(uint8_t *buffer, uint16_t bufferSize, const int bitOfs, const size_t RDD)
{
    uint64_t reg[2];
    if (RDD == 1) { reg[0] = 0x05121100F8ECFEEFull; reg[1] = 0xF8ECFEEFEFEFEFEFull; }   /* single-parity mode has its own seeds */
    else { reg[0] = 0x6572D100E85C2EFFull; reg[1] = 0xE85C2EFFFFFFFFFFull; }
    const uint8_t keep = (uint8_t)(0xff >> (4 - RDD));                                    /* 4+RDD bits per codeword */
    for (int p = 0; p < bitOfs; p++) reg[p & 1] = lfsr_advance(reg[p & 1]);
    for (int j = 0; j < (int)bufferSize; j++) {
        const int p = bitOfs + j;
        buffer[j] ^= (uint8_t)(reg[p & 1] & keep);
        reg[p & 1] = lfsr_advance(reg[p & 1]);
    }
}

/* The sx Hamming codes (LoRaCodes.hpp:222-259, :284-312) and parity checks (:318-323, :335-343) as syndrome masks:
 * parity bit k of a codeword covers the bits in cover[k]; a non-zero syndrome that names a data bit flips it, one
 * that names a parity bit is ignored, anything else is uncorrectable (8,4 only). */
static const unsigned cover[4] = { 0x17, 0x2E, 0x4B, 0x8D };   /* p0: b0 b1 b2 b4 | p1: b1 b2 b3 b5 | p2: b0 b1 b3 b6 | p3: b0 b2 b3 b7 */

static unsigned char decode_hamming84(const unsigned char b, int *error, int *bad)
{
    /* syndrome -> data bit to flip (0 = none), or 0xff = uncorrectable */
    static const unsigned char fix[16] = { 0, 0, 0, 0xff, 0, 0xff, 0xff, 2, 0, 0xff, 0xff, 4, 0xff, 1, 8, 0xff };
    unsigned syn = 0;
    for (int k = 0; k < 4; k++) syn |= (unsigned)parity_of(b & cover[k]) << k;
    if (syn) *error = 1;
    if (fix[syn] == 0xff) { *bad = 1; return b & 0xf; }
    return (b ^ fix[syn]) & 0xf;
}

static unsigned char decode_hamming74(const unsigned char b, int *error)
{
    static const unsigned char fix[8] = { 0, 0, 0, 4, 0, 1, 8, 2 };
    unsigned syn = 0;
    for (int k = 0; k < 3; k++) syn |= (unsigned)parity_of(b & cover[k] & 0x7f) << k;
    if (syn) *error = 1;
    return (b ^ fix[syn]) & 0xf;
}

/* 5/4: one parity bit over the four data bits (b4) */
static unsigned char check_parity54(const unsigned char b, int *error)
{
    if (parity_of(b & 0x1F)) *error = 1;
    return b & 0xf;
}

/* 6/4: the first two parity bits of the Hamming code (b4, b5) */
static unsigned char check_parity64(const unsigned char b, int *error)
{
    if (parity_of(b & cover[0]) | parity_of(b & cover[1])) *error = 1;
    return b & 0xf;
}

/* Diagonal de-interleaver (LoRaCodes.hpp:366-381): block x turns 4+RDD symbols of PPM bits into PPM codewords of
 * 4+RDD bits; bit m of symbol k is bit k of codeword (m + k) mod PPM. */
static void diagonal_deinterleave(const uint16_t *symbols, const size_t numSymbols, uint8_t *codewords, const size_t PPM, const size_t RDD)
{
    const size_t width = 4 + RDD;
    for (size_t blk = 0; blk < numSymbols / width; blk++) {
        uint8_t *cw = codewords + blk * PPM;
        for (size_t k = 0; k < width; k++) {
            const unsigned sym = symbols[blk * width + k];
            size_t dst = k % PPM;
            for (size_t m = 0; m < PPM; m++) {
                cw[dst] |= (uint8_t)(((sym >> m) & 1u) << k);
                dst = dst + 1 == PPM ? 0 : dst + 1;
            }
        }
    }
}

/* the primitives one by one, for exhaustive checks against the reference's (tests/test_oracle_vs_ref.py) */
int lo_code_primitive(int which, int b)
{
    int error = 0, bad = 0, v = 0;
    switch (which) {
    case 0: v = decode_hamming84((unsigned char)b, &error, &bad); break;
    case 1: v = decode_hamming74((unsigned char)b, &error); break;
    case 2: v = check_parity54((unsigned char)b, &error); break;
    case 3: v = check_parity64((unsigned char)b, &error); break;
    case 4: { uint8_t h[3] = { (uint8_t)(b & 0xff), (uint8_t)((b >> 8) & 0xf), 0 }; return header_checksum(h); }
    case 5: return (uint16_t)((uint16_t)b ^ ((uint16_t)b >> 1));
    default: return -1;
    }
    return v | (error ? 0x100 : 0) | (bad ? 0x200 : 0);
}

/* LoRaDecoder.cpp:196-397. Returns the number of output elements written to `out` (bytes; uint16 symbols when
 * interleaving is off), or -1 when the block posts nothing; *dropped is set when it called drop(). */
long lo_decode(const lo_decoder_cfg *c, const uint16_t *syms, size_t nsyms, void *out, int *dropped)
{
    *dropped = 0;
    const size_t PPM = (c->ppm == 0) ? (size_t)c->sf : (size_t)c->ppm;                    /* :201 */
    if (PPM > (size_t)c->sf) return -1;                                                     /* :202 throws */
    if (nsyms < N_HEADER_SYMBOLS) return -1;                                                /* :208 */
    const size_t numSymbols = round_up((unsigned)nsyms, (unsigned)(4 + c->rdd));            /* :210 */
    const size_t numCodewords = (numSymbols / (4 + (size_t)c->rdd)) * PPM;                  /* :211 */
    uint16_t *symbols = (uint16_t *)calloc(numSymbols + 1, sizeof(uint16_t));
    memcpy(symbols, syms, nsyms * sizeof(uint16_t));
    int rdd = c->rdd;                                                                       /* :215 */
    for (size_t i = 0; i < numSymbols; i++) {                                               /* :218-222 */
        uint16_t sym = symbols[i];
        sym = (uint16_t)(sym + (1 << (c->sf - (int)PPM)) / 2);
        sym = (uint16_t)(sym >> (c->sf - (int)PPM));
        sym = (uint16_t)(sym ^ (sym >> 1));
        symbols[i] = sym;
    }
    if (!c->interleaving) {                                                                 /* :264-270 */
        memcpy(out, symbols, numSymbols * sizeof(uint16_t));
        free(symbols);
        return (long)numSymbols;
    }
    uint8_t *codewords = (uint8_t *)calloc(numCodewords + 4, 1);
    {                                                                                       /* :225-255 */
        size_t sOfs = 0, cOfs = 0;
        if (rdd != HEADER_RDD) {
            diagonal_deinterleave(symbols, N_HEADER_SYMBOLS, codewords, PPM, HEADER_RDD);
            if (c->explicit_hdr) whitening_lfsr(codewords + N_HEADER_CODEWORDS, (uint16_t)(PPM - N_HEADER_CODEWORDS), 0, HEADER_RDD);
            else whitening_lfsr(codewords, (uint16_t)PPM, 0, HEADER_RDD);
            cOfs += PPM;
            sOfs += N_HEADER_SYMBOLS;
            if (numSymbols - sOfs > 0) {
                diagonal_deinterleave(symbols + sOfs, numSymbols - sOfs, codewords + cOfs, PPM, (size_t)rdd);
                if (c->explicit_hdr) whitening_lfsr(codewords + cOfs, (uint16_t)(numCodewords - cOfs), (int)(PPM - N_HEADER_CODEWORDS), (size_t)rdd);
                else whitening_lfsr(codewords + cOfs, (uint16_t)(numCodewords - cOfs), (int)PPM, (size_t)rdd);
            }
        } else {
            diagonal_deinterleave(symbols, numSymbols, codewords, PPM, (size_t)rdd);
            if (c->explicit_hdr) whitening_lfsr(codewords + N_HEADER_CODEWORDS, (uint16_t)(numCodewords - N_HEADER_CODEWORDS), 0, (size_t)rdd);
            else whitening_lfsr(codewords, (uint16_t)numCodewords, 0, (size_t)rdd);
        }
    }
    free(symbols);

    int error = 0, bad = 0;                                                                 /* :273-274 */
    const size_t nbytes = (numCodewords + 1) / 2;
    uint8_t *bytes = (uint8_t *)calloc(nbytes + 8, 1);
    size_t dOfs = 0, cOfs = 0, packetLength = 0, dataLength = 0;
    int checkCrc = c->crcc;
    long result = -1;
#define LO_DROP() do { *dropped = 1; goto done; } while (0)
    if (c->explicit_hdr) {                                                                  /* :283-303 */
        bytes[0] = decode_hamming84(codewords[1], &error, &bad) & 0xf;
        bytes[0] |= (uint8_t)(decode_hamming84(codewords[0], &error, &bad) << 4);
        bytes[1] = decode_hamming84(codewords[2], &error, &bad) & 0xf;
        bytes[2] = decode_hamming84(codewords[4], &error, &bad) & 0xf;
        bytes[2] |= (uint8_t)(decode_hamming84(codewords[3], &error, &bad) << 4);
        bytes[2] ^= header_checksum(bytes);
        if (error && c->error_check) LO_DROP();
        if (0 == (bytes[1] & 1)) checkCrc = 0;
        rdd = (bytes[1] >> 1) & 0x7;
        if (rdd > 4) LO_DROP();
        packetLength = bytes[0];
        dataLength = packetLength + ((bytes[1] & 1) ? 5 : 3);
        cOfs = N_HEADER_CODEWORDS;
        dOfs = 6;
    } else {                                                                                /* :304-311 */
        packetLength = (size_t)c->data_length;
        dataLength = c->crcc ? packetLength + 2 : packetLength;
    }
    if (dataLength > nbytes) LO_DROP();                                                     /* :313 */
    for (; cOfs < PPM; cOfs++, dOfs++) {                                                    /* :315-320 */
        if (dOfs & 1) bytes[dOfs >> 1] |= (uint8_t)(decode_hamming84(codewords[cOfs], &error, &bad) << 4);
        else bytes[dOfs >> 1] = decode_hamming84(codewords[cOfs], &error, &bad) & 0xf;
    }
    if (dOfs & 1) {                                                                         /* :322-339 */
        if (rdd == 0) bytes[dOfs >> 1] |= (uint8_t)(codewords[cOfs++] << 4);
        else if (rdd == 1) bytes[dOfs >> 1] |= (uint8_t)(check_parity54(codewords[cOfs++], &error) << 4);
        else if (rdd == 2) bytes[dOfs >> 1] |= (uint8_t)(check_parity64(codewords[cOfs++], &error) << 4);
        else if (rdd == 3) bytes[dOfs >> 1] |= (uint8_t)(decode_hamming74(codewords[cOfs++], &error) << 4);
        else if (rdd == 4) bytes[dOfs >> 1] |= (uint8_t)(decode_hamming84(codewords[cOfs++], &error, &bad) << 4);
        dOfs++;
    }
    dOfs >>= 1;
    if (error && c->error_check) LO_DROP();                                                 /* :342 */
    for (size_t i = dOfs; i < dataLength; i++) {                                            /* :346-361 */
        const uint8_t c0 = codewords[cOfs++], c1 = codewords[cOfs++];
        if (rdd == 0) { bytes[i] = c0 & 0xf; bytes[i] |= (uint8_t)(c1 << 4); }
        else if (rdd == 1) { bytes[i] = check_parity54(c0, &error); bytes[i] |= (uint8_t)(check_parity54(c1, &error) << 4); }
        else if (rdd == 2) { bytes[i] = check_parity64(c0, &error); bytes[i] |= (uint8_t)(check_parity64(c1, &error) << 4); }
        else if (rdd == 3) { bytes[i] = decode_hamming74(c0, &error) & 0xf; bytes[i] |= (uint8_t)(decode_hamming74(c1, &error) << 4); }
        else { bytes[i] = decode_hamming84(c0, &error, &bad) & 0xf; bytes[i] |= (uint8_t)(decode_hamming84(c1, &error, &bad) << 4); }
    }
    if (error && c->error_check) LO_DROP();                                                 /* :363 */
    dOfs = 0;
    if (c->explicit_hdr) {                                                                  /* :367-379 */
        if (bytes[1] & 1) {
            const uint16_t crc = sx1272_data_checksum(bytes + 3, (int)packetLength);
            const uint16_t packetCrc = (uint16_t)(bytes[3 + packetLength] | (bytes[4 + packetLength] << 8));
            if (crc != packetCrc && checkCrc) LO_DROP();
            bytes[3 + packetLength] ^= (uint8_t)crc;
            bytes[4 + packetLength] ^= (uint8_t)(crc >> 8);
        }
        if (!c->hdr) { dOfs = 3; dataLength -= 5; }
    } else if (checkCrc) {                                                                  /* :380-388 */
        const uint16_t crc = sx1272_data_checksum(bytes, c->data_length);
        const uint16_t packetCrc = (uint16_t)(bytes[c->data_length] | (bytes[c->data_length + 1] << 8));
        if (crc != packetCrc) LO_DROP();
        bytes[c->data_length + 0] ^= (uint8_t)crc;
        bytes[c->data_length + 1] ^= (uint8_t)(crc >> 8);
    }
    /* dataLength is a size_t: an explicit packet without the crc flag and a length below 2 wraps around at `-= 5`
     * and the reference then asks for a huge BufferChunk; treated as "nothing posted" here */
    if (dataLength > nbytes + 8) goto done;
    memcpy(out, bytes + dOfs, dataLength);                                                  /* :391-395 */
    result = (long)dataLength;
done:
#undef LO_DROP
    free(codewords);
    free(bytes);
    return result;
}

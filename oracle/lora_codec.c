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

/* LoRaCodes.hpp:131-156 */
static uint8_t header_checksum(const uint8_t *h)
{
    const int a0 = (h[0] >> 4) & 1, a1 = (h[0] >> 5) & 1, a2 = (h[0] >> 6) & 1, a3 = (h[0] >> 7) & 1;
    const int b0 = (h[0] >> 0) & 1, b1 = (h[0] >> 1) & 1, b2 = (h[0] >> 2) & 1, b3 = (h[0] >> 3) & 1;
    const int c0 = (h[1] >> 0) & 1, c1 = (h[1] >> 1) & 1, c2 = (h[1] >> 2) & 1, c3 = (h[1] >> 3) & 1;
    uint8_t res;
    res = (uint8_t)((a0 ^ a1 ^ a2 ^ a3) << 4);
    res |= (a3 ^ b1 ^ b2 ^ b3 ^ c0) << 3;
    res |= (a2 ^ b0 ^ b3 ^ c1 ^ c3) << 2;
    res |= (a1 ^ b0 ^ b2 ^ c0 ^ c1 ^ c2) << 1;
    res |= a0 ^ b1 ^ c0 ^ c1 ^ c2 ^ c3;
    return res;
}

/* LoRaCodes.hpp:158-168 */
static uint16_t crc16sx(uint16_t crc, const uint16_t poly)
{
    for (int i = 0; i < 8; i++) {
        if (crc & 0x8000) crc = (uint16_t)((crc << 1) ^ poly);
        else crc = (uint16_t)(crc << 1);
    }
    return crc;
}

/* LoRaCodes.hpp:170-175 */
static uint8_t xsum8(uint8_t t)
{
    t ^= t >> 4;
    t ^= t >> 2;
    t ^= t >> 1;
    return t & 1;
}

/* LoRaCodes.hpp:181-194 */
static uint16_t sx1272_data_checksum(const uint8_t *data, int length)
{
    uint16_t res = 0;
    uint8_t v = 0xff;
    uint16_t crc = 0;
    for (int i = 0; i < length; i++) {
        crc = crc16sx(res, 0x1021);
        v = (uint8_t)(xsum8(v & 0xB8) | (v << 1));
        res = crc ^ data[i];
    }
    res ^= v;
    v = (uint8_t)(xsum8(v & 0xB8) | (v << 1));
    res ^= (uint16_t)(v << 8);
    return res;
}

/* LoRaCodes.hpp:255-268: the interleaved LFSRs; bufferSize is a uint16_t parameter in the reference */
static void whitening_lfsr(uint8_t *buffer, uint16_t bufferSize, const int bitOfs, const size_t RDD)
{
    static const uint64_t seed1[2] = { 0x6572D100E85C2EFFull, 0xE85C2EFFFFFFFFFFull };
    static const uint64_t seed2[2] = { 0x05121100F8ECFEEFull, 0xF8ECFEEFEFEFEFEFull };
    const uint8_t m = (uint8_t)(0xff >> (4 - RDD));
    uint64_t r[2] = { (1 == RDD) ? seed2[0] : seed1[0], (1 == RDD) ? seed2[1] : seed1[1] };
    int i, j;
    for (i = 0; i < bitOfs; i++)
        r[i & 1] = (r[i & 1] >> 8) | (((r[i & 1] >> 32) ^ (r[i & 1] >> 24) ^ (r[i & 1] >> 16) ^ r[i & 1]) << 56);
    for (j = 0; j < bufferSize; j++, i++) {
        buffer[j] ^= r[i & 1] & m;
        r[i & 1] = (r[i & 1] >> 8) | (((r[i & 1] >> 32) ^ (r[i & 1] >> 24) ^ (r[i & 1] >> 16) ^ r[i & 1]) << 56);
    }
}

/* LoRaCodes.hpp:222-259 */
static unsigned char decode_hamming84(const unsigned char b, int *error, int *bad)
{
    const int b0 = (b >> 0) & 1, b1 = (b >> 1) & 1, b2 = (b >> 2) & 1, b3 = (b >> 3) & 1;
    const int b4 = (b >> 4) & 1, b5 = (b >> 5) & 1, b6 = (b >> 6) & 1, b7 = (b >> 7) & 1;
    const int p0 = b0 ^ b1 ^ b2 ^ b4, p1 = b1 ^ b2 ^ b3 ^ b5, p2 = b0 ^ b1 ^ b3 ^ b6, p3 = b0 ^ b2 ^ b3 ^ b7;
    const int parity = (p0 << 0) | (p1 << 1) | (p2 << 2) | (p3 << 3);
    if (parity != 0) *error = 1;
    switch (parity & 0xf) {
    case 0xD: return (b ^ 1) & 0xf;
    case 0x7: return (b ^ 2) & 0xf;
    case 0xB: return (b ^ 4) & 0xf;
    case 0xE: return (b ^ 8) & 0xf;
    case 0x0: case 0x1: case 0x2: case 0x4: case 0x8: return b & 0xf;
    default: *bad = 1; return b & 0xf;
    }
}

/* LoRaCodes.hpp:284-312 */
static unsigned char decode_hamming74(const unsigned char b, int *error)
{
    const int b0 = (b >> 0) & 1, b1 = (b >> 1) & 1, b2 = (b >> 2) & 1, b3 = (b >> 3) & 1;
    const int b4 = (b >> 4) & 1, b5 = (b >> 5) & 1, b6 = (b >> 6) & 1;
    const int p0 = b0 ^ b1 ^ b2 ^ b4, p1 = b1 ^ b2 ^ b3 ^ b5, p2 = b0 ^ b1 ^ b3 ^ b6;
    const int parity = (p0 << 0) | (p1 << 1) | (p2 << 2);
    if (parity != 0) *error = 1;
    switch (parity) {
    case 0x5: return (b ^ 1) & 0xf;
    case 0x7: return (b ^ 2) & 0xf;
    case 0x3: return (b ^ 4) & 0xf;
    case 0x6: return (b ^ 8) & 0xf;
    default: return b & 0xf;
    }
}

/* LoRaCodes.hpp:318-323 */
static unsigned char check_parity54(const unsigned char b, int *error)
{
    int x = b ^ (b >> 2);
    x = x ^ (x >> 1) ^ (b >> 4);
    if (x & 1) *error = 1;
    return b & 0xf;
}

/* LoRaCodes.hpp:335-343 */
static unsigned char check_parity64(const unsigned char b, int *error)
{
    int x = b ^ (b >> 1) ^ (b >> 2);
    int y = x ^ b ^ (b >> 3);
    x ^= b >> 4;
    y ^= b >> 5;
    if ((x | y) & 1) *error = 1;
    return b & 0xf;
}

/* LoRaCodes.hpp:366-381 */
static void diagonal_deinterleave(const uint16_t *symbols, const size_t numSymbols, uint8_t *codewords, const size_t PPM, const size_t RDD)
{
    for (size_t x = 0; x < numSymbols / (4 + RDD); x++) {
        const size_t cwOff = x * PPM, symOff = x * (4 + RDD);
        for (size_t k = 0; k < 4 + RDD; k++)
            for (size_t m = 0; m < PPM; m++) {
                const size_t i = (m + k) % PPM;
                const int bit = (symbols[symOff + k] >> m) & 1;
                codewords[cwOff + i] |= (uint8_t)(bit << k);
            }
    }
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

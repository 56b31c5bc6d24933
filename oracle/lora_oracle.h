/* TEST INFRASTRUCTURE ONLY (oracle/) -- CPU restatement of the reference hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product path (lora_sdr_amd/) never includes, links or calls it.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * LoRa-SDR tree). Arithmetic is strict IEEE fp32/fp64 op-by-op in the reference's
 * order (built with -ffp-contract=off, no -march, no -ffast-math) so that results are
 * bit-identical to the reference compiled the same way; tests/test_oracle_vs_ref.py
 * pins that against oracle/_ref/libloraref.so (the real sources) and
 * tests/test_oracle_golden.py against the committed fixtures in tests/golden/.
 */
#ifndef LORA_ORACLE_H
#define LORA_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { float re, im; } lo_cf32;

/* ---- kissfft<float> for nfft = 2^k (kissfft.hh:17-52, 71-157) ---- */
typedef struct lo_fft lo_fft;
lo_fft *lo_fft_new(int nfft);
void lo_fft_free(lo_fft *f);
int lo_fft_size(const lo_fft *f);
int lo_fft_stages(const lo_fft *f, int *radix, int *remainder); /* returns count */
const lo_cf32 *lo_fft_twiddles(const lo_fft *f);
void lo_fft_transform(const lo_fft *f, const lo_cf32 *src, lo_cf32 *dst);

/* ---- LoRaDetector<float>::detect (LoRaDetector.hpp:29-64) ---- */
size_t lo_detect(const lo_fft *f, const lo_cf32 *in, lo_cf32 *fftOut,
                 float *power, float *powerAvg, float *fIndex);

/* ---- LoRaDemod tables (LoRaDemod.cpp:97-114) ---- */
void lo_demod_tables(int sf, lo_cf32 *up, lo_cf32 *down, lo_cf32 *fine /* N*128 */);

/* ---- dechirp loop (LoRaDemod.cpp:157-166 / :191-202); returns the advanced index ---- */
int lo_dechirp(int N, const lo_cf32 *in, const lo_cf32 *chirp, const lo_cf32 *fine,
               int fineIdx, float fineErr, lo_cf32 *dec);

/* ---- batch of independent windows: the semantics of the product C-ABI
 *      lorahip_detect_batch (include/lorahip.h), used as its checker ----
 * chirpSel: 0 = up table (conj), 1 = down table, 2 = none (input already dechirped,
 * i.e. the LoRaDetector::feed seam). fineIdx0/fineErr may be NULL (= 0). */
void lo_detect_batch(int sf, const lo_cf32 *iq, size_t nWindows, const int64_t *offsets,
                     const int32_t *chirpSel, const int32_t *fineIdx0, const float *fineErr,
                     uint16_t *sym, float *power, float *powerAvg, float *fIndex,
                     int32_t *fineIdxOut, lo_cf32 *fftOut, lo_cf32 *decOut, int nthreads);

/* ---- genChirp<float> (ChirpGenerator.hpp:22-47) ---- */
int lo_genchirp(lo_cf32 *samps, int N, int ovs, int NN, float f0, int down, float ampl,
                float *phaseAccum);

/* ---- one LoRaMod frame, ovs=1 (LoRaMod.cpp:109-238): 10 up-chirps, 2 sync chirps,
 *      2.25 down-chirps, data chirps, `padding` zero symbols. Returns samples written. ---- */
size_t lo_mod_frame(int sf, unsigned char sync, float ampl, size_t padding,
                    const uint16_t *syms, size_t nsyms, lo_cf32 *out, float *phaseAccum);
size_t lo_mod_frame_len(int sf, size_t padding, size_t nsyms);

/* ---- LoRaDemod block state machine (LoRaDemod.cpp:68-74, 139-143, 145-327) ---- */
enum { LO_FRAMESYNC = 0, LO_DOWNCHIRP0, LO_DOWNCHIRP1, LO_QUARTERCHIRP, LO_DATASYMBOLS };

typedef struct lo_demod lo_demod;
typedef struct {
    int64_t consumed;      /* inPort->consume(total)              */
    int32_t stateBefore;   /* state on entry                      */
    int32_t value;         /* detect() of window 0                */
    float power, powerAvg, snr, fIndex; /* as left at the end of work() (window 1 overwrites) */
    int32_t packetPosted;  /* 1 if a packet was posted this call  */
    int32_t packetLen;     /* symbols in it                       */
    int32_t signalsEmitted;/* 1 at DOWNCHIRP1                     */
    int32_t sigError; float sigPower, sigSnr;
    char label[48];        /* _id ("" = none)                     */
} lo_work_result;

lo_demod *lo_demod_new(int sf);
void lo_demod_free(lo_demod *d);
void lo_demod_set_sync(lo_demod *d, unsigned char sync);
void lo_demod_set_threshold(lo_demod *d, double thresh_dB);
void lo_demod_set_mtu(lo_demod *d, size_t mtu);
void lo_demod_activate(lo_demod *d);
/* one work() call; returns 0 if < 2N available (nothing done), else 1.
 * dec: 2N, fft: N, packet: mtu int16 (all optional / may be NULL). */
int lo_demod_work(lo_demod *d, const lo_cf32 *in, size_t avail, lo_work_result *res,
                  lo_cf32 *dec, lo_cf32 *fft, int16_t *packet);

/* CPU baseline: nStreams independent blocks over contiguous streams, nthreads workers;
 * every stream is processed `repeat` times (fresh block each time); returns the total number
 * of work() calls (= windows dechirped+FFT'd+scanned). */
int64_t lo_demod_bench(int sf, const lo_cf32 *iq, size_t samplesPerStream, int nStreams, int nthreads, int repeat);
/* Parity aid at scale: nStreams independent streams from the zero start state; per stream its call count, its packets (lengths,
 * posting call, symbols back to back) and optionally per call what it consumed and the kind of label it posted (0 none, 1 SYNC,
 * 2 P, 3 DC, 4 QC, 5 S<n>). Returns the total number of calls, -1 if a capacity was too small (that stream's nPackets = -1). */
int64_t lo_demod_run_many(int sf, const lo_cf32 *iq, size_t samplesPerStream, int nStreams, int nthreads,
                          int sync, double thresh, size_t mtu,
                          int32_t *nCalls, int32_t *nPackets, int16_t *pktSyms, size_t symCap,
                          int32_t *pktLens, int32_t *pktCall, size_t pktCap,
                          int32_t *callConsumed, uint8_t *callClass, size_t callCap);


/* ---- receive-side codec (oracle/lora_codec.c): the LoRaDecoder block, LoRaDecoder.cpp:196-397 ---- */
typedef struct lo_decoder_cfg {
    int sf;             /* setSpreadFactor  */
    int ppm;            /* setSymbolSize, 0 = sf */
    int rdd;            /* setCodingRate: "4/4".."4/8" -> 0..4 */
    int crcc;           /* enableCrcc */
    int interleaving;   /* enableInterleaving */
    int error_check;    /* enableErrorCheck */
    int explicit_hdr;   /* enableExplicit */
    int hdr;            /* enableHdr */
    int data_length;    /* setDataLength */
} lo_decoder_cfg;
int lo_code_primitive(int which, int b);   /* 0 hamming84, 1 hamming74, 2 parity54, 3 parity64, 4 header checksum (12 bits), 5 gray */
long lo_decode(const lo_decoder_cfg *c, const uint16_t *syms, size_t nsyms, void *out, int *dropped);

#ifdef __cplusplus
}
#endif
#endif

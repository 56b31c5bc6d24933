/* TEST INFRASTRUCTURE ONLY (oracle/) -- see lora_oracle.h for the rules.
 *
 * Plain-C restatement of the LoRa-SDR demod hot path: kissfft (power-of-two sizes),
 * LoRaDetector::detect, the LoRaDemod tables / dechirp recurrence / state machine,
 * genChirp and the LoRaMod frame layout. Each block cites the reference lines it
 * follows. Complex products are written out as gcc evaluates std::complex<float>
 * operator* for finite operands: (ac - bd, ad + bc), each product and sum rounded
 * separately (no FMA).
 */
#define _GNU_SOURCE
#include "lora_oracle.h"
#include <complex.h>
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

static inline lo_cf32 c_mul(lo_cf32 a, lo_cf32 b)
{
    lo_cf32 r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
static inline lo_cf32 c_add(lo_cf32 a, lo_cf32 b) { lo_cf32 r = { a.re + b.re, a.im + b.im }; return r; }
static inline lo_cf32 c_sub(lo_cf32 a, lo_cf32 b) { lo_cf32 r = { a.re - b.re, a.im - b.im }; return r; }

/***********************************************************************
 * kissfft<float>, forward, nfft = 2^k
 **********************************************************************/
#define LO_MAX_STAGES 32
struct lo_fft {
    int nfft;
    int nstages;
    int radix[LO_MAX_STAGES];
    int remainder[LO_MAX_STAGES];
    lo_cf32 *tw;
};

lo_fft *lo_fft_new(int nfft)
{
    lo_fft *f = (lo_fft *)calloc(1, sizeof(lo_fft));
    f->nfft = nfft;
    f->tw = (lo_cf32 *)malloc(sizeof(lo_cf32) * (size_t)nfft);
    /* kissfft.hh:17-22  fill_twiddles: phinc = -2*acos((float)-1)/nfft, all in float;
     * dst[i] = exp(complex<float>(0, i*phinc)) -> cexpf */
    const float phinc = -2 * acosf((float)-1) / nfft;
    for (int i = 0; i < nfft; ++i) {
        const float complex e = cexpf(CMPLXF(0.0f, i * phinc));
        f->tw[i].re = crealf(e);
        f->tw[i].im = cimagf(e);
    }
    /* kissfft.hh:34-51  factorise: 4's, then 2's, then odd */
    int n = nfft, p = 4;
    do {
        while (n % p) {
            switch (p) {
            case 4: p = 2; break;
            case 2: p = 3; break;
            default: p += 2; break;
            }
            if (p * p > n) p = n;
        }
        n /= p;
        f->radix[f->nstages] = p;
        f->remainder[f->nstages] = n;
        f->nstages++;
    } while (n > 1);
    return f;
}

void lo_fft_free(lo_fft *f) { if (f) { free(f->tw); free(f); } }
int lo_fft_size(const lo_fft *f) { return f->nfft; }
const lo_cf32 *lo_fft_twiddles(const lo_fft *f) { return f->tw; }
int lo_fft_stages(const lo_fft *f, int *radix, int *remainder)
{
    for (int i = 0; i < f->nstages; i++) { if (radix) radix[i] = f->radix[i]; if (remainder) remainder[i] = f->remainder[i]; }
    return f->nstages;
}

/* kissfft.hh:128-135 */
static void kf_bfly2(const lo_fft *f, lo_cf32 *Fout, size_t fstride, int m)
{
    for (int k = 0; k < m; ++k) {
        const lo_cf32 t = c_mul(Fout[m + k], f->tw[k * fstride]);
        Fout[m + k] = c_sub(Fout[k], t);
        Fout[k] = c_add(Fout[k], t);
    }
}

/* kissfft.hh:137-157 (forward: negative_if_inverse = +1) */
static void kf_bfly4(const lo_fft *f, lo_cf32 *Fout, size_t fstride, size_t m)
{
    lo_cf32 s0, s1, s2, s3, s4, s5;
    for (size_t k = 0; k < m; ++k) {
        s0 = c_mul(Fout[k + m], f->tw[k * fstride]);
        s1 = c_mul(Fout[k + 2 * m], f->tw[k * fstride * 2]);
        s2 = c_mul(Fout[k + 3 * m], f->tw[k * fstride * 3]);
        s5 = c_sub(Fout[k], s1);

        Fout[k] = c_add(Fout[k], s1);
        s3 = c_add(s0, s2);
        s4 = c_sub(s0, s2);
        { const lo_cf32 r = { s4.im * 1, -s4.re * 1 }; s4 = r; }

        Fout[k + 2 * m] = c_sub(Fout[k], s3);
        Fout[k] = c_add(Fout[k], s3);
        Fout[k + m] = c_add(s5, s4);
        Fout[k + 3 * m] = c_sub(s5, s4);
    }
}

/* kissfft.hh:83-116 */
static void kf_work(const lo_fft *ff, int stage, lo_cf32 *Fout, const lo_cf32 *f, size_t fstride)
{
    const int p = ff->radix[stage];
    const int m = ff->remainder[stage];
    lo_cf32 *const Fout_beg = Fout;
    lo_cf32 *const Fout_end = Fout + p * m;
    if (m == 1) {
        do { *Fout = *f; f += fstride; } while (++Fout != Fout_end);
    } else {
        do { kf_work(ff, stage + 1, Fout, f, fstride * p); f += fstride; } while ((Fout += m) != Fout_end);
    }
    Fout = Fout_beg;
    if (p == 2) kf_bfly2(ff, Fout, fstride, m);
    else if (p == 4) kf_bfly4(ff, Fout, fstride, (size_t)m);
    else abort(); /* unreachable for nfft = 2^k (radix 3/5/generic: kissfft.hh:159-299) */
}

void lo_fft_transform(const lo_fft *f, const lo_cf32 *src, lo_cf32 *dst)
{
    if (f->nfft == 1) { dst[0] = src[0]; return; }
    kf_work(f, 0, dst, src, 1); /* kissfft.hh:77-80 */
}

/***********************************************************************
 * LoRaDetector<float>::detect  (LoRaDetector.hpp:29-64)
 **********************************************************************/
size_t lo_detect(const lo_fft *f, const lo_cf32 *in, lo_cf32 *fftOut,
                 float *power, float *powerAvg, float *fIndex)
{
    const size_t N = (size_t)f->nfft;
    const float powerScale = (float)(20 * log10((double)N)); /* :18  double -> float member */
    lo_fft_transform(f, in, fftOut);                          /* :32 */
    size_t maxIndex = 0;
    float maxValue = 0;
    double total = 0;
    for (size_t i = 0; i < N; i++) {                          /* :36-48 */
        const float re = fftOut[i].re, im = fftOut[i].im;
        const float mag2 = re * re + im * im;
        total += mag2;
        if (mag2 > maxValue) { maxIndex = i; maxValue = mag2; }
    }
    const float noise = sqrtf((float)(total - maxValue));     /* :50 */
    const float fundamental = sqrtf(maxValue);                /* :51 */
    *powerAvg = 20 * log10f(noise) - powerScale;              /* :53 */
    *power = 20 * log10f(fundamental) - powerScale;           /* :54 */
    const lo_cf32 l = fftOut[maxIndex > 0 ? maxIndex - 1 : N - 1];
    const lo_cf32 r = fftOut[maxIndex < N - 1 ? maxIndex + 1 : 0];
    const float left = cabsf(CMPLXF(l.re, l.im));             /* :56 std::abs -> cabsf */
    const float right = cabsf(CMPLXF(r.re, r.im));            /* :57 */
    const double demon = (2.0 * fundamental) - right - left;  /* :59 */
    if (demon == 0.0) *fIndex = 0.0f;                         /* :60 */
    else *fIndex = (float)(0.5 * (right - left) / demon);     /* :61 */
    return maxIndex;
}

/***********************************************************************
 * LoRaDemod tables (LoRaDemod.cpp:97-114)
 **********************************************************************/
void lo_demod_tables(int sf, lo_cf32 *up, lo_cf32 *down, lo_cf32 *fine)
{
    const size_t N = (size_t)1 << sf;
    const size_t fineSteps = 128;
    float phase = -M_PI;
    double phaseAccum = 0.0;
    for (size_t i = 0; i < N; i++) {
        phaseAccum += phase;
        const double er = 1.0 * cos(phaseAccum), ei = 1.0 * sin(phaseAccum); /* std::polar(1.0, a) */
        if (up) { up[i].re = (float)er; up[i].im = (float)(-ei); }          /* conj */
        if (down) { down[i].re = (float)er; down[i].im = (float)ei; }
        phase += (2 * M_PI) / N;
    }
    if (!fine) return;
    phaseAccum = 0.0;
    phase = 2.0 * M_PI / (N * fineSteps);
    for (size_t i = 0; i < N * fineSteps; i++) {
        phaseAccum += phase;
        fine[i].re = (float)(1.0 * cos(phaseAccum));
        fine[i].im = (float)(1.0 * sin(phaseAccum));
    }
}

/***********************************************************************
 * dechirp loop (LoRaDemod.cpp:157-166; the window-1 copy at :191-202 is identical
 * but works on a temporary index)
 **********************************************************************/
int lo_dechirp(int N, const lo_cf32 *in, const lo_cf32 *chirp, const lo_cf32 *fine,
               int fineIdx, float fineErr, lo_cf32 *dec)
{
    const size_t fineSteps = 128;
    const int M = (int)((size_t)N * fineSteps);
    for (int i = 0; i < N; i++) {
        lo_cf32 d = in[i];
        if (chirp) d = c_mul(d, chirp[i]);
        if (fine) d = c_mul(d, fine[fineIdx]);
        /* _fineTuneIndex -= _finefreqError * _fineSteps;  int -= float*float(size_t) */
        fineIdx = (int)((float)fineIdx - fineErr * (float)fineSteps);
        if (fineIdx < 0) fineIdx += M;
        else if (fineIdx >= M) fineIdx -= M;
        dec[i] = d;
    }
    return fineIdx;
}

/***********************************************************************
 * batch of independent windows (checker for the product C-ABI)
 **********************************************************************/
typedef struct {
    int sf; const lo_cf32 *iq; const int64_t *offsets; const int32_t *chirpSel;
    const int32_t *fineIdx0; const float *fineErr;
    uint16_t *sym; float *power, *powerAvg, *fIndex; int32_t *fineIdxOut;
    lo_cf32 *fftOut, *decOut;
    const lo_cf32 *up, *down, *fine;
    size_t lo, hi;
} batch_job;

static void *batch_body(void *arg)
{
    batch_job *j = (batch_job *)arg;
    const size_t N = (size_t)1 << j->sf;
    lo_fft *f = lo_fft_new((int)N);
    lo_cf32 *dec = (lo_cf32 *)malloc(sizeof(lo_cf32) * N);
    lo_cf32 *fft = (lo_cf32 *)malloc(sizeof(lo_cf32) * N);
    for (size_t w = j->lo; w < j->hi; w++) {
        const lo_cf32 *in = j->iq + (j->offsets ? (size_t)j->offsets[w] : w * N);
        const int sel = j->chirpSel ? j->chirpSel[w] : 0;
        const lo_cf32 *chirp = sel == 0 ? j->up : (sel == 1 ? j->down : NULL);
        const int idx0 = j->fineIdx0 ? j->fineIdx0[w] : 0;
        /* LORAHIP_CHIRP_NONE (the LoRaDetector::feed seam) has no dechirp loop at all: the index does not move */
        const float e = (chirp && j->fineErr) ? j->fineErr[w] : 0.0f;
        const int idx1 = lo_dechirp((int)N, in, chirp, chirp ? j->fine : NULL, idx0, e, dec);
        if (j->fineIdxOut) j->fineIdxOut[w] = idx1;
        if (j->decOut) memcpy(j->decOut + w * N, dec, sizeof(lo_cf32) * N);
        lo_cf32 *out = j->fftOut ? j->fftOut + w * N : fft;
        j->sym[w] = (uint16_t)lo_detect(f, dec, out, &j->power[w], &j->powerAvg[w], &j->fIndex[w]);
    }
    free(dec); free(fft); lo_fft_free(f);
    return NULL;
}

void lo_detect_batch(int sf, const lo_cf32 *iq, size_t nWindows, const int64_t *offsets,
                     const int32_t *chirpSel, const int32_t *fineIdx0, const float *fineErr,
                     uint16_t *sym, float *power, float *powerAvg, float *fIndex,
                     int32_t *fineIdxOut, lo_cf32 *fftOut, lo_cf32 *decOut, int nthreads)
{
    const size_t N = (size_t)1 << sf;
    lo_cf32 *up = (lo_cf32 *)malloc(sizeof(lo_cf32) * N);
    lo_cf32 *down = (lo_cf32 *)malloc(sizeof(lo_cf32) * N);
    lo_cf32 *fine = (lo_cf32 *)malloc(sizeof(lo_cf32) * N * 128);
    lo_demod_tables(sf, up, down, fine);
    const int T = nthreads > 1 ? nthreads : 1;
    batch_job *jobs = (batch_job *)calloc((size_t)T, sizeof(batch_job));
    pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
    for (int t = 0; t < T; t++) {
        batch_job j = { sf, iq, offsets, chirpSel, fineIdx0, fineErr, sym, power, powerAvg, fIndex,
                        fineIdxOut, fftOut, decOut, up, down, fine,
                        nWindows * (size_t)t / (size_t)T, nWindows * (size_t)(t + 1) / (size_t)T };
        jobs[t] = j;
        if (T == 1) batch_body(&jobs[t]);
        else pthread_create(&th[t], NULL, batch_body, &jobs[t]);
    }
    if (T > 1) for (int t = 0; t < T; t++) pthread_join(th[t], NULL);
    free(jobs); free(th); free(up); free(down); free(fine);
}

/***********************************************************************
 * genChirp<float>  (ChirpGenerator.hpp:22-47)
 **********************************************************************/
int lo_genchirp(lo_cf32 *samps, int N, int ovs, int NN, float f0, int down, float ampl,
                float *phaseAccumIO)
{
    float phaseAccum = *phaseAccumIO;
    const float fMin = -M_PI / ovs;
    const float fMax = M_PI / ovs;
    const float fStep = (2 * M_PI) / (N * ovs * ovs);
    float f = fMin + f0;
    int i;
    for (i = 0; i < NN; i++) {
        f += fStep;
        if (f > fMax) f -= (fMax - fMin);
        if (down) phaseAccum -= f; else phaseAccum += f;
        /* std::polar(float rho, float theta) = (rho*cosf, rho*sinf) */
        samps[i].re = ampl * cosf(phaseAccum);
        samps[i].im = ampl * sinf(phaseAccum);
    }
    phaseAccum -= floor(phaseAccum / (2 * M_PI)) * 2 * M_PI;
    *phaseAccumIO = phaseAccum;
    return i;
}

/***********************************************************************
 * LoRaMod frame, ovs = 1  (LoRaMod.cpp:109-238)
 **********************************************************************/
size_t lo_mod_frame_len(int sf, size_t padding, size_t nsyms)
{
    const size_t N = (size_t)1 << sf;
    /* STATE_PADSYMBOLS emits one zero symbol BEFORE it tests `_counter >= _padding` (LoRaMod.cpp:218-224):
     * padding 0 still produces one */
    if (padding == 0) padding = 1;
    return N * (10 + 2 + 2 + nsyms + padding) + N / 4;
}

size_t lo_mod_frame(int sf, unsigned char sync, float ampl, size_t padding,
                    const uint16_t *syms, size_t nsyms, lo_cf32 *out, float *phaseAccum)
{
    const int N = 1 << sf;
    const int NN = N; /* ovs = 1 */
    size_t n = 0;
    *phaseAccum = 0; /* STATE_WAITINPUT: _phaseAccum = 0  (LoRaMod.cpp:135) */
    for (int c = 0; c < 10; c++) n += (size_t)lo_genchirp(out + n, N, 1, NN, 0.0f, 0, ampl, phaseAccum); /* :141-147 */
    { const int sw0 = (sync >> 4) * 8; const float freq = (2 * M_PI * sw0) / NN;                           /* :150-158 */
      n += (size_t)lo_genchirp(out + n, N, 1, NN, freq, 0, ampl, phaseAccum); }
    { const int sw1 = (sync & 0xf) * 8; const float freq = (2 * M_PI * sw1) / NN;                          /* :161-169 */
      n += (size_t)lo_genchirp(out + n, N, 1, NN, freq, 0, ampl, phaseAccum); }
    n += (size_t)lo_genchirp(out + n, N, 1, NN, 0.0f, 1, ampl, phaseAccum);                                /* :172-178 */
    n += (size_t)lo_genchirp(out + n, N, 1, NN, 0.0f, 1, ampl, phaseAccum);                                /* :181-187 */
    n += (size_t)lo_genchirp(out + n, N, 1, NN / 4, 0.0f, 1, ampl, phaseAccum);                            /* :190-197 */
    for (size_t s = 0; s < nsyms; s++) {                                                                   /* :200-215 */
        const int sym = syms[s];
        const float freq = (2 * M_PI * sym) / NN;
        n += (size_t)lo_genchirp(out + n, N, 1, NN, freq, 0, ampl, phaseAccum);
    }
    if (padding == 0) padding = 1;                                                                         /* :220-222: emit, then test */
    for (size_t p = 0; p < padding; p++)                                                                   /* :218-229 */
        for (int i = 0; i < NN; i++) { out[n].re = 0.0f; out[n].im = 0.0f; n++; }
    return n;
}

/***********************************************************************
 * LoRaDemod block  (LoRaDemod.cpp:68-74, 124-143, 145-327)
 **********************************************************************/
struct lo_demod {
    size_t N;
    int sf;
    lo_fft *fft;
    lo_cf32 *up, *down, *fine, *det, *detOut;
    const lo_cf32 *chirpTable;
    unsigned char sync;
    float thresh;
    size_t mtu;
    int state;
    size_t symCount;
    int16_t *outSymbols;
    short prevValue;      /* zero-initialised here; indeterminate in the reference */
    int freqError;        /* idem */
    int fineTuneIndex;
    float finefreqError;  /* idem */
};

lo_demod *lo_demod_new(int sf)
{
    lo_demod *d = (lo_demod *)calloc(1, sizeof(lo_demod));
    d->sf = sf;
    d->N = (size_t)1 << sf;
    d->fft = lo_fft_new((int)d->N);
    d->up = (lo_cf32 *)malloc(sizeof(lo_cf32) * d->N);
    d->down = (lo_cf32 *)malloc(sizeof(lo_cf32) * d->N);
    d->fine = (lo_cf32 *)malloc(sizeof(lo_cf32) * d->N * 128);
    d->det = (lo_cf32 *)malloc(sizeof(lo_cf32) * d->N);
    d->detOut = (lo_cf32 *)malloc(sizeof(lo_cf32) * d->N);
    lo_demod_tables(sf, d->up, d->down, d->fine);
    d->sync = 0x12; d->thresh = -30.0f; d->mtu = 256;   /* :71-73 */
    d->fineTuneIndex = 0;                               /* :116 */
    d->outSymbols = NULL;
    lo_demod_activate(d);
    return d;
}

void lo_demod_free(lo_demod *d)
{
    if (!d) return;
    lo_fft_free(d->fft);
    free(d->up); free(d->down); free(d->fine); free(d->det); free(d->detOut); free(d->outSymbols);
    free(d);
}

void lo_demod_set_sync(lo_demod *d, unsigned char sync) { d->sync = sync; }                 /* :124-127 */
void lo_demod_set_threshold(lo_demod *d, double t) { d->thresh = (float)t; }                /* :129-132 */
void lo_demod_set_mtu(lo_demod *d, size_t mtu) { d->mtu = mtu; }                            /* :134-137 */
void lo_demod_activate(lo_demod *d) { d->state = LO_FRAMESYNC; d->chirpTable = d->up; }     /* :139-143 */

int lo_demod_work(lo_demod *d, const lo_cf32 *in, size_t avail, lo_work_result *res,
                  lo_cf32 *dec, lo_cf32 *fft, int16_t *packet)
{
    const size_t N = d->N;
    if (avail < N * 2) return 0;                                                            /* :148 */
    size_t total = 0;
    lo_cf32 *fftBuff = fft ? fft : d->detOut;
    char id[48]; id[0] = 0;
    memset(res, 0, sizeof(*res));
    res->stateBefore = d->state;

    /* :157-166 */
    d->fineTuneIndex = lo_dechirp((int)N, in, d->chirpTable, d->fine, d->fineTuneIndex, d->finefreqError, d->det);
    if (dec) memcpy(dec, d->det, sizeof(lo_cf32) * N);
    float power = 0, powerAvg = 0, snr = 0, fIndex = 0;
    const size_t value = lo_detect(d->fft, d->det, fftBuff, &power, &powerAvg, &fIndex);    /* :172 */
    snr = power - powerAvg;
    const int squelched = (snr < d->thresh);                                                /* :174 */

    switch (d->state) {
    case LO_FRAMESYNC: {                                                                    /* :179-234 */
        const int syncd = !squelched && (d->prevValue + 4) / 8 == 0;
        const int match0 = (value + 4) / 8 == (unsigned)(d->sync >> 4);
        int match1 = 0;
        if (syncd && match0) {
            /* :189-206: window 1 on a temporary index; detect() overwrites power/powerAvg/fIndex */
            (void)lo_dechirp((int)N, in + N, d->chirpTable, d->fine, d->fineTuneIndex, d->finefreqError, d->det);
            if (dec) memcpy(dec + N, d->det, sizeof(lo_cf32) * N);
            /* detect() without fftOutput: goes to the detector's internal buffer */
            const size_t value1 = lo_detect(d->fft, d->det, d->detOut, &power, &powerAvg, &fIndex);
            match1 = (value1 + 4) / 8 == (unsigned)(d->sync & 0xf);
        }
        if (syncd && match0 && match1) {
            total = 2 * N;
            d->state = LO_DOWNCHIRP0;
            d->chirpTable = d->down;
            snprintf(id, sizeof(id), "SYNC");
        } else if (!squelched) {
            total = N - value;
            d->finefreqError += fIndex;
            snprintf(id, sizeof(id), "P %.4f", (double)fIndex);
        } else {
            total = N;
            d->finefreqError = 0;
            d->fineTuneIndex = 0;
        }
    } break;
    case LO_DOWNCHIRP0: {                                                                   /* :239-248 */
        d->state = LO_DOWNCHIRP1;
        total = N;
        snprintf(id, sizeof(id), "DC");
        int error = (int)value;
        if (value > N / 2) error -= (int)N;
        d->freqError = error;
    } break;
    case LO_DOWNCHIRP1: {                                                                   /* :253-270 */
        d->state = LO_QUARTERCHIRP;
        total = N;
        d->chirpTable = d->up;
        free(d->outSymbols);
        d->outSymbols = (int16_t *)calloc(d->mtu ? d->mtu : 1, sizeof(int16_t));
        int error = (int)value;
        if (value > N / 2) error -= (int)N;
        d->freqError = (d->freqError + error) / 2;
        res->signalsEmitted = 1;
        res->sigError = d->freqError; res->sigPower = power; res->sigSnr = snr;
    } break;
    case LO_QUARTERCHIRP: {                                                                 /* :275-283 */
        d->state = LO_DATASYMBOLS;
        total = N / 4 + (size_t)(d->freqError / 2);
        d->finefreqError += (d->freqError / 2);
        d->symCount = 0;
        snprintf(id, sizeof(id), "QC");
    } break;
    case LO_DATASYMBOLS: {                                                                  /* :288-311 */
        total = N;
        d->outSymbols[d->symCount++] = (int16_t)value;
        if (d->symCount >= d->mtu || squelched) {
            res->packetPosted = 1;
            res->packetLen = (int32_t)d->symCount;
            if (packet) memcpy(packet, d->outSymbols, d->symCount * sizeof(int16_t));
            d->finefreqError = 0;
            d->state = LO_FRAMESYNC;
        }
        snprintf(id, sizeof(id), "S%zu %.4f", d->symCount, (double)fIndex);
    } break;
    }

    d->prevValue = (short)value;                                                            /* :326 */
    res->consumed = (int64_t)total;
    res->value = (int32_t)value;
    res->power = power; res->powerAvg = powerAvg; res->snr = snr; res->fIndex = fIndex;
    memcpy(res->label, id, sizeof(id));
    return 1;
}

/***********************************************************************
 * CPU baseline driver
 **********************************************************************/
typedef struct { int sf; const lo_cf32 *iq; size_t sps; int lo, hi; int repeat; int64_t calls; } bench_job;

static void *bench_body(void *arg)
{
    bench_job *j = (bench_job *)arg;
    const size_t N = (size_t)1 << j->sf;
    lo_cf32 *dec = (lo_cf32 *)malloc(sizeof(lo_cf32) * 2 * N);
    lo_cf32 *fft = (lo_cf32 *)malloc(sizeof(lo_cf32) * N);
    /* one block per worker (its constructor builds the 128*N fine table), re-activated per stream */
    lo_demod *d = lo_demod_new(j->sf);
    for (int rep = 0; rep < (j->repeat > 1 ? j->repeat : 1); rep++)
    for (int s = j->lo; s < j->hi; s++) {
        lo_demod_activate(d);
        /* same start state for every stream (the reference only resets these in its noise branch) */
        d->prevValue = 0; d->freqError = 0; d->fineTuneIndex = 0; d->finefreqError = 0; d->symCount = 0;
        const lo_cf32 *in = j->iq + (size_t)s * j->sps;
        size_t pos = 0;
        lo_work_result r;
        /* like the block, also fill the dec/fft debug buffers (LoRaDemod.cpp:163-164) */
        while (lo_demod_work(d, in + pos, j->sps - pos, &r, dec, fft, NULL)) { pos += (size_t)r.consumed; j->calls++; if (!r.consumed) break; }
    }
    lo_demod_free(d);
    free(dec); free(fft);
    return NULL;
}

int64_t lo_demod_bench(int sf, const lo_cf32 *iq, size_t samplesPerStream, int nStreams, int nthreads, int repeat)
{
    const int T = nthreads > 1 ? nthreads : 1;
    bench_job *jobs = (bench_job *)calloc((size_t)T, sizeof(bench_job));
    pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
    for (int t = 0; t < T; t++) {
        bench_job j = { sf, iq, samplesPerStream, nStreams * t / T, nStreams * (t + 1) / T, repeat, 0 };
        jobs[t] = j;
        if (T == 1) bench_body(&jobs[t]); else pthread_create(&th[t], NULL, bench_body, &jobs[t]);
    }
    int64_t total = 0;
    for (int t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); total += jobs[t].calls; }
    free(jobs); free(th);
    return total;
}

/***********************************************************************
 * Parity aid at scale (the restatement's twin of oracle/ref_driver.cpp::loraref_demod_run_many, same argument meaning): nStreams
 * independent streams, every one from the zero start state, over nthreads. callClass: 0 no label, 1 "SYNC", 2 "P ..", 3 "DC",
 * 4 "QC", 5 "S<n> .." (LoRaDemod.cpp:213,220-224,245,282,302-305).
 **********************************************************************/
typedef struct {
    int sf; const lo_cf32 *iq; size_t sps; int lo, hi;
    int sync; double thresh; size_t mtu;
    int32_t *nCalls, *nPackets; int16_t *pktSyms; size_t symCap; int32_t *pktLens, *pktCall; size_t pktCap;
    int32_t *callConsumed; uint8_t *callClass; size_t callCap;
    int bad;
} many_job;

static void *many_body(void *arg)
{
    many_job *j = (many_job *)arg;
    lo_demod *d = lo_demod_new(j->sf);
    lo_demod_set_sync(d, (unsigned char)j->sync);
    lo_demod_set_threshold(d, j->thresh);
    lo_demod_set_mtu(d, j->mtu);
    int16_t *pkt = (int16_t *)calloc(j->mtu ? j->mtu : 1, sizeof(int16_t));
    for (int s = j->lo; s < j->hi; s++) {
        lo_demod_activate(d);
        d->prevValue = 0; d->freqError = 0; d->fineTuneIndex = 0; d->finefreqError = 0; d->symCount = 0;
        const lo_cf32 *in = j->iq + (size_t)s * j->sps;
        size_t pos = 0, at = 0, np = 0, calls = 0;
        int ok = 1;
        lo_work_result r;
        while (lo_demod_work(d, in + pos, j->sps - pos, &r, NULL, NULL, pkt)) {
            if (j->callConsumed || j->callClass) {
                if (calls >= j->callCap) ok = 0;
                else {
                    if (j->callConsumed) j->callConsumed[(size_t)s * j->callCap + calls] = (int32_t)r.consumed;
                    if (j->callClass) {
                        uint8_t c = 0;
                        if (r.label[0] == 'S' && r.label[1] == 'Y') c = 1;
                        else if (r.label[0] == 'P') c = 2;
                        else if (r.label[0] == 'D') c = 3;
                        else if (r.label[0] == 'Q') c = 4;
                        else if (r.label[0] == 'S') c = 5;
                        j->callClass[(size_t)s * j->callCap + calls] = c;
                    }
                }
            }
            if (r.packetPosted) {
                if (np >= j->pktCap || at + (size_t)r.packetLen > j->symCap) ok = 0;
                else {
                    memcpy(j->pktSyms + (size_t)s * j->symCap + at, pkt, (size_t)r.packetLen * sizeof(int16_t));
                    j->pktLens[(size_t)s * j->pktCap + np] = r.packetLen;
                    if (j->pktCall) j->pktCall[(size_t)s * j->pktCap + np] = (int32_t)calls;
                    at += (size_t)r.packetLen;
                }
                np++;
            }
            calls++;
            pos += (size_t)r.consumed;
            if (!r.consumed) break;
        }
        j->nCalls[s] = (int32_t)calls;
        j->nPackets[s] = ok ? (int32_t)np : -1;
        if (!ok) j->bad = 1;
    }
    free(pkt);
    lo_demod_free(d);
    return NULL;
}

int64_t lo_demod_run_many(int sf, const lo_cf32 *iq, size_t samplesPerStream, int nStreams, int nthreads,
                          int sync, double thresh, size_t mtu,
                          int32_t *nCalls, int32_t *nPackets, int16_t *pktSyms, size_t symCap,
                          int32_t *pktLens, int32_t *pktCall, size_t pktCap,
                          int32_t *callConsumed, uint8_t *callClass, size_t callCap)
{
    const int T = nthreads > 1 ? nthreads : 1;
    many_job *jobs = (many_job *)calloc((size_t)T, sizeof(many_job));
    pthread_t *th = (pthread_t *)calloc((size_t)T, sizeof(pthread_t));
    for (int t = 0; t < T; t++) {
        many_job j = { sf, iq, samplesPerStream, nStreams * t / T, nStreams * (t + 1) / T, sync, thresh, mtu,
                       nCalls, nPackets, pktSyms, symCap, pktLens, pktCall, pktCap, callConsumed, callClass, callCap, 0 };
        jobs[t] = j;
        if (T == 1) many_body(&jobs[t]); else pthread_create(&th[t], NULL, many_body, &jobs[t]);
    }
    int64_t total = 0;
    int bad = 0;
    for (int t = 0; t < T; t++) { if (T > 1) pthread_join(th[t], NULL); bad |= jobs[t].bad; }
    for (int s = 0; s < nStreams; s++) total += nCalls[s];
    free(jobs); free(th);
    return bad ? -1 : total;
}

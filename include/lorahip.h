/* lorahip.h -- C ABI of the MI355X-native LoRa demodulation hot path.
 *
 * Drop-in boundary for the part of myriadrf/LoRa-SDR that BASELINE.json's north_star
 * names: dechirp multiply -> 2^SF-point complex FFT -> |.|^2 arg-max / power / fractional
 * bin. Everything here is plain pointers and sizes; no exceptions cross the boundary,
 * every entry point returns LORAHIP_OK (0) or a negative LORAHIP_E_* code.
 * File:line citations are relative to the LoRa-SDR tree.
 *
 * Three levels, from the inner seam outwards:
 *
 *   1. lorahip_detector_*   one window, host buffers: exactly the semantics of
 *      `template<T> class LoRaDetector` (LoRaDetector.hpp:8-72) so LoRaDemod.cpp can
 *      swap `LoRaDetector<float> _detector` for a thin wrapper.
 *
 *   2. lorahip_detect_batch   many independent symbol windows per call (the data-parallel
 *      axis: channels x windows), device pointers, asynchronous on the context's stream.
 *      Per window it performs LoRaDemod.cpp:157-166 (dechirp with the chirp table and the
 *      fine-tune table, including the int<-float index recurrence) followed by
 *      LoRaDetector::detect (LoRaDetector.hpp:29-64 on kissfft.hh:77-157).
 *
 *   3. lorahip_demod_*   B channels of the `LoRaDemod` block (LoRaDemod.cpp:68-143 setters,
 *      :145-327 work()): same parameters (sf, sync, thresh, mtu), same 5-state frame
 *      machine, same int16 symbol packets; driven by a streaming kernel that walks every
 *      channel's stream on the device (SF6-12), or in lock-step from the host with one batch
 *      launch per work() round (lorahip_demod_set_mode).
 *
 * Around the path (each behind its own entry points further down): the batched modulator + AWGN channel
 * (lorahip_mod_frames, lorahip_add_awgn), the batched decoder (lorahip_decode_packets) with the packet hand-off
 * lorahip_demod_packets_to_device, and the front-end channeliser (lorahip_channelizer_*).
 *
 * Results: symbol indices and FFT bins are bit-identical to the reference CPU path
 * compiled without FMA contraction (the kernels evaluate kissfft's radix-4/2 DIT graph
 * with kissfft's own float twiddles, op for op); power / powerAvg / fIndex agree to
 * float rounding of log10/hypot (tolerances in tests/).
 */
#ifndef LORAHIP_H
#define LORAHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LORAHIP_OK             0
#define LORAHIP_E_INVALID     -1  /* bad argument (NULL, sf out of range, size mismatch)   */
#define LORAHIP_E_NODEVICE    -2  /* no HIP device / device index out of range             */
#define LORAHIP_E_HIP         -3  /* a HIP runtime call failed; see lorahip_last_error()    */
#define LORAHIP_E_NOMEM       -4  /* host or device allocation failed                       */
#define LORAHIP_E_ARCH        -5  /* device is not gfx950 (kernels are built for it only)   */

#define LORAHIP_SF_MIN  6
#define LORAHIP_SF_MAX 12
#define LORAHIP_FINE_STEPS 128     /* LoRaDemod.cpp:69 _fineSteps */

/* chirp selection per window */
#define LORAHIP_CHIRP_UP    0      /* _upChirpTable  = conj(entry)  LoRaDemod.cpp:103 (FRAMESYNC, DATASYMBOLS) */
#define LORAHIP_CHIRP_DOWN  1      /* _downChirpTable = entry       LoRaDemod.cpp:104 (DOWNCHIRP0/1)           */
#define LORAHIP_CHIRP_NONE  2      /* input is already dechirped: the LoRaDetector::feed seam (no dechirp loop:
                                      fine_idx0 / fine_err are ignored, fine_idx_out = fine_idx0)               */

const char *lorahip_strerror(int code);
const char *lorahip_last_error(void);       /* thread-local text of the last LORAHIP_E_HIP */
int lorahip_version(void);                  /* ABI version, currently 4 (1 -> 2: lorahip_work_result grew, level-3 ports and labels;
                                               2 -> 3, additions only: level-3 signals, append runs / lorahip_demod_receive, lorahip_rx_*;
                                               3 -> 4, additions only: lorahip_demod_receive_flush, _run_host_rows, _stream_wait / _stream_follow,
                                               _set_stream_lanes / _stream_lanes, _set_stream_grid, _set_variant, _set_record_capacity,
                                               lorahip_decode_packets_host, lorahip_decode_max_symbols, lorahip_decode_max_data_length, lorahip_demod_receive_signal_rows /
                                               _receive_num_signals: signals in receiver steps, pipelined ones included; async = 3: the resident receiver,
                                               lorahip_demod_resident_active, lorahip_demod_receive_steps) */
int lorahip_device_count(void);             /* number of usable gfx950 devices, 0 if none */
int lorahip_selfcheck(void);                /* host-only: the kernels' compile-time LDS layouts are consistent; no device needed */

/* -------------------------------------------------------------------------------------
 * Host-side tables, exactly the reference's expressions (no device needed):
 *   up/down : N cf32       LoRaDemod.cpp:97-107
 *   fine    : 128*N cf32   LoRaDemod.cpp:108-114
 *   twiddle : N cf32       kissfft.hh:17-22
 * Any pointer may be NULL. Buffers are interleaved (re,im) floats.
 * ------------------------------------------------------------------------------------- */
int lorahip_host_tables(int sf, float *up, float *down, float *fine, float *twiddle);

/* -------------------------------------------------------------------------------------
 * Level 2: batch context. One per (device, SF); not thread-safe (one host thread per
 * context, like one Pothos actor per block). Replaces: LoRaDemod ctor tables
 * (LoRaDemod.cpp:97-116) + LoRaDetector ctor (LoRaDetector.hpp:12-20) + kissfft ctor
 * (kissfft.hh:71-75).
 * ------------------------------------------------------------------------------------- */
typedef struct lorahip_ctx lorahip_ctx;

int lorahip_create(lorahip_ctx **ctx, int device, int sf);
void lorahip_destroy(lorahip_ctx *ctx);
int lorahip_sf(const lorahip_ctx *ctx);
/* Launch on an existing hipStream_t (e.g. torch's current stream) from now on. NULL is HIP's
 * null stream. Until this is called the context uses a private non-blocking stream. */
int lorahip_set_stream(lorahip_ctx *ctx, void *hip_stream);
int lorahip_reset_stream(lorahip_ctx *ctx);          /* back to the private stream */
int lorahip_synchronize(lorahip_ctx *ctx);

/* Kernel variant: 0 = auto (fastest validated for this SF), 1 = generic LDS kernel, 10 = the tuned kernel with every table in LDS.
 * These produce identical symbol indices and FFT bins (the reference's operation graph, no FMA).
 * LORAHIP_VARIANT_FMA (opt-in, level 2 only) is NOT one of them: the same tuned kernels with every complex multiply contracted to
 * one multiply + one FMA. Its bins differ from the CPU build's in the last place or two (far inside the 1e-4 relative tolerance
 * north_star states, symbol indices identical wherever the peak's margin exceeds that), and it is ~20 % fewer vector instructions.
 * It exists to measure what bit-exact bins cost (profiles/r04); nothing selects it by default and level 3 cannot reach it. */
#define LORAHIP_VARIANT_FMA 40
int lorahip_set_variant(lorahip_ctx *ctx, int variant);

/* How the kernels obtain _fineTuneTable[_fineTuneIndex] for windows whose index moves (LoRaDemod.cpp:159-162). Default (0): no
 * table read at all -- the index sequence in closed form, the entry as the product of two small fp64 factor tables held in LDS,
 * which the library has checked against all 128*N entries of the reference's table on this host (lorahip_fine_split_active() == 1;
 * if that check ever failed the tables are not built and the kernels read the table). 1: always read the table in HBM and walk
 * the index chain (the A/B switch; results are bit-identical either way). */
int lorahip_set_fine_gather(lorahip_ctx *ctx, int enable);
int lorahip_fine_split_active(const lorahip_ctx *ctx);
/* host-only self tests of the above (no device needed): 1 if the split reproduces every entry of the fine-tune table of `sf`;
 * the index sequence of one window as the kernels evaluate it (idx_out: N entries, *path = 1 closed form / 0 serial chain) */
int lorahip_fine_split_selftest(int sf);
int lorahip_fine_indices_host(int sf, int32_t idx0, float err, int32_t *idx_out, int32_t *idx_end, int32_t *path);

/* One batch of independent windows. All pointers are DEVICE pointers for
 * lorahip_detect_batch and HOST pointers for lorahip_detect_batch_host.
 *
 * Window w reads N = 2^sf cf32 samples starting at sample index
 *     offsets ? offsets[w] : w * window_stride
 * of `iq` (window_stride 0 means N: back-to-back windows).
 */
typedef struct lorahip_batch {
    size_t struct_size;          /* = sizeof(lorahip_batch), for ABI growth                      */
    /* inputs */
    const float *iq;             /* interleaved cf32 samples                                      */
    size_t n_windows;
    const int64_t *offsets;      /* optional per-window start sample                              */
    size_t window_stride;        /* used when offsets == NULL                                     */
    const int32_t *chirp_sel;    /* optional per-window LORAHIP_CHIRP_*; NULL -> chirp_sel_all    */
    int32_t chirp_sel_all;
    const int32_t *fine_idx0;    /* optional per-window _fineTuneIndex on entry (NULL -> 0)       */
    const float *fine_err;       /* optional per-window _finefreqError (NULL -> 0)                */
    /* outputs (sym/power/power_avg/f_index required, the rest optional) */
    uint16_t *sym;               /* detect() return value                LoRaDetector.hpp:63      */
    float *power;                /*                                      LoRaDetector.hpp:54      */
    float *power_avg;            /*                                      LoRaDetector.hpp:53      */
    float *f_index;              /*                                      LoRaDetector.hpp:56-61   */
    int32_t *fine_idx_out;       /* _fineTuneIndex after the N steps     LoRaDemod.cpp:160-162    */
    float *fft_out;              /* n_windows*N cf32, the "fft" port     LoRaDemod.cpp:154,172    */
    float *dec_out;              /* n_windows*N cf32, the "dec" port     LoRaDemod.cpp:164        */
} lorahip_batch;

int lorahip_detect_batch(lorahip_ctx *ctx, const lorahip_batch *b);       /* async, device ptrs */
/* The same with HOST pointers, synchronous: inputs are staged to the device, the call returns with the outputs filled.
 * `iq` must hold every sample a window reads, i.e. at least
 *     offsets ? max_w(offsets[w]) + N  :  (n_windows - 1) * (window_stride ? window_stride : N) + N
 * samples (that many are copied); offsets must be >= 0, fine_idx0 in [0, 128*N): LORAHIP_E_INVALID otherwise. */
int lorahip_detect_batch_host(lorahip_ctx *ctx, const lorahip_batch *b);

/* -------------------------------------------------------------------------------------
 * Mixed spreading factors in one call (BASELINE configs[3]: 16384 channels, SF = 7 + c mod 6). The reference runs one
 * LoRaDemod block per channel, each with its own SF (LoRaDemod.cpp:119); a launch here is uniform in N, so the scheduler buckets
 * the channels by SF, gives every bucket a level-2 context with its own HIP stream, issues the buckets' launches back to back
 * (they overlap on the device) and joins them on events. Results are bucket-major: the channels of the lowest SF first, each
 * bucket in ascending channel order; lorahip_mixed_rows() gives the row of every channel in the [rows][windows_per_channel]
 * arrays.
 *   create : SF of every channel (6..12)
 *   plan   : channel c's windows_per_channel back-to-back windows start at sample channel_offset[c] of the IQ buffer
 *   detect : asynchronous, device pointers, arrays of rows * windows_per_channel entries; the steady-state (up-chirp) shape
 *   synchronize : returns when every bucket's launch of the last detect has finished
 * ------------------------------------------------------------------------------------- */
typedef struct lorahip_mixed lorahip_mixed;
int lorahip_mixed_create(lorahip_mixed **m, int device, const int32_t *channel_sf, size_t n_channels);
void lorahip_mixed_destroy(lorahip_mixed *m);
size_t lorahip_mixed_num_buckets(const lorahip_mixed *m);
int lorahip_mixed_bucket(const lorahip_mixed *m, size_t i, int32_t *sf, size_t *first_row, size_t *n_channels);
/* bucket i's level-2 context, borrowed (owned by the scheduler): for lorahip_set_variant / timers / lorahip_timer_* on it */
lorahip_ctx *lorahip_mixed_context(const lorahip_mixed *m, size_t i);
int lorahip_mixed_rows(const lorahip_mixed *m, int64_t *row_of_channel);
int lorahip_mixed_plan(lorahip_mixed *m, const int64_t *channel_offset, size_t windows_per_channel);
int lorahip_mixed_detect(lorahip_mixed *m, const float *iq_dev, uint16_t *sym_dev, float *power_dev, float *power_avg_dev, float *f_index_dev);
int lorahip_mixed_synchronize(lorahip_mixed *m);

/* Several devices in one process ("one process, 8 devices, one host thread + stream per device"; the reference
 * runs one LoRaDemod block per channel, LoRaDemod.cpp:119-122, and nothing in it binds those channels to one GPU). Channels are
 * independent units, so there is no data-path collective: lorahip_mixed_create_multi splits the channels over devices[0..n_devices)
 * with lorahip_shard_plan, gives every device its own scheduler (above) and its own host thread that issues that device's launches.
 *   lorahip_shard_plan : host-only. SF buckets from the largest windows down, each cut into n_shards contiguous ranges; the
 *                        count % n_shards left-over channels of a bucket go to the shards holding the fewest bytes so far (weight
 *                        8*2^SF + 14 per window, lowest shard first among equals). The same rule as lora_sdr_amd/shard.py, which the
 *                        one-process-per-GPU form of the split (bench.py under torch.distributed) uses.
 *   create_multi       : devices may repeat (two shards on one GPU are two independent schedulers).
 *   plan               : as above, with channel_offset[c] relative to the IQ buffer of channel c's OWN device.
 *   rows               : channel c's row in the result arrays of its own device; lorahip_mixed_shard_of gives that device's index.
 *   detect_multi       : arrays of n_devices device pointers (entry s lives on devices[s]; entries of shards without channels are
 *                        ignored). Returns when every device's launches are issued; lorahip_mixed_synchronize joins all devices.
 *   lorahip_mixed_shard: shard s's single-device scheduler, borrowed (buckets, contexts, timers); NULL for a shard without channels.
 * On an object made by lorahip_mixed_create these report one shard; lorahip_mixed_detect refuses a multi-device object. */
int lorahip_shard_plan(const int32_t *channel_sf, size_t n_channels, size_t n_shards, int32_t *shard_of_channel);
int lorahip_mixed_create_multi(lorahip_mixed **m, const int *devices, size_t n_devices, const int32_t *channel_sf, size_t n_channels);
size_t lorahip_mixed_num_devices(const lorahip_mixed *m);
int lorahip_mixed_device(const lorahip_mixed *m, size_t shard, int32_t *device, size_t *n_channels);
lorahip_mixed *lorahip_mixed_shard(const lorahip_mixed *m, size_t shard);
int lorahip_mixed_shard_of(const lorahip_mixed *m, int32_t *shard_of_channel);
int lorahip_mixed_detect_multi(lorahip_mixed *m, const float *const *iq_dev, uint16_t *const *sym_dev, float *const *power_dev,
                               float *const *power_avg_dev, float *const *f_index_dev);

/* Pinned host memory for the host-pointer entry points (lorahip_detect_batch_host, lorahip_demod_run, the detector shim): buffers
 * obtained here -- or any hipHostMalloc'ed / hipHostRegister'ed memory -- are read by the DMA engine directly; ordinary memory is
 * gathered through the library's double-buffered pinned staging first (several threads, LORAHIP_UPLOAD_THREADS). NULL on failure. */
void *lorahip_host_alloc(size_t bytes);
void lorahip_host_free(void *p);

/* Time the last `n` launches made through this context between two internal HIP events
 * recorded on the launch stream (bench.py uses this for the roofline line). */
int lorahip_timer_start(lorahip_ctx *ctx);
int lorahip_timer_stop(lorahip_ctx *ctx, float *elapsed_ms);

/* -------------------------------------------------------------------------------------
 * Level 1: LoRaDetector<float> shim (LoRaDetector.hpp:8-72). N must be 2^sf with sf in
 * [LORAHIP_SF_MIN, LORAHIP_SF_MAX]. The object keeps the window (feed() writes it), the bins and
 * detect()'s results in one block of pinned host memory that the device addresses directly:
 * a detect() is one kernel launch and one wait (~19 us; no staged copies). One object is
 * used by one thread at a time, as the reference's detector is.
 * ------------------------------------------------------------------------------------- */
typedef struct lorahip_detector lorahip_detector;

int lorahip_detector_create(lorahip_detector **det, int device, size_t N);   /* LoRaDetector(N)  :12 */
void lorahip_detector_destroy(lorahip_detector *det);
int lorahip_detector_feed(lorahip_detector *det, size_t i, float re, float im); /* feed()       :23 */
/* detect(): returns the arg-max bin through *index; fft_out may be NULL (:29-64) */
int lorahip_detector_detect(lorahip_detector *det, size_t *index, float *power, float *power_avg,
                            float *f_index, float *fft_out);

/* -------------------------------------------------------------------------------------
 * Level 3: B channels of the LoRaDemod block (declared here, see lorahip_demod.cpp).
 *
 * What "identical to the reference block" means here: symbol values and FFT bins are bit-exact by construction; power, powerAvg and
 * fIndex equal the CPU build's to float rounding (<= 2e-5 dB / 2e-6 bins: the logarithms and the fp64 sum are evaluated in a
 * different order), and the frame machine consumes them (`snr < thresh`, `_finefreqError += fIndex`, LoRaDemod.cpp:173-174,219).
 * A value within that rounding of a decision boundary could therefore take the other branch than the CPU build; every trace the
 * tests compare is identical, but at this level the guarantee is empirical.
 * ------------------------------------------------------------------------------------- */
typedef struct lorahip_demod lorahip_demod;

int lorahip_demod_create(lorahip_demod **d, int device, int sf, size_t n_channels); /* make(sf) LoRaDemod.cpp:119 */
void lorahip_demod_destroy(lorahip_demod *d);
/* One level-3 object whose channels each have their own SF, over several devices of the node: what n_channels calls of the
 * reference's make(sf) with different sf give a flow graph (LoRaDemod.cpp:119-122), as ONE handle (BASELINE configs[3]: 16384
 * channels, SF 7..12, 8 GPUs). lorahip_shard_plan assigns every channel a device (devices[] may repeat: two shards on one GPU); on
 * each device the channels of one SF form a PART -- a plain single-SF object with its own context, HIP stream and host thread, so
 * the parts of a device overlap on it and the devices run side by side; there is no data-path collective. Every lorahip_demod_*
 * setter and accessor works on the handle with GLOBAL channel numbers (packets, signals, traces, labels, consumed); the packet
 * queue lists the parts one after the other (device slot ascending, SF ascending inside one), each in its own order. Runs:
 *   lorahip_demod_run                       one host buffer per channel (any SF mix, any number of devices)
 *   lorahip_demod_run_device_segments       all channels' segments in ONE device buffer (objects on one device)
 *   lorahip_demod_run_device_segments_multi iq_dev[n_devices]: channel c's segment lies in the buffer of ITS device slot
 *                                           (lorahip_demod_part_of / lorahip_demod_part say which)
 * Not offered on such a handle (LORAHIP_E_INVALID; use the part's own handle, lorahip_demod_part_handle): lorahip_demod_run_device,
 * the append runs, lorahip_demod_packets_to_device (one decoder configuration per SF), lorahip_demod_set_stream when the object
 * spans several devices; debug ports only for one SF and host buffers. lorahip_demod_kernel_ms / _last_launches report the maximum
 * over the parts, lorahip_demod_work_calls / _near_threshold the sums. On a plain object the part accessors report one part. */
int lorahip_demod_create_mixed(lorahip_demod **d, const int *devices, size_t n_devices, const int32_t *channel_sf, size_t n_channels);
size_t lorahip_demod_num_channels(const lorahip_demod *d);
size_t lorahip_demod_num_parts(const lorahip_demod *d);
int lorahip_demod_part(const lorahip_demod *d, size_t i, int32_t *device, int32_t *sf, size_t *n_channels, int32_t *device_slot);
int lorahip_demod_part_of(const lorahip_demod *d, int32_t *part_of_channel, int32_t *local_channel);   /* [n_channels] each, nullable */
lorahip_demod *lorahip_demod_part_handle(const lorahip_demod *d, size_t i);                             /* borrowed */
int lorahip_demod_set_sync(lorahip_demod *d, unsigned char sync);        /* setSync       :124 */
int lorahip_demod_set_threshold(lorahip_demod *d, double thresh_dB);     /* setThreshold  :129 */
int lorahip_demod_set_mtu(lorahip_demod *d, size_t mtu);                 /* setMTU        :134 */
int lorahip_demod_activate(lorahip_demod *d);                            /* activate()    :139 */
/* How work() rounds are driven: 0 = auto (1 where a streaming kernel exists for the SF, else 2),
 * 1 = on the device: one streaming kernel walks every channel's stream window after window with the frame
 *     machine (LoRaDemod.cpp:176-312) in registers; no host round trip between the windows of a channel,
 * 2 = from the host: one batch launch per lock-step round, the frame machine on the host between launches.
 * Both produce identical packets, traces and signals. */
int lorahip_demod_set_mode(lorahip_demod *d, int mode);
/* Launch on an existing hipStream_t from now on (same meaning as lorahip_set_stream): work queued on that stream before a
 * run -- a channeliser, a modulator, a copy -- is ordered before the run's kernels. A run returns with the stream drained. */
int lorahip_demod_set_stream(lorahip_demod *d, void *hip_stream);
/* Make `hip_stream` (a hipStream_t of the object's device; NULL = the null stream) wait for everything queued so far on the object's
 * launch stream: the way a consumer on a stream of its own reads rows that are "valid in stream order on the launch stream"
 * (lorahip_demod_receive with async = 1 or 2). Costs an event record and a stream wait; allowed while a pipelined step is in flight. */
int lorahip_demod_stream_wait(lorahip_demod *d, void *hip_stream);
/* The other direction: the object's launch stream waits for everything queued so far on `hip_stream` -- the producer of the samples,
 * or a consumer of rows that the next pipelined step will overwrite -- without a host wait. */
int lorahip_demod_stream_follow(lorahip_demod *d, void *hip_stream);
int lorahip_demod_reset_stream(lorahip_demod *d);    /* back to the private stream */
/* kernel variant of the host-driven mode's batch launches (lorahip_set_variant: 0, 1, 10 -- identical results); LORAHIP_VARIANT_FMA
 * is refused (LORAHIP_E_INVALID): level 3 runs the reference's operation graph only */
int lorahip_demod_set_variant(lorahip_demod *d, int variant);
/* The streaming kernels' grid (scheduling only: every result is the same). 0 = the library's choice (one workgroup per channel set;
 * at SF11 the resident number of workgroups, each walking channel after channel: lorahip_wide.hip), < 0 = one workgroup per
 * channel set always, n > 0 = at most n workgroups each walking several sets -- honoured where the build holds such an instance
 * (SF11 and SF12), the default elsewhere. For measurements and tests. */
int lorahip_demod_set_stream_grid(lorahip_demod *d, int max_workgroups);
/* Lanes per channel of the streaming kernels at SF7-9 (scheduling only: every result is the same). A channel is a sequential chain of
 * work() calls; the default geometry gives a lane 16 points of the window (SF7: 8 lanes per channel), which is the cheapest per
 * window and fills the device from 16384 (SF7) / 8192 (SF8) / 4096 (SF9) channels up. A receiver with fewer channels leaves
 * wavefront slots empty, so the library then gives a channel 2x / 4x / 8x the lanes (8 or 4 points per lane: shorter calls, more
 * wavefronts). log2_lanes: 0 = chosen by the channel count (default), < 0 = 16 points per lane always, 4 / 5 / 6 = 16 / 32 / 64 lanes
 * per channel where the build holds that instance (SF7: 4, 5; SF8: 5, 6; SF9: 6), the default geometry elsewhere. For measurements
 * and tests.
 * 16 | l (l = 3, 4, 5: SF7; 4, 5: SF8; 5: SF9): a channel takes TWO groups of 2^l lanes, and the second evaluates the window the NEXT
 * work() call will read if this one consumes exactly N samples and leaves the fine-tune state where a plain call leaves it (inside a
 * packet, on a quiet or aligned FRAMESYNC call, on the first down-chirp); the frame machine makes that second call in the same pass
 * when it finds the channel exactly there, and drops the window otherwise. Chosen by the library only at SF7 with at most half as many
 * channels as the device holds wavefronts at two per SIMD (1024 on an MI355X), where it is worth 4 - 8 %.
 * The parts of a mixed object (lorahip_demod_create_mixed) run side by side on their device: where the choice is the library's, a
 * part counts the wavefronts its sibling parts bring (at 16 points per lane) as taken, and widens only into what is left. */
int lorahip_demod_set_stream_lanes(lorahip_demod *d, int log2_lanes);
/* log2 of the lanes per channel the object's streaming launches run on (the choice above resolved for its channel count and device;
 * SF11 / SF12: 7 / 8, a channel is a workgroup); LORAHIP_E_INVALID for a mixed object (per part: lorahip_demod_part_handle) */
int lorahip_demod_stream_lanes(const lorahip_demod *d);
/* A bound on the streaming kernels' per-launch record capacity (work() calls per channel per launch; 0 = none beyond the library's
 * own sizing). A channel that fills its records stops, and the run resumes it with another launch: results are the same. For tests
 * of that path (it replaces the environment hook LORAHIP_STREAM_CAP of earlier builds). */
int lorahip_demod_set_record_capacity(lorahip_demod *d, size_t max_calls_per_launch);
/* same switch as lorahip_set_fine_gather, for the demodulator's kernels */
int lorahip_demod_set_fine_gather(lorahip_demod *d, int enable);

/* Per-channel outcome of one work() round (what the block would have done on its ports). */
typedef struct lorahip_work_result {
    int64_t consumed;       /* inPort->consume(total)                          :320 */
    int32_t state_before;   /* 0 FRAMESYNC 1 DOWNCHIRP0 2 DOWNCHIRP1 3 QUARTERCHIRP 4 DATASYMBOLS */
    int32_t value;          /* detect() of window 0                            :172 */
    float power, power_avg, snr, f_index;
    int32_t worked;         /* 0 if fewer than 2N samples were available       :148 */
    int32_t packet_len;     /* >0: a packet of that many int16 symbols was posted this round :295-298 */
    int32_t signals;        /* 1 at DOWNCHIRP1: error/power/snr emitted        :267-269 */
    int32_t sig_error;
    float sig_power, sig_snr;
    /* the dechirp state of this call (LoRaDemod.cpp:157-166): what lorahip_demod_get_ports() replays the debug ports from */
    int32_t fine_idx_before;   /* _fineTuneIndex when the call began                                          */
    int32_t fine_idx_after;    /* ... after the N steps of window 0 (where the sync check's window 1 starts :191) */
    float fine_err_before;     /* _finefreqError when the call began                                          */
    int32_t reserved;
} lorahip_work_result;

/* Feed every channel's whole stream (host memory, cf32, n_samples[c] samples each) and run
 * work() rounds in lock-step until no channel has 2N samples left. Packets are appended to
 * the demod's queue. Returns the number of rounds through *rounds (may be NULL). */
int lorahip_demod_run(lorahip_demod *d, const float *const *streams, const size_t *n_samples,
                      int64_t *rounds);
/* Same, but `iq` is one DEVICE buffer holding n_channels streams of samples_per_channel each. */
int lorahip_demod_run_device(lorahip_demod *d, const float *iq_dev, size_t samples_per_channel,
                             int64_t *rounds);
/* Same for a caller that keeps its channels' samples in ONE device buffer at places of its own -- the running receiver behind a
 * channeliser: channel c's stream is the n_samples[c] samples that start at sample first_sample[c] of iq_dev (HOST arrays of
 * n_channels entries; segments may have any lengths, including 0, and need not be ordered). work() leaves fewer than 2N samples
 * of a channel unconsumed (LoRaDemod.cpp:148) and how many it consumed differs from channel to channel (lorahip_demod_consumed), so
 * a caller that appends each new chunk behind the last one advances first_sample[c] by what channel c consumed and presents the
 * remainder together with the new samples -- nothing is copied, nothing is lost at the chunk boundaries. */
int lorahip_demod_run_device_segments(lorahip_demod *d, const float *iq_dev, const int64_t *first_sample, const size_t *n_samples,
                                      int64_t *rounds);
/* HOST buffers that are the rows of ONE block of host memory: channel c's n_samples[c] samples begin at sample first_sample[c] of row c
 * (rows + 2 * c * row_stride floats), first_sample[c] + n_samples[c] <= row_stride. What a framework hands a block whose input buffer
 * manager carves every port's slabs out of one pinned allocation (lorahip_host_alloc; lora_sdr_amd/pothos/LoRaDemodBatch.cpp::
 * getInputBufferManager, the counterpart of LoRaDemod.cpp:346-357): the span of the rows that holds samples crosses PCIe as ONE strided
 * copy straight from the caller's memory -- no staging copy, no per-channel call -- and the run reads per-channel segments of the
 * device copy. Results as lorahip_demod_run on the same samples; synchronous (the rows are the caller's again on return). Ordinary
 * memory works too (the runtime stages it). An object over several parts takes its channels' buffers one by one (lorahip_demod_run). */
int lorahip_demod_run_host_rows(lorahip_demod *d, const float *rows, size_t row_stride, const int64_t *first_sample, const size_t *n_samples,
                                int64_t *rounds);
/* the same for an object that spans several devices (lorahip_demod_create_mixed): one buffer per entry of its device list */
int lorahip_demod_run_device_segments_multi(lorahip_demod *d, const float *const *iq_dev, size_t n_devices, const int64_t *first_sample,
                                            const size_t *n_samples, int64_t *rounds);

/* The RUNNING receiver: channel c's stream is row c of a (n_channels, row_stride) device array of which the first n_valid samples
 * are valid -- a capture, or the output array of a channeliser, that fills chunk by chunk. Each call is one work() of every channel
 * over what has arrived: a channel continues at ITS OWN read position (work() leaves fewer than 2N samples of it unconsumed,
 * LoRaDemod.cpp:148; the positions live on the device, nothing is uploaded per call), its frame machine, fine-tune state and any
 * open packet carry over. n_valid must not shrink from call to call; lorahip_demod_rewind() -- or any other kind of run -- starts
 * new streams at sample 0. After an append run lorahip_demod_consumed() is the channel's absolute read position in its row and the
 * packets' `round` counts the channel's work() calls since the streams began. */
int lorahip_demod_run_device_append(lorahip_demod *d, const float *iq_dev, size_t row_stride, size_t n_valid, int64_t *rounds);
int lorahip_demod_rewind(lorahip_demod *d);
/* One receiver step in ONE call: lorahip_demod_run_device_append, then the packets that completed -- those begun in earlier calls
 * included -- packed ON THE DEVICE into the batched decoder's rows (as lorahip_demod_packets_to_device, rows numbered there too:
 * no upload, no host copy of any symbol), then the queue cleared. *n_packets = packets delivered (LORAHIP_E_INVALID and the packets
 * left queued if that exceeds cap_packets), *work_calls = LoRaDemod::work() calls made by this step, summed over the channels.
 * With async = 1 the call returns without waiting for the packing kernels: the rows are valid in stream order on the object's
 * launch stream (lorahip_demod_set_stream), which is where a decoder that follows would be queued. Signals kept by
 * lorahip_demod_set_signals go to the rows registered with lorahip_demod_receive_signal_rows (below); without such rows they are dropped.
 * With async = 2 the steps are PIPELINED: this step's kernel is launched before the previous step's summary is read, so the host's
 * share of a step (launch latencies, the wait for the summary, the packing launches: ~60 us) overlaps the kernel instead of
 * following it. The price is one step of latency: *n_packets / *work_calls and the rows are those of the PREVIOUS step (0 for the
 * first), valid in stream order on the object's launch stream like with async = 1 (short steps are packed on a side stream beside
 * the running kernel; the launch stream waits for that before anything later); lorahip_demod_receive_flush() delivers the last
 * step's and leaves the pipeline. The first call after anything else touched
 * the object is an ordinary step (its packets delivered at once). While a step is in flight every other entry point that needs the
 * object's state returns LORAHIP_E_INVALID ("flush first"); no trace / ports in this mode (signals: lorahip_demod_receive_signal_rows).
 * Rows that cannot hold the packets that are due (cap_packets too small, null pointers) lose NOTHING: the call returns
 * LORAHIP_E_INVALID with *n_packets = the rows needed, the packets stay in the step's record set on the device, and the next
 * lorahip_demod_receive / _flush whose rows hold them delivers them first, together with the packets of the step launched in
 * between (*n_packets and *work_calls then cover both steps; until then no further kernel is launched -- the samples wait in the
 * caller's array, a later call covers them). As with async = 0/1 a packet longer than sym_stride keeps its true length in
 * nsyms_dev and its first sym_stride symbols (the decoder flags it): sym_stride >= the MTU never truncates.
 * With async = 3 the receiver is RESIDENT (SF7-12; the grid is at most the device's resident set of workgroups, each taking as many
 * channel sets per step as it takes to cover the channels; until the object is in place for it -- the first call of a capture -- the
 * call is an ordinary step): ONE kernel launch stays on the device across the steps. A call copies a 104-byte message into a ring in device memory (the doorbell) and
 * returns the counts of the PREVIOUS step; the kernel's wavefronts poll the ring, work through what arrived with the tables they
 * staged once, pack their own channels' packets (and signals, lorahip_demod_receive_signal_rows) themselves into the rows that came
 * WITH the step's call, and the last workgroup reports two words to pinned host memory: no launch, no helper kernel per step. So: the rows
 * passed to call k are filled by step k and are complete -- in memory, for any stream, a copy engine or, if they are pinned host
 * memory, the host -- when call k + 1 (or lorahip_demod_receive_flush) returns with their *n_packets; keep two sets of rows and
 * alternate (with a depth d > 1, `reserved` below: call k + d reports them, d + 1 sets). Rows are handed out in the order the wavefronts finish: a channel's packets of a step are consecutive and in time order,
 * channels are not sorted (channel_dev says whose a row is). The samples up to n_valid must BE in iq_dev when the call is made (the
 * kernel reads them on its own, not in the order of any stream). Rows too small for a step: the excess is dropped, counted in
 * *n_packets, and the call that reports the step returns LORAHIP_E_INVALID -- size the rows for a step (one packet per channel and
 * frame that can end in it). While the kernel is resident it holds the wavefront slots of its channels -- on a full device nothing else
 * runs until the flush, and a DEVICE-wide synchronise (hipDeviceSynchronize) waits for it --, every other entry point that needs the
 * object returns LORAHIP_E_INVALID, and every wait is bounded: a
 * wavefront that sees no message for 8 s leaves, a host call that sees no report for 5 s tells the kernel to leave and fails (the
 * object then takes ordinary steps). The launch is sized for an EMPTY device: a second object (another part of a mixed receiver, another
 * process) asking for a resident kernel while the first holds the device's slots finds its census incomplete after 2 s, ends its launch and
 * takes ordinary steps from then on -- one resident receiver per device at a time. lorahip_demod_receive_flush(d, rows, ..) reports the
 * last step, ends the kernel and leaves the
 * object as a streaming run leaves it.
 * Ordering of the rows: a step's packets are written in stream order AFTER everything that was queued on the launch stream when
 * the call began, so a consumer (decoder) of the rows handed out by the call before, queued on that stream (or on a stream the
 * launch stream was told to follow before the call, lorahip_demod_stream_follow), has read them before they are overwritten -- one
 * set of rows is enough. lorahip_demod_receive_flush also resumes a channel whose record capacity the
 * last step filled (that step's remaining packets follow the others in the rows). */
typedef struct lorahip_packet_rows {
    size_t struct_size;     /* = sizeof(lorahip_packet_rows) */
    uint16_t *syms_dev; size_t sym_stride;      /* [cap_packets][sym_stride], zero padded */
    int32_t *nsyms_dev;                          /* [cap_packets] */
    int32_t *channel_dev;                        /* [cap_packets], nullable */
    size_t cap_packets;
    int32_t async;          /* 0 wait, 1 rows valid in stream order, 2 pipelined, 3 resident (see above) */
    int32_t reserved;       /* async = 3, at the call that starts the resident kernel: the DEPTH, 0 / 1 (default) .. 3 -- how many steps the
                               receiver may run ahead of the last report. Call k then reports step k - depth (the rows of call k - depth),
                               the flush everything left, and the caller cycles depth + 1 sets of rows. A step ends with its slowest
                               workgroup; with more steps in flight the fast ones work ahead instead of waiting for it. Otherwise 0. */
} lorahip_packet_rows;
int lorahip_demod_receive(lorahip_demod *d, const float *iq_dev, size_t row_stride, size_t n_valid, const lorahip_packet_rows *rows,
                          size_t *n_packets, int64_t *work_calls);
/* The block's signals in a running receiver (the reference emits "error" / "power" / "snr" once per packet at DOWNCHIRP1, LoRaDemod.cpp:
 * 267-269). With lorahip_demod_set_signals(d, 1) AND rows registered here, every lorahip_demod_receive / _receive_flush call delivers
 * the signals of the step whose packets it delivers -- ordinary steps at once, pipelined steps (async = 2) one step late, beside the
 * packets -- into rows [0, n), n = lorahip_demod_receive_num_signals(d): channel, error, power, snr per emission, channels ascending
 * and time ascending inside a channel, written in the same stream order as the packet rows. The arrays may be device memory or pinned
 * host memory (lorahip_host_alloc), which the device writes directly -- what a host consumer (a block's emitSignal) reads after the
 * stream has passed; any array may be NULL. A step emits at most one signal per packet it completes plus one per channel (a frame whose
 * packet is still open): cap >= cap_packets + n_channels always suffices. Rows that cannot hold what is due lose nothing: the call
 * returns LORAHIP_E_INVALID exactly as for packet rows that are too small (pipelined: packets and signals stay in the step's record
 * set; ordinary: they stay queued for lorahip_demod_get_packets / _get_signals). rows = NULL (or no rows registered): receiver steps
 * drop the signals, as before version 4. The registration may be changed before any call, steps in flight or not (it is two pointers
 * and a count): a caller that alternates two sets of packet rows alternates two sets of signal rows the same way. The signals a call
 * delivers go where the registration stood AT that call (ordinary and pipelined steps); a RESIDENT step's signals go, like its packets,
 * where the registration stood when the step was rung, and are complete when the next call reports them. */
typedef struct lorahip_signal_rows {
    size_t struct_size;     /* = sizeof(lorahip_signal_rows) */
    int32_t *channel;       /* [cap] */
    int32_t *error;         /* [cap]  "error": the frequency error in bins (int)      :267 */
    float *power;           /* [cap]  "power"                                         :268 */
    float *snr;             /* [cap]  "snr"                                           :269 */
    size_t cap;
} lorahip_signal_rows;
int lorahip_demod_receive_signal_rows(lorahip_demod *d, const lorahip_signal_rows *rows /* nullable */);
size_t lorahip_demod_receive_num_signals(const lorahip_demod *d);
int lorahip_demod_receive_flush(lorahip_demod *d, const lorahip_packet_rows *rows /* nullable: the last step's packets are dropped */,
                                size_t *n_packets, int64_t *work_calls);
int lorahip_demod_resident_active(const lorahip_demod *d);      /* 1 while the resident kernel (async = 3) is on the device */
/* The resident steps the LAST receive / flush call reported, oldest first: packets[i] / signals[i] of each (the call's *n_packets is
 * their sum). A receive call reports at most one; a flush at a depth d > 1 up to d, each in the rows of its own call. Returns how many. */
size_t lorahip_demod_receive_steps(const lorahip_demod *d, size_t *packets, size_t *signals, size_t cap);

/* The block's signals "error" (int), "power" (float), "snr" (float), emitted once per packet at DOWNCHIRP1 (LoRaDemod.cpp:85-87,
 * 267-269), WITHOUT a per-call trace: with enable = 1 the following runs keep one record per emission -- the kernels evaluate
 * power / snr for that one call of a packet, nothing else changes -- queued like the packets (channels ascending, time ascending
 * inside a channel after a streaming run) and cleared with them by lorahip_demod_clear_packets. `rounds` as for packets. Off by
 * default. Values equal the traced run's sig_error / sig_power / sig_snr. */
int lorahip_demod_set_signals(lorahip_demod *d, int enable);
size_t lorahip_demod_num_signals(const lorahip_demod *d);
int lorahip_demod_get_signals(const lorahip_demod *d, int32_t *channels, int64_t *rounds, int32_t *errors, float *powers, float *snrs, size_t cap);

size_t lorahip_demod_num_packets(const lorahip_demod *d);
/* packet i: channel, round index it was posted in, and length; symbols copied if out != NULL */
int lorahip_demod_get_packet(const lorahip_demod *d, size_t i, int32_t *channel, int64_t *round,
                             size_t *len, int16_t *out, size_t cap);
/* all queued packets at once, in posting order: per packet channel / round / length, and the symbols back to back */
size_t lorahip_demod_num_packet_symbols(const lorahip_demod *d);
int lorahip_demod_get_packets(const lorahip_demod *d, int32_t *channels, int64_t *rounds, int64_t *lens, size_t cap_packets,
                              int16_t *syms, size_t cap_syms);
/* The queued packets in the batched decoder's input layout, on the device. Straight after a run of the streaming mode -- nothing
 * read back to the host yet -- the rows are packed on the device from the kernel's records (no host round trip; rows ordered by
 * channel, then time). That includes packets begun in an earlier run: the symbols of a packet a channel is inside when a run ends
 * stay on the device and the next run's records continue them (packets of at most 4096 symbols; longer ones are handed on through
 * the host queue). Otherwise the rows come from the host queue in lorahip_demod_get_packets' order. Either way: packet p's
 * symbols at syms_dev + p*sym_stride (zero padded), its length in nsyms_dev[p] -- a packet longer than sym_stride keeps its true
 * length there and lorahip_decode_packets() reports -2 for it --, its channel in channel_dev[p] (nullable). *n_packets = number of
 * queued packets (also when cap_packets is too small: LORAHIP_E_INVALID then). Does not clear the queue. */
int lorahip_demod_packets_to_device(lorahip_demod *d, uint16_t *syms_dev, size_t sym_stride, int32_t *nsyms_dev, int32_t *channel_dev,
                                    size_t cap_packets, size_t *n_packets);
void lorahip_demod_clear_packets(lorahip_demod *d);
/* samples of `channel`'s stream the last lorahip_demod_run[_device] consumed (the sum of its consume() calls, :320): a
 * streaming caller presents the unconsumed remainder again in front of the next chunk, as the framework's port buffer does */
int64_t lorahip_demod_consumed(const lorahip_demod *d, size_t channel);
/* the same for every channel at once: out[n_channels] */
int lorahip_demod_consumed_all(const lorahip_demod *d, int64_t *out);
/* total work() calls made (sum over channels) since lorahip_demod_create */
int64_t lorahip_demod_work_calls(const lorahip_demod *d);
/* device time of the streaming kernel launches of the last lorahip_demod_run[_device] (HIP events on the launch stream; 0 in the
 * host-driven mode): what the level-3 roofline line of bench.py is computed from */
double lorahip_demod_kernel_ms(const lorahip_demod *d);
/* streaming kernel launches the last lorahip_demod_run[_device] took (0 in the host-driven mode). One is the rule; a run whose
 * per-launch record buffers filled is resumed (records drained in between, which costs more than the launch), and the capacity the
 * following runs of this object are given grows with what the run needed -- a receiver settles at one launch per run */
int lorahip_demod_last_launches(const lorahip_demod *d);
/* The runtime signal for the caveat at the top of this section: how many decisions since the last activate() sat so close to their
 * boundary that the last-place differences between this library's power / powerAvg / fIndex and a CPU build's could have flipped
 * them. Counted, never altered.
 *   near_squelch : work() calls in FRAMESYNC / DATASYMBOLS (the states that consume `snr < thresh`, LoRaDemod.cpp:174,217,291)
 *                  with |snr - thresh| <= 4e-5 dB
 *   near_step    : windows dechirped with a moving fine-tune index whose step _finefreqError * 128 (LoRaDemod.cpp:160) lies
 *                  within 6e-5 of an integer, i.e. where a last-place difference in the fIndex values accumulated in
 *                  _finefreqError (:219) would move the reference's int <- float truncation
 * Both 0 means: every branch taken has a margin far above float rounding, so the identity with the CPU block holds by margin for
 * that run, not only by observation. Either pointer may be NULL. */
int lorahip_demod_near_threshold(const lorahip_demod *d, int64_t *near_squelch, int64_t *near_step);
/* Debug ports of the block, opt-in like the trace (they triple the HBM traffic): what LoRaDemod::work() writes to its "raw", "dec"
 * and "fft" outputs (LoRaDemod.cpp:81-83,163-164,172,320-324), per channel, for the NEXT runs. Device buffers owned by the caller:
 *   fft_dev  [n_channels][fft_cap_frames][N] cf32   one frame of N bins per work() call: window 0's FFT (:172, produce(N) :324)
 *   dec_dev  [n_channels][dec_cap_samples]   cf32   `total` dechirped samples per call (:164, produce(total) :322): window 0's first
 *                                                   min(total, N), and window 1 when the call consumed 2N (the sync check, :189-206)
 *   raw_dev  [n_channels][raw_cap_samples]   cf32   the samples consumed (:163, :321)
 * Any pointer may be NULL (that port stays off); p == NULL switches all off. Bins and samples are bit-identical to the reference's.
 * What exceeds a capacity is dropped; lorahip_demod_port_counts() reports what the last run produced. Labels: see below.
 * The ports are replayed from the per-call trace, so a run with ports on keeps one internally even when lorahip_demod_set_trace was
 * never enabled: that trace lives for the run only (it is not visible through lorahip_demod_get_trace / _trace_len / _get_labels
 * and does not accumulate), and like any traced run it drains the records to the host, i.e. the device-resident packet hand-off
 * of lorahip_demod_packets_to_device takes its host-queue path. */
typedef struct lorahip_demod_ports {
    size_t struct_size;     /* = sizeof(lorahip_demod_ports) */
    float *fft_dev; size_t fft_cap_frames;
    float *dec_dev; size_t dec_cap_samples;
    float *raw_dev; size_t raw_cap_samples;
    int32_t host_buffers;   /* 1: the three pointers are HOST buffers (a framework's port buffers): the library keeps device
                               buffers of the same shape and copies what a run produced into them before the run returns */
    int32_t reserved;
} lorahip_demod_ports;
int lorahip_demod_set_ports(lorahip_demod *d, const lorahip_demod_ports *p);
int lorahip_demod_port_counts(const lorahip_demod *d, size_t channel, size_t *fft_frames, size_t *dec_samples, size_t *raw_samples);
/* The stream labels work() posts at index 0 of what each call produces on raw / dec / fft (LoRaDemod.cpp:314-319) -- "SYNC",
 * "P <fIndex>", "DC", "QC", "S<n> <fIndex>", or none ("") -- for every call in `channel`'s trace (lorahip_demod_set_trace), as
 * NUL-terminated strings back to back, formatted like the reference's (fixed, 4 decimals). Label k sits at element k*N of the fft
 * port and at element sum(consumed[0..k)) of raw / dec. *n_calls = number of strings, *bytes = bytes needed; buf may be NULL. */
int lorahip_demod_get_labels(const lorahip_demod *d, size_t channel, char *buf, size_t cap, size_t *n_calls, size_t *bytes);
/* optional trace of every work() call: enable before run, then read back */
int lorahip_demod_set_trace(lorahip_demod *d, int enable);
size_t lorahip_demod_trace_len(const lorahip_demod *d, size_t channel);
int lorahip_demod_get_trace(const lorahip_demod *d, size_t channel, lorahip_work_result *out, size_t cap);

/* -------------------------------------------------------------------------------------
 * Synthetic IQ directly in HBM (bench / tests input; ChirpGenerator.hpp:22-47 semantics in
 * closed form): window w of channel c = ampl * upchirp(sym[c*S+w]) (+ AWGN of per-component
 * sigma `noise_sigma`, counter-based RNG keyed by seed). iq_dev: B*S*N cf32.
 * ------------------------------------------------------------------------------------- */
int lorahip_synth_symbols(lorahip_ctx *ctx, float *iq_dev, const uint16_t *sym_dev,
                          size_t n_windows, float ampl, float noise_sigma, uint64_t seed);

/* -------------------------------------------------------------------------------------
 * Batched modulator + channel (the step before the path: SURVEY.md section 8f #3). n_frames packets of nsyms
 * uint16 symbols each -> the samples the LoRaMod block produces for them (LoRaMod.cpp:109-238 on
 * ChirpGenerator.hpp:22-47, ovs = 1): 10 up-chirps, the two sync-word chirps, 2 1/4 down-chirps, one chirp
 * per symbol, max(padding,1) zero symbols; one running float phase accumulator per frame. Frame f is written
 * at iq_dev + f*frame_stride samples; frame_stride >= lorahip_mod_frame_len(sf, nsyms, padding).
 * ------------------------------------------------------------------------------------- */
size_t lorahip_mod_frame_len(int sf, size_t nsyms, size_t padding);
int lorahip_mod_frames(lorahip_ctx *ctx, float *iq_dev, size_t frame_stride, const uint16_t *syms_dev,
                       size_t n_frames, size_t nsyms, unsigned char sync, float ampl, size_t padding);
/* complex AWGN of per-component standard deviation sigma added in place (counter-based generator keyed by seed) */
int lorahip_add_awgn(lorahip_ctx *ctx, float *iq_dev, size_t n_samples, float sigma, uint64_t seed);

/* -------------------------------------------------------------------------------------
 * Batched decoder (the step after the path: SURVEY.md section 8f #2): what the LoRaDecoder block does with one symbol
 * message (LoRaDecoder.cpp:196-397 on LoRaCodes.hpp), for n_packets messages at once. Parameters and defaults are the
 * block's (LoRaDecoder.cpp:98-110, setters :134-191): coding rate "4/4".."4/8" = rdd 0..4.
 *   packet p: nsyms_dev[p] symbols at syms_dev + p*sym_stride (sym_stride <= lorahip_decode_max_symbols() = 16384)
 *   out_len_dev[p]: number of output elements posted at out_dev + p*out_stride -- bytes, or uint16 symbols when
 *       interleaving is off --, -1 if the block posts nothing (fewer than 8 symbols, or dropped), -2 if the row does not hold the
 *       symbols the packet's length needs (nsyms_dev[p] > sym_stride: the caller's rows are too short -- an error, see
 *       lorahip_decode_max_symbols); dropped_dev[p] = 1 where the block calls drop() (the "dropped" signal).
 *   out_stride: even, >= 2*(sym_stride + 8). ctx supplies the device and the stream only (any SF).
 * ------------------------------------------------------------------------------------- */
typedef struct lorahip_decoder_cfg {
    size_t struct_size;     /* = sizeof(lorahip_decoder_cfg) */
    int32_t sf;             /* setSpreadFactor      default 10 */
    int32_t ppm;            /* setSymbolSize        default 0 = sf */
    int32_t rdd;            /* setCodingRate        default 4 ("4/8") */
    int32_t crcc;           /* enableCrcc           default 0 */
    int32_t interleaving;   /* enableInterleaving   default 1 */
    int32_t error_check;    /* enableErrorCheck     default 0 */
    int32_t explicit_hdr;   /* enableExplicit       default 1 */
    int32_t hdr;            /* enableHdr            default 0 */
    int32_t data_length;    /* setDataLength        default 8 */
} lorahip_decoder_cfg;
int lorahip_decode_packets(lorahip_ctx *ctx, const lorahip_decoder_cfg *cfg, const uint16_t *syms_dev, size_t sym_stride,
                           const int32_t *nsyms_dev, size_t n_packets, uint8_t *out_dev, size_t out_stride,
                           int32_t *out_len_dev, int32_t *dropped_dev);
/* The same with every buffer in HOST memory (staged through the context's pinned buffer; synchronous): what a host-side block calls
 * -- lora_sdr_amd/pothos/LoRaDecoderBatch.cpp collects the messages waiting on its inputs into rows and posts the bytes. */
int lorahip_decode_packets_host(lorahip_ctx *ctx, const lorahip_decoder_cfg *cfg, const uint16_t *syms, size_t sym_stride,
                                const int32_t *nsyms, size_t n_packets, uint8_t *out, size_t out_stride, int32_t *out_len,
                                int32_t *dropped);
/* the longest row (symbols): sym_stride <= this (16384). A packet of ANY length up to its row decodes as the reference decodes it: the
 * reference de-interleaves every codeword of a message but only those the announced length needs reach the output (LoRaDecoder.cpp:
 * 315-361) -- at most 2 * 260 with an explicit header --, so the kernel's working set follows the configuration, not the packet. A packet
 * LONGER than its row (nsyms_dev[p] > sym_stride) still decodes when the symbols its length needs lie inside the row; otherwise
 * out_len = -2: the caller's rows are too short, which a caller must treat as an error (LoRaDecoderBatch.cpp throws). */
int lorahip_decode_max_symbols(void);
/* without a header: the largest data_length (4096 bytes); a larger one is refused with LORAHIP_E_INVALID */
int lorahip_decode_max_data_length(void);

/* -------------------------------------------------------------------------------------
 * Front-end channeliser (the step before the path: SURVEY.md section 8f #4). NOT a reference component: the
 * reference's topologies put Pothos' /comms/rotate and a decimating FIR in front of every LoRaDemod block; this
 * does that for K channels of one wideband stream at once and writes the [channel][time] layout
 * lorahip_demod_run_device() reads. Definition (x[n] = the wideband stream since the last reset, x[n<0] = 0,
 * w_k = lorahip_channelizer_phase_inc(freq[k]), D = decim, L = n_taps, h = taps):
 *
 *     n_m    = (m + 1) D - 1                                  (every D new samples make one output)
 *     y_k[m] = sum_{j<L} h[j] x[n_m - j] exp(-2 pi i frac(w_k (n_m - j) / 2^64))
 *
 * i.e. mix channel k's centre frequency (freq[k] cycles per input sample) down to 0, low-pass with h, keep every
 * D-th sample. Evaluated in fp32 (fused multiply-add) on taps pre-rotated in double; the phase is a 64-bit
 * counter, so there is no drift and a stream cut into arbitrary chunks gives bit-identical outputs to one call.
 * run(): consumes n_in samples, writes *n_out = lorahip_channelizer_out_count(c, n_in) samples per channel at
 * out_dev + k*out_stride (complex64, out_stride in samples >= *n_out). decim*(256 + n_taps/decim) samples must
 * fit the LDS (decim <= 64 for short filters).
 * ------------------------------------------------------------------------------------- */
typedef struct lorahip_channelizer lorahip_channelizer;
uint64_t lorahip_channelizer_phase_inc(double freq);          /* floor(frac(freq) * 2^64) */
int lorahip_channelizer_create(lorahip_channelizer **out, lorahip_ctx *ctx, size_t n_channels, const double *freq,
                               size_t decim, const float *taps, size_t n_taps);
void lorahip_channelizer_destroy(lorahip_channelizer *c);
int lorahip_channelizer_reset(lorahip_channelizer *c);
size_t lorahip_channelizer_out_count(const lorahip_channelizer *c, size_t n_in);
int lorahip_channelizer_run(lorahip_channelizer *c, const float *wide_dev, size_t n_in, float *out_dev,
                            size_t out_stride, size_t *n_out);
/* A batch of independent captures in one launch (recordings, antennas): capture s is the n_in samples at wide_dev + s*capture_stride,
 * processed like a fresh stream (from sample 0, zero history; bit-identical to reset() + run() on it); its channel k goes to
 * out_dev + (s*n_channels + k)*out_stride, *n_out = n_in / decim samples each. The object's own stream state is not touched.
 * n_captures <= 65535. */
int lorahip_channelizer_run_captures(lorahip_channelizer *c, const float *wide_dev, size_t n_captures, size_t capture_stride,
                                     size_t n_in, float *out_dev, size_t out_stride, size_t *n_out);

/* Measurement aid: one read-only streaming pass over n_bytes of device memory (pattern 0: linear
 * 16 B per lane; 1: the access shape of the tuned SF7 kernel). Time it with lorahip_timer_*; the
 * result is the practical HBM ceiling the roofline fraction can be compared with. */
int lorahip_membw_probe(lorahip_ctx *ctx, const float *buf_dev, size_t n_bytes, int pattern, int blocks_per_cu);

#ifdef __cplusplus
}
#endif
#endif /* LORAHIP_H */

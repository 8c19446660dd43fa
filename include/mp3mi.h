/* libmp3mi -- MI355X-native MPEG-1 Layer III encoding hot path, C ABI.
 *
 * Two surfaces:
 *
 * 1. The batched API (mp3mi_batch_*): thousands of independent streams per call, device
 *    pointers in, device pointers out.  This is where the throughput is.
 *
 * 2. The reference's own per-frame call surface (section "drop-in symbols" below): the seven
 *    external-linkage functions that the reference's driver calls for Layer III
 *    (/root/reference/src/musicin.c:751-786, 803), with identical names, argument meaning and
 *    in-place side effects, backed by one hidden default stream that runs the same kernels with
 *    n_streams = 1.  A maintainer links musicin.o + common.o against libmp3mi.so instead of
 *    l3psy.o encode.o(filterbank part) mdct.o loop.o l3bitstream.o -- see INTEGRATION.md.
 *
 * Every function fails loudly (non-zero return / abort with a message for the void drop-in
 * symbols, like the reference's exit()/abort()) when no gfx950 device is usable: there is no
 * CPU fallback in this library.
 *
 * Threads ("one thread or process per device"):
 *   - A batch object (mp3mi_batch, mp3mi_l12_batch) belongs to ONE thread at a time: calls on the same object must not
 *     overlap; the caller serialises them (any thread may make the next call once the previous one has returned --
 *     every entry point selects the batch's own device for its duration and restores the caller's).
 *   - DIFFERENT batch objects are independent: they may be created, used and destroyed concurrently from different
 *     threads, on the same device or on different ones.  A batch owns its HIP streams, events and device buffers;
 *     the only library-wide state is the construction of the constant tables, which the library serialises itself
 *     (csrc/tables_host.cpp).  Two batches on one device share its compute units: correct, each at a part of
 *     the rate (tests/test_threads.py: two threads, two batches, one device, interleaved calls, both bit-exact).
 *   - mp3mi_encode_host / mp3mi_encode_host_ex / mp3mi_encode_host_async create a batch of their own per call: reentrant.
 *   - The drop-in symbols (section 2 above) keep the reference's contract: ONE stream per process, one caller at a
 *     time -- the reference's own functions hold their state in function statics (src/l3psy.c:130-160, src/loop.c:240).
 *   - The mp3mi_debug_* accessors of diagnostic builds read device-global counters: single-threaded use only.
 */
#ifndef MP3MI_H
#define MP3MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ batched API */

typedef struct mp3mi_batch mp3mi_batch;

enum {
    MP3MI_OK = 0,
    MP3MI_ERR_ARG = -1,       /* unsupported rate / bitrate / channel count (what the reference refuses) */
    MP3MI_ERR_NO_DEVICE = -2, /* no usable GPU */
    MP3MI_ERR_HIP = -3,       /* a HIP call failed; message on stderr */
    MP3MI_ERR_NOMEM = -4,
    MP3MI_ERR_TABLES = -5,    /* the init tables do not hash to their pinned values (csrc/tables_pins.h): a damaged build;
                                 the bitstream would not be bit-exact, so nothing is encoded */
    MP3MI_ERR_REFERENCE_ABORT = -6 /* mp3mi_batch_sync / mp3mi_encode_host: the work completed, but for at least one stream the
                                 REFERENCE would have died on the input (an assertion of its code fails); that stream's
                                 out_len is 0, every other stream's output is valid -- see mp3mi_batch_stream_status */
};

/* Scheduling and scratch options of a batch.  The defaults are right for production use; the fields exist for
 * measurements (tools/) and tests.  Zero-initialise, set struct_size = sizeof, or call mp3mi_batch_options_default. */
typedef struct mp3mi_batch_options {
    uint32_t struct_size;     /* sizeof(mp3mi_batch_options) of the caller's build */
    uint32_t scratch_mb;      /* budget of the per-chunk scratch buffers in MiB (~78 KB per frame and stream); 0 = 32768 */
    int32_t chunk_frames;     /* upper limit of a chunk's length in frames; 0 = whatever the budget allows */
    uint32_t test_flags;      /* MP3MI_TEST_*: force the exact tier of the two-tier decisions (as mp3mi_batch_set_test_flags) */
    int32_t call_overlap;     /* a call's feed-forward kernels may start beside the loop kernels of the call before: -1 default (on), 0, 1 */
    int32_t gate;             /* the start census that orders the two HIP streams' kernels on the chip: -1 default (on), 0, 1 */
    int32_t placement;        /* streams placed on SIMDs by their cost in the chunk before: -1 default (on for >= 2 streams per SIMD), 0, 1 */
    int32_t loop_part_streams; /* streams per part (multiple of 64); 0 = the resident wavefronts of the device */
    int32_t y_after_loop;     /* the filterbank / MDCT / prep kernels of a chunk wait for the loop kernel before it: -1 default (only for a part larger than the resident wavefronts), 0, 1 */
    int32_t psy_beside;       /* what of the psychoacoustic stage runs beside a loop kernel: -1 default (k_cw + k_part + k_psy beside a
                                 resident loop kernel), 0 nothing, 1 k_cw + k_part + k_psy, 2 k_psy only */
    int32_t dropin_lookahead; /* the drop-in symbols' look-ahead (mp3mi_dropin.h): -1 default = 2 the filterbank's (and mdct_sub behind it: memory of the current frame only), 0 none, 1 all (buffer lifetime requirement: mp3mi_dropin.h),
                                 3 L3psycho_anal's only, 4 all but iteration_loop's / III_format_bitstream's.  Not a property of a batch: the hidden default stream of the drop-in symbols
                                 reads it through mp3mi_batch_options_from_env (MP3MI_DROPIN_LOOKAHEAD) */
    int32_t call_hold;        /* the LAST loop kernel of a call waits (on the device, at most 20 ms -- 0.4 ms per frame of a chunk where that is more, at most 200 ms --) until the call after it has run its first
                                 transforms, or until a call that waits for results lets it go (sync, stream_status, timing, flush, reset,
                                 destroy): calls issued back to back then lose no pipeline fill: -1 default (on), 0, 1 */
    int32_t dropin_stats;     /* the drop-in symbols print, at III_FlushBitstream, the frames they served, the time from the first frame's
                                 first call to the flush and the waits for the device: 0 default, 1 (MP3MI_DROPIN_STATS) */
    uint32_t abi;             /* MP3MI_OPTIONS_ABI of the header the caller was built against.  The struct's size alone does not tell two
                                 layouts apart (round 5 replaced a field in the middle and kept the size): a caller built against another layout is
                                 refused (MP3MI_ERR_ARG) instead of having its fields read as their neighbours */
} mp3mi_batch_options;
#define MP3MI_OPTIONS_ABI 6u  /* raised whenever the struct's layout or a field's meaning changes; new fields go at the END */
/* mp3mi_batch_create_ex returns MP3MI_ERR_ARG for a value outside the ranges named above (the three-state fields take
 * -1, 0, 1; psy_beside -1 .. 2; dropin_lookahead -1 .. 4; loop_part_streams a multiple of 64; unknown test flags). */
void mp3mi_batch_options_default(mp3mi_batch_options *opt);
/* The same, then overridden by the MP3MI_* environment variables that tools/ and tests/ use (MP3MI_SCRATCH_MB,
 * MP3MI_CHUNK_FRAMES, MP3MI_{NOISE,PHASE,PSY,QUANT,PREP,CW}_EXACT, MP3MI_CALL_OVERLAP, MP3MI_NO_GATE, MP3MI_NO_PLACE,
 * MP3MI_LOOP_PART_STREAMS, MP3MI_CALL_HOLD, MP3MI_Y_AFTER_LOOP, MP3MI_PSY_BESIDE, MP3MI_DROPIN_LOOKAHEAD, MP3MI_DROPIN_STATS).  This is the ONLY place the library
 * reads its environment: mp3mi_batch_create calls it once; mp3mi_batch_create_ex never does. */
void mp3mi_batch_options_from_env(mp3mi_batch_options *opt);

/* Creates an encoder for n_streams independent streams that share sample rate and channel
 * count.  rate_hz in {44100, 48000, 32000}; channels 1 or 2; kbps points to n_streams MPEG-1
 * Layer III bitrates (32..320) or is NULL, in which case every stream uses kbps_all.
 * max_frames bounds n_frames of later encode calls.  Replaces the set-up part of
 * /root/reference/src/musicin.c:456-581 (parse_args defaults, hdr_to_frps, slots per frame). */
int mp3mi_batch_create(mp3mi_batch **out, int n_streams, int rate_hz, int channels,
                       const int *kbps, int kbps_all, int max_frames);
/* The same with explicit options (NULL = the defaults); reads no environment variable. */
int mp3mi_batch_create_ex(mp3mi_batch **out, int n_streams, int rate_hz, int channels,
                          const int *kbps, int kbps_all, int max_frames, const mp3mi_batch_options *opt);
void mp3mi_batch_destroy(mp3mi_batch *b);

/* Bytes to reserve per stream in the output buffer for n_frames frames. */
size_t mp3mi_batch_out_stride(const mp3mi_batch *b, int n_frames);

/* Encodes n_frames whole frames of every stream, from a fresh encoder state, including the
 * final flush (III_FlushBitstream + close_bit_stream_w, musicin.c:802-805).
 *   pcm_dev     device pointer, int16, [n_streams][n_frames*1152][channels] (WAV sample order)
 *   out_dev     device pointer, [n_streams][out_stride] bytes
 *   out_len_dev device pointer, [n_streams] uint32: bytes produced per stream
 * Work is enqueued on the batch's stream; call mp3mi_batch_sync before reading results. */
int mp3mi_batch_encode(mp3mi_batch *b, const int16_t *pcm_dev, int n_frames, uint8_t *out_dev,
                       size_t out_stride, uint32_t *out_len_dev);
int mp3mi_batch_sync(mp3mi_batch *b);

/* Inputs the reference DIES on.  A few assertions of the reference's Layer III code fail on real inputs
 * (tests/golden/coverage_notes.json, "reference_aborts", with the fixtures that reach them):
 *   MP3MI_STREAM_ABORT_GLOBAL_GAIN  assert( cod_info->global_gain < 256 ), /root/reference/src/loop.c:358 -- a granule with
 *                                   exact-zero lines beside a few tiny ones (a click behind digital silence)
 *   MP3MI_STREAM_ABORT_HUFF_BITS    assert( max_bits >= 0 ) in inner_loop, src/loop.c:579 -- scalefactor bits above the
 *                                   granule's budget (48 kHz, 32 kbps, stereo, short blocks)
 *   MP3MI_STREAM_ABORT_FLUSH_SLOT   assert( l ) in get_side_info, src/formatBitstream.c:390, reached from BF_FlushBitstream
 *                                   when the last main data ends exactly on a slot boundary with headers still queued
 * The reference's process ends there and leaves no usable file.  A batch cannot end for one stream: the stream's status
 * records the first such event, its out_len becomes 0 for that call and every later one, the other streams are not
 * affected, and mp3mi_batch_sync returns MP3MI_ERR_REFERENCE_ABORT once -- the first sync after the call in which the
 * event happened (whole-file and streaming calls alike; the final flush reports what it finds itself), while the
 * status is there to be read.  (The drop-in symbols abort() with the reference's message, as the reference does.)
 * mp3mi_batch_stream_status waits for the work issued so far and copies the status of every stream of the most recent
 * streams (since the last reset / whole-file call; after mp3mi_batch_flush: of the streams it ended, until the next
 * encode starts new ones) to status_host[n_streams]: 0, or code | frame << 8 where frame is the
 * index of the frame it happened in (the number of frames for the final flush; the field is 22 bits wide and saturates
 * at 4194303, ~30 h of audio fed call by call).  Returns the number of streams with a
 * non-zero status, or a negative MP3MI_ERR_*. */
enum { MP3MI_STREAM_OK = 0, MP3MI_STREAM_ABORT_GLOBAL_GAIN = 1, MP3MI_STREAM_ABORT_HUFF_BITS = 2, MP3MI_STREAM_ABORT_FLUSH_SLOT = 3 };
int mp3mi_batch_stream_status(mp3mi_batch *b, int32_t *status_host);

/* Streaming: the reference is a frame-streaming encoder (/root/reference/src/musicin.c:585-805); these calls encode
 * a stream piece by piece with everything it carries from frame to frame kept in the batch (psychoacoustic
 * history and thresholds, the filterbank's and the FFT window's past samples, the bit reservoir, the main data
 * that is formatted but whose slot is still open).
 *   mp3mi_batch_encode_next  encodes the NEXT n_frames frames of every stream: pcm_dev holds only these frames,
 *                            [n_streams][n_frames*1152][channels].  out_dev / out_len_dev receive, per stream, the
 *                            file bytes that became final with this call -- in file order, so concatenating the
 *                            outputs of successive calls and of the final flush gives the stream's file.  (Up to
 *                            511 bytes of main data, plus the headers in between, stay behind in the reservoir's
 *                            open slots until later frames fill them: src/formatBitstream.c:52-120.)  The first call
 *                            after create, reset, flush or a whole-file encode starts new streams.
 *   mp3mi_batch_flush        III_FlushBitstream + close_bit_stream_w (musicin.c:802-805): delivers what is left and
 *                            ends the streams.  out_stride >= 2049 here.
 *   mp3mi_batch_reset        abandons the current streams: fresh encoder state.
 * mp3mi_batch_encode (above) is the same encoder run over a whole stream in one call.  Ragged batches and
 * streaming do not combine: a stream's last partial frame is zero-filled by the caller (src/encode.c:162-166).
 * out_stride >= mp3mi_batch_out_stride(b, n_frames). */
int mp3mi_batch_encode_next(mp3mi_batch *b, const int16_t *pcm_dev, int n_frames, uint8_t *out_dev,
                            size_t out_stride, uint32_t *out_len_dev);
int mp3mi_batch_flush(mp3mi_batch *b, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev);
int mp3mi_batch_reset(mp3mi_batch *b);

/* Ragged batch: stream s has n_samples_dev[s] valid samples per channel (0 <= n <= n_frames*1152) in
 * its row of pcm_dev (row pitch n_frames*1152*channels as above).  As the reference's get_audio /
 * read_samples do (/root/reference/src/encode.c:123-269, zero fill :162-166), the last partial frame
 * is zero-filled and the stream ends after ceil(n/1152) frames; out_len_dev[s] is its own file
 * length (0 for a stream without samples). */
int mp3mi_batch_encode_ragged(mp3mi_batch *b, const int16_t *pcm_dev, const int32_t *n_samples_dev, int n_frames,
                              uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev);

/* Header bits the reference's driver sets from -c, -o and -d (/root/reference/src/musicin.c:263-275,
 * written at src/l3bitstream.c:330-334): copyright 0/1, original 0/1, emphasis 0..3.  Applies to later
 * encode calls of this batch. */
int mp3mi_batch_set_header(mp3mi_batch *b, int copyright, int original, int emphasis);

/* Header mode field, -m of the reference's driver (/root/reference/src/musicin.c:226-234, src/common.h:233-236):
 * stereo or dual channel for two-channel batches, mono for one-channel ones (the default follows the channel
 * count).  Dual channel changes nothing but the header field -- the reference's Layer III encoder treats the two
 * channels independently in every mode.  Joint stereo is refused (MP3MI_ERR_ARG) as the reference refuses it for
 * Layer III (src/musicin.c:548-552). */
enum { MP3MI_MODE_STEREO = 0, MP3MI_MODE_JOINT_STEREO = 1, MP3MI_MODE_DUAL_CHANNEL = 2, MP3MI_MODE_MONO = 3 };
int mp3mi_batch_set_mode(mp3mi_batch *b, int mode);

/* Error protection, -e of the reference's driver: the protection bit of the header is cleared and a 16-bit CRC
 * word follows the header, which the reference never computes for Layer III and writes as zero
 * (/root/reference/src/l3bitstream.c:312, 325, 338-342); the side information grows by 16 bits, so the mean bits
 * per granule shrink (src/musicin.c:744-746).  Reproduced bit for bit -- including the zero CRC. */
int mp3mi_batch_set_error_protection(mp3mi_batch *b, int on);

/* Milliseconds spent inside the dominant (iteration loop) kernel and inside all kernels during
 * the last encode call, measured with HIP events on the batch's stream. */
int mp3mi_batch_last_timing(mp3mi_batch *b, float *loop_kernel_ms, float *all_kernels_ms,
                            int *loop_kernel_launches);

/* The same summed over all encode calls since the batch was created (waits for the calls issued so far): calls
 * may be issued back to back without a sync in between -- a call's feed-forward kernels then run beside the loop
 * kernels of the call before -- and their timing read once at the end. */
int mp3mi_batch_total_timing(mp3mi_batch *b, double *loop_kernel_ms, double *all_kernels_ms,
                             long *loop_kernel_launches, long *calls);

/* Host buffers in, host buffers out, OVERLAPPED with the encode -- what the reference's driver does frame by frame with
 * get_audio / read_samples and fwrite (/root/reference/src/encode.c:123-269, src/common.c:843-868), for a whole batch:
 * the call's PCM crosses PCIe chunk by chunk on a copy stream while the chunks before it are encoded, and the call's
 * file bytes come back in one copy behind its last formatter; with calls issued back to back the next call's PCM goes up
 * and this call's bytes come down beside the next call's kernels (two calls may be in flight; the batch keeps two
 * device copies of PCM and output).  out_stride = mp3mi_batch_out_stride(b, max_frames) makes the download one plain copy.
 *   pcm_host: [n_streams][n_frames*1152][channels]; out_host: [n_streams][out_stride], out_stride >= n_frames * the largest
 *   frame size + 1; out_len_host: [n_streams].  A whole-file call like mp3mi_batch_encode: every stream starts afresh.
 * Asynchronous: returns once everything is enqueued; the buffers must stay valid -- and the results are there -- when
 * mp3mi_batch_sync returns (which reports MP3MI_ERR_REFERENCE_ABORT as for device calls).  Page-locked buffers
 * (mp3mi_host_alloc, hipHostMalloc, hipHostRegister) move at the full PCIe rate beside the kernels; pageable memory
 * works, but the runtime stages it and the call blocks while it does.  bench.py --host-io measures this path. */
int mp3mi_batch_encode_host_async(mp3mi_batch *b, const int16_t *pcm_host, int n_frames, uint8_t *out_host, size_t out_stride,
                                  uint32_t *out_len_host);
/* Bytes moved and time spent inside the copies (HIP events on the two copy streams) over all host-buffer calls since the
 * batch was created; waits for the calls issued so far. */
typedef struct mp3mi_host_io_stats {
    double h2d_bytes, d2h_bytes; /* PCM up, file bytes down */
    double h2d_ms, d2h_ms;       /* summed durations of the copies */
    long calls;
} mp3mi_host_io_stats;
int mp3mi_batch_host_io_stats(mp3mi_batch *b, mp3mi_host_io_stats *st);
/* page-locked host memory for the call above, for callers that do not link HIP themselves; NULL on failure */
void *mp3mi_host_alloc(size_t bytes);
void mp3mi_host_free(void *p);

/* Host-buffer convenience wrapper (tests, smoke): a batch of its own per call, mp3mi_batch_encode_host_async, sync.
 * pcm: [n_streams][n_frames*1152*channels]; out: [n_streams][out_stride]; out_len: [n_streams]. */
int mp3mi_encode_host(int n_streams, int rate_hz, int channels, const int *kbps, int kbps_all,
                      const int16_t *pcm, int n_frames, uint8_t *out, size_t out_stride,
                      uint32_t *out_len);

/* The same with per-stream sample counts (n_samples, host, may be NULL) and header bits
 * (mp3mi_batch_encode_ragged, mp3mi_batch_set_header). */
int mp3mi_encode_host_ex(int n_streams, int rate_hz, int channels, const int *kbps, int kbps_all,
                         const int16_t *pcm, const int32_t *n_samples, int n_frames, int copyright,
                         int original, int emphasis, uint8_t *out, size_t out_stride, uint32_t *out_len);

/* Stage seams of the LAST chunk of the last encode call, copied to host memory (parity tests
 * compare them with oracle/stage_dump.h).  what: 0 = psychoacoustic records (mp3mi_psy_out),
 * 1 = xr (f64[576] per granule-channel), 2 = quantised values (int16[576]), 3 = side info
 * (mp3mi_frame_side per frame), 4 = raw subband samples (f64[576], enabled by
 * mp3mi_batch_debug_enable), 5 = the loop's stateless head (csrc/mp3mi_dev.h: mp3mi_loop_prep, 472 bytes per
 * granule-channel); the psychoacoustic transforms' outputs (src/subs.c:38-123): 6 = long energies (f32, rows of 544
 * per granule-channel: lines 0..512, the rest padding), 7 = short energies (f32[3][129]), 8 = raw lines (f32[312]:
 * (re, im) of short lines 2..51 of the three windows, then re[6], im[6] of long lines 0..5).  Returns the number of
 * bytes written, or a negative error. */
long mp3mi_batch_debug_fetch(mp3mi_batch *b, int what, void *host_dst, size_t cap);
void mp3mi_batch_debug_enable(mp3mi_batch *b, int on);

/* Two-tier decisions: six places decide from a cheap value with a proven error bound and
 * repeat the computation the reference's way only when the decision could depend on the last bits.  Each bit
 * forces the second (exact) tier everywhere; the emitted bytes must not change (tests/test_gpu_tiers.py).  The
 * environment variables MP3MI_{NOISE,PHASE,PSY,QUANT,PREP}_EXACT=1 set the same bits at mp3mi_batch_create. */
enum {
    MP3MI_TEST_NOISE_EXACT = 1,  /* calc_noise: the reference's sequential band sums (k_loop) */
    MP3MI_TEST_PHASE_EXACT = 2,  /* phases: correctly rounded atan2 (k_cw) */
    MP3MI_TEST_PSY_EXACT = 4,    /* masking threshold: dm_log / dm_exp (k_psy) */
    MP3MI_TEST_QUANT_EXACT = 8,  /* quantiser: every line against the (i - 0.4054)^(4/3) table (k_loop) */
    MP3MI_TEST_PREP_EXACT = 16,  /* the loop's stateless head by k_prep for every record, the reference's walk with correctly
                                    rounded logs, instead of k_mdct's tail */
    MP3MI_TEST_CW_EXACT = 32,    /* unpredictability: correctly rounded sines and cosines for every record (k_cw) */
    MP3MI_TEST_ALL_EXACT = 63,
    MP3MI_TEST_PREP_LIST = 64    /* not a tier: k_mdct's tail lists every third record as undecided and spoils what it wrote for
                                    it, so that the loop reads what k_prep computed through the list (the path of the ~1e-7
                                    records the tail really cannot decide) */
};
int mp3mi_batch_set_test_flags(mp3mi_batch *b, unsigned flags);

/* Deterministic synthetic PCM (benchmarks / tests): interleaved int16, n_per_ch samples per channel. */
void mp3mi_synth_pcm(int16_t *out, long n_per_ch, int channels, int rate_hz, uint32_t stream,
                     uint32_t seed);

/* Self-test hook: evaluates function fn of the device math layer (see csrc/k_debug.hip) on the
 * GPU for n host-side arguments.  Used by tests to prove device == host bit for bit. */
int mp3mi_debug_dmath(int fn, const double *x, const double *y, double *out, size_t n);

/* Self-test hook: maximum relative error, on this device, of the three approximate expressions the quantiser's
 * first tier is built from (csrc/k_debug.hip): out[0] raw sqrt(a * raw sqrt(a)) vs a^(3/4) over all 2^24 floats
 * of [1, 4); out[1] raw exp2 at the 801 step sizes the search can ask for; out[2] raw exp2 over 2^24 arguments
 * of [-80, 80].  tests/test_gpu_tiers.py asserts that their sum stays inside the guard band's 7e-7 budget. */
int mp3mi_debug_fastmath_bounds(double out[3]);
/* What v_cvt_pknorm_u16_f32 -- the rounding step of the quantiser's first tier -- returns on this device, over every float of the
 * range the quantiser feeds it (csrc/k_debug.hip): out[0] = max |n - a * 65535| (the proof in csrc/k_loop.hip needs <= 0.5),
 * out[1] = non-monotone neighbours, out[2] = clamp / half mismatches (both 0). */
int mp3mi_debug_pknorm_bound(double out[3]);
/* diagnostics: of the (granule, channel) records of the last call's LAST chunk, how many needed the second tier of
 * the unpredictability (k_part's check); *n_records receives their number.  Call after
 * mp3mi_batch_sync. */
int mp3mi_batch_debug_cw_fixups(mp3mi_batch *b, int *n_listed, int *n_records);
/* likewise: the records of the last item whose loop-prep values k_mdct's tail could not decide (k_prep recomputed them) */
int mp3mi_batch_debug_prep_fixups(mp3mi_batch *b, int *n_listed);

/* Library / device identification string for logs. */
const char *mp3mi_version(void);
/* sha256 (first 16 hex digits) over the sources the library was built from (csrc/Makefile): measurements carry it, and
 * bench.py only quotes counter figures of a committed profile that was taken on the same sources. */
const char *mp3mi_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif

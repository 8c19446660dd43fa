/* Layers I and II: device-side declarations -- table block, geometry of a chunk, record layouts.
 *
 * The path (reference: src/musicin.c:620-704, src/encode.c:419-1444, src/psy.c) is feed-forward: a frame depends on
 * the frames before it only through PCM history -- the filterbank's 480 taps, the FFT window, and the two (Layer I:
 * three) earlier FFT passes whose magnitudes and phases predict the current one.  There is no bit reservoir.  So
 * (stream, frame) is the parallel axis everywhere and a chunk recomputes the passes just before it instead of
 * carrying state.
 *
 *   k_fft12      (k_fft.hip)  the 1024-point FFT of every pass (stream, pass q, all channels), and from the spectrum
 *                             energy, r = sqrt(energy), phi = (float) atan2 of every line      src/subs.c:38-123, src/psy.c:258-286
 *   k12_psy                   unpredictability, partitions, spreading, masking, thresholds,
 *                             signal-to-mask ratio of the 32 subbands                          src/psy.c:282-386
 *   k_filter     (k_fbmdct.hip, as for Layer III) subband samples
 *   k12_alloc    (k_l12.hip)  scale factors, transmission pattern, joint-stereo bound, bit allocation, CRC,
 *                             quantisation and the frame's bits                                 src/encode.c:512-1416
 *
 * A PASS is one call of the psychoacoustic model's inner loop (src/psy.c:247): Layer II runs two per frame (576 new
 * samples each), Layer I one (384).  Pass q of a stream (q = frame * layer + i) sees savebuf[0 .. 1023] = the samples
 * [spp (q + 1) - span, spp (q + 1) - span + 1024) with spp = 576, span = 1056 (Layer II) or 384, 1024 (Layer I).
 */
#ifndef MP3MI_L12_DEV_H
#define MP3MI_L12_DEV_H

#include "mp3mi_dev.h"

#define L12_HBLK 513
#define L12_ROW 520  /* row pitch of the per-line arrays in floats (16-byte multiples) */
#define L12_CB 63
#define L12_PCM_HIST 1792 /* samples per channel a streaming call needs from before its first sample: the FFT window of the third
                             pass before it (Layer I: 2 * 384 + 1024; Layer II: 576 + 1056 = 1632) */

/* read-only tables, one block in device memory per batch (tables_host.cpp: mp3mi_build_tables_l12) */
typedef struct {
    int32_t rate_idx, layer, npart, pad0;
    /* psychoacoustic model 2, src/psy.c:151-228 */
    float spread_r[64][64];       /* spread_r[j][k] = s[j][k]: lane j of k12_psy keeps its row in registers */
    float cbval[64], rnorm[64], bmaxv[64]; /* bmaxv[j] = bmax[(unsigned) (cbval[j] + 0.5)] */
    float rn_nl[64];              /* rnorm[j] * numlines[j] (float product, src/psy.c:347), 0 where the reference takes nb = 0 */
    double tmn[64];
    float absthr[L12_ROW];
    int16_t part_first[64 + 1];   /* first line of partition b; [npart] = 513 */
    uint8_t partition[L12_ROW];
    /* scale factors, quantisation, allocation: src/common.c:127-150, src/encode.c:777-780, 1195-1205 */
    double multiple[64];
    double snr[18], qa[17], qb[17]; /* as the layer's first frame leaves them (Layer I rearranges: src/encode.c:900-905, 1226-1231) */
    uint16_t alloc[4][32][16][4];   /* Tables B.2a-d: {steps, bits, group, quant}; Layer I does not use it */
    int32_t sblimit[4];
} mp3mi_tables_l12;

/* a chunk of a call: frames [f0, f0 + nf) of every stream */
struct l12_geom {
    int n_streams, channels, layer, rate_idx;
    int n_frames;            /* frames of the call (rows of the PCM buffer hold n_frames * spf samples per channel) */
    int f0, nf;
    int lb;                  /* passes before the chunk's first that are recomputed: 2 (Layer II), 3 (Layer I) */
    int np;                  /* passes of the chunk incl. the lb before it: nf * layer + lb */
    int spf, spp, span;      /* samples per frame / per pass, and the span of savebuf: 1152 576 1056 | 384 384 1024 */
    int g0, n_gran;          /* what k_filter computed for this chunk: 18-slot granules [g0 - 1, g0 + n_gran) of the UNDELAYED
                                filterbank (slot n consumes samples [32 n, 32 n + 32)) */
    int slot0;               /* filterbank slot of the chunk's first frame's first slot, relative to slot 0 of granule g0 - 1 */
    int actual_mode;         /* the driver's -m: 0 stereo, 1 joint stereo, 2 dual channel, 3 mono */
    int crc, hdr_flags;      /* error protection; bit 3 copyright, bit 2 original, bits 1-0 emphasis */
    int test_flags;          /* MP3MI_TEST_PHASE_EXACT, _PSY_EXACT, _CW_EXACT: the second tier everywhere (tests) */
    const int32_t *n_samples; /* device, [n_streams]: valid samples per channel (ragged batch) or NULL */
    /* streaming (mp3mi_l12_batch_encode_next): the call continues streams that earlier calls began */
    long fabs0;              /* frames of every stream encoded by earlier calls */
    const int16_t *hist;     /* device, [n_streams][L12_PCM_HIST][channels]: the samples before the call's first (zeros at the
                                start of a stream), or NULL */
    int whole_file;          /* the call is the whole stream: the last frame's wavefront adds the file's last byte */
};

/* per stream, device: what depends on the stream's bitrate */
struct l12_stream_cfg {
    int32_t bitrate_index, frame_bits, table, sblimit;
};

/* per frame, device: the seams tests compare with oracle/stage_dump_l12.h (written only when the batch's debug
 * switch is on) */
struct l12_frame_dbg {
    double ltmin[2][32];
    int32_t scalar[2][3][32], j_scale[3][32], scfsi[2][32], bit_alloc[2][32];
    int32_t mode, mode_ext, jsbound, sblimit, adb_left, crc, pad[2];
};

#ifdef __cplusplus
void mp3mi_launch_fft12(const mp3mi_tables *T, const l12_geom &g, const int16_t *pcm, float *erp, hipStream_t st);
void mp3mi_launch_l12_psy(const mp3mi_tables_l12 *T, const l12_geom &g, const float *erp, float *snr, hipStream_t st);
void mp3mi_launch_l12_alloc(const mp3mi_tables_l12 *T, const l12_geom &g, const l12_stream_cfg *cfg, const double *sbs,
                            const float *snr, uint8_t *out, size_t out_stride, uint32_t *out_len, l12_frame_dbg *dbg, hipStream_t st);
void mp3mi_launch_l12_hist_save(const l12_geom &g, const int16_t *pcm, const int16_t *hist_in, int16_t *hist_out, const int16_t *fb_in,
                                int16_t *fb_out, hipStream_t st);
void mp3mi_launch_l12_flush(int n_streams, uint8_t *out, size_t out_stride, uint32_t *out_len, hipStream_t st);
extern "C" int mp3mi_build_tables_l12(mp3mi_tables_l12 *T, int rate_idx, int layer);
#endif

#endif

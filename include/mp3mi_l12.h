/* libmp3mi -- MPEG-1 Layer I and Layer II encoding on the MI355X, C ABI.
 *
 * The batched counterpart of the Layer I / II cases of the reference's frame loop
 * (/root/reference/src/musicin.c:620-704) with psychoacoustic model 2 (-p 2, the default; src/psy.c):
 *
 *   reference call (per frame, per stream)                              here (per batch)
 *   get_audio                         src/encode.c:187-269              the caller's PCM rows, read on the device
 *   window_subband / filter_subband   src/encode.c:287-409              k_filter (shared with Layer III)
 *   I_/II_scale_factor_calc, I_/II_combine_LR, II_transmission_pattern  src/encode.c:469-691          k12_alloc
 *   psycho_anal                       src/psy.c:36-421                  k_fft12, k12_psy
 *   I_/II_main_bit_allocation         src/encode.c:782-1172             k12_alloc
 *   I_/II_CRC_calc, encode_info, encode_CRC, *_encode_bit_alloc, *_encode_scale,
 *   *_subband_quantization, *_sample_encoding, put1bit                  src/common.c:1251-1327, src/encode.c:418-437, 695-748, 1174-1430   k12_alloc
 *   close_bit_stream_w                src/common.c:968                  the file's one byte past the last frame
 *
 * The emitted bytes are bit-exact to the reference's `encode -l 1|2` on the same PCM (tests/test_gpu_l12.py,
 * tests/golden/l12_*).  Psychoacoustic model 1 (-p 1) cannot be offered: the reference as shipped cannot run it
 * (src/tonal.c:86-150 reads table files -- "2cb0", "2th0", ... -- that the repository does not contain, and exits).
 * MPEG-2 LSF rates are refused as the reference's psycho_anal refuses them (src/psy.c:131-136).  Frames are never
 * padded (src/musicin.c:566-581 drops the fraction before looking at it).
 *
 * No CPU fallback: every entry point returns MP3MI_ERR_NO_DEVICE when HIP has no device.  Error codes: mp3mi.h.
 * Threads: as for mp3mi_batch (mp3mi.h, "Threads"): one thread at a time per mp3mi_l12_batch, different batches independent.
 */
#ifndef MP3MI_L12_H
#define MP3MI_L12_H

#include "mp3mi.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mp3mi_l12_batch mp3mi_l12_batch;

/* layer 1 or 2; rate_hz in {44100, 48000, 32000}; channels 1 or 2; kbps: n_streams bitrates of the layer's table
 * (src/common.c:122-123: Layer I 32..448 in steps of 32, Layer II 32 48 56 64 80 96 112 128 160 192 224 256 320 384) or
 * NULL = kbps_all for every stream.  max_frames bounds n_frames of later calls; a frame is 384 (Layer I) or 1152
 * (Layer II) samples per channel.  scratch_mb: budget of the inter-kernel buffers in MiB, 0 = 32768. */
int mp3mi_l12_batch_create(mp3mi_l12_batch **out, int layer, int n_streams, int rate_hz, int channels,
                           const int *kbps, int kbps_all, int max_frames, unsigned scratch_mb);
void mp3mi_l12_batch_destroy(mp3mi_l12_batch *b);

/* the driver's -m (MP3MI_MODE_*; joint stereo IS available for these layers: src/encode.c:882-948), -e (a computed
 * CRC-16 here, src/common.c:1251-1327), -c -o -d */
int mp3mi_l12_batch_set_mode(mp3mi_l12_batch *b, int mode);
int mp3mi_l12_batch_set_error_protection(mp3mi_l12_batch *b, int on);
int mp3mi_l12_batch_set_header(mp3mi_l12_batch *b, int copyright, int original, int emphasis);

/* bytes to reserve per stream for n_frames frames */
size_t mp3mi_l12_batch_out_stride(const mp3mi_l12_batch *b, int n_frames);

/* Encodes n_frames frames of every stream from a fresh encoder state, the file's last byte included.
 *   pcm_dev     device, int16 [n_streams][n_frames * (384 | 1152)][channels] (WAV sample order)
 *   n_samples_dev  device, [n_streams] valid samples per channel of each stream, or NULL: all.  The last partial frame
 *               is zero-filled and the stream ends after ceil(n / frame) frames (src/encode.c:123-185)
 *   out_dev     device, [n_streams][out_stride] bytes;  out_len_dev  device, [n_streams] uint32
 * Work is enqueued on the batch's stream; mp3mi_l12_batch_sync waits for it. */
int mp3mi_l12_batch_encode(mp3mi_l12_batch *b, const int16_t *pcm_dev, const int32_t *n_samples_dev, int n_frames,
                           uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev);
int mp3mi_l12_batch_sync(mp3mi_l12_batch *b);

/* Streaming, as mp3mi_batch_encode_next / _flush / _reset of mp3mi.h: the reference is a frame-streaming encoder
 * (src/musicin.c:585-705), and for these layers a stream carries nothing from frame to frame but PCM history (the
 * filterbank's taps, the FFT windows of the passes that predict the next one) -- kept in the batch.
 *   mp3mi_l12_batch_encode_next  the NEXT n_frames frames of every stream: pcm_dev holds only these frames; out_dev /
 *                                out_len_dev receive their bytes (whole frames: nothing stays behind).  The first call after
 *                                create, reset, flush or a whole-file encode starts new streams.
 *   mp3mi_l12_batch_flush        close_bit_stream_w: the file's one byte past the last frame (src/common.c:843-868); ends the streams
 *   mp3mi_l12_batch_reset        abandons the streams in progress
 * Concatenating the outputs of the calls and of the flush gives the bytes of mp3mi_l12_batch_encode over the whole stream. */
int mp3mi_l12_batch_encode_next(mp3mi_l12_batch *b, const int16_t *pcm_dev, int n_frames, uint8_t *out_dev, size_t out_stride,
                                uint32_t *out_len_dev);
int mp3mi_l12_batch_flush(mp3mi_l12_batch *b, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev);
int mp3mi_l12_batch_reset(mp3mi_l12_batch *b);

/* Two-tier decisions (mp3mi.h, MP3MI_TEST_PHASE_EXACT | _PSY_EXACT | _CW_EXACT): force the exact tier; bytes must not change */
int mp3mi_l12_batch_set_test_flags(mp3mi_l12_batch *b, unsigned flags);

/* Stage seams of the LAST chunk of the last call for tests (oracle/stage_dump_l12.h): enable before encoding, then
 * fetch [n_streams][frames of that chunk] records of mp3mi_l12_frame_seams; returns bytes written or a negative error;
 * *first_frame receives the chunk's first frame. */
typedef struct mp3mi_l12_frame_seams {
    double ltmin[2][32];
    int32_t scalar[2][3][32], j_scale[3][32], scfsi[2][32], bit_alloc[2][32];
    int32_t mode, mode_ext, jsbound, sblimit, adb_left, crc, pad[2];
} mp3mi_l12_frame_seams;
void mp3mi_l12_batch_debug_enable(mp3mi_l12_batch *b, int on);
long mp3mi_l12_batch_debug_fetch(mp3mi_l12_batch *b, void *host_dst, size_t cap, int *first_frame, int *n_chunk_frames);

/* milliseconds inside the kernels of all encode calls since create (HIP events on the batch's stream), and the calls */
int mp3mi_l12_batch_total_timing(mp3mi_l12_batch *b, double *all_kernels_ms, long *calls);
/* the same per kernel, HIP events around every launch: [0] k_fft12, [1] k12_psy, [2] k_filter,
 * [3] k12_alloc -- milliseconds and launches since create */
int mp3mi_l12_batch_kernel_timing(mp3mi_l12_batch *b, double ms[4], long launches[4]);

/* Host-buffer convenience wrapper (tests, smoke): pcm [n_streams][n_frames * frame * channels], n_samples may be NULL;
 * mode MP3MI_MODE_* or -1 = by the channel count. */
int mp3mi_l12_encode_host(int layer, int n_streams, int rate_hz, int channels, const int *kbps, int kbps_all, int mode,
                          int error_protection, const int16_t *pcm, const int32_t *n_samples, int n_frames,
                          uint8_t *out, size_t out_stride, uint32_t *out_len);

#ifdef __cplusplus
}
#endif
#endif

/* Internal host-side declarations shared by the translation units of libmp3mi.so. */
#ifndef MP3MI_HOST_H
#define MP3MI_HOST_H

#include "mp3mi_dev.h"

#ifdef __cplusplus
extern "C" {
#endif

/* tables_host.cpp: fill the table block for MPEG-1 sampling_frequency code rate_idx
 * (0 = 44.1 kHz, 1 = 48 kHz, 2 = 32 kHz).  Returns 0 on success. */
int mp3mi_build_tables(mp3mi_tables *T, int rate_idx);
/* hashes of the table members as this host builds them (test / pin generation; tables_host.cpp) */
int mp3mi_tables_digest(int rate_idx, uint64_t *hashes, const char **names, int cap);

#ifdef __cplusplus
}

/* kernel launchers (one per .hip file); all take device pointers */
struct mp3mi_geom {
#if defined(MP3MI_ULP_CENSUS)
    size_t census_cb_stride; /* diagnostic build: the shadow sums of census site UC_CW_REACH sit this many floats (and twice as many) behind cb_all */
#endif
    int n_streams, channels, rate_idx;
    int n_frames;       /* frames per stream in this call (PCM extent = n_frames*1152 per channel) */
    int f0, nf;         /* frames [f0, f0+nf) form the current chunk */
    int g0, n_gran;     /* granules [g0, g0+n_gran) of the chunk: 2*f0, 2*nf for the batch path */
    int hdr_mode;       /* header mode field (src/common.h:233-236): 0 stereo, 2 dual, 3 mono */
    int hdr_flags;      /* bit 3 copyright, bit 2 original, bits 1-0 emphasis, bits 5-4 mode_ext */
    int crc;            /* error protection on: protection bit 0 and the 16-bit CRC word after the header -- which the
                           reference never computes for Layer III and writes as 0 (src/l3bitstream.c:312, 338-342); side
                           info grows by 16 bits (src/musicin.c:744-746) */
    const int32_t *n_samples; /* device, [n_streams]: valid samples per channel of each stream (ragged batch), or NULL: all n_frames*1152.
                                 Samples beyond it read as zero and frames beyond ceil(n/1152) are not encoded (src/encode.c:162-166) */
    /* streaming (mp3mi_batch_encode_next): the call continues streams that earlier calls began */
    long fabs0;         /* frames of every stream encoded by earlier calls = index of the call's first frame in its stream */
    const int16_t *hist; /* device, [n_streams][MP3MI_PCM_HIST][channels]: the samples before the call's first one (zeros at the
                           start of a stream), or NULL: nothing precedes the call */
    const int64_t *out_base; /* device, [n_streams]: file position of byte 0 of the stream's output row, or NULL: 0 */
    int whole_file;     /* the call is the whole stream: k_format also finishes the file (length incl. flush + close) */
    long pcm_pitch;     /* samples per channel in a stream's row of the PCM buffer; 0 = n_frames * 1152 (Layers I / II frames are
                           not 1152 samples: l12_batch.cpp sets it for k_filter) */
    int test_flags;     /* bit 0: k_loop takes the exact (sequential) noise sums only (MP3MI_NOISE_EXACT=1, tests);
                           bit 1: k_cw takes the correctly rounded atan2 only (MP3MI_PHASE_EXACT=1, tests);
                           bit 2: k_psy takes dm_log / dm_exp only (MP3MI_PSY_EXACT=1, tests);
                           bit 3: k_loop's quantiser takes the exact table search only (MP3MI_QUANT_EXACT=1, tests);
                           bit 4: k_cw takes the correctly rounded sines and cosines for every record (MP3MI_CW_EXACT=1, tests) */
};

static inline mp3mi_geom mp3mi_make_geom(int n_streams, int channels, int rate_idx, int n_frames, int f0, int nf)
{
    mp3mi_geom g;
    g.n_streams = n_streams; g.channels = channels; g.rate_idx = rate_idx; g.n_frames = n_frames;
    g.f0 = f0; g.nf = nf; g.g0 = 2 * f0; g.n_gran = 2 * nf;
    g.hdr_mode = (channels == 1) ? 3 : 0;
    g.hdr_flags = 0;
    g.crc = 0;
    g.fabs0 = 0; g.hist = NULL; g.out_base = NULL; g.whole_file = 1;
    g.test_flags = 0;
    g.pcm_pitch = 0;
    g.n_samples = NULL;
#if defined(MP3MI_ULP_CENSUS)
    g.census_cb_stride = 0;
#endif
    return g;
}

void mp3mi_launch_fft(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm,
                      float *energy_l, float *energy_s, float *bins, double *cw_mid, float *hist6, hipStream_t st, int which = 3);
/* records whose unpredictability needs its second tier (k_part lists them, k_cw_fix and k_part's second run work
 * through the list); device memory, mp3mi_cw_fixlist_bytes(records) */
struct mp3mi_cw_fixlist {
    unsigned count, cap, pad[2];
    unsigned list[1]; /* cap entries */
};
static inline size_t mp3mi_cw_fixlist_bytes(size_t n_rec) { return sizeof(mp3mi_cw_fixlist) + n_rec * sizeof(unsigned); }
void mp3mi_launch_cw_fix_reset(mp3mi_cw_fixlist *fix, unsigned cap, hipStream_t st);
void mp3mi_launch_cw_fix(const mp3mi_geom &g, const float *bins, double *cw_mid, float *hist6, const mp3mi_cw_fixlist *fix, hipStream_t st);
void mp3mi_launch_psy(const mp3mi_tables *T, const mp3mi_geom &g, const float *energy_l,
                      const float *energy_s, double *cw_mid, float *hist6, const float *bins, mp3mi_cw_fixlist *fix,
                      void *psy_state, double *eb_all, float *cb_all, mp3mi_psy_out *out, hipStream_t st, int which = 3);
/* the 32 new samples of a window_subband call, as a kernel argument (k_dropin.hip) */
struct mp3mi_dropin_samples { int16_t v[32]; };
void mp3mi_launch_filter(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm, double *sbs, double *sb_dbg, hipStream_t st);
/* records whose loop-prep values k_mdct's tail could not decide (k_fbmdct.hip): k_prep works through the list;
 * device memory, mp3mi_prep_fixlist_bytes(records) */
struct mp3mi_prep_fixlist {
    unsigned count, pad[3];
    unsigned list[1]; /* one entry per record of a launch at most */
};
static inline size_t mp3mi_prep_fixlist_bytes(size_t n_rec) { return sizeof(mp3mi_prep_fixlist) + n_rec * sizeof(unsigned); }
/* prep / fix NULL: the spectrum only (fix->count is zeroed by the caller, on the same stream) */
void mp3mi_launch_mdct(const mp3mi_tables *T, const mp3mi_geom &g, const mp3mi_psy_out *psy, const double *sbs, double *xr,
                       mp3mi_loop_prep *prep, mp3mi_prep_fixlist *fix, hipStream_t st);
size_t mp3mi_sbs_bytes(const mp3mi_geom &g); /* subband samples between k_filter and k_mdct */
/* fix != NULL: the records it lists (a fixed small grid walks the list); NULL: every record of the launch */
void mp3mi_launch_prep(const mp3mi_tables *T, const mp3mi_geom &g, const double *xr,
                       const mp3mi_psy_out *psy, mp3mi_loop_prep *prep, const mp3mi_prep_fixlist *fix, int force_exact, hipStream_t st);
/* stream placement of k_loop (k_loop.hip: loop_place_stream); all pointers NULL = stream == blockIdx */
#define MP3MI_PLACE_KEYS 8192
struct mp3mi_loop_place {
    const int *order;      /* [n_streams] sorted position -> stream (k_rank) */
    int *cost;             /* [n_streams] cost of each stream in this launch (input of the next k_rank) */
    unsigned *taken;       /* [n_streams] zero before the launch */
    unsigned *simd_slots;  /* [MP3MI_PLACE_KEYS] zero before the launch: arrivals per SIMD */
    unsigned *simd_idx;    /* [MP3MI_PLACE_KEYS] zero before the launch: arrival ticket of the SIMD + 1 */
    unsigned *ticket, *scan; /* zero before the launch */
    int n_simd;            /* SIMDs of the device */
};
void mp3mi_launch_rank(const int *cost, int *order, int n, hipStream_t st);
void mp3mi_launch_loop(const mp3mi_tables *T, const mp3mi_geom &g, const double *xr,
                       const mp3mi_psy_out *psy, const mp3mi_loop_prep *prep, const int32_t *bits_per_frame,
                       void *loop_state, int16_t *ix, mp3mi_frame_side *side, unsigned *gate_count, mp3mi_loop_place place,
                       hipStream_t st);
/* bounded wait (one wavefront) until k_loop's start census reaches `target` -- see k_loop.hip */
int mp3mi_loop_resident(void);
void mp3mi_launch_gate(const unsigned *count, unsigned target, unsigned max_ticks, hipStream_t st);
void mp3mi_launch_hold(const unsigned *flag, unsigned ticket, unsigned max_ticks, hipStream_t st);
void mp3mi_launch_hold_release(unsigned *flag, unsigned ticket, hipStream_t st);
/* streaming plumbing around k_format (k_format.hip): the bytes of a stream that are not final yet -- the unfilled
   part of the bit reservoir's slots and the headers in between, at most MP3MI_CARRY_BYTES -- wait in `carry` */
#define MP3MI_CARRY_BYTES 2048
void mp3mi_launch_carry_in(int n_streams, const uint8_t *carry, const int32_t *carry_len, uint8_t *out, size_t out_stride, hipStream_t st);
/* loop_state: the streams' mp3mi_loop_state records (k_loop.hip) as words -- word 0 is ResvSize, the last word the
   stream's status (MP3MI_DEV_ABORT_*); voided: counts the streams whose file a call voided because of it */
void mp3mi_launch_stream_tail(const mp3mi_geom &g, int flush, int32_t *loop_state, int loop_state_words, const int32_t *bits_per_frame,
                              uint8_t *out, size_t out_stride, int64_t *out_base, uint8_t *carry, int32_t *carry_len, uint32_t *out_len,
                              unsigned *voided, hipStream_t st);
/* status[s] = the status word of stream s (a gather out of the strided state records) */
void mp3mi_launch_status_gather(int n_streams, const int32_t *loop_state, int loop_state_words, int32_t *status, hipStream_t st);
void mp3mi_launch_hist_save(const mp3mi_geom &g, const int16_t *pcm, int16_t *hist, hipStream_t st);
/* loop_state / voided as for mp3mi_launch_stream_tail (a whole-file call ends the streams in k_format); NULL: no status */
void mp3mi_launch_format(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *ix,
                         const mp3mi_frame_side *side, const int32_t *bits_per_frame,
                         const int32_t *bitrate_index, uint8_t *out, size_t out_stride,
                         uint32_t *out_len, int32_t *loop_state, int loop_state_words, unsigned *voided, hipStream_t st);
/* Compute units of the CURRENT device (launch geometry of the persistent kernels).  Asked per call: batches of
   several devices may be driven from one process, by several threads (mp3mi.h, "Threads") -- no cached static. */
static inline int mp3mi_current_cu_count(void)
{
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    return n;
}
#endif

#endif

// Host side of the Layer I / II batch (include/mp3mi_l12.h): set-up as the reference's driver does it
// (src/musicin.c:528-581, src/common.c:291-347), scratch sizing, and one pipeline of kernels per chunk of frames:
//
//   k_fft12 -> k12_psy ;  k_filter ;  k12_alloc
//
// Frames are independent but for PCM history (l12_dev.h), so a chunk is just these four launches in stream order, and
// chunks follow each other on the batch's one HIP stream.  No CPU fallback.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mp3mi_host.h"
#include "l12_dev.h"
#include "mp3mi_l12.h"

#define CHK(call)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "mp3mi: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MP3MI_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

static const int L12_BITRATES[2][15] = { /* src/common.c:122-123 */
    {0, 32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448},
    {0, 32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384}};

struct mp3mi_l12_batch {
    int device;
    int layer, n_streams, rate_idx, rate_hz, channels, max_frames, chunk_frames;
    int spf, spp, span, lb;
    int mode, crc, hdr_flags;
    unsigned test_flags;
    int debug;
    int max_frame_bytes;
    std::vector<l12_stream_cfg> cfg_h;
    hipStream_t stream;
    hipEvent_t ev0, ev1;
    std::vector<hipEvent_t> kev; // five per chunk of the last call: around the four launches
    int kev_chunks;
    double kernel_ms[4];
    long kernel_launches[4];
    bool timing_open;
    double total_ms;
    long calls;
    mp3mi_tables *T3;          // window, FFT program, filterbank tables (shared with Layer III)
    mp3mi_tables_l12 *T;
    l12_stream_cfg *cfg;
    float *erp, *snr;
    double *sbs;
    l12_frame_dbg *dbg;
    int dbg_f0, dbg_nf;
    // streaming: PCM history of the streams in progress, two copies taken in turn (k12_hist_save is out of place)
    int16_t *hist[2], *fb_hist[2];
    int hist_cur;
    long fabs0;              // frames every stream has been given by mp3mi_l12_batch_encode_next since the last reset
};

static int l12_have_device(void)
{
    int n = 0;
    return hipGetDeviceCount(&n) == hipSuccess && n > 0;
}

struct l12_device_scope {
    int prev;
    bool ok;
    explicit l12_device_scope(int dev) : prev(-1), ok(true)
    {
        int cur = -1;
        if (hipGetDevice(&cur) != hipSuccess) { ok = false; return; }
        if (cur != dev) {
            if (hipSetDevice(dev) != hipSuccess) { ok = false; return; }
            prev = cur;
        }
    }
    ~l12_device_scope() { if (prev >= 0) (void) hipSetDevice(prev); }
};
#define ON_DEVICE(b)                                                                                     \
    l12_device_scope dev_scope_((b)->device);                                                            \
    if (!dev_scope_.ok) { fprintf(stderr, "mp3mi: cannot select device %d\n", (b)->device); return MP3MI_ERR_HIP; }

static size_t l12_per_frame_bytes(int layer, int channels, int spf)
{
    const size_t rec = (size_t) 3 * L12_ROW * sizeof(float) + 32 * sizeof(float);
    return (size_t) layer * channels * rec + (size_t) spf * 8 * channels;
}

extern "C" int mp3mi_l12_batch_create(mp3mi_l12_batch **out, int layer, int n_streams, int rate_hz, int channels,
                                      const int *kbps, int kbps_all, int max_frames, unsigned scratch_mb)
{
    static const double s_freq[3] = {44.1, 48, 32}; // src/common.c:113
    if (!out) return MP3MI_ERR_ARG;
    *out = NULL;
    if ((layer != 1 && layer != 2) || n_streams < 1 || max_frames < 1 || (channels != 1 && channels != 2)) return MP3MI_ERR_ARG;
    int ri;
    if (rate_hz == 44100) ri = 0;
    else if (rate_hz == 48000) ri = 1;
    else if (rate_hz == 32000) ri = 2;
    else return MP3MI_ERR_ARG; // (MPEG-2 LSF rates: psycho_anal exits on them, src/psy.c:131-136)
    if (!l12_have_device()) {
        fprintf(stderr, "mp3mi: no usable HIP device -- this library has no CPU fallback\n");
        return MP3MI_ERR_NO_DEVICE;
    }
    mp3mi_l12_batch *b = new mp3mi_l12_batch();
    memset((void *) &b->stream, 0, sizeof(b->stream));
    b->layer = layer; b->n_streams = n_streams; b->rate_idx = ri; b->rate_hz = rate_hz; b->channels = channels;
    b->max_frames = max_frames;
    b->spf = layer == 1 ? 384 : 1152; b->spp = layer == 1 ? 384 : 576; b->span = layer == 1 ? 1024 : 1056;
    b->lb = layer == 1 ? 3 : 2;
    b->mode = channels == 1 ? MP3MI_MODE_MONO : MP3MI_MODE_STEREO;
    b->crc = 0; b->hdr_flags = 0; b->test_flags = 0; b->debug = 0;
    b->timing_open = false; b->total_ms = 0.0; b->calls = 0; b->kev_chunks = 0;
    for (int i = 0; i < 4; i++) { b->kernel_ms[i] = 0.0; b->kernel_launches[i] = 0; }
    b->ev0 = b->ev1 = 0; b->T3 = NULL; b->T = NULL; b->cfg = NULL;
    b->erp = b->snr = NULL; b->sbs = NULL; b->dbg = NULL; b->dbg_f0 = b->dbg_nf = 0;
    b->hist[0] = b->hist[1] = b->fb_hist[0] = b->fb_hist[1] = NULL; b->hist_cur = 0; b->fabs0 = 0;
    if (hipGetDevice(&b->device) != hipSuccess) { delete b; return MP3MI_ERR_HIP; }
    b->cfg_h.resize((size_t) n_streams);
    b->max_frame_bytes = 0;
    for (int s = 0; s < n_streams; s++) {
        const int k = kbps ? kbps[s] : kbps_all;
        int bi;
        for (bi = 1; bi < 15; bi++)
            if (L12_BITRATES[layer - 1][bi] == k) break;
        if (bi == 15) { delete b; return MP3MI_ERR_ARG; }
        l12_stream_cfg &c = b->cfg_h[(size_t) s];
        c.bitrate_index = bi;
        // src/musicin.c:562-569: slots per frame, the fraction dropped (and with it every padding decision)
        const int whole_SpF = (int) (((double) b->spf / s_freq[ri]) * ((double) k / (double) (layer == 1 ? 32 : 8)));
        c.frame_bits = whole_SpF * (layer == 1 ? 32 : 8);
        if (layer == 2) { // pick_table, src/common.c:291-318
            const int br_per_ch = k / channels, sfrq = (int) s_freq[ri];
            if ((sfrq == 48 && br_per_ch >= 56) || (br_per_ch >= 56 && br_per_ch <= 80)) c.table = 0;
            else if (sfrq != 48 && br_per_ch >= 96) c.table = 1;
            else if (sfrq != 32 && br_per_ch <= 48) c.table = 2;
            else c.table = 3;
            static const int SBL[4] = {27, 30, 8, 12};
            c.sblimit = SBL[c.table];
        } else { c.table = 0; c.sblimit = 32; }
        if (c.frame_bits / 8 > b->max_frame_bytes) b->max_frame_bytes = c.frame_bits / 8;
    }
    const size_t budget = (size_t) (scratch_mb ? scratch_mb : 32768u) << 20;
    // per stream the chunk's buffers also hold the lb warm-up passes and three extra filterbank granules, whatever its length
    const size_t per_stream_fixed = (size_t) b->lb * channels * ((size_t) 3 * L12_ROW * sizeof(float) + 32 * sizeof(float)) + (size_t) 4 * channels * 576 * sizeof(double);
    const size_t per_stream_budget = budget / (size_t) n_streams;
    long cf = per_stream_budget > per_stream_fixed ? (long) ((per_stream_budget - per_stream_fixed) / l12_per_frame_bytes(layer, channels, b->spf)) : 1;
    if (cf < 1) cf = 1;
    if (cf > max_frames) cf = max_frames;
    b->chunk_frames = (int) cf;

    int rc = MP3MI_OK;
    auto build = [&]() -> int {
        std::vector<char> t3_store(sizeof(mp3mi_tables)), t12_store(sizeof(mp3mi_tables_l12));
        mp3mi_tables *T3h = (mp3mi_tables *) t3_store.data();
        mp3mi_tables_l12 *T12h = (mp3mi_tables_l12 *) t12_store.data();
        int trc = mp3mi_build_tables(T3h, ri);
        if (trc == -8) return MP3MI_ERR_TABLES;
        if (trc != 0) return MP3MI_ERR_ARG;
        trc = mp3mi_build_tables_l12(T12h, ri, layer);
        if (trc == -8) return MP3MI_ERR_TABLES;
        if (trc != 0) return MP3MI_ERR_ARG;
        for (int s = 0; s < n_streams; s++)
            if (layer == 2 && b->cfg_h[(size_t) s].sblimit != T12h->sblimit[b->cfg_h[(size_t) s].table]) return MP3MI_ERR_TABLES;
        CHK(hipStreamCreate(&b->stream));
        CHK(hipEventCreate(&b->ev0));
        CHK(hipEventCreate(&b->ev1));
        CHK(hipMalloc((void **) &b->T3, sizeof(mp3mi_tables)));
        CHK(hipMalloc((void **) &b->T, sizeof(mp3mi_tables_l12)));
        CHK(hipMalloc((void **) &b->cfg, sizeof(l12_stream_cfg) * (size_t) n_streams));
        CHK(hipMemcpy(b->T3, T3h, sizeof(mp3mi_tables), hipMemcpyHostToDevice));
        CHK(hipMemcpy(b->T, T12h, sizeof(mp3mi_tables_l12), hipMemcpyHostToDevice));
        CHK(hipMemcpy(b->cfg, b->cfg_h.data(), sizeof(l12_stream_cfg) * (size_t) n_streams, hipMemcpyHostToDevice));
        const size_t np = (size_t) cf * layer + (size_t) b->lb, nrec = (size_t) n_streams * np * (size_t) channels;
        const size_t ngran = ((size_t) cf * (size_t) (b->spf / 32) + 17) / 18 + 3; // granules of 18 slots k_filter may be asked for
        CHK(hipMalloc((void **) &b->erp, nrec * 3 * L12_ROW * sizeof(float)));
        CHK(hipMalloc((void **) &b->snr, nrec * 32 * sizeof(float)));
        CHK(hipMalloc((void **) &b->sbs, (size_t) n_streams * ngran * (size_t) channels * 576 * sizeof(double)));
        return MP3MI_OK;
    };
    rc = build();
    if (rc != MP3MI_OK) { mp3mi_l12_batch_destroy(b); return rc; }
    *out = b;
    return MP3MI_OK;
}

extern "C" void mp3mi_l12_batch_destroy(mp3mi_l12_batch *b)
{
    if (!b) return;
    {
        l12_device_scope sc(b->device);
        if (b->stream) (void) hipStreamSynchronize(b->stream);
        (void) hipFree(b->T3); (void) hipFree(b->T); (void) hipFree(b->cfg); (void) hipFree(b->erp);
        (void) hipFree(b->snr); (void) hipFree(b->sbs); (void) hipFree(b->dbg);
        for (int i = 0; i < 2; i++) { (void) hipFree(b->hist[i]); (void) hipFree(b->fb_hist[i]); }
        if (b->ev0) (void) hipEventDestroy(b->ev0);
        if (b->ev1) (void) hipEventDestroy(b->ev1);
        for (hipEvent_t e : b->kev) (void) hipEventDestroy(e);
        if (b->stream) (void) hipStreamDestroy(b->stream);
    }
    delete b;
}

extern "C" int mp3mi_l12_batch_set_mode(mp3mi_l12_batch *b, int mode)
{
    if (!b || mode < 0 || mode > 3) return MP3MI_ERR_ARG;
    if ((b->channels == 1) != (mode == MP3MI_MODE_MONO)) return MP3MI_ERR_ARG;
    b->mode = mode;
    return MP3MI_OK;
}
extern "C" int mp3mi_l12_batch_set_error_protection(mp3mi_l12_batch *b, int on)
{
    if (!b) return MP3MI_ERR_ARG;
    b->crc = on != 0;
    return MP3MI_OK;
}
extern "C" int mp3mi_l12_batch_set_header(mp3mi_l12_batch *b, int copyright, int original, int emphasis)
{
    if (!b || emphasis < 0 || emphasis > 3) return MP3MI_ERR_ARG;
    b->hdr_flags = ((copyright != 0) << 3) | ((original != 0) << 2) | emphasis;
    return MP3MI_OK;
}
extern "C" int mp3mi_l12_batch_set_test_flags(mp3mi_l12_batch *b, unsigned flags)
{
    if (!b) return MP3MI_ERR_ARG;
    b->test_flags = flags;
    return MP3MI_OK;
}
extern "C" size_t mp3mi_l12_batch_out_stride(const mp3mi_l12_batch *b, int n_frames)
{
    if (!b || n_frames < 0) return 0;
    // (two-channel Layer I below its fields' size: the reference's frames outgrow their slots, k_l12.hip)
    size_t fb = (size_t) b->max_frame_bytes;
    if (b->layer == 1 && b->channels == 2 && fb < 38) fb = 38;
    return ((size_t) n_frames * fb + 1 + 255) / 256 * 256;
}
extern "C" void mp3mi_l12_batch_debug_enable(mp3mi_l12_batch *b, int on) { if (b) b->debug = on != 0; }

static int l12_close_timing(mp3mi_l12_batch *b)
{
    if (!b->timing_open) return MP3MI_OK;
    float ms = 0.0f;
    CHK(hipEventSynchronize(b->ev1));
    CHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    b->total_ms += (double) ms;
    for (int c = 0; c < b->kev_chunks; c++)
        for (int k = 0; k < 4; k++) {
            CHK(hipEventElapsedTime(&ms, b->kev[(size_t) c * 5 + k], b->kev[(size_t) c * 5 + k + 1]));
            b->kernel_ms[k] += (double) ms;
            b->kernel_launches[k]++;
        }
    b->kev_chunks = 0;
    b->timing_open = false;
    return MP3MI_OK;
}

static int l12_encode_impl(mp3mi_l12_batch *b, const int16_t *pcm_dev, const int32_t *n_samples_dev, int n_frames,
                           uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev, bool whole_file)
{
    if (!b || !pcm_dev || !out_dev || !out_len_dev || n_frames < 1 || n_frames > b->max_frames) return MP3MI_ERR_ARG;
    if (out_stride < mp3mi_l12_batch_out_stride(b, n_frames)) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    { const int rc = l12_close_timing(b); if (rc != MP3MI_OK) return rc; }
    const int S = b->n_streams, C = b->channels, layer = b->layer;
    if (b->debug && !b->dbg) CHK(hipMalloc((void **) &b->dbg, sizeof(l12_frame_dbg) * (size_t) S * (size_t) b->chunk_frames));
    CHK(hipEventRecord(b->ev0, b->stream));
    {
        const size_t want = 5 * (size_t) ((n_frames + b->chunk_frames - 1) / b->chunk_frames);
        while (b->kev.size() < want) {
            hipEvent_t e;
            CHK(hipEventCreate(&e));
            b->kev.push_back(e);
        }
    }
    // equal chunks (as batch.cpp cuts a Layer III call): a short remainder -- 191 + 191 + 1 frames of the 383-frame bench call --
    // costs four full launches over all streams and recomputes the warm-up passes for one frame
    const int nchunks = (n_frames + b->chunk_frames - 1) / b->chunk_frames, cfr = (n_frames + nchunks - 1) / nchunks;
    int chunk = 0;
    for (int f0 = 0; f0 < n_frames; f0 += cfr, chunk++) {
        hipEvent_t *ke = &b->kev[(size_t) chunk * 5];
        const int nf = n_frames - f0 < cfr ? n_frames - f0 : cfr;
        l12_geom g;
        memset(&g, 0, sizeof(g));
        g.n_streams = S; g.channels = C; g.layer = layer; g.rate_idx = b->rate_idx;
        g.n_frames = n_frames; g.f0 = f0; g.nf = nf;
        g.lb = b->lb; g.np = nf * layer + b->lb;
        g.spf = b->spf; g.spp = b->spp; g.span = b->span;
        g.actual_mode = b->mode; g.crc = b->crc; g.hdr_flags = b->hdr_flags; g.test_flags = (int) b->test_flags;
        g.n_samples = n_samples_dev;
        g.whole_file = whole_file ? 1 : 0;
        g.fabs0 = whole_file ? 0 : b->fabs0;
        g.hist = (!whole_file && b->fabs0 > 0) ? b->hist[b->hist_cur] : NULL; // (nothing precedes a stream's first call: zeros)
        // the filterbank slots of the chunk: frame f's slot u is slot (spf / 32) f + u of the stream, Layer I's two slots
        // EARLIER (get_audio holds 64 samples back, src/encode.c:224-247); k_filter computes whole 18-slot granules
        const long slots = b->spf / 32, first = slots * f0 - (layer == 1 ? 2 : 0), last = slots * (f0 + nf) - 1 - (layer == 1 ? 2 : 0);
        const long gf = first >= 0 ? first / 18 : -((-first + 17) / 18), gl = last >= 0 ? last / 18 : -((-last + 17) / 18);
        g.g0 = (int) gf + 1;                 // granule g0 - 1 is "granule slot 0" of k_filter
        g.n_gran = (int) (gl - gf);          // granules [g0, g0 + n_gran) follow it
        g.slot0 = (int) (first - 18 * gf);
        mp3mi_geom fg = mp3mi_make_geom(S, C, b->rate_idx, n_frames, f0, nf);
        fg.g0 = g.g0; fg.n_gran = g.n_gran;
        fg.pcm_pitch = (long) n_frames * b->spf;
        fg.n_samples = n_samples_dev;
        fg.fabs0 = g.fabs0;
        fg.hist = g.hist ? b->fb_hist[b->hist_cur] : NULL;
        CHK(hipEventRecord(ke[0], b->stream));
        mp3mi_launch_fft12(b->T3, g, pcm_dev, b->erp, b->stream);
        CHK(hipEventRecord(ke[1], b->stream));
        mp3mi_launch_l12_psy(b->T, g, b->erp, b->snr, b->stream);
        CHK(hipEventRecord(ke[2], b->stream));
        mp3mi_launch_filter(b->T3, fg, pcm_dev, b->sbs, NULL, b->stream);
        CHK(hipEventRecord(ke[3], b->stream));
        mp3mi_launch_l12_alloc(b->T, g, b->cfg, b->sbs, b->snr, out_dev, out_stride, out_len_dev, b->debug ? b->dbg : NULL, b->stream);
        CHK(hipEventRecord(ke[4], b->stream));
        b->dbg_f0 = f0; b->dbg_nf = nf;
    }
    if (!whole_file) { // the samples the next call will need from before its first
        l12_geom g;
        memset(&g, 0, sizeof(g));
        g.n_streams = S; g.channels = C; g.n_frames = n_frames; g.spf = b->spf;
        for (int i = 0; i < 2; i++) {
            if (!b->hist[i]) {
                CHK(hipMalloc((void **) &b->hist[i], sizeof(int16_t) * L12_PCM_HIST * (size_t) C * (size_t) S));
                CHK(hipMalloc((void **) &b->fb_hist[i], sizeof(int16_t) * MP3MI_PCM_HIST * (size_t) C * (size_t) S));
                CHK(hipMemsetAsync(b->hist[i], 0, sizeof(int16_t) * L12_PCM_HIST * (size_t) C * (size_t) S, b->stream));
                CHK(hipMemsetAsync(b->fb_hist[i], 0, sizeof(int16_t) * MP3MI_PCM_HIST * (size_t) C * (size_t) S, b->stream));
            }
        }
        if (b->fabs0 == 0) { // a stream's first call: what precedes it is silence (the reference's zero-filled buffers)
            CHK(hipMemsetAsync(b->hist[b->hist_cur], 0, sizeof(int16_t) * L12_PCM_HIST * (size_t) C * (size_t) S, b->stream));
            CHK(hipMemsetAsync(b->fb_hist[b->hist_cur], 0, sizeof(int16_t) * MP3MI_PCM_HIST * (size_t) C * (size_t) S, b->stream));
        }
        mp3mi_launch_l12_hist_save(g, pcm_dev, b->hist[b->hist_cur], b->hist[b->hist_cur ^ 1], b->fb_hist[b->hist_cur], b->fb_hist[b->hist_cur ^ 1], b->stream);
        b->hist_cur ^= 1;
        b->fabs0 += n_frames;
    } else b->fabs0 = 0;
    CHK(hipEventRecord(b->ev1, b->stream));
    b->kev_chunks = chunk;
    b->timing_open = true;
    b->calls++;
    CHK(hipGetLastError());
    return MP3MI_OK;
}

extern "C" int mp3mi_l12_batch_encode(mp3mi_l12_batch *b, const int16_t *pcm_dev, const int32_t *n_samples_dev, int n_frames,
                                      uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev)
{
    return l12_encode_impl(b, pcm_dev, n_samples_dev, n_frames, out_dev, out_stride, out_len_dev, true);
}

extern "C" int mp3mi_l12_batch_encode_next(mp3mi_l12_batch *b, const int16_t *pcm_dev, int n_frames, uint8_t *out_dev, size_t out_stride,
                                           uint32_t *out_len_dev)
{
    return l12_encode_impl(b, pcm_dev, NULL, n_frames, out_dev, out_stride, out_len_dev, false);
}

extern "C" int mp3mi_l12_batch_flush(mp3mi_l12_batch *b, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev)
{
    if (!b || !out_dev || !out_len_dev || out_stride < 1) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    mp3mi_launch_l12_flush(b->n_streams, out_dev, out_stride, out_len_dev, b->stream);
    b->fabs0 = 0;
    CHK(hipGetLastError());
    return MP3MI_OK;
}

extern "C" int mp3mi_l12_batch_reset(mp3mi_l12_batch *b)
{
    if (!b) return MP3MI_ERR_ARG;
    b->fabs0 = 0;
    return MP3MI_OK;
}

extern "C" int mp3mi_l12_batch_sync(mp3mi_l12_batch *b)
{
    if (!b) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    CHK(hipStreamSynchronize(b->stream));
    return l12_close_timing(b);
}

extern "C" int mp3mi_l12_batch_total_timing(mp3mi_l12_batch *b, double *all_kernels_ms, long *calls)
{
    if (!b) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    CHK(hipStreamSynchronize(b->stream));
    { const int rc = l12_close_timing(b); if (rc != MP3MI_OK) return rc; }
    if (all_kernels_ms) *all_kernels_ms = b->total_ms;
    if (calls) *calls = b->calls;
    return MP3MI_OK;
}

extern "C" int mp3mi_l12_batch_kernel_timing(mp3mi_l12_batch *b, double ms[4], long launches[4])
{
    if (!b || !ms || !launches) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    CHK(hipStreamSynchronize(b->stream));
    { const int rc = l12_close_timing(b); if (rc != MP3MI_OK) return rc; }
    for (int i = 0; i < 4; i++) { ms[i] = b->kernel_ms[i]; launches[i] = b->kernel_launches[i]; }
    return MP3MI_OK;
}

extern "C" long mp3mi_l12_batch_debug_fetch(mp3mi_l12_batch *b, void *host_dst, size_t cap, int *first_frame, int *n_chunk_frames)
{
    static_assert(sizeof(mp3mi_l12_frame_seams) == sizeof(l12_frame_dbg), "seam record of mp3mi_l12.h and l12_dev.h");
    if (!b || !host_dst || !b->dbg) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    CHK(hipStreamSynchronize(b->stream));
    const size_t bytes = sizeof(l12_frame_dbg) * (size_t) b->n_streams * (size_t) b->dbg_nf;
    if (cap < bytes) return MP3MI_ERR_ARG;
    CHK(hipMemcpy(host_dst, b->dbg, bytes, hipMemcpyDeviceToHost));
    if (first_frame) *first_frame = b->dbg_f0;
    if (n_chunk_frames) *n_chunk_frames = b->dbg_nf;
    return (long) bytes;
}

extern "C" int mp3mi_l12_encode_host(int layer, int n_streams, int rate_hz, int channels, const int *kbps, int kbps_all, int mode,
                                     int error_protection, const int16_t *pcm, const int32_t *n_samples, int n_frames,
                                     uint8_t *out, size_t out_stride, uint32_t *out_len)
{
    if (!pcm || !out || !out_len || n_frames < 1 || n_streams < 1) return MP3MI_ERR_ARG;
    mp3mi_l12_batch *b = NULL;
    int rc = mp3mi_l12_batch_create(&b, layer, n_streams, rate_hz, channels, kbps, kbps_all, n_frames, 0);
    if (rc != MP3MI_OK) return rc;
    if (mode >= 0) rc = mp3mi_l12_batch_set_mode(b, mode);
    if (rc == MP3MI_OK) rc = mp3mi_l12_batch_set_error_protection(b, error_protection);
    if (rc != MP3MI_OK || out_stride < mp3mi_l12_batch_out_stride(b, n_frames)) { mp3mi_l12_batch_destroy(b); return MP3MI_ERR_ARG; }
    const size_t spf = layer == 1 ? 384 : 1152;
    const size_t pcm_bytes = (size_t) n_streams * (size_t) n_frames * spf * (size_t) channels * sizeof(int16_t);
    int16_t *pcm_d = NULL;
    uint8_t *out_d = NULL;
    uint32_t *len_d = NULL;
    int32_t *ns_d = NULL;
    rc = MP3MI_ERR_HIP;
    if (hipMalloc((void **) &pcm_d, pcm_bytes) == hipSuccess && hipMalloc((void **) &out_d, out_stride * (size_t) n_streams) == hipSuccess &&
        hipMalloc((void **) &len_d, sizeof(uint32_t) * (size_t) n_streams) == hipSuccess &&
        hipMalloc((void **) &ns_d, sizeof(int32_t) * (size_t) n_streams) == hipSuccess &&
        hipMemcpy(pcm_d, pcm, pcm_bytes, hipMemcpyHostToDevice) == hipSuccess &&
        (!n_samples || hipMemcpy(ns_d, n_samples, sizeof(int32_t) * (size_t) n_streams, hipMemcpyHostToDevice) == hipSuccess)) {
        rc = mp3mi_l12_batch_encode(b, pcm_d, n_samples ? ns_d : NULL, n_frames, out_d, out_stride, len_d);
        if (rc == MP3MI_OK) rc = mp3mi_l12_batch_sync(b);
        if (rc == MP3MI_OK && (hipMemcpy(out, out_d, out_stride * (size_t) n_streams, hipMemcpyDeviceToHost) != hipSuccess ||
                               hipMemcpy(out_len, len_d, sizeof(uint32_t) * (size_t) n_streams, hipMemcpyDeviceToHost) != hipSuccess))
            rc = MP3MI_ERR_HIP;
    }
    (void) hipFree(pcm_d); (void) hipFree(out_d); (void) hipFree(len_d); (void) hipFree(ns_d);
    mp3mi_l12_batch_destroy(b);
    return rc;
}

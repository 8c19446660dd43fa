// Host side of the batched encoder: owns the device buffers, cuts the frames of a call into
// chunks that fit the scratch budget, and enqueues the kernels of every chunk on TWO HIP streams:
//
//   front stream (low priority):  k_fft | k_cw k_part k_psy k_filter k_mdct (k_prep: the records k_mdct lists)   of chunk c+1
//   loop  stream (high priority): k_loop k_format                                   of chunk c
//
// k_loop keeps one wavefront per stream resident for a whole chunk (four per SIMD) and leaves room for
// one more wavefront per SIMD: the feed-forward kernels of the next chunk run there, behind a gate that
// lets k_loop become resident first -- all but the FFTs, which take a whole CU's LDS per workgroup and run
// between two k_loop launches.  The three buffers that cross from the front stream to the loop stream
// (psy, xr, prep) are double-buffered; events order producer/consumer.  Calls overlap the same way.
//
//   k_fft     (stream, granule, channel)  psy FFTs                 feed-forward
//   k_psy     (stream, channel)           thresholds / block type  serial over granules
//   k_fbmdct  (stream, channel, granules) filterbank + MDCT        feed-forward, needs block type
//   k_loop    (stream)                    iteration loop           serial over frames
//   k_format  (stream, frame)             bitstream formatting     independent per frame
//
// Mirrors the Layer III case of the reference's frame loop (src/musicin.c:708-788) for every
// stream at once.  No CPU fallback: every entry point returns an error when HIP cannot run.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "mp3mi_host.h"
#include "mp3mi.h"

size_t mp3mi_psy_state_size(void);
size_t mp3mi_loop_state_size(void);

#define CHK(call)                                                                              \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            fprintf(stderr, "mp3mi: %s failed: %s (%s:%d)\n", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return MP3MI_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

struct mp3mi_batch {
    int device;              // the HIP device everything of this batch lives on (current at create time)
    int n_streams, rate_idx, rate_hz, channels, max_frames, chunk_frames;
    std::vector<int> bits_per_frame_h, bitrate_index_h;
    int max_frame_bytes;
    hipStream_t stream;      // front stream: feed-forward kernels (and the initial memsets)
    hipStream_t lstream;     // loop stream: k_loop + k_format
    // A batch of more streams than k_loop holds resident is cut into PARTS (contiguous stream ranges); a part's frames of
    // one chunk are an "item", and the items go through the two HIP streams one after the other (encode_impl).
    // Per (double-buffer slot, part), index slot * n_parts + part:
    std::vector<hipEvent_t> ev_front; // the item's front kernels are done
    std::vector<hipEvent_t> ev_loop;  // its k_loop is done (the slot's region of this part may be overwritten)
    std::vector<char> slot_used;      // ev_loop has been recorded: the region's last reader is a k_loop that may still run
    int n_parts, part_streams;
    hipEvent_t ev_done;      // everything of the previous encode call is done
    hipEvent_t ev_hist;      // the front stream's last work of a call (the PCM history hand-over) is enqueued
    bool have_done;
    bool overlap_calls;      // a call's front stream does not wait for the call before it (MP3MI_CALL_OVERLAP=0: it does)
    // The last k_loop of a call is HELD on the device (k_hold, k_loop.hip) until the next call's first transforms are through:
    // hold_flag is one word of host memory mapped into the device's address space, hold_seq the ticket of the hold in
    // force (tickets only grow), held = a hold is in force that neither a next call nor the host has let go yet
    unsigned *hold_flag_h, *hold_flag_d;
    unsigned hold_seq;
    bool held, hold_calls;
    int slot_base;           // parity of the double-buffer slot the next call's chunk 0 takes
    unsigned *gate_count;    // start census of k_loop's wavefronts (device memory, only ever grows), NULL = gate off
    unsigned gate_total;     // census value once every wavefront launched so far has started
    unsigned gate_first;     // ... the same (kept for the gate's target: the census once the LAST launch is resident)
    int *place_order, *place_cost; // k_loop stream placement (mp3mi_loop_place), NULL = off
    unsigned *place_zero;    // taken[n] + simd_slots + simd_idx + ticket + scan, zeroed before every k_loop
    int n_simd;
    mp3mi_batch_options opt; // as given at create time (defaults resolved where a field says "-1 default")
    unsigned *voided;        // device counter: streams whose file a call voided (the reference dies on them), since the last sync
    int32_t *status_dev;     // [S]: what mp3mi_batch_stream_status copies out; between a flush and the next encode / reset it HOLDS the ended streams' status
    bool status_kept;        // status_dev holds the status of the streams the last flush ended (their state is reset)
    int prep_exact;          // MP3MI_TEST_PREP_EXACT: k_prep over every record, its second tier only, instead of k_mdct's tail (tests)
    int test_flags;          // mp3mi_geom::test_flags
    int hdr_flags;           // copyright << 3 | original << 2 | emphasis (src/l3bitstream.c:330-334)
    int hdr_mode;            // header mode field: 0 stereo, 2 dual channel, 3 mono (src/common.h:233-236)
    int crc;                 // error protection (-e): zero CRC word after the header, as the reference writes it
    int last_slot;
    mp3mi_tables *T;
    int32_t *bits_per_frame, *bitrate_index;
    float *energy_l, *energy_s, *hist6, *fft_bins;
    double *cw_mid, *xr[2], *sbs, *sb_dbg, *part_eb;
    mp3mi_cw_fixlist *cw_fix; // the (granule, channel) records whose unpredictability needs its second tier (k_part)
    float *part_cb;
    mp3mi_psy_out *psy[2];
    mp3mi_loop_prep *prep[2];
    mp3mi_prep_fixlist *prep_fix; // the records k_mdct's tail could not decide (k_prep works through the list); front stream only
    void *psy_state, *loop_state;
    int16_t *ix;
    mp3mi_frame_side *side;
    // streaming (encode_next / flush): what a stream carries from call to call besides psy_state / loop_state
    long frames_done;        // frames of every stream encoded since the last reset
    bool fresh;              // reset since the last encode (or never encoded): state buffers are zero
    int16_t *pcm_hist;       // [S][MP3MI_PCM_HIST][C]: the samples before the next call's first
    int64_t *out_base;       // [S]: file bytes delivered so far
    uint8_t *carry;          // [S][MP3MI_CARRY_BYTES]: file bytes formatted but not final yet
    int32_t *carry_len;      // [S]
    int debug, last_nf;
    // Host-buffer calls (mp3mi_batch_encode_host_async): the call's PCM goes up and its file bytes come down chunk by
    // chunk on two copy streams of their own, beside the kernels; two calls may be in flight, so the device copies of
    // PCM and output exist twice (slot = call number & 1).  Created with the first such call.
    struct host_io {
        bool ready;
        hipStream_t h2d, d2h;
        int16_t *pcm[2];          // [S][max_frames * 1152][C]
        uint8_t *out[2];          // [S][out_stride]
        uint32_t *len[2];         // [S]
        size_t out_stride;
        hipEvent_t pcm_free[2], out_free[2]; // the slot's PCM has been read by its last kernel / its output by its last copy
        bool pcm_used[2], out_used[2];
        std::vector<hipEvent_t> ev_fmt[2];   // per chunk: the chunk's formatter is done
        std::vector<hipEvent_t> t_up[2], t_dn[2]; // per chunk two events around the copy (the second is what consumers wait for)
        int n_chunks[2];
        double bytes_up[2], bytes_dn[2];
        bool pending[2];
        unsigned call_no;
        double tot_up_bytes, tot_dn_bytes, tot_up_ms, tot_dn_ms;
        long tot_calls;
    } hio;
    // HIP-event timing of the calls: two sets taken in turn, so that a call can be issued while the one before still
    // runs; a set is read out (harvested) when its turn comes again -- which also keeps the host at most two calls
    // ahead of the device -- or when the timing is asked for
    struct timing_set {
        hipEvent_t ev0, ev1;
        std::vector<hipEvent_t> loop_ev;
        int launches; // bracketed spans (one per chunk)
        int kernels;  // k_loop launches inside them
        bool pending;
    } ts[2];
    unsigned call_no;
    float last_loop_ms, last_all_ms;
    int last_launches;
    double tot_loop_ms, tot_all_ms;
    long tot_launches, tot_calls;
};

// Every entry point runs on the batch's own device whatever the calling thread's current device is, and leaves
// the caller's current device as it found it.
struct device_scope {
    int prev;
    bool ok;
    explicit device_scope(int dev) : prev(-1), ok(true)
    {
        if (hipGetDevice(&prev) != hipSuccess) { ok = false; prev = -1; return; }
        if (prev != dev && hipSetDevice(dev) != hipSuccess) ok = false;
    }
    ~device_scope() { if (prev >= 0) (void) hipSetDevice(prev); }
};
#define ON_DEVICE(b)                                                                                     \
    device_scope dev_scope_((b)->device);                                                                \
    if (!dev_scope_.ok) { fprintf(stderr, "mp3mi: cannot select device %d\n", (b)->device); return MP3MI_ERR_HIP; }

static const int BITRATES[15] = {0, 32, 40, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320}; // src/common.c:124

static int have_device(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return 0;
    return 1;
}

#if !defined(MP3MI_SOURCE_HASH)
#define MP3MI_SOURCE_HASH "unknown"
#endif
extern "C" const char *mp3mi_source_hash(void) { return MP3MI_SOURCE_HASH; }

extern "C" const char *mp3mi_version(void)
{
#if defined(MP3MI_EMU)
    return "libmp3mi 0.1 (TEST BUILD: wave emulator, not the product)";
#else
    return "libmp3mi 0.1 (gfx950 HIP)";
#endif
}

static_assert((int) MP3MI_STREAM_ABORT_GLOBAL_GAIN == MP3MI_DEV_ABORT_GLOBAL_GAIN && (int) MP3MI_STREAM_ABORT_HUFF_BITS == MP3MI_DEV_ABORT_HUFF_BITS &&
                  (int) MP3MI_STREAM_ABORT_FLUSH_SLOT == MP3MI_DEV_ABORT_FLUSH_SLOT,
              "status codes of mp3mi.h and mp3mi_dev.h");

extern "C" void mp3mi_batch_destroy(mp3mi_batch *b);

// The host lets the held k_loop of the last call go (mp3mi_batch::held): whoever is about to WAIT for that call's results,
// or to put work on the front stream that waits for it, calls this first.
static void hold_release(mp3mi_batch *b)
{
    if (!b->held) return;
    // (this hold, not the ones before it: k_hold reads the word of ITS ticket -- a ring of eight, so that a release the device
    // has not looked at yet is not overwritten by the next one: the host runs ahead of the device through chains of calls
    // that never wait, e.g. encode_next / flush / encode_next / flush)
    __atomic_store_n(b->hold_flag_h + 1 + (b->hold_seq & 7u), b->hold_seq, __ATOMIC_RELEASE);
    b->held = false;
}

extern "C" void mp3mi_batch_options_default(mp3mi_batch_options *o)
{
    if (!o) return;
    memset(o, 0, sizeof(*o));
    o->struct_size = (uint32_t) sizeof(*o);
    o->abi = MP3MI_OPTIONS_ABI;
    o->call_overlap = o->gate = o->placement = o->y_after_loop = o->psy_beside = o->dropin_lookahead = o->call_hold = -1;
}

// The one place the library reads its environment (mp3mi.h): the knobs of tools/ and tests/.
extern "C" void mp3mi_batch_options_from_env(mp3mi_batch_options *o)
{
    if (!o) return;
    mp3mi_batch_options_default(o);
    const char *e;
    auto on = [](const char *v) { return v && atoi(v) != 0; };
    if ((e = getenv("MP3MI_SCRATCH_MB")) && atol(e) > 0) o->scratch_mb = (uint32_t) atol(e);
    if ((e = getenv("MP3MI_CHUNK_FRAMES")) && atol(e) > 0) o->chunk_frames = (int32_t) atol(e);
    if (on(getenv("MP3MI_NOISE_EXACT"))) o->test_flags |= MP3MI_TEST_NOISE_EXACT;
    if (on(getenv("MP3MI_PHASE_EXACT"))) o->test_flags |= MP3MI_TEST_PHASE_EXACT;
    if (on(getenv("MP3MI_PSY_EXACT"))) o->test_flags |= MP3MI_TEST_PSY_EXACT;
    if (on(getenv("MP3MI_QUANT_EXACT"))) o->test_flags |= MP3MI_TEST_QUANT_EXACT;
    if (on(getenv("MP3MI_PREP_EXACT"))) o->test_flags |= MP3MI_TEST_PREP_EXACT;
    if (on(getenv("MP3MI_CW_EXACT"))) o->test_flags |= MP3MI_TEST_CW_EXACT;
    if ((e = getenv("MP3MI_CALL_OVERLAP"))) o->call_overlap = atoi(e) != 0;
    if (on(getenv("MP3MI_NO_GATE"))) o->gate = 0;
    if (on(getenv("MP3MI_NO_PLACE"))) o->placement = 0;
    if ((e = getenv("MP3MI_LOOP_PART_STREAMS")) && atoi(e) >= 64) o->loop_part_streams = atoi(e) / 64 * 64;
    if ((e = getenv("MP3MI_Y_AFTER_LOOP"))) o->y_after_loop = atoi(e) != 0;
    if ((e = getenv("MP3MI_PSY_BESIDE"))) o->psy_beside = atoi(e) == 2 ? 2 : (atoi(e) ? 1 : 0);
    if ((e = getenv("MP3MI_CALL_HOLD"))) o->call_hold = atoi(e) != 0;
    if (on(getenv("MP3MI_DROPIN_STATS"))) o->dropin_stats = 1;
    if ((e = getenv("MP3MI_DROPIN_LOOKAHEAD")) && atoi(e) >= 0 && atoi(e) <= 4) o->dropin_lookahead = atoi(e);
}

// Fills *b step by step; on any failure the caller destroys the partially built object (every pointer and handle
// starts out null, and mp3mi_batch_destroy skips what was never created).
static int batch_build(mp3mi_batch *b, int n_streams, int rate_hz, int channels, const int *kbps, int kbps_all, int max_frames,
                       const mp3mi_batch_options &opt)
{
    static const double s_freq[3] = {44.1, 48, 32}; // src/common.c:113
    int ri;
    if (rate_hz == 44100) ri = 0;
    else if (rate_hz == 48000) ri = 1;
    else if (rate_hz == 32000) ri = 2;
    else return MP3MI_ERR_ARG; // the reference's L3psycho_anal exits on anything else (src/l3psy.c:170-176)
    if (n_streams <= 0 || max_frames <= 0 || (channels != 1 && channels != 2)) return MP3MI_ERR_ARG;
    CHK(hipGetDevice(&b->device));
    b->n_streams = n_streams; b->rate_idx = ri; b->rate_hz = rate_hz; b->channels = channels;
    b->max_frames = max_frames; b->debug = 0; b->last_nf = 0;
    b->bits_per_frame_h.resize(n_streams);
    b->bitrate_index_h.resize(n_streams);
    b->max_frame_bytes = 0;
    for (int s = 0; s < n_streams; s++) {
        const int k = kbps ? kbps[s] : kbps_all;
        int bi;
        for (bi = 1; bi < 15; bi++)
            if (BITRATES[bi] == k) break;
        if (bi == 15) return MP3MI_ERR_ARG;
        // slots per frame, never padded (src/musicin.c:562-581)
        const int whole_SpF = (int) (((double) 1152 / s_freq[ri]) * ((double) k / 8.0));
        b->bitrate_index_h[s] = bi;
        b->bits_per_frame_h[s] = 8 * whole_SpF;
        if (whole_SpF > b->max_frame_bytes) b->max_frame_bytes = whole_SpF;
    }
    // chunk size from a scratch budget (bytes per frame and stream of the per-chunk buffers)
    const size_t per_gc = MP3MI_HBLK_P * 4 + MP3MI_PART_P * 12 + 3 * MP3MI_HBLK_S * 4 + MP3MI_FFT_BINS * 4 + 50 * 8 + 12 * 4 +
                          2 * (sizeof(mp3mi_psy_out) + sizeof(mp3mi_loop_prep) + 576 * 8) + 576 * 8 + 576 * 2;
    const size_t per_frame = per_gc * 2 * (size_t) channels + sizeof(mp3mi_frame_side);
    b->opt = opt;
    const size_t budget = (size_t) (opt.scratch_mb ? opt.scratch_mb : 32768u) << 20;
    long cf = (long) (budget / (per_frame * (size_t) n_streams));
    if (cf < 1) cf = 1;
    if (cf > max_frames) cf = max_frames;
    if (opt.chunk_frames > 0 && opt.chunk_frames < cf) cf = opt.chunk_frames;
    b->chunk_frames = (int) cf;

    std::vector<char> Th_store(sizeof(mp3mi_tables)); // host copy of the tables, released on every path
    mp3mi_tables *Th = (mp3mi_tables *) Th_store.data();
    {
        const int trc = mp3mi_build_tables(Th, ri);
        if (trc == -8) return MP3MI_ERR_TABLES; // the host's libm does not reproduce the pinned tables (tables_host.cpp)
        if (trc != 0) return MP3MI_ERR_ARG;
    }
    const size_t ngc = (size_t) n_streams * 2 * (size_t) cf * (size_t) channels;
    {
        int least = 0, greatest = 0;
        CHK(hipDeviceGetStreamPriorityRange(&least, &greatest));
        CHK(hipStreamCreateWithPriority(&b->stream, hipStreamDefault, least));
        CHK(hipStreamCreateWithPriority(&b->lstream, hipStreamDefault, greatest));
    }
    {   // parts: as few as hold the batch with at most mp3mi_loop_resident() streams each, equal in size (a multiple of
        // 64); options.loop_part_streams overrides (tests)
        const int resident = mp3mi_loop_resident();
        int np = (n_streams + resident - 1) / resident;
        int ps = ((n_streams + np - 1) / np + 63) / 64 * 64;
        if (opt.loop_part_streams >= 64) ps = opt.loop_part_streams;
        b->part_streams = ps;
        b->n_parts = (n_streams + ps - 1) / ps;
        b->ev_front.assign(2 * (size_t) b->n_parts, (hipEvent_t) 0);
        b->ev_loop.assign(2 * (size_t) b->n_parts, (hipEvent_t) 0);
        b->slot_used.assign(2 * (size_t) b->n_parts, 0);
        for (size_t i = 0; i < b->ev_front.size(); i++) {
            CHK(hipEventCreateWithFlags(&b->ev_front[i], hipEventDisableTiming));
            CHK(hipEventCreateWithFlags(&b->ev_loop[i], hipEventDisableTiming));
        }
    }
    CHK(hipEventCreateWithFlags(&b->ev_done, hipEventDisableTiming));
    CHK(hipEventCreateWithFlags(&b->ev_hist, hipEventDisableTiming));
    b->have_done = false;
    b->last_slot = 0;
    b->overlap_calls = opt.call_overlap != 0;
    b->slot_base = 0;
    b->test_flags = (int) (opt.test_flags & 15u) | ((opt.test_flags & MP3MI_TEST_CW_EXACT) ? 16 : 0) | ((opt.test_flags & MP3MI_TEST_PREP_LIST) ? 32 : 0);
    b->prep_exact = (opt.test_flags & MP3MI_TEST_PREP_EXACT) ? 1 : 0;
    b->hdr_flags = 0;
    b->hdr_mode = (channels == 1) ? 3 : 0;
    b->crc = 0;
    b->gate_count = NULL; b->gate_total = 0; b->gate_first = 0;
    b->hold_calls = opt.call_hold != 0 && opt.call_overlap != 0 && opt.gate != 0;
    if (b->hold_calls) {
        CHK(hipHostMalloc((void **) &b->hold_flag_h, 64, hipHostMallocMapped));
        for (int i = 0; i < 16; i++) b->hold_flag_h[i] = 0;
        CHK(hipHostGetDevicePointer((void **) &b->hold_flag_d, b->hold_flag_h, 0));
    }
    if (opt.gate != 0) {
        CHK(hipMalloc((void **) &b->gate_count, 2 * sizeof(unsigned))); // [0] start census, [1] frames finished in this launch
        CHK(hipMemset(b->gate_count, 0, 2 * sizeof(unsigned)));
    }
    CHK(hipMalloc((void **) &b->voided, sizeof(unsigned)));
    CHK(hipMemset(b->voided, 0, sizeof(unsigned)));
    CHK(hipMalloc((void **) &b->status_dev, sizeof(int32_t) * (size_t) n_streams));
    b->place_order = NULL; b->place_cost = NULL; b->place_zero = NULL; b->n_simd = 0;
    {
        hipDeviceProp_t prop;
        int dev = 0;
        CHK(hipGetDevice(&dev));
        CHK(hipGetDeviceProperties(&prop, dev));
        b->n_simd = prop.multiProcessorCount * 4;
        if (opt.placement != 0 && (n_streams >= 2 * b->n_simd || opt.placement == 1)) { // placement only matters when SIMDs hold several streams
            CHK(hipMalloc((void **) &b->place_order, sizeof(int) * n_streams));
            CHK(hipMalloc((void **) &b->place_cost, sizeof(int) * n_streams));
            CHK(hipMalloc((void **) &b->place_zero, sizeof(unsigned) * ((size_t) n_streams + 2 * MP3MI_PLACE_KEYS + 2)));
        }
    }
    CHK(hipMalloc((void **) &b->T, sizeof(mp3mi_tables)));
    CHK(hipMemcpy(b->T, Th, sizeof(mp3mi_tables), hipMemcpyHostToDevice));
    CHK(hipMalloc((void **) &b->bits_per_frame, sizeof(int32_t) * n_streams));
    CHK(hipMalloc((void **) &b->bitrate_index, sizeof(int32_t) * n_streams));
    CHK(hipMemcpy(b->bits_per_frame, b->bits_per_frame_h.data(), sizeof(int32_t) * n_streams, hipMemcpyHostToDevice));
    CHK(hipMemcpy(b->bitrate_index, b->bitrate_index_h.data(), sizeof(int32_t) * n_streams, hipMemcpyHostToDevice));
    CHK(hipMalloc((void **) &b->energy_l, ngc * MP3MI_HBLK_P * sizeof(float)));
    CHK(hipMalloc((void **) &b->part_eb, ngc * MP3MI_PART_P * sizeof(double)));
#if defined(MP3MI_ULP_CENSUS) // (diagnostic build: two shadow copies behind the sums, mp3mi_geom::census_cb_stride)
    CHK(hipMalloc((void **) &b->part_cb, 3 * ngc * MP3MI_PART_P * sizeof(float)));
    CHK(hipMemset(b->part_cb, 0, 3 * ngc * MP3MI_PART_P * sizeof(float)));
#else
    CHK(hipMalloc((void **) &b->part_cb, ngc * MP3MI_PART_P * sizeof(float)));
#endif
    CHK(hipMalloc((void **) &b->energy_s, ngc * 3 * MP3MI_HBLK_S * sizeof(float)));
    CHK(hipMalloc((void **) &b->hist6, ngc * 12 * sizeof(float)));
    CHK(hipMalloc((void **) &b->fft_bins, ngc * MP3MI_FFT_BINS * sizeof(float)));
    CHK(hipMalloc((void **) &b->cw_mid, ngc * 50 * sizeof(double)));
    CHK(hipMalloc((void **) &b->cw_fix, mp3mi_cw_fixlist_bytes(ngc)));
    CHK(hipMemset(b->cw_fix, 0, sizeof(mp3mi_cw_fixlist)));
    CHK(hipMalloc((void **) &b->prep_fix, mp3mi_prep_fixlist_bytes(ngc)));
    CHK(hipMemset(b->prep_fix, 0, sizeof(mp3mi_prep_fixlist)));
    for (int i = 0; i < 2; i++) {
        CHK(hipMalloc((void **) &b->xr[i], ngc * 576 * sizeof(double)));
        CHK(hipMalloc((void **) &b->psy[i], ngc * sizeof(mp3mi_psy_out)));
        CHK(hipMalloc((void **) &b->prep[i], ngc * sizeof(mp3mi_loop_prep)));
    }
    CHK(hipMalloc((void **) &b->sbs, (ngc + (size_t) n_streams * channels) * 576 * sizeof(double)));
    CHK(hipMalloc((void **) &b->ix, ngc * 576 * sizeof(int16_t)));
    CHK(hipMalloc((void **) &b->side, (size_t) n_streams * (size_t) cf * sizeof(mp3mi_frame_side)));
    CHK(hipMalloc((void **) &b->psy_state, mp3mi_psy_state_size() * (size_t) n_streams * channels));
    CHK(hipMalloc((void **) &b->loop_state, mp3mi_loop_state_size() * (size_t) n_streams));
    CHK(hipMalloc((void **) &b->pcm_hist, sizeof(int16_t) * MP3MI_PCM_HIST * (size_t) channels * (size_t) n_streams));
    CHK(hipMalloc((void **) &b->out_base, sizeof(int64_t) * (size_t) n_streams));
    CHK(hipMalloc((void **) &b->carry, (size_t) MP3MI_CARRY_BYTES * (size_t) n_streams));
    CHK(hipMalloc((void **) &b->carry_len, sizeof(int32_t) * (size_t) n_streams));
    CHK(hipMemset(b->out_base, 0, sizeof(int64_t) * (size_t) n_streams)); // a flush before the first encode delivers nothing
    CHK(hipMemset(b->carry_len, 0, sizeof(int32_t) * (size_t) n_streams));
    b->frames_done = 0;
    b->fresh = false;
    b->sb_dbg = NULL;
    for (int i = 0; i < 2; i++) {
        CHK(hipEventCreate(&b->ts[i].ev0));
        CHK(hipEventCreate(&b->ts[i].ev1));
        b->ts[i].launches = 0;
        b->ts[i].pending = false;
    }
    b->call_no = 0;
    b->last_loop_ms = b->last_all_ms = 0;
    b->last_launches = 0;
    b->tot_loop_ms = b->tot_all_ms = 0;
    b->tot_launches = b->tot_calls = 0;
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_create(mp3mi_batch **out, int n_streams, int rate_hz, int channels, const int *kbps,
                                  int kbps_all, int max_frames)
{
    mp3mi_batch_options opt;
    mp3mi_batch_options_from_env(&opt); // once per batch; nothing else in the library reads the environment
    return mp3mi_batch_create_ex(out, n_streams, rate_hz, channels, kbps, kbps_all, max_frames, &opt);
}

extern "C" int mp3mi_batch_create_ex(mp3mi_batch **out, int n_streams, int rate_hz, int channels, const int *kbps,
                                     int kbps_all, int max_frames, const mp3mi_batch_options *opt_in)
{
    if (!out) return MP3MI_ERR_ARG;
    *out = NULL;
    mp3mi_batch_options opt;
    mp3mi_batch_options_default(&opt);
    if (opt_in) {
        if (opt_in->struct_size != sizeof(opt) || opt_in->abi != MP3MI_OPTIONS_ABI) return MP3MI_ERR_ARG; // another version of the header
        opt = *opt_in;
        auto tri = [](int v) { return v >= -1 && v <= 1; }; // -1 default, 0 off, 1 on
        if ((opt.test_flags & ~(unsigned) (MP3MI_TEST_ALL_EXACT | MP3MI_TEST_PREP_LIST)) || opt.chunk_frames < 0 || opt.loop_part_streams < 0 ||
            (opt.loop_part_streams % 64) != 0 || opt.psy_beside < -1 || opt.psy_beside > 2 || opt.dropin_lookahead < -1 || opt.dropin_lookahead > 4 || (opt.dropin_stats != 0 && opt.dropin_stats != 1) || !tri(opt.call_overlap) || !tri(opt.gate) ||
            !tri(opt.placement) || !tri(opt.y_after_loop) || !tri(opt.call_hold))
            return MP3MI_ERR_ARG;
    }
    // argument errors first: they are the caller's, whatever the machine
    if (rate_hz != 44100 && rate_hz != 48000 && rate_hz != 32000) return MP3MI_ERR_ARG; // src/l3psy.c:170-176 exits on anything else
    if (n_streams <= 0 || max_frames <= 0 || (channels != 1 && channels != 2)) return MP3MI_ERR_ARG;
    for (int s = 0; s < n_streams; s++) {
        const int k = kbps ? kbps[s] : kbps_all;
        int bi;
        for (bi = 1; bi < 15; bi++)
            if (BITRATES[bi] == k) break;
        if (bi == 15) return MP3MI_ERR_ARG;
        if (!kbps) break;
    }
    if (!have_device()) {
        fprintf(stderr, "mp3mi: no HIP device available -- this library has no CPU path\n");
        return MP3MI_ERR_NO_DEVICE;
    }
    mp3mi_batch *b = new mp3mi_batch(); // value-initialised: every pointer, handle and counter starts at zero
    const int rc = batch_build(b, n_streams, rate_hz, channels, kbps, kbps_all, max_frames, opt);
    if (rc != MP3MI_OK) {
        mp3mi_batch_destroy(b); // frees whatever was allocated before the failure
        return rc;
    }
    *out = b; // published only when complete
    return MP3MI_OK;
}

extern "C" void mp3mi_batch_destroy(mp3mi_batch *b)
{
    if (!b) return;
    device_scope ds(b->device);
    hold_release(b);
    if (b->stream) hipStreamSynchronize(b->stream);
    if (b->lstream) hipStreamSynchronize(b->lstream);
    void *bufs[] = {b->T, b->bits_per_frame, b->bitrate_index, b->energy_l, b->energy_s, b->hist6, b->fft_bins, b->cw_mid, b->cw_fix,
                    b->part_eb, b->part_cb, b->xr[0], b->xr[1], b->psy[0], b->psy[1], b->prep[0], b->prep[1], b->prep_fix, b->sbs, b->ix, b->side,
                    b->psy_state, b->loop_state, b->pcm_hist, b->out_base, b->carry, b->carry_len, b->gate_count, b->place_order, b->place_cost, b->place_zero, b->sb_dbg, b->voided, b->status_dev};
    for (void *p : bufs)
        if (p) hipFree(p);
    for (hipEvent_t e : b->ev_front) if (e) hipEventDestroy(e);
    for (hipEvent_t e : b->ev_loop) if (e) hipEventDestroy(e);
    hipEvent_t evs[] = {b->ts[0].ev0, b->ts[0].ev1, b->ts[1].ev0, b->ts[1].ev1, b->ev_done, b->ev_hist};
    for (hipEvent_t e : evs)
        if (e) hipEventDestroy(e);
    for (int k = 0; k < 2; k++)
        for (size_t i = 0; i < b->ts[k].loop_ev.size(); i++) hipEventDestroy(b->ts[k].loop_ev[i]);
    {   // (whatever of the host-buffer state exists, also after an allocation that failed half-way)
        if (b->hio.h2d) hipStreamSynchronize(b->hio.h2d);
        if (b->hio.d2h) hipStreamSynchronize(b->hio.d2h);
        for (int i = 0; i < 2; i++) {
            if (b->hio.pcm[i]) hipFree(b->hio.pcm[i]);
            if (b->hio.out[i]) hipFree(b->hio.out[i]);
            if (b->hio.len[i]) hipFree(b->hio.len[i]);
            if (b->hio.pcm_free[i]) hipEventDestroy(b->hio.pcm_free[i]);
            if (b->hio.out_free[i]) hipEventDestroy(b->hio.out_free[i]);
            for (std::vector<hipEvent_t> *v : {&b->hio.ev_fmt[i], &b->hio.t_up[i], &b->hio.t_dn[i]})
                for (hipEvent_t e : *v) hipEventDestroy(e);
        }
        if (b->hio.h2d) hipStreamDestroy(b->hio.h2d);
        if (b->hio.d2h) hipStreamDestroy(b->hio.d2h);
    }
    if (b->stream) hipStreamDestroy(b->stream);
    if (b->lstream) hipStreamDestroy(b->lstream);
    if (b->hold_flag_h) hipHostFree(b->hold_flag_h);
    delete b;
}

extern "C" size_t mp3mi_batch_out_stride(const mp3mi_batch *b, int n_frames)
{
    // whole frames, the byte under construction that close writes, and -- for streaming calls -- the bytes an
    // earlier call formatted but could not deliver yet, which lead the row
    size_t n = (size_t) n_frames * (size_t) b->max_frame_bytes + 1 + MP3MI_CARRY_BYTES;
    return (n + 255) & ~(size_t) 255;
}

extern "C" void mp3mi_batch_debug_enable(mp3mi_batch *b, int on) { b->debug = on; }

extern "C" int mp3mi_batch_set_test_flags(mp3mi_batch *b, unsigned flags)
{
    if (!b || (flags & ~(unsigned) (MP3MI_TEST_ALL_EXACT | MP3MI_TEST_PREP_LIST))) return MP3MI_ERR_ARG;
    b->test_flags = (int) (flags & 15u) | ((flags & MP3MI_TEST_CW_EXACT) ? 16 : 0) | ((flags & MP3MI_TEST_PREP_LIST) ? 32 : 0);
    b->prep_exact = (flags & MP3MI_TEST_PREP_EXACT) ? 1 : 0;
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_set_header(mp3mi_batch *b, int copyright, int original, int emphasis)
{
    if (!b || (copyright & ~1) || (original & ~1) || (emphasis & ~3)) return MP3MI_ERR_ARG;
    b->hdr_flags = (copyright << 3) | (original << 2) | emphasis;
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_set_mode(mp3mi_batch *b, int mode)
{
    if (!b) return MP3MI_ERR_ARG;
    // -m s / d / m of the reference's driver (src/musicin.c:226-234); joint stereo is refused for Layer III by the
    // reference itself (src/musicin.c:548-552), and the mode has to fit the channel count
    const bool ok = (b->channels == 2 && (mode == MP3MI_MODE_STEREO || mode == MP3MI_MODE_DUAL_CHANNEL)) || (b->channels == 1 && mode == MP3MI_MODE_MONO);
    if (!ok) return MP3MI_ERR_ARG;
    b->hdr_mode = mode;
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_set_error_protection(mp3mi_batch *b, int on)
{
    if (!b || (on & ~1)) return MP3MI_ERR_ARG;
    b->crc = on;
    return MP3MI_OK;
}

struct host_call { // a call on host buffers (mp3mi_batch_encode_host_async): where the PCM comes from and the results go
    const int16_t *pcm;
    uint8_t *out;
    size_t out_stride;
    uint32_t *out_len;
    int slot;
};
static int encode_impl(mp3mi_batch *b, const int16_t *pcm_dev, const int32_t *n_samples_dev, int n_frames, uint8_t *out_dev,
                       size_t out_stride, uint32_t *out_len_dev, bool whole_file, const host_call *hc = NULL);

// reads a timing set out (waits for its call to finish)
static int harvest_timing(mp3mi_batch *b, int k)
{
    mp3mi_batch::timing_set &ts = b->ts[k];
    if (!ts.pending) return MP3MI_OK;
    CHK(hipEventSynchronize(ts.ev1));
    float tot = 0, loop = 0;
    CHK(hipEventElapsedTime(&tot, ts.ev0, ts.ev1));
    for (int c = 0; c < ts.launches; c++) {
        float ms = 0;
        CHK(hipEventElapsedTime(&ms, ts.loop_ev[2 * c], ts.loop_ev[2 * c + 1]));
        loop += ms;
    }
    b->last_loop_ms = loop; b->last_all_ms = tot; b->last_launches = ts.kernels;
    b->tot_loop_ms += loop; b->tot_all_ms += tot; b->tot_launches += ts.kernels; b->tot_calls++;
    ts.pending = false;
    return MP3MI_OK;
}

// Fresh encoder state for every stream: what the reference's function statics and the caller's buffers hold when
// its main() starts (all zero).  Enqueued on the front stream behind whatever is still running.
static int reset_impl(mp3mi_batch *b)
{
    const int S = b->n_streams, C = b->channels;
    if (b->have_done) CHK(hipStreamWaitEvent(b->stream, b->ev_done, 0)); // the previous call's kernels may still be running on the loop stream
    CHK(hipMemsetAsync(b->psy_state, 0, mp3mi_psy_state_size() * (size_t) S * C, b->stream));
    CHK(hipMemsetAsync(b->loop_state, 0, mp3mi_loop_state_size() * (size_t) S, b->stream));
    CHK(hipMemsetAsync(b->pcm_hist, 0, sizeof(int16_t) * MP3MI_PCM_HIST * (size_t) C * (size_t) S, b->stream));
    CHK(hipMemsetAsync(b->out_base, 0, sizeof(int64_t) * (size_t) S, b->stream));
    CHK(hipMemsetAsync(b->carry_len, 0, sizeof(int32_t) * (size_t) S, b->stream));
    b->frames_done = 0;
    b->fresh = true;
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_reset(mp3mi_batch *b)
{
    if (!b) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    b->status_kept = false; // an explicit reset starts new streams: what a flush kept of the ones it ended goes with them
    hold_release(b); // (the front stream is about to wait for the call before)
    return reset_impl(b);
}

extern "C" int mp3mi_batch_encode(mp3mi_batch *b, const int16_t *pcm_dev, int n_frames, uint8_t *out_dev,
                                  size_t out_stride, uint32_t *out_len_dev)
{
    return encode_impl(b, pcm_dev, NULL, n_frames, out_dev, out_stride, out_len_dev, true);
}

extern "C" int mp3mi_batch_encode_ragged(mp3mi_batch *b, const int16_t *pcm_dev, const int32_t *n_samples_dev, int n_frames,
                                         uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev)
{
    if (!n_samples_dev) return MP3MI_ERR_ARG;
    return encode_impl(b, pcm_dev, n_samples_dev, n_frames, out_dev, out_stride, out_len_dev, true);
}

extern "C" int mp3mi_batch_encode_next(mp3mi_batch *b, const int16_t *pcm_dev, int n_frames, uint8_t *out_dev,
                                       size_t out_stride, uint32_t *out_len_dev)
{
    return encode_impl(b, pcm_dev, NULL, n_frames, out_dev, out_stride, out_len_dev, false);
}

extern "C" int mp3mi_batch_flush(mp3mi_batch *b, uint8_t *out_dev, size_t out_stride, uint32_t *out_len_dev)
{
    if (!b || !out_dev || !out_len_dev || out_stride < (size_t) MP3MI_CARRY_BYTES + 1) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    hold_release(b);
    if (b->fresh) { // nothing was encoded since the reset: no file body (the reference would write one byte; see mp3mi.h)
        CHK(hipMemsetAsync(out_len_dev, 0, sizeof(uint32_t) * (size_t) b->n_streams, b->lstream));
        return MP3MI_OK;
    }
    mp3mi_geom g = mp3mi_make_geom(b->n_streams, b->channels, b->rate_idx, 0, 0, 0);
    g.fabs0 = b->frames_done;
    g.crc = b->crc;
    mp3mi_launch_stream_tail(g, 1, (int32_t *) b->loop_state, (int) (mp3mi_loop_state_size() / 4), b->bits_per_frame, out_dev, out_stride,
                             b->out_base, b->carry, b->carry_len, out_len_dev, b->voided, b->lstream); // behind the last call's k_format
    CHK(hipGetLastError());
    // the streams' status words go with the state the reset clears: keep what mp3mi_batch_stream_status is asked for
    // after the flush (until the next encode starts new streams)
    mp3mi_launch_status_gather(b->n_streams, (const int32_t *) b->loop_state, (int) (mp3mi_loop_state_size() / 4), b->status_dev, b->lstream);
    CHK(hipGetLastError());
    b->status_kept = true;
    CHK(hipEventRecord(b->ev_done, b->lstream));
    b->have_done = true;
    return reset_impl(b); // the streams are over: the next encode_next starts new ones
}

static int encode_impl(mp3mi_batch *b, const int16_t *pcm_dev, const int32_t *n_samples_dev, int n_frames, uint8_t *out_dev,
                       size_t out_stride, uint32_t *out_len_dev, bool whole_file, const host_call *hc)
{
    if (!b || !pcm_dev || !out_dev || !out_len_dev || n_frames <= 0 || n_frames > b->max_frames) return MP3MI_ERR_ARG;
    if (out_stride < (size_t) n_frames * (size_t) b->max_frame_bytes + 1 + (whole_file ? 0 : MP3MI_CARRY_BYTES)) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    const int S = b->n_streams, C = b->channels;
    if (b->debug && !b->sb_dbg) {
        const size_t ngc = (size_t) S * 2 * (size_t) b->chunk_frames * (size_t) C;
        CHK(hipMalloc((void **) &b->sb_dbg, ngc * 576 * sizeof(double)));
    }
    // The call before this one may still be running.  Its kernels and this call's share nothing but the batch's own
    // buffers, and every one of those is either touched on ONE stream only (in-order: the FFT outputs, the subband
    // samples, the psy state and the PCM history on the front stream; ix, the side information, the loop state, the
    // carry and the caller's output on the loop stream) or double-buffered across the two (psy / xr / prep: the slots
    // go on alternating from call to call, and a slot's writer waits for the k_loop that read it last, ev_loop).  So
    // this call's feed-forward kernels start at once and fill the chip while the last k_loop of the call before -- 4096
    // wavefronts, serial, nothing beside them -- runs out: back-to-back calls lose no pipeline fill.
    // (options.call_overlap = 0: the front stream waits for the whole call before, as reset / flush still do.)
    if (b->have_done && !b->overlap_calls) {
        hold_release(b);
        CHK(hipStreamWaitEvent(b->stream, b->ev_done, 0));
    }
    // a whole-file call starts every stream afresh; a streaming call continues (the first one after create / reset /
    // flush / a whole-file call starts afresh too)
    if (whole_file || (b->frames_done == 0 && !b->fresh)) {
        CHK(hipMemsetAsync(b->psy_state, 0, mp3mi_psy_state_size() * (size_t) S * C, b->stream));
        CHK(hipMemsetAsync(b->pcm_hist, 0, sizeof(int16_t) * MP3MI_PCM_HIST * (size_t) C * (size_t) S, b->stream));
        CHK(hipMemsetAsync(b->loop_state, 0, mp3mi_loop_state_size() * (size_t) S, b->lstream));
        CHK(hipMemsetAsync(b->out_base, 0, sizeof(int64_t) * (size_t) S, b->lstream));
        CHK(hipMemsetAsync(b->carry_len, 0, sizeof(int32_t) * (size_t) S, b->lstream));
        b->frames_done = 0;
    }
    const long fabs0 = b->frames_done;
    b->fresh = false;
    b->status_kept = false;
    if (b->place_cost) CHK(hipMemsetAsync(b->place_cost, 0, sizeof(int) * (size_t) S, b->lstream)); // first chunk: order = identity
    if (hc && b->hio.out_used[hc->slot]) CHK(hipStreamWaitEvent(b->lstream, b->hio.out_free[hc->slot], 0)); // the call two before this one copied out of it
    CHK(hipMemsetAsync(out_dev, 0, out_stride * (size_t) S, b->lstream)); // (behind the formatter of the call before: it may be the same buffer)
    if (!whole_file && fabs0 > 0) { // the bytes earlier calls formatted but could not deliver lead the rows
        mp3mi_launch_carry_in(S, b->carry, b->carry_len, out_dev, out_stride, b->lstream);
        CHK(hipGetLastError());
    }
    const int nchunks = (n_frames + b->chunk_frames - 1) / b->chunk_frames;
    const int P = b->n_parts, n_items = nchunks * P;
    if (harvest_timing(b, (int) (b->call_no & 1)) != MP3MI_OK) return MP3MI_ERR_HIP; // (the call before the last one)
    mp3mi_batch::timing_set &ts = b->ts[b->call_no & 1];
    while ((int) ts.loop_ev.size() < 2 * n_items) {
        hipEvent_t e;
        CHK(hipEventCreate(&e));
        ts.loop_ev.push_back(e);
    }
    ts.launches = n_items;
    ts.kernels = 0;
    CHK(hipEventRecord(ts.ev0, b->stream));
    // The unit of scheduling is an ITEM: the frames of one chunk of one PART of the streams (a batch of at most
    // mp3mi_loop_resident() streams -- 4096 on an MI355X -- is one part).  Items go through the pipeline one after the
    // other, chunk by chunk and within a chunk part by part, exactly as the chunks of a one-part batch do: every buffer
    // is stream-major, so an item is the same kernels with their pointers advanced to the part's first stream.
    //
    //   front stream:  FFT(k+1) | gate | k_cw k_part k_psy (k+1) | k_filter k_mdct k_prep (k+1) | FFT(k+2) ...
    //   loop stream:              k_rank k_loop(k) ................................................ k_format(k)
    //
    // Two kinds of front-end kernels cannot share the chip with k_loop as it starts: k_fft takes a whole CU's LDS per
    // workgroup, and k_psy's wavefronts live long (serial over granules), so whichever is in flight when a k_loop is
    // launched keeps its workgroups from starting.  The FFTs of item k+1 therefore run BETWEEN two k_loop launches; all
    // the rest of item k+1 runs beside k_loop(k), behind a gate that lets k_loop become resident first, in the order of
    // how little each loses at one wavefront per SIMD; what does not fit beside k_loop (the tail of k_mdct, k_prep)
    // runs behind it, alone and fast.  (Until round 3 the feed-forward kernels ran over ALL streams of a chunk and only
    // k_loop was cut into parts, with a hand-made assignment of kernels to parts for two and for four parts; with three
    // parts k_psy was still running when the second part started and that part took 65 ms instead of 37:
    // profiles/r03_experiments.txt.  12 288 / 16 384 / 20 000 streams: 5.8 -> ... M frames/s.)
    const int cfr = (n_frames + nchunks - 1) / nchunks; // equal chunks: a short last one would run without overlap
    struct item_view {
        mp3mi_geom g;      // of the part: n_streams, n_samples / hist / out_base advanced
        int slot, ev;      // double-buffer slot of the chunk; index of the (slot, part) events
        size_t s0, rec0;   // first stream; (granule, channel) records of the chunk before the part
    };
    auto view = [&](int k) {
        item_view v;
        const int c = k / P, part = k % P;
        const int f0 = c * cfr;
        const int nf = (n_frames - f0 < cfr) ? n_frames - f0 : cfr;
        v.s0 = (size_t) part * (size_t) b->part_streams;
        const int n = S - (int) v.s0 < b->part_streams ? S - (int) v.s0 : b->part_streams;
        mp3mi_geom g = mp3mi_make_geom(n, C, b->rate_idx, n_frames, f0, nf);
        g.test_flags = b->test_flags;
        g.n_samples = n_samples_dev ? n_samples_dev + v.s0 : NULL;
        g.hdr_flags |= b->hdr_flags;
        g.hdr_mode = b->hdr_mode;
        g.crc = b->crc;
        g.fabs0 = fabs0;
        g.hist = b->pcm_hist + v.s0 * MP3MI_PCM_HIST * (size_t) C;
        g.out_base = whole_file ? NULL : b->out_base + v.s0;
        g.whole_file = whole_file ? 1 : 0;
#if defined(MP3MI_ULP_CENSUS)
        g.census_cb_stride = (size_t) S * 2 * (size_t) b->chunk_frames * (size_t) C * MP3MI_PART_P;
#endif
        v.g = g;
        v.slot = (c + b->slot_base) & 1;
        v.ev = v.slot * P + part;
        v.rec0 = v.s0 * 2 * (size_t) nf * (size_t) C;
        return v;
    };
    const size_t pcm_pitch = (size_t) n_frames * 1152 * (size_t) C; // int16 per stream in the caller's buffer
    if (hc) {
        // The whole call's PCM, chunk by chunk, in order on the upload stream: a 2-D copy per chunk (every stream's samples of
        // the chunk's frames; the layout on the device is the caller's).  The chunk's first kernel waits for its copy, so chunk
        // c + 1 crosses PCIe while chunk c is encoded -- and, with calls issued back to back, the next call's first chunk
        // while this call's last one is.
        mp3mi_batch::host_io &H = b->hio;
        const int sl = hc->slot;
        if (H.pcm_used[sl]) CHK(hipStreamWaitEvent(H.h2d, H.pcm_free[sl], 0)); // (the kernels of the call two before this one)
        while ((int) H.t_up[sl].size() < 2 * nchunks) {
            hipEvent_t e;
            for (std::vector<hipEvent_t> *v : {&H.t_up[sl], &H.t_dn[sl]}) { CHK(hipEventCreate(&e)); v->push_back(e); }
            if (H.t_up[sl].size() % 2 == 0) { CHK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); H.ev_fmt[sl].push_back(e); }
        }
        H.n_chunks[sl] = nchunks;
        H.bytes_up[sl] = H.bytes_dn[sl] = 0.0;
        for (int c = 0; c < nchunks; c++) {
            const int f0 = c * cfr, nf = (n_frames - f0 < cfr) ? n_frames - f0 : cfr;
            const size_t off = (size_t) f0 * 1152 * (size_t) C, width = (size_t) nf * 1152 * (size_t) C * sizeof(int16_t);
            CHK(hipEventRecord(H.t_up[sl][2 * c], H.h2d));
            CHK(hipMemcpy2DAsync(H.pcm[sl] + off, pcm_pitch * sizeof(int16_t), hc->pcm + off, pcm_pitch * sizeof(int16_t), width, (size_t) S,
                                 hipMemcpyHostToDevice, H.h2d));
            CHK(hipEventRecord(H.t_up[sl][2 * c + 1], H.h2d));
            H.bytes_up[sl] += (double) width * S;
        }
    }
    const size_t psy_state_bytes = mp3mi_psy_state_size() * (size_t) C, loop_state_bytes = mp3mi_loop_state_size();
    // which: 1 the FFTs, 2 k_cw, 4 the partition sums (k_part), 8 k_psy
    auto stage_x = [&](int k, int which) -> int {
        const item_view v = view(k);
        const size_t r = v.rec0;
        if (hc && (which & 1) && k % P == 0) CHK(hipStreamWaitEvent(b->stream, b->hio.t_up[hc->slot][2 * (k / P) + 1], 0)); // the chunk's PCM is up
        if (which & 3) {
            mp3mi_launch_fft(b->T, v.g, pcm_dev + v.s0 * pcm_pitch, b->energy_l + r * MP3MI_HBLK_P, b->energy_s + r * 3 * MP3MI_HBLK_S,
                             b->fft_bins + r * MP3MI_FFT_BINS, b->cw_mid + r * 50, b->hist6 + r * 12, b->stream, which & 3);
            CHK(hipGetLastError());
        }
        // the k_loop that read this region of the slot last (two chunks ago, maybe in the call before)
        if ((which & 8) && b->slot_used[v.ev]) CHK(hipStreamWaitEvent(b->stream, b->ev_loop[v.ev], 0));
        if (which & 12) {
            mp3mi_launch_psy(b->T, v.g, b->energy_l + r * MP3MI_HBLK_P, b->energy_s + r * 3 * MP3MI_HBLK_S, b->cw_mid + r * 50, b->hist6 + r * 12,
                             b->fft_bins + r * MP3MI_FFT_BINS, b->cw_fix, (char *) b->psy_state + v.s0 * psy_state_bytes, b->part_eb + r * MP3MI_PART_P,
                             b->part_cb + r * MP3MI_PART_P, b->psy[v.slot] + r, b->stream, (which >> 2) & 3);
            CHK(hipGetLastError());
        }
        return MP3MI_OK;
    };
    // which of stage X runs beside the k_loop before the item's own, and whether the item's stage Y waits for that k_loop (below)
    auto y_after = [&](const item_view &iv) {
        return b->opt.y_after_loop >= 0 ? b->opt.y_after_loop != 0 : iv.g.n_streams > mp3mi_loop_resident();
    };
    auto beside_of = [&](const item_view &iv) {
        int bs = y_after(iv) ? 0 : 14;
        if (b->opt.psy_beside >= 0) bs = b->opt.psy_beside == 2 ? 8 : (b->opt.psy_beside ? 14 : 0);
        return bs;
    };
    // The call before this one may have left its last k_loop HELD (k_hold): this call's first item then takes the place
    // "the next item" has inside a call -- its transforms run now, in front of that k_loop, the hold is let go behind them,
    // and the rest of the item runs beside that k_loop, behind the gate, like every item after it.
    bool joined = false;
    {
        const item_view v0 = view(0);
        if (b->held && !y_after(v0) && b->gate_count) {
            if (stage_x(0, 15 & ~beside_of(v0)) != MP3MI_OK) return MP3MI_ERR_HIP;
            mp3mi_launch_hold_release(b->hold_flag_d, b->hold_seq, b->stream);
            CHK(hipGetLastError());
            b->held = false;
            joined = true;
        } else {
            hold_release(b);
            if (stage_x(0, 15) != MP3MI_OK) return MP3MI_ERR_HIP;
        }
    }
    for (int k = 0; k < n_items; k++) {
        const item_view v = view(k);
        const mp3mi_geom &g = v.g;
        const size_t r = v.rec0;
        const bool follows = k >= 1 || joined; // a k_loop runs (or is about to) that this item's kernels go beside
        // A part larger than the resident wavefronts (options.loop_part_streams, tests only) keeps every SIMD full to its
        // end, so the feed-forward kernels of the next item find no freed slots beside it, only cycles to take: there
        // stage Y waits for k_loop.
        const bool y_after_loop = y_after(v);
        // what of stage X runs beside k_loop (bits as for stage_x): all but the FFTs -- k_cw, k_part, k_psy are small in
        // registers and LDS, the item's FFTs were done before that launch started, and the region k_psy writes was read
        // last by the launch before it: 238.2 vs 247.2 ms per 4096 x 383 step with them between the launches.
        // options.psy_beside = 0 / 1 / 2: nothing / all three / k_psy only.
        const int beside = beside_of(v);
        // ---- front stream: everything that does not depend on the bit reservoir ----
        if (k >= 1 && y_after_loop) { // (the k_loop before this item's: view(k - 1))
            const item_view pv = view(k - 1);
            CHK(hipStreamWaitEvent(b->stream, b->ev_loop[pv.ev], 0));
        } else if (follows && b->gate_count) // this item's kernels run behind k_loop(k - 1), once that is resident (<= 300 us)
            mp3mi_launch_gate(b->gate_count, b->gate_first - 16u, 30000u, b->stream);
        if (beside && follows && stage_x(k, beside) != MP3MI_OK) return MP3MI_ERR_HIP;
        mp3mi_launch_filter(b->T, g, pcm_dev + v.s0 * pcm_pitch, b->sbs + v.s0 * (size_t) (g.n_gran + 1) * (size_t) C * 576,
                            b->debug ? b->sb_dbg + r * 576 : NULL, b->stream);
        CHK(hipGetLastError());
        // k_mdct's tail also computes the loop's stateless head and lists the records it could not decide; k_prep works
        // through the list (MP3MI_TEST_PREP_EXACT: through every record, the reference's way)
        CHK(hipMemsetAsync(&b->prep_fix->count, 0, sizeof(unsigned), b->stream));
        mp3mi_launch_mdct(b->T, g, b->psy[v.slot] + r, b->sbs + v.s0 * (size_t) (g.n_gran + 1) * (size_t) C * 576, b->xr[v.slot] + r * 576,
                          b->prep[v.slot] + r, b->prep_fix, b->stream);
        CHK(hipGetLastError());
        mp3mi_launch_prep(b->T, g, b->xr[v.slot] + r * 576, b->psy[v.slot] + r, b->prep[v.slot] + r, b->prep_exact ? NULL : b->prep_fix, b->prep_exact, b->stream);
        CHK(hipGetLastError());
        if (k + 1 < n_items) {
            // (The next item's transforms take whole CUs' LDS: they run BETWEEN two k_loop launches.  Nothing but the queue behind
            // a busy chip sees to that -- this item's kernels take longer than the k_loop they run beside -- and an explicit wait
            // for that k_loop's end costs 1.4 ms per step, because the transforms then no longer slip in as its last CUs drain:
            // profiles/r05_experiments.txt, E7.)
            if (stage_x(k + 1, 15 & ~beside_of(view(k + 1))) != MP3MI_OK) return MP3MI_ERR_HIP;
        }
        CHK(hipEventRecord(b->ev_front[v.ev], b->stream));
        // ---- loop stream: the serial search and the formatter ----
        // (what k_loop needs of its own stream is prepared AHEAD of the wait for the front stream -- in the shadow of the transforms that
        // run between two k_loop launches, not behind them: the ranking of the part's streams by their cost in the chunk before,
        // 0.19 ms for 4096 streams, and the two counters' reset)
        mp3mi_loop_place place = {NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0};
        if (b->place_order) { // rank the part's streams by their cost in the previous chunk, hand the tables to k_loop
            mp3mi_launch_rank(b->place_cost + v.s0, b->place_order + v.s0, g.n_streams, b->lstream);
            CHK(hipGetLastError());
            CHK(hipMemsetAsync(b->place_zero, 0, sizeof(unsigned) * ((size_t) S + 2 * MP3MI_PLACE_KEYS + 2), b->lstream));
            place.order = b->place_order + v.s0; place.cost = b->place_cost + v.s0; place.taken = b->place_zero;
            place.simd_slots = b->place_zero + S; place.simd_idx = place.simd_slots + MP3MI_PLACE_KEYS;
            place.ticket = place.simd_idx + MP3MI_PLACE_KEYS; place.scan = place.ticket + 1;
            place.n_simd = b->n_simd;
        }
        if (b->gate_count) CHK(hipMemsetAsync(b->gate_count + 1, 0, sizeof(unsigned), b->lstream));
        CHK(hipStreamWaitEvent(b->lstream, b->ev_front[v.ev], 0));
        if (b->hold_calls && k == n_items - 1) {
            // the call's LAST k_loop: held until the next call's first transforms are through (its first item then runs beside
            // this launch), the host lets go (hold_release), or the bound has passed: 20 ms, more for chunks so long that what the
            // front stream still holds of this item plus the next call's transforms take longer (0.4 ms per frame of a chunk)
            b->hold_seq++;
            const unsigned hold_ticks = 100000u * (unsigned) (g.nf * 4 / 10 > 20 ? (g.nf * 4 / 10 < 200 ? g.nf * 4 / 10 : 200) : 20); // 100 MHz
            mp3mi_launch_hold(b->hold_flag_d, b->hold_seq, hold_ticks, b->lstream);
            CHK(hipGetLastError());
            b->held = true;
        }
        CHK(hipEventRecord(ts.loop_ev[2 * k], b->lstream));
        {
            const int n = g.n_streams;
            b->gate_total += (unsigned) n; // a wavefront per stream counts itself in
            b->gate_first = b->gate_total; // the census once this launch is resident: the next item's kernels start behind it
            mp3mi_launch_loop(b->T, g, b->xr[v.slot] + r * 576, b->psy[v.slot] + r, b->prep[v.slot] + r, b->bits_per_frame + v.s0,
                              (char *) b->loop_state + v.s0 * loop_state_bytes, b->ix + r * 576, b->side + v.s0 * (size_t) g.nf,
                              b->gate_count, place, b->lstream);
            CHK(hipGetLastError());
        }
        ts.kernels += 1;
        CHK(hipEventRecord(ts.loop_ev[2 * k + 1], b->lstream));
        CHK(hipEventRecord(b->ev_loop[v.ev], b->lstream));
        b->slot_used[v.ev] = 1;
        mp3mi_launch_format(b->T, g, b->ix + r * 576, b->side + v.s0 * (size_t) g.nf, b->bits_per_frame + v.s0, b->bitrate_index + v.s0,
                            out_dev + v.s0 * out_stride, out_stride, out_len_dev + v.s0, (int32_t *) ((char *) b->loop_state + v.s0 * loop_state_bytes),
                            (int) (loop_state_bytes / 4), b->voided, b->lstream);
        CHK(hipGetLastError());
        if (hc && k == n_items - 1) {
            // The call's file bytes, behind its last formatter, as ONE copy: rows of the caller's stride when that is the device
            // buffer's (mp3mi_batch_out_stride(b, max_frames): a plain copy, which the DMA engines take), a 2-D copy
            // otherwise.  (A window of columns per chunk -- the bytes that became final with it -- was measured first: the
            // runtime runs such rectangles as copy KERNELS, 11 ms each beside the encoder's own, and the step lost 56 ms;
            // profiles/r04_experiments.txt.  With calls issued back to back this copy runs beside the next call's kernels.)
            mp3mi_batch::host_io &H = b->hio;
            const int sl = hc->slot;
            const size_t row = hc->out_stride < out_stride ? hc->out_stride : out_stride;
            CHK(hipEventRecord(H.ev_fmt[sl][0], b->lstream));
            CHK(hipStreamWaitEvent(H.d2h, H.ev_fmt[sl][0], 0));
            CHK(hipEventRecord(H.t_dn[sl][0], H.d2h));
            if (hc->out_stride == out_stride)
                CHK(hipMemcpyAsync(hc->out, out_dev, out_stride * (size_t) S, hipMemcpyDeviceToHost, H.d2h));
            else
                CHK(hipMemcpy2DAsync(hc->out, hc->out_stride, out_dev, out_stride, row, (size_t) S, hipMemcpyDeviceToHost, H.d2h));
            CHK(hipMemcpyAsync(hc->out_len, out_len_dev, sizeof(uint32_t) * (size_t) S, hipMemcpyDeviceToHost, H.d2h));
            CHK(hipEventRecord(H.t_dn[sl][1], H.d2h));
            H.bytes_dn[sl] += (double) row * S;
            CHK(hipEventRecord(H.out_free[sl], H.d2h));
            H.out_used[sl] = true;
        }
        b->last_nf = g.nf;
        b->last_slot = v.slot;
    }
    auto geom_of = [&](int c) { // the whole batch's geometry of chunk c (hand-over kernels below)
        const int f0 = c * cfr;
        const int nf = (n_frames - f0 < cfr) ? n_frames - f0 : cfr;
        mp3mi_geom g = mp3mi_make_geom(S, C, b->rate_idx, n_frames, f0, nf);
        g.test_flags = b->test_flags;
        g.n_samples = n_samples_dev;
        g.hdr_flags |= b->hdr_flags;
        g.hdr_mode = b->hdr_mode;
        g.crc = b->crc;
        g.fabs0 = fabs0;
        g.hist = b->pcm_hist;
        g.out_base = whole_file ? NULL : b->out_base;
        g.whole_file = whole_file ? 1 : 0;
        return g;
    };
    {   // hand over to the next call: PCM history (front stream: behind the last kernels that read the old one) and,
        // for a streaming call, what became final / what waits (loop stream: behind the last k_format)
        const mp3mi_geom g = geom_of(0);
        mp3mi_launch_hist_save(g, pcm_dev, b->pcm_hist, b->stream);
        CHK(hipGetLastError());
        if (!whole_file) {
            mp3mi_launch_stream_tail(g, 0, (int32_t *) b->loop_state, (int) (mp3mi_loop_state_size() / 4), b->bits_per_frame, out_dev,
                                     out_stride, b->out_base, b->carry, b->carry_len, out_len_dev, b->voided, b->lstream);
            CHK(hipGetLastError());
        }
        CHK(hipEventRecord(b->ev_hist, b->stream));
        CHK(hipStreamWaitEvent(b->lstream, b->ev_hist, 0)); // ev_done below then covers both streams
    }
    if (hc) { // (lstream has just joined the front stream: every kernel that reads the slot's PCM is ahead of this)
        CHK(hipEventRecord(b->hio.pcm_free[hc->slot], b->lstream));
        b->hio.pcm_used[hc->slot] = true;
        b->hio.pending[hc->slot] = true;
    }
    b->slot_base = (b->slot_base + nchunks) & 1;
    b->frames_done = whole_file ? 0 : fabs0 + n_frames; // a whole-file call leaves finished streams behind
    CHK(hipEventRecord(ts.ev1, b->lstream));
    ts.pending = true;
    b->call_no++;
    CHK(hipEventRecord(b->ev_done, b->lstream));
    b->have_done = true;
    CHK(hipGetLastError());
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_sync(mp3mi_batch *b)
{
    if (!b) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    hold_release(b);
    CHK(hipStreamSynchronize(b->stream));
    CHK(hipStreamSynchronize(b->lstream));
    if (b->hio.ready) { // host-buffer calls: the results are in the caller's memory when this returns
        CHK(hipStreamSynchronize(b->hio.h2d));
        CHK(hipStreamSynchronize(b->hio.d2h));
    }
    CHK(hipGetLastError());
    // streams whose file was voided since the last sync: the reference dies on those inputs (mp3mi.h)
    unsigned voided = 0;
    CHK(hipMemcpy(&voided, b->voided, sizeof(voided), hipMemcpyDeviceToHost));
    if (voided) {
        CHK(hipMemset(b->voided, 0, sizeof(unsigned)));
        return MP3MI_ERR_REFERENCE_ABORT;
    }
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_stream_status(mp3mi_batch *b, int32_t *status_host)
{
    if (!b || !status_host) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    hold_release(b);
    // (the state of the most recent streams: a whole-file call leaves it behind, the next call's reset clears it; a
    // flush gathers it before its reset)
    if (!b->status_kept) {
        mp3mi_launch_status_gather(b->n_streams, (const int32_t *) b->loop_state, (int) (mp3mi_loop_state_size() / 4), b->status_dev, b->lstream);
        CHK(hipGetLastError());
    }
    CHK(hipStreamSynchronize(b->stream));
    CHK(hipStreamSynchronize(b->lstream));
    CHK(hipMemcpy(status_host, b->status_dev, sizeof(int32_t) * (size_t) b->n_streams, hipMemcpyDeviceToHost));
    int n = 0;
    for (int s = 0; s < b->n_streams; s++) n += status_host[s] != 0;
    return n;
}

extern "C" int mp3mi_batch_debug_cw_fixups(mp3mi_batch *b, int *n_listed, int *n_records)
{
    if (!b || !n_listed || !n_records) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    hold_release(b);
    mp3mi_cw_fixlist h;
    CHK(hipMemcpy(&h, b->cw_fix, sizeof(h), hipMemcpyDeviceToHost));
    *n_listed = (int) h.count;
    *n_records = (int) h.cap;
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_debug_prep_fixups(mp3mi_batch *b, int *n_listed)
{
    if (!b || !n_listed) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    hold_release(b);
    mp3mi_prep_fixlist h;
    if (hipStreamSynchronize(b->stream) != hipSuccess) return MP3MI_ERR_HIP;
    CHK(hipMemcpy(&h, b->prep_fix, sizeof(h), hipMemcpyDeviceToHost));
    *n_listed = (int) h.count;
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_last_timing(mp3mi_batch *b, float *loop_kernel_ms, float *all_kernels_ms, int *launches)
{
    if (!b || !b->have_done || b->call_no == 0) return MP3MI_ERR_ARG; // nothing has been encoded yet
    ON_DEVICE(b);
    hold_release(b);
    if (harvest_timing(b, (int) (b->call_no & 1)) != MP3MI_OK || harvest_timing(b, (int) ((b->call_no - 1) & 1)) != MP3MI_OK) return MP3MI_ERR_HIP;
    if (loop_kernel_ms) *loop_kernel_ms = b->last_loop_ms;
    if (all_kernels_ms) *all_kernels_ms = b->last_all_ms;
    if (launches) *launches = b->last_launches;
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_total_timing(mp3mi_batch *b, double *loop_kernel_ms, double *all_kernels_ms, long *launches, long *calls)
{
    if (!b) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    hold_release(b);
    if (harvest_timing(b, (int) (b->call_no & 1)) != MP3MI_OK || harvest_timing(b, (int) ((b->call_no - 1) & 1)) != MP3MI_OK) return MP3MI_ERR_HIP;
    if (loop_kernel_ms) *loop_kernel_ms = b->tot_loop_ms;
    if (all_kernels_ms) *all_kernels_ms = b->tot_all_ms;
    if (launches) *launches = b->tot_launches;
    if (calls) *calls = b->tot_calls;
    return MP3MI_OK;
}

extern "C" long mp3mi_batch_debug_fetch(mp3mi_batch *b, int what, void *host_dst, size_t cap)
{
    if (!b || !host_dst) return MP3MI_ERR_ARG;
    const size_t ngc = (size_t) b->n_streams * 2 * (size_t) b->last_nf * (size_t) b->channels;
    const void *src = NULL;
    size_t n = 0;
    switch (what) {
    case 0: src = b->psy[b->last_slot]; n = ngc * sizeof(mp3mi_psy_out); break;
    case 1: src = b->xr[b->last_slot]; n = ngc * 576 * sizeof(double); break;
    case 2: src = b->ix; n = ngc * 576 * sizeof(int16_t); break;
    case 3: src = b->side; n = (size_t) b->n_streams * (size_t) b->last_nf * sizeof(mp3mi_frame_side); break;
    case 4: src = b->sb_dbg; n = ngc * 576 * sizeof(double); break;
    case 5: src = b->prep[b->last_slot]; n = ngc * sizeof(mp3mi_loop_prep); break;
    // the transforms' outputs as k_fft hands them to k_cw / k_part / k_psy (the direct FFT seam: oracle/fft_seam.h)
    case 6: src = b->energy_l; n = ngc * MP3MI_HBLK_P * sizeof(float); break;
    case 7: src = b->energy_s; n = ngc * 3 * MP3MI_HBLK_S * sizeof(float); break;
    case 8: src = b->fft_bins; n = ngc * MP3MI_FFT_BINS * sizeof(float); break;
    default: return MP3MI_ERR_ARG;
    }
    if (!src || n > cap) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    hold_release(b);
    if (hipStreamSynchronize(b->stream) != hipSuccess || hipStreamSynchronize(b->lstream) != hipSuccess) return MP3MI_ERR_HIP;
    if (hipMemcpy(host_dst, src, n, hipMemcpyDeviceToHost) != hipSuccess) return MP3MI_ERR_HIP;
    return (long) n;
}

// ---- host buffers in, host buffers out, overlapped with the encode ("end-to-end with PCIe") ----
// What the reference's driver does per frame with get_audio / read_samples (src/encode.c:123-269) and fwrite: here the
// PCM of a whole call crosses PCIe chunk by chunk on a copy stream while the chunks before it are encoded, and the
// file bytes come back behind each chunk's formatter (encode_impl).
static int host_io_harvest(mp3mi_batch *b, int sl)
{
    mp3mi_batch::host_io &H = b->hio;
    if (!H.pending[sl]) return MP3MI_OK;
    const int n = H.n_chunks[sl];
    CHK(hipEventSynchronize(H.t_dn[sl][1]));
    CHK(hipEventSynchronize(H.t_up[sl][2 * n - 1]));
    float ms = 0;
    for (int c = 0; c < n; c++) {
        CHK(hipEventElapsedTime(&ms, H.t_up[sl][2 * c], H.t_up[sl][2 * c + 1]));
        H.tot_up_ms += ms;
    }
    CHK(hipEventElapsedTime(&ms, H.t_dn[sl][0], H.t_dn[sl][1]));
    H.tot_dn_ms += ms;
    H.tot_up_bytes += H.bytes_up[sl];
    H.tot_dn_bytes += H.bytes_dn[sl];
    H.tot_calls++;
    H.pending[sl] = false;
    return MP3MI_OK;
}

// The copy streams with the first host-buffer call, a slot's device copies with the first call that uses the slot: a
// one-shot call (mp3mi_encode_host) pays for one slot, the second exists once two calls are in flight.  What a failed
// allocation leaves behind is freed by mp3mi_batch_destroy.
static int host_io_init(mp3mi_batch *b, int sl)
{
    mp3mi_batch::host_io &H = b->hio;
    const size_t S = (size_t) b->n_streams;
    if (!H.ready) {
        H.out_stride = mp3mi_batch_out_stride(b, b->max_frames);
        if (!H.h2d) CHK(hipStreamCreateWithFlags(&H.h2d, hipStreamNonBlocking));
        if (!H.d2h) CHK(hipStreamCreateWithFlags(&H.d2h, hipStreamNonBlocking));
        H.ready = true;
    }
    if (!H.pcm[sl]) CHK(hipMalloc((void **) &H.pcm[sl], S * (size_t) b->max_frames * 1152 * (size_t) b->channels * sizeof(int16_t)));
    if (!H.out[sl]) CHK(hipMalloc((void **) &H.out[sl], S * H.out_stride));
    if (!H.len[sl]) CHK(hipMalloc((void **) &H.len[sl], S * sizeof(uint32_t)));
    if (!H.pcm_free[sl]) CHK(hipEventCreateWithFlags(&H.pcm_free[sl], hipEventDisableTiming));
    if (!H.out_free[sl]) CHK(hipEventCreateWithFlags(&H.out_free[sl], hipEventDisableTiming));
    return MP3MI_OK;
}

extern "C" int mp3mi_batch_encode_host_async(mp3mi_batch *b, const int16_t *pcm_host, int n_frames, uint8_t *out_host, size_t out_stride,
                                             uint32_t *out_len_host)
{
    if (!b || !pcm_host || !out_host || !out_len_host || n_frames <= 0 || n_frames > b->max_frames) return MP3MI_ERR_ARG;
    if (out_stride < (size_t) n_frames * (size_t) b->max_frame_bytes + 1) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    mp3mi_batch::host_io &H = b->hio;
    const int sl = (int) (H.call_no & 1);
    if (host_io_init(b, sl) != MP3MI_OK) return MP3MI_ERR_HIP;
    if (host_io_harvest(b, sl) != MP3MI_OK) return MP3MI_ERR_HIP; // (waits for the call two before this one: at most two in flight)
    const host_call hc = {pcm_host, out_host, out_stride, out_len_host, sl};
    const int rc = encode_impl(b, H.pcm[sl], NULL, n_frames, H.out[sl], H.out_stride, H.len[sl], true, &hc);
    if (rc == MP3MI_OK) H.call_no++; // (a call that failed took no slot)
    return rc;
}

extern "C" int mp3mi_batch_host_io_stats(mp3mi_batch *b, mp3mi_host_io_stats *st)
{
    if (!b || !st) return MP3MI_ERR_ARG;
    ON_DEVICE(b);
    memset(st, 0, sizeof(*st));
    if (!b->hio.ready) return MP3MI_OK;
    hold_release(b);
    if (host_io_harvest(b, 0) != MP3MI_OK || host_io_harvest(b, 1) != MP3MI_OK) return MP3MI_ERR_HIP;
    st->calls = b->hio.tot_calls;
    st->h2d_bytes = b->hio.tot_up_bytes; st->d2h_bytes = b->hio.tot_dn_bytes;
    st->h2d_ms = b->hio.tot_up_ms; st->d2h_ms = b->hio.tot_dn_ms;
    return MP3MI_OK;
}

// page-locked host memory for the calls above (a caller that does not link HIP itself)
extern "C" void *mp3mi_host_alloc(size_t bytes)
{
    void *p = NULL;
    if (!have_device() || hipHostMalloc(&p, bytes, 0) != hipSuccess) return NULL;
    return p;
}
extern "C" void mp3mi_host_free(void *p) { if (p) (void) hipHostFree(p); }

static int encode_host_impl(int n_streams, int rate_hz, int channels, const int *kbps, int kbps_all, const int16_t *pcm,
                            const int32_t *n_samples, int n_frames, int hdr, uint8_t *out, size_t out_stride, uint32_t *out_len)
{
    mp3mi_batch *b = NULL;
    int rc = mp3mi_batch_create(&b, n_streams, rate_hz, channels, kbps, kbps_all, n_frames);
    if (rc != MP3MI_OK) return rc;
    if (hdr >= 0) rc = mp3mi_batch_set_header(b, (hdr >> 3) & 1, (hdr >> 2) & 1, hdr & 3);
    const size_t pcm_bytes = (size_t) n_streams * (size_t) n_frames * 1152 * (size_t) channels * sizeof(int16_t);
    int16_t *pcm_d = NULL;
    uint8_t *out_d = NULL;
    uint32_t *len_d = NULL;
    int32_t *ns_d = NULL;
    if (rc == MP3MI_OK && !n_samples) { // whole streams: the overlapped host path (pageable buffers here: staged by the runtime)
        rc = mp3mi_batch_encode_host_async(b, pcm, n_frames, out, out_stride, out_len);
        if (rc == MP3MI_OK) rc = mp3mi_batch_sync(b); // (MP3MI_ERR_REFERENCE_ABORT: done, the outputs are delivered -- mp3mi.h)
        mp3mi_batch_destroy(b);
        return rc;
    }
    if (rc == MP3MI_OK) rc = MP3MI_ERR_HIP;
    if (rc == MP3MI_ERR_HIP && hipMalloc((void **) &pcm_d, pcm_bytes) == hipSuccess &&
        hipMalloc((void **) &out_d, out_stride * n_streams) == hipSuccess &&
        hipMalloc((void **) &len_d, sizeof(uint32_t) * n_streams) == hipSuccess &&
        hipMalloc((void **) &ns_d, sizeof(int32_t) * n_streams) == hipSuccess &&
        hipMemcpy(pcm_d, pcm, pcm_bytes, hipMemcpyHostToDevice) == hipSuccess &&
        (!n_samples || hipMemcpy(ns_d, n_samples, sizeof(int32_t) * n_streams, hipMemcpyHostToDevice) == hipSuccess)) {
        rc = n_samples ? mp3mi_batch_encode_ragged(b, pcm_d, ns_d, n_frames, out_d, out_stride, len_d)
                       : mp3mi_batch_encode(b, pcm_d, n_frames, out_d, out_stride, len_d);
        if (rc == MP3MI_OK) rc = mp3mi_batch_sync(b);
        // MP3MI_ERR_REFERENCE_ABORT means "done, and some stream is an input the reference dies on": that stream's out_len is
        // 0 and every other stream's output is valid (mp3mi.h), so the results are delivered and the code is kept
        if ((rc == MP3MI_OK || rc == MP3MI_ERR_REFERENCE_ABORT) &&
            (hipMemcpy(out, out_d, out_stride * n_streams, hipMemcpyDeviceToHost) != hipSuccess ||
             hipMemcpy(out_len, len_d, sizeof(uint32_t) * n_streams, hipMemcpyDeviceToHost) != hipSuccess))
            rc = MP3MI_ERR_HIP;
    }
    if (pcm_d) hipFree(pcm_d);
    if (out_d) hipFree(out_d);
    if (len_d) hipFree(len_d);
    if (ns_d) hipFree(ns_d);
    mp3mi_batch_destroy(b);
    return rc;
}

extern "C" int mp3mi_encode_host(int n_streams, int rate_hz, int channels, const int *kbps, int kbps_all,
                                 const int16_t *pcm, int n_frames, uint8_t *out, size_t out_stride,
                                 uint32_t *out_len)
{
    return encode_host_impl(n_streams, rate_hz, channels, kbps, kbps_all, pcm, NULL, n_frames, -1, out, out_stride, out_len);
}

extern "C" int mp3mi_encode_host_ex(int n_streams, int rate_hz, int channels, const int *kbps, int kbps_all,
                                    const int16_t *pcm, const int32_t *n_samples, int n_frames, int copyright,
                                    int original, int emphasis, uint8_t *out, size_t out_stride, uint32_t *out_len)
{
    if ((copyright & ~1) || (original & ~1) || (emphasis & ~3)) return MP3MI_ERR_ARG;
    return encode_host_impl(n_streams, rate_hz, channels, kbps, kbps_all, pcm, n_samples, n_frames,
                            (copyright << 3) | (original << 2) | emphasis, out, out_stride, out_len);
}

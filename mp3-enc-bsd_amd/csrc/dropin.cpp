// The reference's per-frame Layer III call surface (include/mp3mi_dropin.h) over the HIP
// kernels: one hidden default stream, n_streams = 1.  Host code only marshals caller-owned
// arrays to and from the device -- through host-mapped buffers that the kernels read and write in place (a call is
// its launches and ONE wait; a few kilobytes per call cross the bus either way) -- and reproduces the caller-visible side effects of each
// function (savebuf shift, buffer pointer advance, in-place sign flips, back pointer); every
// number that ends up in the bitstream is computed by the kernels.
//
// Error behaviour mirrors the reference: these functions return void, so failures print a
// message and exit/abort (src/l3psy.c:170-176, src/l3psy.c:665-666, asserts throughout).
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mp3mi_host.h"
#include "mp3mi.h"
#include "mp3mi_dropin.h"

size_t mp3mi_psy_state_size(void);
size_t mp3mi_loop_state_size(void);
void mp3mi_launch_filter_subband(const mp3mi_tables *T, const double *z, double *s, hipStream_t st);
void mp3mi_launch_window_filter(const mp3mi_tables *T, double *ring, int off, const mp3mi_dropin_samples &in, double *zs, hipStream_t st);
void mp3mi_launch_mdct_sub(const mp3mi_tables *T, const double *sb_in, double *sb_out, const int32_t *bt, double *xr, double *xr_dev, int stereo, int mode_gr,
                           unsigned *zero_me, unsigned *flag, unsigned seq, unsigned *count, hipStream_t st);
void mp3mi_launch_format_marked(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *ix, const mp3mi_frame_side *side,
                                const int32_t *bits_per_frame, const int32_t *bitrate_index, uint8_t *out, size_t out_stride, uint32_t *out_len,
                                unsigned *flag, unsigned seq_before, unsigned seq_done, int16_t *ix_host, mp3mi_frame_side *side_host, hipStream_t st);
void mp3mi_launch_window_filter_frame(const mp3mi_tables *T, const double *ring, int off0_a, int off0_b, const int16_t *samples, int n_ch, int n_slots,
                                      double *zs, double *sb_out, hipStream_t st);
void mp3mi_launch_dropin_done(unsigned *flag, unsigned seq, hipStream_t st);
void mp3mi_launch_part_wave(const mp3mi_tables *T, const mp3mi_geom &g, const float *energy_l, const double *cw_mid, const float *hist6,
                            const void *psy_state, double *eb_all, float *cb_all, hipStream_t st);
void mp3mi_launch_prep_tail(const mp3mi_tables *T, const mp3mi_geom &g, const double *xr, const mp3mi_psy_out *psy, mp3mi_loop_prep *prep,
                            mp3mi_prep_fixlist *fix, hipStream_t st);

#define DIE(...)                                   \
    do {                                           \
        fprintf(stderr, "mp3mi (drop-in): ");      \
        fprintf(stderr, __VA_ARGS__);              \
        fprintf(stderr, "\n");                     \
        abort();                                   \
    } while (0)
#define HIPOK(call)                                                          \
    do {                                                                     \
        hipError_t e_ = (call);                                              \
        if (e_ != hipSuccess) DIE("%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

namespace {

const int WIN_FRAMES = 16; // frames of history kept in the formatter's byte window

// mirrors mp3mi_loop_state in k_loop.hip (all int32)
struct loop_state_host {
    int32_t ResvSize;
    int32_t sc_en_tot[2][2], sc_en[2][2][21], sc_xm[2][2][21], sc_xrmax[2][2];
    int32_t addr[2][2][3];
    int32_t ref_abort;
};

// a buffer that crosses the boundary: host-mapped, h for the host code, d for the kernels
template <typename T> struct io_buf {
    T *h = nullptr, *d = nullptr;
    void alloc(size_t bytes)
    {
        HIPOK(hipHostMalloc((void **) &h, bytes, hipHostMallocMapped));
        HIPOK(hipHostGetDevicePointer((void **) &d, h, 0));
        memset(h, 0, bytes);
    }
};

struct DropIn {
    bool ready = false;
    int rate_idx = -1;
    hipStream_t st = 0;
    mp3mi_tables *T = nullptr;
    // psy
    io_buf<int16_t> pcm;
    float *el = nullptr, *es = nullptr, *h6 = nullptr, *bins = nullptr, *part_cb = nullptr;
    double *part_eb = nullptr;
    double *cw = nullptr;
    mp3mi_cw_fixlist *cw_fix = nullptr;
    void *psy_state = nullptr;
    io_buf<mp3mi_psy_out> psy1;
    // filterbank
    io_buf<double> ring; // [2][512]: the filterbank's sample history, host-mapped: the kernels of the call-by-call service
                         // write it, and so does the host when it closes a look-ahead (window_ahead_close)
    double *z_d = nullptr, *s_d = nullptr;
    int off[2] = {0, 0};
    // window_subband computes the filter_subband that follows it in the same launch: z[512] and s[32] arrive in a
    // host-mapped buffer, and filter_subband hands s out when the z it is given is still the one it got (compared)
    double *zs_h = nullptr, *zs_d = nullptr;
    double last_z[512], last_s[32];
    bool have_s = false;
    // LOOK-AHEAD (round 4).  The reference's Layer III frame loop (src/musicin.c:751-769) calls L3psycho_anal four times and
    // window_subband / filter_subband 72 times per frame -- 76 launches with a wait each when every call is served on its own.
    // Both work on memory the caller has ALREADY handed over:
    //   * window_subband: the four L3psycho_anal calls of a frame were given &buffer[ch][0] and &buffer[ch][576]; when a
    //     channel's first window_subband of the frame starts at that same &buffer[ch][0], the frame's 36 slots of BOTH
    //     channels are computed in one launch and handed out call by call -- as long as every call's pointer is where the
    //     previous one left it and its 32 samples still are what was read ahead.  Anything else (a caller that moves or
    //     rewrites its buffer between calls, the Layer I / II loops, which never call L3psycho_anal) is served call by
    //     call as before, from a ring that is first brought to where the handed-out slots left it.
    //   * L3psycho_anal: when a channel's two calls of the LAST frame were given p and p + 576 and this frame's first one
    //     is given the same p again, both granules are analysed in one launch; the second call is served from it if its
    //     pointer, its 576 samples and the delay line are what was read ahead, otherwise the channel's state is put back
    //     (a device copy taken before the launch) and granule 0 is analysed again alone.
    // MP3MI_DROPIN_LOOKAHEAD=0 turns both off.
    bool lookahead = true, lookahead_psy = false; // (MP3MI_DROPIN_LOOKAHEAD: 0 none, 1 all, 2 the filterbank's only, 3 L3psycho_anal's only, 4 all but iteration_loop's / III_format_bitstream's)
    struct win_ahead {
        bool valid = false;
        const short *p0 = nullptr; // where the channel's slot 0 was read
        int next = 0;              // slots handed out
    } wa[2];
    io_buf<int16_t> wa_smp;        // [2][1152]: the samples read ahead
    io_buf<double> wa_zs;          // [2][36][544]: z and s of every slot
    const short *psy_ptr[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; // this frame's L3psycho_anal buffers, [ch][gr]
    int psy_seen[2] = {0, 0};      // bit gr: seen since the channel's last window look-ahead
    const short *psy_prev[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}}; // the frame before
    struct psy_ahead { // per channel: what the frame's look-ahead holds for the channel's calls still to come
        bool valid[2] = {false, false}; // [granule]: analysed ahead, not handed out yet
        const short *p[2] = {nullptr, nullptr}; // where the call's 576 samples were read
        short smp[2][576];         // ... and what they were
        short save_before[2][1344]; // the delay line the call must find (before its own shift)
        mp3mi_psy_out out[2];
        int served = 0;            // granules of the frame handed out from the look-ahead so far (replayed on a fall-back)
    } pa[2];
    const short *psy_save_ptr[2] = {nullptr, nullptr}; // the delay lines (savebuf) the channels' calls were given last
    // mdct_sub ahead of its call: the filterbank look-ahead knows the frame's subband samples, L3psycho_anal has handed out its
    // block types -- the transform is launched right behind the filterbank kernel (one wait for both) and mdct_sub hands its
    // result out IF the caller's L3SBS and block types are what it was computed from (compared value by value)
    int bt_pred[2][2] = {{0, 0}, {0, 0}}; // [gr][ch], as handed out by this frame's L3psycho_anal calls
    int bt_known = 0;                     // bit gr * 2 + ch
    struct { bool valid = false; int n_ch = 0; double prev[2][576]; } spec; // prev: granule slot 0 of the L3SBS before the launch
    io_buf<mp3mi_psy_out> psy2;    // the look-ahead's results: [granule][channel]
    void *psy_snap = nullptr;      // [2] psychoacoustic state before a two-granule launch
    io_buf<unsigned> done_flag;    // what the host spins on at the end of a call's launches (k_dropin_done)
    unsigned done_seq = 0, done_seen = 0; // marks put on the stream so far / the latest one a call has waited for
    long n_launch_waits = 0;       // (statistics: marks a call had to wait for, mp3mi_dropin_waits)
    long n_loop_ahead = 0, n_fmt_ahead = 0; // (statistics: iteration_loop / III_format_bitstream calls served from a launch ahead of the call)
    // iteration_loop and III_format_bitstream AHEAD of their calls.  k_format is launched right behind k_loop, from the records
    // k_loop leaves on the device; III_format_bitstream hands its bytes out if the caller's l3_enc, side information and
    // scalefactors still are what iteration_loop returned.  And when the frame's filterbank look-ahead has launched mdct_sub
    // (spec) and L3psycho_anal's look-ahead holds all of the frame's records, both follow on the stream at once -- with the
    // frame length, header bits and channel count of the frame BEFORE -- and iteration_loop hands the result out if its
    // arguments (pe, ratio, block types, spectrum, mean_bits, header) are exactly what was read.  Anything else: the loop's
    // state and the formatter's byte window are put back (copies taken before the launch) and the call is served on its own.
    bool lookahead_loop = false;
    struct frame_ahead {
        bool loop_pending = false, fmt_pending = false; // launched, not handed out yet
        bool have_last = false;        // the parameters below are a served frame's
        int C = 0, crc = 0, bitsPerFrame = 0, bitrate_index = 0, mode = 0, hdr_flags = 0;
        const mp3mi_psy_out *rec_h = nullptr; // (host view of) the records k_loop read
        unsigned seq_loop = 0, seq_fmt = 0;
        loop_state_host state_before;
        uint8_t *win_before = nullptr;
        size_t win_before_bytes = 0;
    } fa;
    int psy2_n_ch = 0, psy2_served = 0; // the frame's L3psycho_anal look-ahead: channels in D.psy2, records handed out as foreseen
    bool win_for_run = false;      // the formatter's window exists for this run of frames (since the last III_FlushBitstream)
    long win_slid_for = -1;        // frame the window has been positioned for
    bool stats = false;            // options.dropin_stats: a line at III_FlushBitstream
    double t_first = 0.0;          // when the first frame's first call came in (seconds, steady clock)
    long frames_total = 0;
    // mdct: the caller's L3SBS as mdct_sub finds it (sbuf[sb_cur]: block 0 = what the call before left there) and as it leaves it
    // (sbuf[sb_cur ^ 1]); the two change places with every call (k_dropin.hip, k_mdct_sub)
    io_buf<double> sbuf[2], xr;
    int sb_cur = 0;
    unsigned *mdct_count = nullptr; // (k_mdct_sub's workgroups count themselves out)
    // device-memory copies of what a frame's launches ahead hand to each other (the host-mapped buffers are for the host: a
    // kernel that reads them in dependent steps waits for the bus every time): the spectrum (k_mdct_sub -> k_loop), the
    // quantised values and the side information (k_loop -> k_format_marked, which writes the host's copies)
    double *xr_dev = nullptr;
    int16_t *ix_dev = nullptr;
    mp3mi_frame_side *side_dev = nullptr;
    io_buf<int32_t> bt;
    // loop
    io_buf<mp3mi_psy_out> psy4;
    mp3mi_loop_prep *prep4 = nullptr;
    mp3mi_prep_fixlist *prep_fix = nullptr;
    io_buf<int16_t> ix;
    io_buf<mp3mi_frame_side> side;
    io_buf<loop_state_host> loop_state;
    io_buf<int32_t> bits, bri;
    bool loop_first = true;
    // format
    io_buf<uint8_t> win; // the formatter's byte window: k_format places the frames, emit() reads them in place
    uint32_t *len_d = nullptr;
    size_t win_bytes = 0;
    long frames_done = 0, abs_emitted = 0, m_end = 0;
    int frame_bytes = 0, si_bytes = 0;
    void (*putbits)(Bit_stream_struc *, unsigned int, int) = nullptr;
    Bit_stream_struc *bs = nullptr;
};

DropIn D;

void ensure(int rate_idx)
{
    if (D.ready) {
        if (rate_idx != D.rate_idx) DIE("sampling frequency changed between calls");
        return;
    }
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) DIE("no HIP device available -- this library has no CPU path");
    mp3mi_tables *Th = (mp3mi_tables *) malloc(sizeof(mp3mi_tables));
    if (!Th || mp3mi_build_tables(Th, rate_idx) != 0) DIE("cannot build tables for rate index %d", rate_idx);
    HIPOK(hipStreamCreate(&D.st));
    HIPOK(hipMalloc((void **) &D.T, sizeof(mp3mi_tables)));
    HIPOK(hipMemcpy(D.T, Th, sizeof(mp3mi_tables), hipMemcpyHostToDevice));
    free(Th);
    D.pcm.alloc(2 * 2304 * sizeof(int16_t)); // (two channels interleaved at most)
    HIPOK(hipMalloc((void **) &D.el, 4 * MP3MI_HBLK_P * sizeof(float))); // (four records: a frame's two granules of two channels in one launch)
    HIPOK(hipMalloc((void **) &D.part_eb, 4 * MP3MI_PART_P * sizeof(double)));
    HIPOK(hipMalloc((void **) &D.part_cb, 4 * MP3MI_PART_P * sizeof(float)));
    HIPOK(hipMalloc((void **) &D.es, 4 * 3 * MP3MI_HBLK_S * sizeof(float)));
    HIPOK(hipMalloc((void **) &D.h6, 4 * 12 * sizeof(float)));
    HIPOK(hipMalloc((void **) &D.bins, 4 * MP3MI_FFT_BINS * sizeof(float)));
    HIPOK(hipMalloc((void **) &D.cw, 4 * 50 * sizeof(double)));
    HIPOK(hipMalloc((void **) &D.cw_fix, mp3mi_cw_fixlist_bytes(4)));
    HIPOK(hipMalloc((void **) &D.psy_state, 2 * mp3mi_psy_state_size()));
    HIPOK(hipMemset(D.psy_state, 0, 2 * mp3mi_psy_state_size()));
    D.psy1.alloc(sizeof(mp3mi_psy_out));
    D.ring.alloc(2 * 512 * sizeof(double));
    HIPOK(hipMalloc((void **) &D.z_d, 512 * sizeof(double)));
    HIPOK(hipMalloc((void **) &D.s_d, 32 * sizeof(double)));
    HIPOK(hipHostMalloc((void **) &D.zs_h, (512 + 32) * sizeof(double), hipHostMallocMapped));
    HIPOK(hipHostGetDevicePointer((void **) &D.zs_d, D.zs_h, 0));
    D.done_flag.alloc(64);
    D.wa_smp.alloc(2 * 1152 * sizeof(int16_t));
    D.wa_zs.alloc((size_t) 2 * 36 * 544 * sizeof(double));
    D.psy2.alloc(4 * sizeof(mp3mi_psy_out));
    HIPOK(hipMalloc(&D.psy_snap, 2 * mp3mi_psy_state_size()));
    {
        mp3mi_batch_options o; // (the one place the library reads its environment: batch.cpp)
        mp3mi_batch_options_from_env(&o);
        // Default 2: the filterbank's look-ahead (with mdct_sub behind it) reads only memory handed over during the CURRENT
        // frame.  L3psycho_anal's (and with it the loop's and the formatter's launches ahead, which need its records) reads
        // buffers at the addresses REMEMBERED from the frame before -- fine under the reference's driver, whose buffers are
        // static arrays, a use-after-free under a caller that allocates or rotates them per frame: opt-in (1), with the
        // lifetime requirement stated in mp3mi_dropin.h.
        const int v = o.dropin_lookahead < 0 ? 2 : o.dropin_lookahead;
        D.lookahead = v == 1 || v == 2 || v == 4;
        D.lookahead_psy = v == 1 || v == 3 || v == 4;
        D.lookahead_loop = v == 1;
        D.stats = o.dropin_stats == 1;
    }
    D.sbuf[0].alloc(sizeof(L3SBS));
    D.sbuf[1].alloc(sizeof(L3SBS));
    HIPOK(hipMalloc((void **) &D.mdct_count, sizeof(unsigned)));
    HIPOK(hipMalloc((void **) &D.xr_dev, 4 * 576 * sizeof(double)));
    HIPOK(hipMalloc((void **) &D.ix_dev, 4 * 576 * sizeof(int16_t)));
    HIPOK(hipMalloc((void **) &D.side_dev, sizeof(mp3mi_frame_side)));
    HIPOK(hipMemset(D.mdct_count, 0, sizeof(unsigned)));
    D.xr.alloc(4 * 576 * sizeof(double));
    D.bt.alloc(4 * sizeof(int32_t));
    D.psy4.alloc(4 * sizeof(mp3mi_psy_out));
    HIPOK(hipMalloc((void **) &D.prep4, 4 * sizeof(mp3mi_loop_prep)));
    HIPOK(hipMalloc((void **) &D.prep_fix, mp3mi_prep_fixlist_bytes(4)));
    D.ix.alloc(4 * 576 * sizeof(int16_t));
    D.side.alloc(sizeof(mp3mi_frame_side));
    if (sizeof(loop_state_host) != mp3mi_loop_state_size()) DIE("internal: loop state layout mismatch");
    D.loop_state.alloc(sizeof(loop_state_host));
    D.bits.alloc(sizeof(int32_t));
    D.bri.alloc(sizeof(int32_t));
    HIPOK(hipMalloc((void **) &D.len_d, sizeof(uint32_t)));
    D.rate_idx = rate_idx;
    D.ready = true;
}

int rate_index_of(double sfreq)
{
    const unsigned i = (unsigned) (sfreq + 0.5); // src/l3psy.c:168-176
    switch (i) {
    case 44100: return 0;
    case 48000: return 1;
    case 32000: return 2;
    default:
        printf("error, invalid sampling frequency: %d Hz\n", i);
        exit(-1);
    }
}

long emitted_upto(long m, int slot, int frame_bytes, int si_bytes)
{ // file bytes that are final once m bytes of main data have been written
    if (m == 0) return 0;
    return ((m - 1) / slot) * (long) frame_bytes + si_bytes + ((m - 1) % slot) + 1;
}

void emit(long upto, long base)
{
    if (upto <= D.abs_emitted) return;
    const size_t n = (size_t) (upto - D.abs_emitted);
    if (D.abs_emitted - base < 0 || (size_t) (upto - base) > D.win_bytes) DIE("internal: formatter window too small");
    const uint8_t *src = D.win.h + (D.abs_emitted - base); // (the stream was waited for by the caller)
    for (size_t i = 0; i < n; i++) D.putbits(D.bs, src[i], 8);
    D.abs_emitted = upto;
}

} // namespace

static double now_s()
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

// dropin_wait: waits for everything launched on the hidden stream so far; dropin_mark / dropin_wait_for: for everything launched up
// to a mark.  A flag in host-mapped memory, stored by a one-thread kernel behind the
// launches, is seen by the spinning host a few microseconds after the store; hipStreamSynchronize alone takes ~25 us to
// come back, and a frame has four such waits.  The spin is bounded (a device that does not answer within two seconds is the
// runtime's business).
static unsigned dropin_mark()
{
    const unsigned seq = ++D.done_seq;
    mp3mi_launch_dropin_done(D.done_flag.d, seq, D.st);
    return seq;
}

// (marks are waited for in the order they were put)
static void dropin_wait_for(unsigned seq)
{
    if ((int) (D.done_seen - seq) >= 0) return;
    volatile unsigned *f = D.done_flag.h;
    const double t0 = now_s();
    for (unsigned spin = 0; (int) (*f - seq) < 0; spin++)
        if ((spin & 0xfffu) == 0xfffu && now_s() - t0 > 2.0) break;
    if ((int) (*f - seq) < 0) HIPOK(hipStreamSynchronize(D.st));
    D.done_seen = seq;
    D.n_launch_waits++;
}

static void dropin_wait() { dropin_wait_for(dropin_mark()); }

static void psy_wait() { dropin_wait(); }
static void frame_chain_launch(const mp3mi_psy_out *rec_d, const mp3mi_psy_out *rec_h, bool with_format, bool behind_mdct);
static void frame_chain_cancel();
static bool format_setup(int frame_bytes, int si_bytes);

// the kernels behind one L3psycho_anal launch (g: one or two granules of a mono pseudo-stream).  The unpredictability comes
// from the correctly rounded sines throughout (k_cw's second tier: MP3MI_TEST_CW_EXACT) and the partition sums from
// k_part_wave, a lane per partition: with a record or two per launch, k_part's lane per RECORD and its two runs around the
// list of doubtful records cost 4 x 64 us per frame.
static void psy_launch(mp3mi_geom g, int chn, mp3mi_psy_out *dst_d)
{
    void *state = (char *) D.psy_state + (size_t) chn * mp3mi_psy_state_size();
    g.test_flags |= 16; // k_cw: second tier for every record
    mp3mi_launch_fft(D.T, g, D.pcm.d, D.el, D.es, D.bins, D.cw, D.h6, D.st);
    mp3mi_launch_part_wave(D.T, g, D.el, D.cw, D.h6, state, D.part_eb, D.part_cb, D.st);
    mp3mi_launch_psy(D.T, g, D.el, D.es, D.cw, D.h6, D.bins, D.cw_fix, state, D.part_eb, D.part_cb, dst_d, D.st, 2);
}

// one granule of one channel, the reference's way: the delay line as the caller holds it (already shifted)
static void psy_one_granule(const short *savebuf, int chn, mp3mi_psy_out *dst_d)
{
    // present the 1344-sample window to k_fft as granule 2 of a mono pseudo-stream: its window
    // starts at sample 576*2 - 768 = 384
    memset(D.pcm.h, 0, 2304 * sizeof(int16_t));
    memcpy(D.pcm.h + 384, savebuf, 1344 * sizeof(int16_t));
    mp3mi_geom g = mp3mi_make_geom(1, 1, D.rate_idx, 2, 1, 1);
    g.g0 = 2;
    g.n_gran = 1;
    psy_launch(g, chn, dst_d);
}

static void psy_hand_out(const mp3mi_psy_out &o, double ratio_d[21], double ratio_ds[12][3], double *pe, gr_info *cod_info, int gr, int ch)
{
    D.bt_pred[gr][ch] = o.block_type;
    D.bt_known |= 1 << (gr * 2 + ch);
    memcpy(ratio_d, o.ratio_l, sizeof(o.ratio_l));
    memcpy(ratio_ds, o.ratio_s, sizeof(o.ratio_s));
    *pe = o.pe;
    cod_info->block_type = (unsigned) o.block_type;
    cod_info->window_switching_flag = (o.block_type == 0) ? 0 : 1;
    cod_info->mixed_block_flag = 0;
}

// The look-ahead of a frame: both granules of the channels [first, first + n_ch) in ONE launch.  P[c]: the channel's 1152
// samples; delay[c]: its delay line as its first call of the frame finds it (for the channel whose call this is: see the
// caller).  Fills D.pa[c] and leaves the results in D.psy2.h[granule * n_ch + (c - first)].
static void psy_ahead_start(int first, int n_ch, const short *const P[2], const short *const delay_before[2])
{
    const size_t ss = mp3mi_psy_state_size();
    HIPOK(hipMemcpyAsync((char *) D.psy_snap + (size_t) first * ss, (char *) D.psy_state + (size_t) first * ss, (size_t) n_ch * ss, hipMemcpyDeviceToDevice, D.st));
    // the pseudo-stream of a channel: granule 2's window is samples [384, 1728) = the delay line after the first call's
    // shift, granule 3's [960, 2304) = the same shifted by the second granule's samples
    memset(D.pcm.h, 0, (size_t) 384 * n_ch * sizeof(int16_t));
    for (int c = first; c < first + n_ch; c++) {
        DropIn::psy_ahead &A = D.pa[c];
        const int k = c - first;
        for (int g = 0; g < 2; g++) {
            A.valid[g] = true;
            A.p[g] = P[c] + 576 * g;
            memcpy(A.smp[g], P[c] + 576 * g, sizeof(A.smp[g]));
        }
        memcpy(A.save_before[0], delay_before[c], sizeof(A.save_before[0]));
        memcpy(A.save_before[1], delay_before[c] + 576, 768 * sizeof(short)); // the first call's shift (src/l3psy.c:477-481)
        memcpy(A.save_before[1] + 768, P[c], 576 * sizeof(short));
        A.served = 0;
        for (int i = 0; i < 1344; i++) D.pcm.h[(384 + i) * n_ch + k] = A.save_before[1][i];
        for (int i = 0; i < 576; i++) D.pcm.h[(1728 + i) * n_ch + k] = P[c][576 + i];
    }
    mp3mi_geom g = mp3mi_make_geom(1, n_ch, D.rate_idx, 2, 1, 1);
    g.g0 = 2;
    g.n_gran = 2;
    psy_launch(g, first, D.psy2.d);
    psy_wait();
    for (int c = first; c < first + n_ch; c++)
        for (int gr = 0; gr < 2; gr++) D.pa[c].out[gr] = D.psy2.h[gr * n_ch + (c - first)];
}

extern "C" void L3psycho_anal(short int *buffer, short int savebuf[1344], int chn, int lay, float snr32[32],
                              double sfreq, double ratio_d[21], double ratio_ds[12][3], double *pe,
                              gr_info *cod_info)
{
    (void) snr32;
    if (lay != 3) DIE("L3psycho_anal: layer %d is not served by this library", lay);
    if (chn < 0 || chn > 1) DIE("L3psycho_anal: channel %d", chn);
    ensure(rate_index_of(sfreq));
    if (D.t_first == 0.0) D.t_first = now_s();
    frame_chain_cancel(); // (a frame's loop / formatter launched ahead whose calls never came: the next frame starts from the state they found)
    DropIn::psy_ahead &A = D.pa[chn];
    // which of the frame's calls is this (for the window look-ahead's bookkeeping)?
    const bool first_of_frame = !(D.psy_seen[chn] & 1) || (D.psy_seen[chn] & 2);
    const short *const prev0 = D.psy_ptr[chn][0], *const prev1 = D.psy_ptr[chn][1]; // the frame before, if this is a first call
    const int ga = A.valid[0] ? 0 : (A.valid[1] ? 1 : -1); // the granule the look-ahead expects next for this channel
    const bool foreseen = ga >= 0 && buffer == A.p[ga] && memcmp(buffer, A.smp[ga], sizeof(A.smp[ga])) == 0 &&
                          memcmp(savebuf, A.save_before[ga], sizeof(A.save_before[ga])) == 0;
    // delay line: drop the oldest 576 samples, append the new ones (src/l3psy.c:477-481)
    memmove(savebuf, savebuf + 576, 768 * sizeof(short));
    memcpy(savebuf + 768, buffer, 576 * sizeof(short));
    if (first_of_frame) {
        D.psy_prev[chn][0] = prev0; D.psy_prev[chn][1] = prev1;
        D.psy_ptr[chn][0] = buffer; D.psy_ptr[chn][1] = nullptr;
        D.psy_seen[chn] = 1;
    } else {
        D.psy_ptr[chn][1] = buffer;
        D.psy_seen[chn] |= 2;
    }
    D.psy_save_ptr[chn] = savebuf;
    if (ga >= 0) {
        if (foreseen) { // the call the look-ahead was made for
            A.valid[ga] = false;
            A.served = ga + 1;
            D.psy2_served++;
            psy_hand_out(A.out[ga], ratio_d, ratio_ds, pe, cod_info, first_of_frame ? 0 : 1, chn);
            return;
        }
        // not what was read ahead: the channel's state goes back to before the look-ahead, the granules handed out from it
        // are analysed again one by one (from the delay lines they found), then this call on its own
        const size_t ss = mp3mi_psy_state_size();
        D.psy2_n_ch = 0; // (the frame's records are no longer all the look-ahead's)
        HIPOK(hipMemcpyAsync((char *) D.psy_state + (size_t) chn * ss, (char *) D.psy_snap + (size_t) chn * ss, ss, hipMemcpyDeviceToDevice, D.st));
        for (int q = 0; q < A.served; q++) {
            psy_one_granule(A.save_before[q + 1], chn, D.psy1.d); // (served <= 1: the delay line after the first call = what the second must find)
            psy_wait(); // (D.pcm is host memory the kernels read: the next launch's samples may only be written now)
        }
        A.valid[0] = A.valid[1] = false;
        psy_one_granule(savebuf, chn, D.psy1.d);
        psy_wait();
        psy_hand_out(*D.psy1.h, ratio_d, ratio_ds, pe, cod_info, first_of_frame ? 0 : 1, chn);
        return;
    }
    // Nothing was read ahead for this channel.  A first call of a frame starts a look-ahead when the channel's calls of the
    // frame BEFORE were given p and p + 576 and this one is given the same p: [p, p + 1152) is memory the caller has handed
    // over before.  The other channel joins in when its own calls of the frame before show the same pattern -- its buffer
    // and its delay line were handed over then, at the addresses remembered -- and nothing of it is pending.
    if (D.lookahead_psy && first_of_frame && prev0 == buffer && prev1 == buffer + 576) {
        const int o = chn ^ 1;
        const bool both = chn == 0 && D.psy_ptr[o][0] && D.psy_ptr[o][1] == D.psy_ptr[o][0] + 576 && D.psy_save_ptr[o] &&
                          !D.pa[o].valid[0] && !D.pa[o].valid[1];
        // this channel's delay line "as its first call found it": the shift above has already happened and the 576 oldest
        // samples are gone, but the look-ahead only derives the line AFTER the shift from it -- which is savebuf as it stands
        short before[1344];
        memset(before, 0, 576 * sizeof(short));
        memcpy(before + 576, savebuf, 768 * sizeof(short));
        const short *P[2] = {nullptr, nullptr}, *DL[2] = {nullptr, nullptr};
        P[chn] = buffer; DL[chn] = before;
        if (both) { P[o] = D.psy_ptr[o][0]; DL[o] = D.psy_save_ptr[o]; }
        psy_ahead_start(both ? 0 : chn, both ? 2 : 1, P, DL);
        D.psy2_n_ch = (both || chn == 0) ? (both ? 2 : 1) : 0; // D.psy2 holds the frame's records [granule][channel] from channel 0 on
        D.psy2_served = 1;
        A.valid[0] = false; // this call is the channel's first: served now
        A.served = 1;
        psy_hand_out(A.out[0], ratio_d, ratio_ds, pe, cod_info, 0, chn);
        return;
    }
    D.psy2_n_ch = 0;
    psy_one_granule(savebuf, chn, D.psy1.d);
    psy_wait();
    psy_hand_out(*D.psy1.h, ratio_d, ratio_ds, pe, cod_info, first_of_frame ? 0 : 1, chn);
}

// the ring of channel k is brought to where the slots handed out so far left it; the look-ahead of the channel ends
static void window_ahead_close(int k)
{
    DropIn::win_ahead &W = D.wa[k];
    if (!W.valid) return;
    if (W.next > 0) {
        // what W.next calls of window_subband would have left in the ring (src/encode.c:306-312; only the last sixteen
        // slots' samples survive): written by the host -- nothing is in flight on the stream: every launch is waited for --
        // where round 4's first version launched a kernel for it (14 us, twice a frame)
        double *ring = D.ring.h + 512 * k;
        const int16_t *smp = D.wa_smp.h + 1152 * k;
        for (int q = W.next > 16 ? W.next - 16 : 0; q < W.next; q++)
            for (int j = 0; j < 32; j++) ring[(31 - j + D.off[k] - 32 * q) & 511] = (double) smp[32 * q + j] * (1.0 / 32768.0);
        D.off[k] = (D.off[k] - 32 * W.next) & 511; // 480 = -32 mod 512 per slot (src/encode.c:313-314)
    }
    W.valid = false;
}

extern "C" void window_subband(short **buffer, double z[512], int k)
{
    // The reference's Layer I / II frame loops (src/musicin.c:620-704) call the filterbank without ever calling
    // L3psycho_anal: the analysis window and the matrixing coefficients do not depend on the sampling frequency, so the
    // hidden stream is set up with any rate's tables (a later L3psycho_anal at another rate -- another layer in the same
    // process -- is refused)
    if (!D.ready) ensure(0);
    if (k < 0 || k > 1) DIE("window_subband: channel %d", k);
    DropIn::win_ahead &W = D.wa[k];
    if (W.valid) { // is this the call that was foreseen: the pointer where the last one left it, the samples unchanged?
        if (*buffer == W.p0 + 32 * W.next && memcmp(*buffer, D.wa_smp.h + 1152 * k + 32 * W.next, 32 * sizeof(int16_t)) == 0) {
            const double *zs = D.wa_zs.h + ((size_t) k * 36 + W.next) * 544;
            *buffer += 32; // src/encode.c:307
            memcpy(z, zs, 512 * sizeof(double));
            memcpy(D.last_z, zs, 512 * sizeof(double));
            memcpy(D.last_s, zs + 512, 32 * sizeof(double));
            D.have_s = true;
            if (++W.next == 36) window_ahead_close(k);
            return;
        }
        window_ahead_close(k); // no: from here on call by call
    }
    // the frame's first slot of this channel, at the very buffer both of the channel's L3psycho_anal calls were given?
    if (D.lookahead && D.psy_seen[k] == 3 && *buffer == D.psy_ptr[k][0] && D.psy_ptr[k][1] == D.psy_ptr[k][0] + 576) {
        // ... then [*buffer, *buffer + 1152) has been handed over, and so has the other channel's if its calls were seen:
        // both channels' 36 slots in one launch
        const int other = k ^ 1;
        const bool both = D.psy_seen[other] == 3 && D.psy_ptr[other][1] == D.psy_ptr[other][0] + 576 && !D.wa[other].valid;
        const int first = both ? 0 : k, n_ch = both ? 2 : 1;
        for (int c = first; c < first + n_ch; c++) {
            memcpy(D.wa_smp.h + 1152 * c, D.psy_ptr[c][0], 1152 * sizeof(int16_t));
            D.wa[c].valid = true;
            D.wa[c].p0 = D.psy_ptr[c][0];
            D.wa[c].next = 0;
            D.psy_seen[c] = 0; // (the next frame's calls have to be seen again)
        }
        // the frame's mdct_sub behind the same wait, when all its inputs are in hand: both granules' block types of the
        // channels covered (every channel of the frame: first == 0) and the previous frame's granule in D.sb
        bool spec = first == 0;
        for (int c = 0; c < n_ch; c++) spec = spec && (D.bt_known & (1 << c)) && (D.bt_known & (1 << (2 + c)));
        io_buf<double> &SI = D.sbuf[D.sb_cur], &SO = D.sbuf[D.sb_cur ^ 1];
        if (spec) {
            for (int c = 0; c < n_ch; c++) {
                memcpy(D.spec.prev[c], SI.h + (size_t) c * 3 * 576, sizeof(D.spec.prev[c]));
                D.bt.h[c] = D.bt_pred[0][c];
                D.bt.h[2 + c] = D.bt_pred[1][c];
            }
        }
        mp3mi_launch_window_filter_frame(D.T, D.ring.d + 512 * first, D.off[first], D.off[first + n_ch - 1], D.wa_smp.d + 1152 * first, n_ch, 36,
                                         D.wa_zs.d + (size_t) first * 36 * 544, spec ? SI.d : NULL, D.st);
        // (the transform tells the host itself when it is through, and empties the list of the kernels that may follow it)
        unsigned seq_fb = 0;
        if (spec) {
            seq_fb = ++D.done_seq;
            mp3mi_launch_mdct_sub(D.T, SI.d, SO.d, D.bt.d, D.xr.d, D.xr_dev, n_ch, 2, &D.prep_fix->count, D.done_flag.d, seq_fb, D.mdct_count, D.st);
        } else
            seq_fb = dropin_mark();
        D.spec.valid = spec;
        D.spec.n_ch = n_ch;
        D.bt_known = 0;
        // ... and the frame's iteration_loop and III_format_bitstream behind that, when L3psycho_anal's look-ahead holds all of
        // the frame's records (every one handed out as foreseen) and a frame has been served before: its frame length, header
        // bits and channel count are taken for this one's (DropIn::frame_ahead).  They run while the host hands out the 72 slots.
        DropIn::frame_ahead &A = D.fa;
        if (spec && D.lookahead_loop && A.have_last && !A.loop_pending && !A.fmt_pending && n_ch == A.C && D.psy2_n_ch == n_ch &&
            D.psy2_served == 2 * n_ch && D.frames_done > 0 && format_setup(A.bitsPerFrame / 8, (32 + 16 * A.crc + (A.C == 2 ? 256 : 136)) / 8))
            frame_chain_launch(D.psy2.d, D.psy2.h, true, true);
        D.psy2_n_ch = 0; // (used, or not usable)
        dropin_wait_for(seq_fb);
        window_subband(buffer, z, k); // handed out from what was just computed
        return;
    }
    mp3mi_dropin_samples in;
    memcpy(in.v, *buffer, sizeof(in.v));
    *buffer += 32; // src/encode.c:307
    mp3mi_launch_window_filter(D.T, D.ring.d + 512 * k, D.off[k], in, D.zs_d, D.st);
    dropin_wait();
    memcpy(z, D.zs_h, 512 * sizeof(double));
    memcpy(D.last_z, D.zs_h, 512 * sizeof(double));
    memcpy(D.last_s, D.zs_h + 512, 32 * sizeof(double));
    D.have_s = true;
    D.off[k] = (D.off[k] + 480) & 511; // src/encode.c:313-314
}

// (statistics for tests and tools: waits for the device since the process started)
extern "C" long mp3mi_dropin_waits(void) { return D.n_launch_waits; }

extern "C" void filter_subband(double z[512], double s[32])
{
    if (!D.ready) ensure(0);
    if (D.have_s && memcmp(z, D.last_z, sizeof(D.last_z)) == 0) { // the z that window_subband just returned: its s is here already
        memcpy(s, D.last_s, sizeof(D.last_s));
        return;
    }
    HIPOK(hipMemcpyAsync(D.z_d, z, 512 * sizeof(double), hipMemcpyHostToDevice, D.st));
    mp3mi_launch_filter_subband(D.T, D.z_d, D.s_d, D.st);
    HIPOK(hipMemcpyAsync(s, D.s_d, 32 * sizeof(double), hipMemcpyDeviceToHost, D.st));
    HIPOK(hipStreamSynchronize(D.st));
}

extern "C" void mdct_sub(L3SBS *sb_sample, double (*mdct_freq)[2][576], int stereo, III_side_info_t *l3_side, int mode_gr)
{
    if (!D.ready) DIE("mdct_sub called before L3psycho_anal fixed the sampling frequency");
    if (mode_gr != 2) DIE("mdct_sub: MPEG-2 LSF (mode_gr = %d) is outside this library's scope", mode_gr);
    int32_t bt[4] = {0, 0, 0, 0};
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < stereo; ch++) bt[gr * 2 + ch] = (int32_t) l3_side->gr[gr].ch[ch].tt.block_type;
    if (D.spec.valid) { // launched behind the frame's filterbank look-ahead: is this the call it was computed for?
        D.spec.valid = false;
        bool same = stereo == D.spec.n_ch;
        const double *caller = (const double *) sb_sample;
        for (int ch = 0; same && ch < stereo; ch++) {
            same = bt[ch] == D.bt.h[ch] && bt[2 + ch] == D.bt.h[2 + ch] && memcmp(caller + (size_t) ch * 3 * 576, D.spec.prev[ch], sizeof(D.spec.prev[ch])) == 0;
            for (int i = 0; same && i < 36; i++) // slot i of the frame: what filter_subband handed out for it
                same = memcmp(caller + (size_t) ch * 3 * 576 + 576 + (size_t) i * 32, D.wa_zs.h + ((size_t) ch * 36 + i) * 544 + 512, 32 * sizeof(double)) == 0;
        }
        if (same) {
            const double *res = D.sbuf[D.sb_cur ^ 1].h;
            for (int ch = 0; ch < stereo; ch++) {
                memcpy((double *) sb_sample + (size_t) ch * 3 * 576, res + (size_t) ch * 3 * 576, 3 * 576 * sizeof(double));
                for (int gr = 0; gr < 2; gr++) memcpy(mdct_freq[gr][ch], D.xr.h + ((size_t) gr * stereo + ch) * 576, 576 * sizeof(double));
            }
            D.sb_cur ^= 1;
            return;
        }
    }
    frame_chain_cancel(); // (a loop launched ahead read the spectrum this call is about to replace)
    memcpy(D.sbuf[D.sb_cur].h, sb_sample, sizeof(L3SBS));
    memcpy(D.bt.h, bt, sizeof(bt));
    mp3mi_launch_mdct_sub(D.T, D.sbuf[D.sb_cur].d, D.sbuf[D.sb_cur ^ 1].d, D.bt.d, D.xr.d, NULL, stereo, mode_gr, NULL, NULL, 0, D.mdct_count, D.st);
    dropin_wait();
    D.sb_cur ^= 1;
    memcpy(sb_sample, D.sbuf[D.sb_cur].h, sizeof(L3SBS));
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < stereo; ch++) memcpy(mdct_freq[gr][ch], D.xr.h + ((size_t) gr * stereo + ch) * 576, 576 * sizeof(double));
}

// ---- k_loop and k_format of a frame, back to back on the stream (DropIn::frame_ahead) ----

// the formatter's byte window of this run of frames (a run ends with III_FlushBitstream); false: the frame length changed
static bool format_setup(int frame_bytes, int si_bytes)
{
    if (D.frames_done == 0 && !D.win_for_run) {
        D.frame_bytes = frame_bytes;
        D.si_bytes = si_bytes;
        D.win_bytes = (size_t) (WIN_FRAMES + 1) * frame_bytes;
        if (D.win.h) HIPOK(hipHostFree(D.win.h));
        D.win.alloc(D.win_bytes);
        D.abs_emitted = 0;
        D.m_end = 0;
        D.win_for_run = true;
        D.win_slid_for = -1;
        return true;
    }
    return frame_bytes == D.frame_bytes && si_bytes == D.si_bytes;
}

// slide the byte window so that frame n sits at index min(n, WIN_FRAMES) (once per frame; the window is host memory that the
// device writes: moved between two launches, with nothing in flight)
static void format_position_window(long n)
{
    if (D.win_slid_for == n) return;
    if (n > WIN_FRAMES) {
        memmove(D.win.h, D.win.h + D.frame_bytes, D.win_bytes - (size_t) D.frame_bytes);
        memset(D.win.h + D.win_bytes - (size_t) D.frame_bytes, 0, (size_t) D.frame_bytes);
    }
    D.win_slid_for = n;
}

static void format_launch(int C, int crc, int bitsPerFrame, int bitrate_index, int mode, int hdr_flags, bool marked = false, unsigned seq_before = 0, unsigned seq_done = 0)
{
    const long n = D.frames_done;
    const int widx = (int) (n < WIN_FRAMES ? n : WIN_FRAMES);
    *D.bits.h = bitsPerFrame;
    *D.bri.h = bitrate_index;
    mp3mi_geom g = mp3mi_make_geom(1, C, D.rate_idx, 1 << 30, widx, 1);
    g.hdr_mode = mode;
    g.crc = crc;
    g.hdr_flags = hdr_flags;
    if (marked) mp3mi_launch_format_marked(D.T, g, D.ix_dev, D.side_dev, D.bits.d, D.bri.d, D.win.d, D.win_bytes, D.len_d, D.done_flag.d, seq_before, seq_done, D.ix.d, D.side.d, D.st);
    else mp3mi_launch_format(D.T, g, D.ix.d, D.side.d, D.bits.d, D.bri.d, D.win.d, D.win_bytes, D.len_d, NULL, 0, NULL, D.st);
}

// Launches the frame's k_loop (records rec_d, spectrum D.xr) and, when with_format, its k_format behind it, with the parameters in
// D.fa; copies of the loop's state and of the byte window are taken first.  Nothing may be in flight that writes either.
static void frame_chain_launch(const mp3mi_psy_out *rec_d, const mp3mi_psy_out *rec_h, bool with_format, bool behind_mdct)
{
    DropIn::frame_ahead &A = D.fa;
    memcpy(&A.state_before, D.loop_state.h, sizeof(A.state_before));
    A.rec_h = rec_h;
    *D.bits.h = A.bitsPerFrame;
    mp3mi_geom g = mp3mi_make_geom(1, A.C, D.rate_idx, 1, 0, 1);
    g.crc = A.crc;
    // behind_mdct: k_mdct_sub, launched just before, has emptied the list and left a copy of the spectrum in device memory
    const double *xr = behind_mdct ? D.xr_dev : D.xr.d;
    // with the formatter behind it k_loop writes to device memory; k_format_marked writes the host's copies
    int16_t *ix = with_format ? D.ix_dev : D.ix.d;
    mp3mi_frame_side *side = with_format ? D.side_dev : D.side.d;
    // the loop's stateless head of the spectrum: k_mdct's tail as a kernel of its own, then the reference's walk for
    // the records it lists as undecided (k_prep.hip; none, practically)
    if (!behind_mdct) HIPOK(hipMemsetAsync(&D.prep_fix->count, 0, sizeof(unsigned), D.st));
    mp3mi_launch_prep_tail(D.T, g, xr, rec_d, D.prep4, D.prep_fix, D.st);
    mp3mi_launch_prep(D.T, g, xr, rec_d, D.prep4, D.prep_fix, 0, D.st);
    mp3mi_launch_loop(D.T, g, xr, rec_d, D.prep4, D.bits.d, D.loop_state.d, ix, side, NULL, mp3mi_loop_place{NULL, NULL, NULL, NULL, NULL, NULL, NULL, 0}, D.st);
    A.loop_pending = true;
    A.fmt_pending = false;
    if (!with_format) A.seq_loop = dropin_mark();
    if (with_format) {
        format_position_window(D.frames_done);
        if (A.win_before_bytes < D.win_bytes) {
            free(A.win_before);
            A.win_before = (uint8_t *) malloc(D.win_bytes);
            if (!A.win_before) DIE("out of memory");
            A.win_before_bytes = D.win_bytes;
        }
        memcpy(A.win_before, D.win.h, D.win_bytes);
        // (the formatter tells the host itself: as it starts, that k_loop is through; as it ends, that the bytes are in place)
        A.seq_loop = ++D.done_seq;
        A.seq_fmt = ++D.done_seq;
        format_launch(A.C, A.crc, A.bitsPerFrame, A.bitrate_index, A.mode, A.hdr_flags, true, A.seq_loop, A.seq_fmt);
        A.fmt_pending = true;
    }
}

// what was launched ahead and not handed out is undone: the loop's state and the byte window as they were before the launch
static void frame_chain_cancel()
{
    DropIn::frame_ahead &A = D.fa;
    if (!A.loop_pending && !A.fmt_pending) return;
    dropin_wait_for(A.fmt_pending ? A.seq_fmt : A.seq_loop);
    if (A.loop_pending) memcpy(D.loop_state.h, &A.state_before, sizeof(A.state_before));
    if (A.fmt_pending) memcpy(D.win.h, A.win_before, D.win_bytes);
    A.loop_pending = A.fmt_pending = false;
}

static int header_flags_of(const layer *info) { return ((info->mode_ext & 3) << 4) | ((info->copyright & 1) << 3) | ((info->original & 1) << 2) | (info->emphasis & 3); }

extern "C" void iteration_loop(double pe[][2], double xr_org[2][2][576], III_psy_ratio *ratio, III_side_info_t *l3_side,
                               int l3_enc[2][2][576], int mean_bits, int stereo, double xr_dec[2][2][576],
                               III_scalefac_t *scalefac, frame_params *fr_ps, int ancillary_pad, int bitsPerFrame)
{
    static unsigned no_partition_table[4] = {0, 0, 0, 0};
    (void) xr_dec;
    (void) ancillary_pad;
    layer *info = fr_ps->header;
    if (info->version != 1) DIE("iteration_loop: MPEG-2 LSF is outside this library's scope");
    ensure(info->sampling_frequency);
    const int C = stereo;
    const int crc = info->error_protection ? 1 : 0; // 16 more bits of side information (src/musicin.c:744-746)
    if (mean_bits != (bitsPerFrame - (32 + 16 * crc + (C == 1 ? 136 : 256))) / 2) DIE("iteration_loop: unexpected mean_bits %d", mean_bits);
    DropIn::frame_ahead &A = D.fa;
    const bool first_call = D.loop_first;
    if (D.loop_first) { // src/loop.c:250-257
        l3_side->main_data_begin = 0;
        D.loop_first = false;
    }
    const int frame_bytes = bitsPerFrame / 8, si_bytes = (32 + 16 * crc + (C == 2 ? 256 : 136)) / 8;
    bool served = false;
    if (A.loop_pending) { // launched behind the frame's filterbank look-ahead: is this the call it was computed for?
        bool same = !first_call && C == A.C && crc == A.crc && bitsPerFrame == A.bitsPerFrame && (int) info->bitrate_index == A.bitrate_index &&
                    (int) info->mode == A.mode && header_flags_of(info) == A.hdr_flags;
        for (int gr = 0; same && gr < 2; gr++)
            for (int ch = 0; same && ch < C; ch++) {
                const mp3mi_psy_out &r = A.rec_h[gr * C + ch];
                same = memcmp(&r.pe, &pe[gr][ch], sizeof(double)) == 0 && memcmp(r.ratio_l, ratio->l[gr][ch], sizeof(r.ratio_l)) == 0 &&
                       memcmp(r.ratio_s, ratio->s[gr][ch], sizeof(r.ratio_s)) == 0 && r.block_type == (int32_t) l3_side->gr[gr].ch[ch].tt.block_type &&
                       memcmp(D.xr.h + ((size_t) gr * C + ch) * 576, xr_org[gr][ch], 576 * sizeof(double)) == 0;
            }
        if (same) {
            dropin_wait_for(A.seq_loop);
            A.loop_pending = false;
            served = true;
            D.n_loop_ahead++;
        } else
            frame_chain_cancel();
    } else
        frame_chain_cancel(); // (a formatter launched ahead whose call never came)
    if (!served) {
        // records in the batch layout [gr][ch] for one stream, one frame
        mp3mi_psy_out *rec = D.psy4.h;
        double (*xr)[576] = (double (*)[576]) D.xr.h;
        memset(rec, 0, 4 * sizeof(mp3mi_psy_out));
        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < C; ch++) {
                mp3mi_psy_out *r = &rec[gr * C + ch];
                r->pe = pe[gr][ch];
                memcpy(r->ratio_l, ratio->l[gr][ch], sizeof(r->ratio_l));
                memcpy(r->ratio_s, ratio->s[gr][ch], sizeof(r->ratio_s));
                r->block_type = (int32_t) l3_side->gr[gr].ch[ch].tt.block_type;
                memcpy(xr[gr * C + ch], xr_org[gr][ch], sizeof(xr[0]));
            }
        A.C = C; A.crc = crc; A.bitsPerFrame = bitsPerFrame;
        A.bitrate_index = (int) info->bitrate_index; A.mode = (int) info->mode; A.hdr_flags = header_flags_of(info);
        // the formatter behind the loop, from the records the loop leaves on the device (a frame length other than the run's:
        // III_format_bitstream will refuse it, in its own words)
        frame_chain_launch(D.psy4.d, D.psy4.h, D.lookahead_loop && format_setup(frame_bytes, si_bytes), false);
        dropin_wait_for(A.seq_loop);
        A.loop_pending = false;
    }
    A.have_last = true;
    const int16_t (*ix)[576] = (const int16_t (*)[576]) D.ix.h;
    const mp3mi_frame_side &sd = *D.side.h;
    const loop_state_host &ls = *D.loop_state.h;
    // inputs the reference dies on: so does this call, with the reference's own words (its assert() prints the
    // expression and abort()s)
    if ((ls.ref_abort & 255) == MP3MI_DEV_ABORT_GLOBAL_GAIN) DIE("iteration_loop: Assertion `cod_info->global_gain < 256' failed (src/loop.c:358)");
    if ((ls.ref_abort & 255) == MP3MI_DEV_ABORT_HUFF_BITS) DIE("inner_loop: Assertion `max_bits >= 0' failed (src/loop.c:579)");
    l3_side->resvDrain = sd.resvDrain;
    for (int ch = 0; ch < C; ch++)
        for (int b = 0; b < 4; b++) l3_side->scfsi[ch][b] = (unsigned) sd.scfsi[ch][b];
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < C; ch++) {
            const mp3mi_gr_side *s = &sd.gr[gr][ch];
            gr_info *g2 = &l3_side->gr[gr].ch[ch].tt;
            const bool shortb = s->window_switching_flag && s->block_type == 2;
            for (int i = 0; i < 576; i++) { // still non-negative here; signs are applied by III_format_bitstream
                const int v = ix[gr * C + ch][i];
                l3_enc[gr][ch][i] = v < 0 ? -v : v;
            }
            g2->part2_3_length = (unsigned) s->part2_3_length;
            g2->big_values = (unsigned) s->big_values;
            g2->count1 = (unsigned) s->count1;
            g2->global_gain = (unsigned) s->global_gain;
            g2->scalefac_compress = (unsigned) s->scalefac_compress;
            for (int k = 0; k < 3; k++) { g2->table_select[k] = (unsigned) s->table_select[k]; g2->subblock_gain[k] = 0; }
            g2->region0_count = (unsigned) s->region0_count;
            g2->region1_count = (unsigned) s->region1_count;
            g2->preflag = (unsigned) s->preflag;
            g2->scalefac_scale = 0;
            g2->count1table_select = (unsigned) s->count1table_select;
            g2->part2_length = (unsigned) s->part2_length;
            g2->sfb_lmax = shortb ? 0 : 21; // gr_deco, src/loop.c:2063-2081
            g2->sfb_smax = shortb ? 0 : 12;
            g2->address1 = (unsigned) ls.addr[gr][ch][0];
            g2->address2 = (unsigned) ls.addr[gr][ch][1];
            g2->address3 = (unsigned) ls.addr[gr][ch][2];
            g2->quantizerStepSize = (double) (s->global_gain - 210); // global_gain = nint(q + 210), src/loop.c:357
            g2->sfb_partition_table = no_partition_table;
            for (int k = 0; k < 4; k++) g2->slen[k] = 0;
            for (int i = 0; i < 21; i++) scalefac->l[gr][ch][i] = shortb ? 0 : s->scalefac[i];
            for (int i = 0; i < 13; i++)
                for (int w = 0; w < 3; w++) scalefac->s[gr][ch][i][w] = (shortb && i < 12) ? s->scalefac[i * 3 + w] : 0;
        }
}

extern "C" void III_format_bitstream(int bitsPerFrame, frame_params *fr_ps, int l3_enc[2][2][576], III_side_info_t *l3_side,
                                     III_scalefac_t *scalefac, Bit_stream_struc *bs, double (*xr)[2][576], char *ancillary,
                                     int anc_bits)
{
    (void) ancillary;
    layer *info = fr_ps->header;
    if (anc_bits != 0) DIE("III_format_bitstream: ancillary data is not supported");
    const int crc = info->error_protection ? 1 : 0; // the reference's zero CRC word (src/l3bitstream.c:312, 338-342)
    ensure(info->sampling_frequency);
    const int C = fr_ps->stereo;
    if (!D.putbits) {
        D.putbits = (void (*)(Bit_stream_struc *, unsigned int, int)) dlsym(RTLD_DEFAULT, "putbits");
        if (!D.putbits) DIE("host program does not export putbits() (link it with -rdynamic)");
    }
    D.bs = bs;
    const int frame_bytes = bitsPerFrame / 8, si_bytes = (32 + 16 * crc + (C == 2 ? 256 : 136)) / 8, slot = frame_bytes - si_bytes;
    if (!format_setup(frame_bytes, si_bytes)) DIE("III_format_bitstream: frame length changed (padding is never used by the reference driver)");
    DropIn::frame_ahead &A = D.fa;
    int16_t (*ix)[576] = (int16_t (*)[576]) D.ix.h;
    mp3mi_frame_side &sd = *D.side.h;
    const long n = D.frames_done;
    const int widx = (int) (n < WIN_FRAMES ? n : WIN_FRAMES);
    const long base = (n - widx) * (long) frame_bytes;
    bool served = false;
    if (A.loop_pending) frame_chain_cancel(); // (an iteration_loop launched ahead whose call never came)
    if (A.fmt_pending) {
        // launched behind the frame's k_loop, from the records that kernel left in D.ix / D.side: is everything this call is given
        // what iteration_loop handed out -- the values (their signs come from xr: src/l3bitstream.c:115-125), the side
        // information, the scalefactors, the header?
        bool same = C == A.C && crc == A.crc && bitsPerFrame == A.bitsPerFrame && (int) info->bitrate_index == A.bitrate_index &&
                    (int) info->mode == A.mode && header_flags_of(info) == A.hdr_flags && (int) l3_side->main_data_begin == sd.main_data_begin &&
                    (int) l3_side->resvDrain == sd.resvDrain;
        for (int ch = 0; same && ch < C; ch++)
            for (int b = 0; same && b < 4; b++) same = (int32_t) l3_side->scfsi[ch][b] == sd.scfsi[ch][b];
        for (int gr = 0; same && gr < 2; gr++)
            for (int ch = 0; same && ch < C; ch++) {
                const gr_info *g2 = &l3_side->gr[gr].ch[ch].tt;
                const mp3mi_gr_side *s = &sd.gr[gr][ch];
                const bool shortb = g2->window_switching_flag && g2->block_type == 2;
                same = s->part2_3_length == (int32_t) g2->part2_3_length && s->big_values == (int32_t) g2->big_values && s->count1 == (int32_t) g2->count1 &&
                       s->global_gain == (int32_t) g2->global_gain && s->scalefac_compress == (int32_t) g2->scalefac_compress &&
                       s->window_switching_flag == (int32_t) g2->window_switching_flag && s->block_type == (int32_t) g2->block_type &&
                       s->region0_count == (int32_t) g2->region0_count && s->region1_count == (int32_t) g2->region1_count &&
                       s->preflag == (int32_t) g2->preflag && s->count1table_select == (int32_t) g2->count1table_select && s->part2_length == (int32_t) g2->part2_length;
                for (int k = 0; same && k < 3; k++) same = s->table_select[k] == (int32_t) g2->table_select[k];
                if (shortb) for (int i = 0; same && i < 36; i++) same = s->scalefac[i] == scalefac->s[gr][ch][i / 3][i % 3];
                else for (int i = 0; same && i < 21; i++) same = s->scalefac[i] == scalefac->l[gr][ch][i];
                const int16_t *q = ix[gr * C + ch];
                for (int i = 0; same && i < 576; i++) {
                    int v = l3_enc[gr][ch][i];
                    if (xr[gr][ch][i] < 0 && v > 0) v = -v;
                    same = v == (int) q[i];
                }
            }
        if (same) {
            dropin_wait_for(A.seq_fmt);
            A.fmt_pending = false;
            served = true;
            D.n_fmt_ahead++;
        } else
            frame_chain_cancel();
    }
    // signs of the spectrum go onto the quantised values, in place (src/l3bitstream.c:115-125)
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < C; ch++)
            for (int i = 0; i < 576; i++)
                if (xr[gr][ch][i] < 0 && l3_enc[gr][ch][i] > 0) l3_enc[gr][ch][i] *= -1;
    if (!served) {
        memset(&sd, 0, sizeof(sd));
        for (int gr = 0; gr < 2; gr++)
            for (int ch = 0; ch < C; ch++) {
                for (int i = 0; i < 576; i++) ix[gr * C + ch][i] = (int16_t) l3_enc[gr][ch][i];
                const gr_info *g2 = &l3_side->gr[gr].ch[ch].tt;
                mp3mi_gr_side *s = &sd.gr[gr][ch];
                const bool shortb = g2->window_switching_flag && g2->block_type == 2;
                s->part2_3_length = (int32_t) g2->part2_3_length; s->big_values = (int32_t) g2->big_values;
                s->count1 = (int32_t) g2->count1; s->global_gain = (int32_t) g2->global_gain;
                s->scalefac_compress = (int32_t) g2->scalefac_compress;
                s->window_switching_flag = (int32_t) g2->window_switching_flag; s->block_type = (int32_t) g2->block_type;
                for (int k = 0; k < 3; k++) s->table_select[k] = (int32_t) g2->table_select[k];
                s->region0_count = (int32_t) g2->region0_count; s->region1_count = (int32_t) g2->region1_count;
                s->preflag = (int32_t) g2->preflag; s->count1table_select = (int32_t) g2->count1table_select;
                s->part2_length = (int32_t) g2->part2_length;
                if (shortb) for (int i = 0; i < 36; i++) s->scalefac[i] = scalefac->s[gr][ch][i / 3][i % 3];
                else for (int i = 0; i < 21; i++) s->scalefac[i] = scalefac->l[gr][ch][i];
            }
        sd.main_data_begin = l3_side->main_data_begin;
        sd.resvDrain = l3_side->resvDrain;
        for (int ch = 0; ch < C; ch++)
            for (int b = 0; b < 4; b++) sd.scfsi[ch][b] = (int32_t) l3_side->scfsi[ch][b];
        format_position_window(n);
        format_launch(C, crc, bitsPerFrame, (int) info->bitrate_index, (int) info->mode, header_flags_of(info));
        dropin_wait();
    }
    // bytes that are final now = everything up to the end of this frame's main data
    long bits = sd.resvDrain;
    for (int gr = 0; gr < 2; gr++)
        for (int ch = 0; ch < C; ch++) bits += sd.gr[gr][ch].part2_3_length;
    const long m0 = n * (long) slot - l3_side->main_data_begin;
    if (m0 != D.m_end) DIE("III_format_bitstream: back pointer %d does not match the reservoir", l3_side->main_data_begin);
    D.m_end = m0 + bits / 8;
    emit(emitted_upto(D.m_end, slot, frame_bytes, si_bytes), base);
    D.frames_done = n + 1;
    D.frames_total++;
    // nextBackPtr (src/formatBitstream.c:78-79)
    l3_side->main_data_begin = (int) (D.frames_done * (long) slot - D.m_end);
}

extern "C" void III_FlushBitstream(void)
{
    if (!D.ready) return;
    frame_chain_cancel(); // (launched ahead for calls that never came)
    if (D.frames_done == 0) return;
    if (D.stats && D.t_first != 0.0) {
        const double dt = now_s() - D.t_first;
        fprintf(stderr, "mp3mi drop-in: %ld frames in %.4f s from the first frame's first call = %.1f frames/s, %ld waits for the device, %ld / %ld frames' loops / formatters launched ahead of their calls\n", D.frames_total, dt,
                (double) D.frames_total / dt, D.n_launch_waits, D.n_loop_ahead, D.n_fmt_ahead);
    }
    const int slot = D.frame_bytes - D.si_bytes;
    const long rem = ((D.m_end + slot - 1) / slot) * slot - D.m_end;
    {   // BF_FlushBitstream's remainder call asks for a header the queue no longer has (k_format.hip, fmt_flush_dies)
        const long written = (D.m_end + slot - 1) / slot, queued = D.frames_done - written;
        if (queued >= 1 && written * slot == D.m_end && ((queued * (long) slot * 8) % 32) == 0)
            DIE("get_side_info: Assertion `l' failed (src/formatBitstream.c:390, from BF_FlushBitstream)");
    }
    const long total = D.frames_done * (long) D.frame_bytes - rem;
    const long widx = D.frames_done - 1 < WIN_FRAMES ? D.frames_done - 1 : WIN_FRAMES;
    const long base = (D.frames_done - 1 - widx) * (long) D.frame_bytes;
    emit(total, base);
    // src/formatBitstream.c:112-119: the formatter starts over
    D.frames_done = 0;
    D.abs_emitted = 0;
    D.m_end = 0;
    D.win_for_run = false;
}

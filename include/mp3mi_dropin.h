/* libmp3mi -- the reference encoder's per-frame Layer III call surface, C ABI.
 *
 * The reference (lieff/mp3-enc-bsd) has no plugin / FFI layer: its Layer III "interface" is the
 * seven external-linkage functions its driver calls once per frame
 * (/root/reference/src/musicin.c:751-786, 803).  libmp3mi.so exports exactly those names with
 * the same signatures, argument meaning, in-place side effects and error behaviour (message +
 * exit/abort), backed by one hidden default stream that runs the same gfx950 kernels as the
 * batched API with n_streams = 1.  Linking the reference's unchanged musicin.o + common.o against
 * libmp3mi.so instead of l3psy.o / mdct.o / loop.o / l3bitstream.o / formatBitstream.o /
 * reservoir.o / subs.o / pow_nint.o / huffman.o (and the two filterbank functions of encode.o)
 * yields a byte-identical MP3 (tests/test_dropin.py; recipe in INTEGRATION.md).
 *
 * The type definitions below restate the reference's layouts (file:line cited); a translation
 * unit that already includes the reference's own headers defines
 * MP3MI_DROPIN_USE_REFERENCE_HEADERS first and only gets the prototypes' documentation.
 *
 * One stream per process, not thread-safe -- exactly like the reference.  Throughput comes from
 * mp3mi_batch_* (include/mp3mi.h); these symbols exist so the library drops in under musicin.c.
 */
#ifndef MP3MI_DROPIN_H
#define MP3MI_DROPIN_H

#ifdef __cplusplus
extern "C" {
#endif

#ifndef MP3MI_DROPIN_USE_REFERENCE_HEADERS

/* src/l3side.h:60-87 */
typedef struct {
    unsigned part2_3_length;
    unsigned big_values;
    unsigned count1;
    unsigned global_gain;
    unsigned scalefac_compress;
    unsigned window_switching_flag;
    unsigned block_type;
    unsigned mixed_block_flag;
    unsigned table_select[3];
    int subblock_gain[3];
    unsigned region0_count;
    unsigned region1_count;
    unsigned preflag;
    unsigned scalefac_scale;
    unsigned count1table_select;
    unsigned part2_length;
    unsigned sfb_lmax;
    unsigned sfb_smax;
    unsigned address1;
    unsigned address2;
    unsigned address3;
    double quantizerStepSize;
    unsigned *sfb_partition_table;
    unsigned slen[4];
} gr_info;

/* src/l3side.h:89-99 */
typedef struct {
    int main_data_begin;
    unsigned private_bits;
    int resvDrain;
    unsigned scfsi[2][4];
    struct {
        struct gr_info_s {
            gr_info tt;
        } ch[2];
    } gr[2];
} III_side_info_t;

/* src/l3side.h:41-44 */
typedef struct {
    double l[2][2][21];
    double s[2][2][12][3];
} III_psy_ratio;

/* src/l3side.h:103-106 */
typedef struct {
    int l[2][2][22];
    int s[2][2][13][3];
} III_scalefac_t;

/* src/common.h:285-298 */
typedef struct {
    int version;
    int lay;
    int error_protection;
    int bitrate_index;
    int sampling_frequency;
    int padding;
    int extension;
    int mode;
    int mode_ext;
    int copyright;
    int original;
    int emphasis;
} layer;

/* src/common.h:302-310 (alloc is a pointer to the Layer II allocation table; unused by Layer III) */
typedef struct {
    layer *header;
    int actual_mode;
    void *alloc;
    int tab_num;
    int stereo;
    int jsbound;
    int sblimit;
} frame_params;

/* src/mdct.h:20 */
typedef double L3SBS[2][3][18][32];

/* src/common.h:344-357; only ever passed through to the host program's putbits() */
typedef struct bit_stream_struc Bit_stream_struc;

#endif /* MP3MI_DROPIN_USE_REFERENCE_HEADERS */

/* replaces src/l3psy.c:53 (declared src/l3psy.h:33).  Shifts savebuf by 576 and appends the 576
 * new samples, returns the PREVIOUS call's ratios for this channel, the current perceptual
 * entropy and the (delayed) block type in cod_info.  lay must be 3; snr32 is unused. */
void L3psycho_anal(short int *buffer, short int savebuf[1344], int chn, int lay, float snr32[32],
                   double sfreq, double ratio_d[21], double ratio_ds[12][3], double *pe,
                   gr_info *cod_info);

/* replaces src/encode.c:287 (declared src/encoder.h:182).  Consumes 32 samples from *buffer
 * (advancing it), returns the 512 windowed samples of channel k in z. */
void window_subband(short **buffer, double z[512], int k);

/* replaces src/encode.c:361 (declared src/encoder.h:184).  512 windowed samples -> 32 subband samples. */
void filter_subband(double z[512], double s[32]);

/* replaces src/mdct.c:25 (declared src/mdct.h:22).  In place: negates odd slots of odd subbands of
 * the new granules, copies the last granule to slot 0; writes mdct_freq[gr][ch][576]. */
void mdct_sub(L3SBS *sb_sample, double (*mdct_freq)[2][576], int stereo, III_side_info_t *l3_side,
              int mode_gr);

/* replaces src/loop.c:232 (declared src/loop.h:48).  Fills l3_enc (non-negative), l3_side and
 * scalefac; xr_dec is not touched; mean_bits must equal (bitsPerFrame - side info bits) / 2. */
void iteration_loop(double pe[][2], double xr_org[2][2][576], III_psy_ratio *ratio,
                    III_side_info_t *l3_side, int l3_enc[2][2][576], int mean_bits, int stereo,
                    double xr_dec[2][2][576], III_scalefac_t *scalefac, frame_params *fr_ps,
                    int ancillary_pad, int bitsPerFrame);

/* replaces src/l3bitstream.c:67 (declared src/l3bitstream.h:21).  Applies the signs of xr to
 * l3_enc in place, emits the bytes that become final with this frame through the host program's
 * putbits(bs, value, nbits) (src/common.c:1134) and stores the next back pointer in
 * l3_side->main_data_begin.  ancillary data is not supported (the driver passes NULL, 0). */
void III_format_bitstream(int bitsPerFrame, frame_params *fr_ps, int l3_enc[2][2][576],
                          III_side_info_t *l3_side, III_scalefac_t *scalefac, Bit_stream_struc *bs,
                          double (*xr)[2][576], char *ancillary, int anc_bits);

/* replaces src/l3bitstream.c:165.  Emits the queued headers and zero main data up to the end of
 * the stream and resets the formatter. */
void III_FlushBitstream(void);

/* LOOK-AHEAD.  Served one call at a time, a frame of the reference's loop costs 79 waits for the device.  The library
 * therefore reads ahead -- only in memory the caller has already handed over, or in what it has handed out itself.
 * BY DEFAULT (MP3MI_DROPIN_LOOKAHEAD unset = 2) only in memory handed over during the CURRENT frame: the filterbank's
 * look-ahead and mdct_sub behind it.  MP3MI_DROPIN_LOOKAHEAD=1 adds L3psycho_anal's look-ahead and the launches ahead of
 * iteration_loop / III_format_bitstream that feed on it.  BUFFER LIFETIME REQUIREMENT of that mode: the sample buffers
 * and delay lines given to L3psycho_anal must stay allocated, at the same addresses, from one frame to the next (the
 * library reads &buffer[ch][576 ..] and the other channel's buffer and savebuf at the addresses it was given in the
 * frame BEFORE; changed CONTENT is detected and handled, a freed or moved buffer is a read of memory the caller no
 * longer owns).  The reference's driver satisfies it: its buffers are static arrays (src/musicin.c:475-503).
 *   - L3psycho_anal: when a channel's two calls of the frame BEFORE were given p and p + 576 and the first call of this
 *     frame is given the same p, both granules are analysed at once -- and the other channel's with them when its calls of
 *     the frame before show the same pattern (its buffer and its delay line were handed over then, at the addresses
 *     remembered): one launch for the frame's four calls;
 *   - window_subband / filter_subband: the four L3psycho_anal calls of a frame were given &buffer[ch][0] and
 *     &buffer[ch][576] (src/musicin.c:754-758); when a channel's first window_subband of the frame starts at that same
 *     &buffer[ch][0], the 36 slots of both channels are computed in one launch and handed out call by call;
 *   - mdct_sub, iteration_loop, III_format_bitstream: launched right behind that filterbank kernel -- the transform from the
 *     subband samples it produced and the block types L3psycho_anal handed out, the loop from the transform's spectrum and
 *     L3psycho_anal's records with the frame length, header bits and channel count of the frame before, the formatter from
 *     the loop's records -- and each served if the call's arguments are exactly what its launch read: L3SBS and block types;
 *     pe, ratio, spectrum, mean_bits, header; l3_enc, side information, scalefactors, the spectrum's signs, header.
 * Every served call first checks that its pointer is where the previous call left it and that its inputs are still what
 * was read; a caller that moves or rewrites its buffers, or changes what it was handed between two calls, gets the
 * call-by-call service, with the library's state (psychoacoustic state, bit reservoir, the formatter's bytes) put back to
 * where the served calls left it: tests/test_dropin.py, oracle/dropin_probe.c, oracle/dropin_probe_frame.c -- bit-exact
 * either way.  With all of it (mode 1): two launches and two waits per frame instead of 79 under the reference's driver (the kernels that end a stage
 * store a flag in host-mapped memory; the host spins on it).  MP3MI_DROPIN_LOOKAHEAD = 0 none, 1 all (see the lifetime
 * requirement above), 2 (default) / 3 the filterbank's / L3psycho_anal's only, 4 all but the third family (mp3mi_batch_options_from_env); MP3MI_DROPIN_STATS=1: a line at
 * III_FlushBitstream.  mp3mi_dropin_waits: stages a call waited for so far (tests, tools). */
long mp3mi_dropin_waits(void);

#ifdef __cplusplus
}
#endif
#endif

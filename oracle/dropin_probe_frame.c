/* TEST INFRASTRUCTURE -- not part of the product path.
 *
 * Calls all seven per-frame Layer III functions in the order of the reference's frame loop (src/musicin.c:751-786) -- and
 * NOT the way that loop behaves in between.  The drop-in library (csrc/dropin.cpp) launches mdct_sub, iteration_loop and
 * III_format_bitstream AHEAD of their calls, from what it has handed out itself, and must notice a caller whose arguments
 * are not that:
 *   - frames 1 mod 5: a subband sample is changed between filter_subband and mdct_sub;
 *   - frames 2 mod 5: three lines of the spectrum are changed between mdct_sub and iteration_loop;
 *   - frames 3 mod 5: a perceptual entropy is changed between L3psycho_anal and iteration_loop;
 *   - frames 4 mod 7: the spectrum III_format_bitstream takes its signs from is negated after iteration_loop;
 *   - frames 5 mod 7: the header's copyright bit changes between iteration_loop and III_format_bitstream;
 *   - every third frame is handed over in ANOTHER buffer.
 * Every returned value goes to the dump, the bitstream to a file.  Linked once against the unmodified reference objects
 * (oracle/Makefile: _ref/dropin_probe_frame_ref) and once against the library (_ref/dropin_probe_frame, _emu): dumps and
 * files must be equal.
 *
 * usage: dropin_probe_frame dump.bin out.mp3 [frames]
 * Only compiled where /root/reference exists (its headers give the prototypes); nothing of the reference travels as source.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "encoder.h"
#include "l3psy.h"
#include "mdct.h"
#include "loop.h"
#include "l3bitstream.h"

/* globals the reference objects expect from their driver (src/musicin.c:148-156) */
FILE *musicin;
Bit_stream_struc bs;
char *programName = "dropin_probe_frame";
int iswav = 0;
int littleData = 0;
int streaming_input = 0;
unsigned long frameNum = 0;

extern void III_FlushBitstream(void);
extern long mp3mi_dropin_waits(void) __attribute__((weak));

static unsigned lcg = 777u;
static short next_sample(int f, int n)
{
    lcg = lcg * 1664525u + 1013904223u;
    {
        const int tone = (int) (5000.0 * ((n * (f % 5 + 2)) % 48 - 24) / 24.0);
        const int noise = (int) ((lcg >> 16) & 0x7ff) - 1024;
        const int burst = (f % 4 == 1 && n > 640 && n < 700) ? (((lcg >> 8) & 1) ? 11000 : -11000) : 0;
        return (short) (tone + noise + burst);
    }
}

int main(int argc, char **argv)
{
    typedef double IN[2][HAN_SIZE];
    static short buf_a[2][1152], buf_b[2][1152];
    static short sam[2][1344];
    static IN win_que;
    static L3SBS l3_sb_sample;
    static double xr[2][2][576], xr_dec[2][2][576], pe[2][2];
    static int l3_enc[2][2][576];
    static III_psy_ratio ratio;
    static III_side_info_t l3_side;
    static III_scalefac_t scalefac;
    static frame_params fr_ps;
    static layer info;
    FLOAT snr32[32];
    short *win_buf[2];
    FILE *dump;
    const int stereo = 2, mode_gr = 2;
    int frames = 12, f, gr, ch, j, i;
    if (argc < 3) { fprintf(stderr, "usage: %s dump.bin out.mp3 [frames]\n", argv[0]); return 2; }
    if (argc > 3) frames = atoi(argv[3]);
    dump = fopen(argv[1], "wb");
    if (!dump) { perror(argv[1]); return 1; }
    memset(&info, 0, sizeof(info));
    info.version = 1; /* MPEG-1 */
    info.lay = 3;
    info.error_protection = 0;
    info.bitrate_index = 9; /* 128 kbps */
    info.sampling_frequency = 0; /* 44.1 kHz */
    info.padding = 0;
    info.mode = MPG_MD_STEREO;
    info.mode_ext = 0;
    info.copyright = 0;
    info.original = 0;
    info.emphasis = 0;
    fr_ps.header = &info;
    fr_ps.tab_num = -1;
    fr_ps.alloc = NULL;
    hdr_to_frps(&fr_ps);
    open_bit_stream_w(&bs, argv[2], BUFFER_SIZE);
    memset(sam, 0, sizeof(sam));
    for (f = 0; f < frames; f++) {
        short (*buf)[1152] = (f % 3 == 2) ? buf_b : buf_a;
        const int whole_SpF = 417; /* (int) (1152 / 44.1 * 128 / 8): src/musicin.c:561-567 */
        const int bitsPerFrame = 8 * whole_SpF, mean_bits = (bitsPerFrame - (32 + 256)) / mode_gr;
        frameNum++;
        for (ch = 0; ch < 2; ch++)
            for (i = 0; i < 1152; i++) buf[ch][i] = next_sample(f + 2 * ch, i);
        for (gr = 0; gr < mode_gr; gr++)
            for (ch = 0; ch < stereo; ch++)
                L3psycho_anal(&buf[ch][gr * 576], &sam[ch][0], ch, 3, snr32, 44100.0, &ratio.l[gr][ch][0], &ratio.s[gr][ch][0], &pe[gr][ch],
                              &l3_side.gr[gr].ch[ch].tt);
        win_buf[0] = &buf[0][0];
        win_buf[1] = &buf[1][0];
        for (gr = 0; gr < mode_gr; gr++)
            for (ch = 0; ch < stereo; ch++)
                for (j = 0; j < 18; j++) {
                    window_subband(&win_buf[ch], &win_que[ch][0], ch);
                    filter_subband(&win_que[ch][0], &l3_sb_sample[ch][gr + 1][j][0]);
                }
        if (f % 5 == 1) l3_sb_sample[1][2][7][5] = l3_sb_sample[1][2][7][5] * 0.5 + 0.001;
        mdct_sub(&l3_sb_sample, xr, stereo, &l3_side, mode_gr);
        fwrite(xr, sizeof(xr), 1, dump);
        fwrite(l3_sb_sample, sizeof(l3_sb_sample), 1, dump);
        if (f % 5 == 2) { xr[0][1][10] *= 1.5; xr[1][0][200] = -xr[1][0][200]; xr[1][1][3] += 0.01; }
        if (f % 5 == 3) pe[1][0] += 50.0;
        iteration_loop(pe, xr, &ratio, &l3_side, l3_enc, mean_bits, stereo, xr_dec, &scalefac, &fr_ps, 0, bitsPerFrame);
        fwrite(l3_enc, sizeof(l3_enc), 1, dump);
        fwrite(&scalefac, sizeof(scalefac), 1, dump);
        for (gr = 0; gr < mode_gr; gr++)
            for (ch = 0; ch < stereo; ch++) {
                const gr_info *g = &l3_side.gr[gr].ch[ch].tt;
                const unsigned v[14] = {g->part2_3_length, g->big_values, g->count1, g->global_gain, g->scalefac_compress, g->table_select[0], g->table_select[1],
                                        g->table_select[2], g->region0_count, g->region1_count, g->preflag, g->count1table_select, g->part2_length, g->block_type};
                fwrite(v, sizeof(v), 1, dump);
            }
        fwrite(&l3_side.main_data_begin, sizeof(l3_side.main_data_begin), 1, dump);
        fwrite(&l3_side.resvDrain, sizeof(l3_side.resvDrain), 1, dump);
        if (f % 7 == 4)
            for (i = 0; i < 576; i++) xr[0][0][i] = -xr[0][0][i];
        if (f % 7 == 5) info.copyright = 1;
        III_format_bitstream(bitsPerFrame, &fr_ps, l3_enc, &l3_side, &scalefac, &bs, xr, NULL, 0);
        info.copyright = 0;
        fwrite(l3_enc, sizeof(l3_enc), 1, dump); /* (signs applied in place) */
        fwrite(&l3_side.main_data_begin, sizeof(l3_side.main_data_begin), 1, dump);
    }
    III_FlushBitstream();
    close_bit_stream_w(&bs);
    fclose(dump);
    if (mp3mi_dropin_waits) printf("waits %ld frames %d\n", mp3mi_dropin_waits(), frames);
    return 0;
}

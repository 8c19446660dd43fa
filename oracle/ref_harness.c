/* TEST INFRASTRUCTURE -- not part of the product path.
 *
 * Our own driver around the UNMODIFIED reference objects (compiled from
 * /root/reference/src where they lie; see oracle/Makefile target `ref`).  It
 * replays the Layer III call sequence of src/musicin.c:585-805 and writes one
 * stage_dump_t per frame, so the CPU restatement (oracle/mp3_oracle.c) can be
 * pinned stage by stage against the real reference.
 *
 * usage: ref_harness in.wav out.mp3 <rate_hz> <kbps> <s|m|d>[e][c][o] [dump.bin]
 *        ref_harness_fft (the same source with -DFFT_SEAM): ... [dump.bin] [fft_seam.bin]
 *        (the mode letter as the driver's -m; e / c / o = its -e, -c, -o options, src/musicin.c:263-275)
 *
 * Only compiled when /root/reference exists (this container); nothing here
 * travels as source of the reference.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include "common.h"
#include "encoder.h"
#include "l3psy.h"
#include "mdct.h"
#include "loop.h"
#include "l3bitstream.h"
#include "stage_dump.h"
#ifdef FFT_SEAM
/* The direct FFT seam (oracle/fft_seam.h): the reference's own fft() -- an external-linkage function of src/subs.c that
 * the unmodified l3psy.o calls -- is intercepted at link time (-Wl,--wrap=fft) and what it returns is copied.  A call of
 * L3psycho_anal makes four: the long transform, then the three short ones (src/l3psy.c:494, 527). */
#include "fft_seam.h"
static fft_seam_t fft_rec;
static int fft_calls;
void __real_fft(float x_real[], float x_imag[], float energy[], float phi[], int N);
void __wrap_fft(float x_real[], float x_imag[], float energy[], float phi[], int N)
{
    int i;
    __real_fft(x_real, x_imag, energy, phi, N);
    if (N == 1024) {
        memcpy(fft_rec.energy_l, energy, sizeof(fft_rec.energy_l));
        for (i = 0; i < 6; i++) {
            fft_rec.phi_l[i] = phi[i];
            fft_rec.re_l[i] = x_real[i];
            fft_rec.im_l[i] = i ? x_real[N - i] : 0.0f;
        }
        fft_calls = 1;
    } else {
        int w = fft_calls++ - 1;
        if (w < 0 || w > 2) { fprintf(stderr, "ref_harness: unexpected fft call order\n"); exit(3); }
        memcpy(fft_rec.energy_s[w], energy, sizeof(fft_rec.energy_s[w]));
        for (i = 0; i < 50; i++) {
            fft_rec.phi_s[w][i] = phi[2 + i];
            fft_rec.re_s[w][i] = x_real[2 + i];
            fft_rec.im_s[w][i] = x_real[N - 2 - i];
        }
    }
}
#endif

/* globals the reference objects expect from its driver (src/musicin.c:148-156) */
FILE *musicin;
Bit_stream_struc bs;
char *programName = "ref_harness";
int iswav = 0;
int littleData = 0;
int streaming_input = 0;
unsigned long frameNum = 0;

extern void III_FlushBitstream(void);

int main(int argc, char **argv)
{
    typedef double IN[2][HAN_SIZE];
    static short buffer[2][1152];
    static short sam[2][1344];
    static double xr[2][2][576], xr_dec[2][2][576], pe[2][2];
    static int l3_enc[2][2][576];
    static III_psy_ratio ratio;
    static III_side_info_t l3_side;
    static III_scalefac_t scalefac;
    static stage_dump_t d;
    FLOAT snr32[32];
    L3SBS *l3_sb_sample = (L3SBS *) mem_alloc(sizeof(L3SBS), "l3_sb_sample");
    IN *win_que = (IN *) mem_alloc(sizeof(IN), "win_que");
    short *win_buf[2];
    frame_params fr_ps;
    layer info;
    FILE *dump = NULL;
#ifdef FFT_SEAM
    FILE *fft_dump = NULL;
#endif
    struct stat sb;
    unsigned long num_samples;
    int stereo, whole_SpF, gr, ch, j, i, k, kbps;
    long rate;

    if (argc < 6) {
        fprintf(stderr, "usage: %s in.wav out.mp3 rate_hz kbps s|m|d[e][c][o] [dump.bin]\n", argv[0]);
        return 2;
    }
    rate = atol(argv[3]);
    kbps = atoi(argv[4]);
    memset(&info, 0, sizeof(info));
    memset(snr32, 0, sizeof(snr32));
    fr_ps.header = &info;
    fr_ps.tab_num = -1;
    fr_ps.alloc = NULL;
    info.lay = 3;
    info.mode = (argv[5][0] == 'm') ? MPG_MD_MONO : (argv[5][0] == 'd') ? MPG_MD_DUAL_CHANNEL : MPG_MD_STEREO;
    info.mode_ext = 0;
    info.error_protection = strchr(argv[5] + 1, 'e') != NULL;
    info.copyright = strchr(argv[5] + 1, 'c') != NULL;
    info.original = strchr(argv[5] + 1, 'o') != NULL;
    info.sampling_frequency = SmpFrqIndex(rate, &info.version);
    info.bitrate_index = BitrateIndex(3, kbps, info.version);
    if (info.sampling_frequency < 0 || info.bitrate_index < 0 || info.version != 1) return 2;
    if (argc > 6) dump = fopen(argv[6], "wb");
#ifdef FFT_SEAM
    if (argc > 7) fft_dump = fopen(argv[7], "wb"); /* [dump.bin] [fft_seam.bin] */
#endif

    musicin = fopen(argv[1], "rb");
    if (!musicin) { perror(argv[1]); return 1; }
    iswav = 1;
    fseek(musicin, 0x2c, SEEK_SET);
    fstat(fileno(musicin), &sb);
    num_samples = (sb.st_size - 0x2c) / 2;
    open_bit_stream_w(&bs, argv[2], BUFFER_SIZE);
    hdr_to_frps(&fr_ps);
    stereo = fr_ps.stereo;
    whole_SpF = (int) (((double) 1152 / s_freq[info.version][info.sampling_frequency]) *
                       ((double) bitrate[info.version][2][info.bitrate_index] / 8.0));
    info.padding = 0;

    while (get_audio(musicin, buffer, num_samples, stereo, &info) > 0) {
        int bitsPerFrame = 8 * whole_SpF;
        int sideinfo_len = 32 + (stereo == 1 ? 136 : 256) + (info.error_protection ? 16 : 0); /* src/musicin.c:728-746 */
        int mean_bits = (bitsPerFrame - sideinfo_len) / 2;
        memset(&d, 0, sizeof(d));
        d.magic = STAGE_DUMP_MAGIC;
        d.frame_index = (int) frameNum;
        frameNum++;
        win_buf[0] = &buffer[0][0];
        win_buf[1] = &buffer[1][0];

        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < stereo; ch++) {
                L3psycho_anal(&buffer[ch][gr * 576], &sam[ch][0], ch, 3, snr32,
                              s_freq[info.version][info.sampling_frequency] * 1000.0,
                              &ratio.l[gr][ch][0], &ratio.s[gr][ch][0], &pe[gr][ch],
                              &l3_side.gr[gr].ch[ch].tt);
                d.psy_block_type[gr][ch] = l3_side.gr[gr].ch[ch].tt.block_type;
#ifdef FFT_SEAM
                if (fft_dump) {
                    if (fft_calls != 4) { fprintf(stderr, "ref_harness: %d fft calls in one L3psycho_anal\n", fft_calls); return 3; }
                    fwrite(&fft_rec, sizeof(fft_rec), 1, fft_dump);
                }
                fft_calls = 0;
#endif
            }
        memcpy(d.pe, pe, sizeof(pe));
        memcpy(d.ratio_l, ratio.l, sizeof(ratio.l));
        memcpy(d.ratio_s, ratio.s, sizeof(ratio.s));

        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < stereo; ch++)
                for (j = 0; j < 18; j++) {
                    window_subband(&win_buf[ch], &(*win_que)[ch][0], ch);
                    filter_subband(&(*win_que)[ch][0], &(*l3_sb_sample)[ch][gr + 1][j][0]);
                }
        for (ch = 0; ch < stereo; ch++)
            for (gr = 0; gr < 2; gr++)
                memcpy(d.sb_sample[ch][gr], (*l3_sb_sample)[ch][gr + 1], sizeof(d.sb_sample[0][0]));

        mdct_sub(l3_sb_sample, xr, stereo, &l3_side, 2);
        memcpy(d.xr, xr, sizeof(xr));

        iteration_loop(pe, xr, &ratio, &l3_side, l3_enc, mean_bits, stereo, xr_dec,
                       &scalefac, &fr_ps, 0, bitsPerFrame);
        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < stereo; ch++) {
                gr_info *g = &l3_side.gr[gr].ch[ch].tt;
                memcpy(d.l3_enc[gr][ch], l3_enc[gr][ch], sizeof(l3_enc[0][0]));
                d.gi[gr][ch].part2_3_length = g->part2_3_length;
                d.gi[gr][ch].big_values = g->big_values;
                d.gi[gr][ch].count1 = g->count1;
                d.gi[gr][ch].global_gain = g->global_gain;
                d.gi[gr][ch].scalefac_compress = g->scalefac_compress;
                d.gi[gr][ch].window_switching_flag = g->window_switching_flag;
                d.gi[gr][ch].block_type = g->block_type;
                d.gi[gr][ch].mixed_block_flag = g->mixed_block_flag;
                for (k = 0; k < 3; k++) {
                    d.gi[gr][ch].table_select[k] = g->table_select[k];
                    d.gi[gr][ch].subblock_gain[k] = g->subblock_gain[k];
                }
                d.gi[gr][ch].region0_count = g->region0_count;
                d.gi[gr][ch].region1_count = g->region1_count;
                d.gi[gr][ch].preflag = g->preflag;
                d.gi[gr][ch].scalefac_scale = g->scalefac_scale;
                d.gi[gr][ch].count1table_select = g->count1table_select;
                d.gi[gr][ch].part2_length = g->part2_length;
                for (i = 0; i < 22; i++) d.scalefac_l[gr][ch][i] = scalefac.l[gr][ch][i];
                for (i = 0; i < 13; i++)
                    for (k = 0; k < 3; k++) d.scalefac_s[gr][ch][i][k] = scalefac.s[gr][ch][i][k];
            }
        d.main_data_begin = l3_side.main_data_begin;
        d.resvDrain = l3_side.resvDrain;
        for (ch = 0; ch < stereo; ch++)
            for (i = 0; i < 4; i++) d.scfsi[ch][i] = l3_side.scfsi[ch][i];

        III_format_bitstream(bitsPerFrame, &fr_ps, l3_enc, &l3_side, &scalefac, &bs, xr, NULL, 0);
        if (dump) fwrite(&d, sizeof(d), 1, dump);
    }
    III_FlushBitstream();
    close_bit_stream_w(&bs);
    if (dump) fclose(dump);
#ifdef FFT_SEAM
    if (fft_dump) fclose(fft_dump);
#endif
    fclose(musicin);
    return 0;
}

/* TEST INFRASTRUCTURE -- not part of the product path.
 *
 * Our own driver around the UNMODIFIED reference objects (compiled from /root/reference/src where they lie; see
 * oracle/Makefile target `ref`).  It replays the Layer I / Layer II call sequence of src/musicin.c:585-704 with
 * psychoacoustic model 2 and writes one stage_dump_l12_t per frame, so that the CPU restatement
 * (oracle/mp12_oracle.inc) can be pinned seam by seam against the real reference.
 *
 * usage: ref_harness_l12 in.wav out.mpg <layer> <rate_hz> <kbps> <s|m|d|j>[e][c][o] [dump.bin]
 *        (the mode letter as the driver's -m; e / c / o = its -e, -c, -o options, src/musicin.c:263-275)
 * psycho_anal opens "out.dat" in the working directory (src/psy.c:116): run it in a scratch directory.
 *
 * Only compiled when /root/reference exists (this container); nothing here travels as source of the reference.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include "common.h"
#include "encoder.h"
#include "stage_dump_l12.h"

/* globals the reference objects expect from its driver (src/musicin.c:148-156) */
FILE *musicin;
Bit_stream_struc bs;
char *programName = "ref_harness_l12";
int iswav = 0;
int littleData = 0;
int streaming_input = 0;
unsigned long frameNum = 0;

int main(int argc, char **argv)
{
    typedef double SBS[2][3][SCALE_BLOCK][SBLIMIT];
    typedef double JSBS[3][SCALE_BLOCK][SBLIMIT];
    typedef double IN[2][HAN_SIZE];
    typedef unsigned int SUB[2][3][SCALE_BLOCK][SBLIMIT];
    static short buffer[2][1152];
    static short sam[2][1344];
    static unsigned int bit_alloc[2][SBLIMIT], scfsi[2][SBLIMIT], scalar[2][3][SBLIMIT], j_scale[3][SBLIMIT];
    static double ltmin[2][SBLIMIT], max_sc[2][SBLIMIT];
    static stage_dump_l12_t d;
    static unsigned int crc;
    FLOAT snr32[32];
    SBS *sb_sample = (SBS *) mem_alloc(sizeof(SBS), "sb_sample");
    JSBS *j_sample = (JSBS *) mem_alloc(sizeof(JSBS), "j_sample");
    IN *win_que = (IN *) mem_alloc(sizeof(IN), "win_que");
    SUB *subband = (SUB *) mem_alloc(sizeof(SUB), "subband");
    short *win_buf[2];
    frame_params fr_ps;
    layer info;
    FILE *dump = NULL;
    struct stat sb;
    unsigned long num_samples, bitsPerSlot, samplesPerFrame;
    int stereo, whole_SpF, i, j, k, kbps, adb, error_protection;
    long rate;

    if (argc < 7) {
        fprintf(stderr, "usage: %s in.wav out.mpg layer rate_hz kbps s|m|d|j[e][c][o] [dump.bin]\n", argv[0]);
        return 2;
    }
    memset(&info, 0, sizeof(info));
    memset(snr32, 0, sizeof(snr32));
    fr_ps.header = &info;
    fr_ps.tab_num = -1;
    fr_ps.alloc = NULL;
    info.lay = atoi(argv[3]);
    rate = atol(argv[4]);
    kbps = atoi(argv[5]);
    if (info.lay != 1 && info.lay != 2) return 2;
    info.mode = (argv[6][0] == 'm') ? MPG_MD_MONO : (argv[6][0] == 'd') ? MPG_MD_DUAL_CHANNEL
              : (argv[6][0] == 'j') ? MPG_MD_JOINT_STEREO : MPG_MD_STEREO;
    info.mode_ext = 0;
    info.error_protection = strchr(argv[6] + 1, 'e') != NULL;
    info.copyright = strchr(argv[6] + 1, 'c') != NULL;
    info.original = strchr(argv[6] + 1, 'o') != NULL;
    info.sampling_frequency = SmpFrqIndex(rate, &info.version);
    info.bitrate_index = BitrateIndex(info.lay, kbps, info.version);
    if (info.sampling_frequency < 0 || info.bitrate_index < 0 || info.version != 1) return 2;
    if (argc > 7) dump = fopen(argv[7], "wb");

    musicin = fopen(argv[1], "rb");
    if (!musicin) { perror(argv[1]); return 1; }
    iswav = 1;
    fseek(musicin, 0x2c, SEEK_SET);
    fstat(fileno(musicin), &sb);
    num_samples = (sb.st_size - 0x2c) / 2;
    open_bit_stream_w(&bs, argv[2], BUFFER_SIZE);
    hdr_to_frps(&fr_ps);
    stereo = fr_ps.stereo;
    error_protection = info.error_protection;
    if (info.lay == 1) { bitsPerSlot = 32; samplesPerFrame = 384; }
    else { bitsPerSlot = 8; samplesPerFrame = 1152; }
    whole_SpF = (int) (((double) samplesPerFrame / s_freq[info.version][info.sampling_frequency]) *
                       ((double) bitrate[info.version][info.lay - 1][info.bitrate_index] / (double) bitsPerSlot));
    info.padding = 0; /* src/musicin.c:566-581: the fraction is dropped before it is looked at */

    while (get_audio(musicin, buffer, num_samples, stereo, &info) > 0) {
        memset(&d, 0, sizeof(d));
        d.magic = STAGE_DUMP_L12_MAGIC;
        d.frame_index = (int) frameNum;
        frameNum++;
        win_buf[0] = &buffer[0][0];
        win_buf[1] = &buffer[1][0];
        adb = whole_SpF * bitsPerSlot;
        if (info.lay == 1) { /* src/musicin.c:620-658 */
            for (j = 0; j < SCALE_BLOCK; j++)
                for (k = 0; k < stereo; k++) {
                    window_subband(&win_buf[k], &(*win_que)[k][0], k);
                    filter_subband(&(*win_que)[k][0], &(*sb_sample)[k][0][j][0]);
                }
            I_scale_factor_calc(*sb_sample, scalar, stereo);
            if (fr_ps.actual_mode == MPG_MD_JOINT_STEREO) {
                I_combine_LR(*sb_sample, *j_sample);
                I_scale_factor_calc(j_sample, &j_scale, 1);
            }
            put_scale(scalar, &fr_ps, max_sc);
            for (k = 0; k < stereo; k++) {
                psycho_anal(&buffer[k][0], &sam[k][0], k, info.lay, snr32,
                            (FLOAT) s_freq[info.version][info.sampling_frequency] * 1000);
                for (i = 0; i < SBLIMIT; i++) ltmin[k][i] = (double) snr32[i];
            }
            I_main_bit_allocation(ltmin, bit_alloc, &adb, &fr_ps);
            if (error_protection) I_CRC_calc(&fr_ps, bit_alloc, &crc);
        } else { /* src/musicin.c:662-704 */
            for (i = 0; i < 3; i++)
                for (j = 0; j < SCALE_BLOCK; j++)
                    for (k = 0; k < stereo; k++) {
                        window_subband(&win_buf[k], &(*win_que)[k][0], k);
                        filter_subband(&(*win_que)[k][0], &(*sb_sample)[k][i][j][0]);
                    }
            II_scale_factor_calc(*sb_sample, scalar, stereo, fr_ps.sblimit);
            pick_scale(scalar, &fr_ps, max_sc);
            if (fr_ps.actual_mode == MPG_MD_JOINT_STEREO) {
                II_combine_LR(*sb_sample, *j_sample, fr_ps.sblimit);
                II_scale_factor_calc(j_sample, &j_scale, 1, fr_ps.sblimit);
            }
            for (k = 0; k < stereo; k++) {
                psycho_anal(&buffer[k][0], &sam[k][0], k, info.lay, snr32,
                            (FLOAT) s_freq[info.version][info.sampling_frequency] * 1000);
                for (i = 0; i < SBLIMIT; i++) ltmin[k][i] = (double) snr32[i];
            }
            II_transmission_pattern(scalar, scfsi, &fr_ps);
            II_main_bit_allocation(ltmin, scfsi, bit_alloc, &adb, &fr_ps);
            if (error_protection) II_CRC_calc(&fr_ps, bit_alloc, scfsi, &crc);
        }
        memcpy(d.sb_sample, *sb_sample, sizeof(d.sb_sample));
        memcpy(d.ltmin, ltmin, sizeof(ltmin));
        for (k = 0; k < 2; k++)
            for (i = 0; i < 32; i++) {
                for (j = 0; j < 3; j++) d.scalar[k][j][i] = (int32_t) scalar[k][j][i];
                d.scfsi[k][i] = (int32_t) scfsi[k][i];
                d.bit_alloc[k][i] = (int32_t) bit_alloc[k][i];
            }
        for (j = 0; j < 3; j++)
            for (i = 0; i < 32; i++) d.j_scale[j][i] = (int32_t) j_scale[j][i];
        d.mode = info.mode; d.mode_ext = info.mode_ext; d.jsbound = fr_ps.jsbound; d.sblimit = fr_ps.sblimit;
        d.adb_left = adb; d.crc = error_protection ? (int32_t) crc : 0;
        encode_info(&fr_ps, &bs);
        if (error_protection) encode_CRC(crc, &bs);
        if (info.lay == 1) {
            I_encode_bit_alloc(bit_alloc, &fr_ps, &bs);
            I_encode_scale(scalar, bit_alloc, &fr_ps, &bs);
            I_subband_quantization(scalar, *sb_sample, j_scale, *j_sample, bit_alloc, *subband, &fr_ps);
            I_sample_encoding(*subband, bit_alloc, &fr_ps, &bs);
        } else {
            II_encode_bit_alloc(bit_alloc, &fr_ps, &bs);
            II_encode_scale(bit_alloc, scfsi, scalar, &fr_ps, &bs);
            II_subband_quantization(scalar, *sb_sample, j_scale, *j_sample, bit_alloc, *subband, &fr_ps);
            II_sample_encoding(*subband, bit_alloc, &fr_ps, &bs);
        }
        for (i = 0; i < adb; i++) put1bit(&bs, 0);
        if (dump) fwrite(&d, sizeof(d), 1, dump);
    }
    close_bit_stream_w(&bs);
    if (dump) fclose(dump);
    fclose(musicin);
    return 0;
}

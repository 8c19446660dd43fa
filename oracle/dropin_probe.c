/* TEST INFRASTRUCTURE -- not part of the product path.
 *
 * Calls L3psycho_anal / window_subband / filter_subband in the order of the reference's Layer III frame loop
 * (src/musicin.c:751-769) -- and NOT the way that loop behaves in between: the drop-in library (csrc/dropin.cpp) reads
 * ahead in memory the caller has handed over (a channel's second granule at its first L3psycho_anal call, a frame's 36
 * slots of both channels at the first window_subband call) and must notice when a caller does not leave that memory alone:
 *   - odd frames: granule 1 of channel 0 is rewritten between the channel's two L3psycho_anal calls;
 *   - frames 1 mod 4: a sample of a LATER slot of channel 1 is rewritten between two window_subband calls;
 *   - frames 3 mod 5: channel 0's buffer pointer is set back by 32 samples in the middle of the frame;
 *   - every third frame is handed over in ANOTHER buffer.
 * Every returned value is written to the dump.  Linked once against the unmodified reference objects (oracle/Makefile:
 * _ref/dropin_probe_ref) and once against the library (_ref/dropin_probe, _ref/dropin_probe_emu): the dumps must be equal.
 *
 * usage: dropin_probe dump.bin [frames]
 * Only compiled where /root/reference exists (its headers give the prototypes); nothing of the reference travels as source.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "common.h"
#include "encoder.h"
#include "l3psy.h"

/* globals the reference objects expect from their driver (src/musicin.c:148-156) */
FILE *musicin;
Bit_stream_struc bs;
char *programName = "dropin_probe";
int iswav = 0;
int littleData = 0;
int streaming_input = 0;
unsigned long frameNum = 0;

/* the library says how often it waited for the device (absent in the reference link) */
extern long mp3mi_dropin_waits(void) __attribute__((weak));

static unsigned lcg = 12345u;
static short next_sample(int f, int n)
{
    lcg = lcg * 1664525u + 1013904223u;
    /* a tone, noise, and a burst in some frames so that block types switch */
    {
        const int tone = (int) (6000.0 * ((n * (f % 7 + 3)) % 64 - 32) / 32.0);
        const int noise = (int) ((lcg >> 16) & 0x3ff) - 512;
        const int burst = (f % 4 == 2 && n > 700 && n < 760) ? (((lcg >> 8) & 1) ? 9000 : -9000) : 0;
        return (short) (tone + noise + burst);
    }
}

int main(int argc, char **argv)
{
    typedef double IN[2][HAN_SIZE];
    static short buf_a[2][1152], buf_b[2][1152];
    static short sam[2][1344];
    static IN win_que;
    static double s[32];
    FLOAT snr32[32];
    short *win_buf[2];
    FILE *dump;
    int frames = 9, f, gr, ch, j, i;
    if (argc < 2) { fprintf(stderr, "usage: %s dump.bin [frames]\n", argv[0]); return 2; }
    if (argc > 2) frames = atoi(argv[2]);
    dump = fopen(argv[1], "wb");
    if (!dump) { perror(argv[1]); return 1; }
    memset(sam, 0, sizeof(sam));
    for (f = 0; f < frames; f++) {
        short (*buf)[1152] = (f % 3 == 2) ? buf_b : buf_a;
        for (ch = 0; ch < 2; ch++)
            for (i = 0; i < 1152; i++) buf[ch][i] = next_sample(f + 3 * ch, i);
        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < 2; ch++) {
                static gr_info gi;
                double ratio_l[21], ratio_s[12][3], pe = 0.0;
                int bt;
                if ((f & 1) && gr == 1 && ch == 0)
                    for (i = 100; i < 140; i++) buf[0][576 + i] = (short) (buf[0][576 + i] ^ 0x155); /* after (gr 0, ch 0) was served */
                memset(&gi, 0, sizeof(gi));
                L3psycho_anal(&buf[ch][gr * 576], &sam[ch][0], ch, 3, snr32, 44100.0, ratio_l, &ratio_s[0], &pe, &gi);
                bt = (int) gi.block_type;
                fwrite(ratio_l, sizeof(ratio_l), 1, dump);
                fwrite(ratio_s, sizeof(ratio_s), 1, dump);
                fwrite(&pe, sizeof(pe), 1, dump);
                fwrite(&bt, sizeof(bt), 1, dump);
            }
        win_buf[0] = &buf[0][0];
        win_buf[1] = &buf[1][0];
        for (gr = 0; gr < 2; gr++)
            for (ch = 0; ch < 2; ch++)
                for (j = 0; j < 18; j++) {
                    if (f % 4 == 1 && gr == 1 && ch == 1 && j == 5) buf[1][576 + 32 * 7 + 3] = (short) (buf[1][576 + 32 * 7 + 3] + 77); /* a later slot */
                    if (f % 5 == 3 && gr == 0 && ch == 0 && j == 9) win_buf[0] -= 32; /* the same 32 samples once more */
                    window_subband(&win_buf[ch], &win_que[ch][0], ch);
                    filter_subband(&win_que[ch][0], s);
                    fwrite(&win_que[ch][0], sizeof(double), HAN_SIZE, dump);
                    fwrite(s, sizeof(s), 1, dump);
                }
    }
    fclose(dump);
    if (mp3mi_dropin_waits) printf("waits %ld frames %d\n", mp3mi_dropin_waits(), frames);
    return 0;
}

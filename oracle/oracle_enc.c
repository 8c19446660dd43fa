/* TEST INFRASTRUCTURE -- command-line front end of the CPU restatement.
 * usage: oracle_enc in.wav out.mp3 rate_hz kbps s|m [dump.bin [max_frames]]
 * The 44-byte WAV header is skipped blindly, as src/musicin.c:357-362 does.
 * With "-t N" as first args it encodes the input N times and prints frames/s (CPU baseline).
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "mp3_oracle.h"

int main(int argc, char **argv)
{
    int reps = 1, timing = 0;
    FILE *f;
    long sz;
    int16_t *pcm;
    uint8_t *out = NULL;
    size_t n, len = 0;
    int rate, kbps, ch, max_dumps = 0, r;
    stage_dump_t *dumps = NULL;
    struct timespec t0, t1;
    if (argc > 2 && !strcmp(argv[1], "-t")) { reps = atoi(argv[2]); timing = 1; argv += 2; argc -= 2; }
    if (argc < 6) { fprintf(stderr, "usage: oracle_enc [-t N] in.wav out.mp3 rate_hz kbps s|m [dump.bin [max_frames]]\n"); return 2; }
    f = fopen(argv[1], "rb");
    if (!f) { perror(argv[1]); return 1; }
    fseek(f, 0, SEEK_END); sz = ftell(f); fseek(f, 0x2c, SEEK_SET);
    n = (size_t) (sz - 0x2c) / 2;
    pcm = (int16_t *) malloc(n * 2 + 2);
    if (fread(pcm, 2, n, f) != n) { fprintf(stderr, "short read\n"); return 1; }
    fclose(f);
    rate = atoi(argv[3]); kbps = atoi(argv[4]); ch = (argv[5][0] == 'm') ? 1 : 2;
    if (argc > 6) {
        max_dumps = (argc > 7) ? atoi(argv[7]) : (int) (n / (1152 * (size_t) ch) + 1);
        dumps = (stage_dump_t *) calloc((size_t) max_dumps, sizeof(stage_dump_t));
    }
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (r = 0; r < reps; r++) {
        free(out);
        len = mp3o_encode_pcm(rate, kbps, ch, pcm, n, &out, dumps, max_dumps);
    }
    clock_gettime(CLOCK_MONOTONIC, &t1);
    if (!out) { fprintf(stderr, "unsupported configuration\n"); return 2; }
    f = fopen(argv[2], "wb");
    fwrite(out, 1, len, f);
    fclose(f);
    if (dumps) {
        size_t frames = (n + 1152 * (size_t) ch - 1) / (1152 * (size_t) ch);
        if (frames > (size_t) max_dumps) frames = (size_t) max_dumps;
        f = fopen(argv[6], "wb");
        fwrite(dumps, sizeof(stage_dump_t), frames, f);
        fclose(f);
    }
    if (timing) {
        double dt = (double) (t1.tv_sec - t0.tv_sec) + 1e-9 * (double) (t1.tv_nsec - t0.tv_nsec);
        double frames = (double) reps * (double) ((n + 1152 * (size_t) ch - 1) / (1152 * (size_t) ch));
        printf("{\"frames\": %.0f, \"seconds\": %.6f, \"frames_per_s\": %.2f}\n", frames, dt, frames / dt);
    }
    return 0;
}

#ifndef MP3MI_H
#define MP3MI_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
void mp3mi_synth_pcm(int16_t *out, long n_per_ch, int channels, int rate_hz,
                     uint32_t stream, uint32_t seed);
#ifdef __cplusplus
}
#endif
#endif

/* TEST INFRASTRUCTURE -- not part of the product path.
 *
 * What the psychoacoustic model's transforms hand to the rest of L3psycho_anal, per call (frame, granule, channel):
 * `energy` and `phi` as fft() / enphinew() leave them (src/subs.c:38-123; call sites src/l3psy.c:494, 527) and the raw
 * spectrum lines behind them (x_real[i], x_real[N - i] after rsfft: phi = atan2(-im, re), src/subs.c:78).  Only what
 * L3psycho_anal reads: the long transform's 513 energies and lines 0..5 (src/l3psy.c:496-512), the three short
 * transforms' 129 energies each and lines 2..51 (k = (j + 2) >> 2 for j = 6, 10 .. 202: src/l3psy.c:531-549).
 * Written identically by
 *   - oracle/ref_harness.c built with -DFFT_SEAM (our driver around the UNMODIFIED reference objects: the reference's
 *     own fft() is intercepted at link time, -Wl,--wrap=fft, and its outputs copied), and
 *   - oracle/mp3_oracle.c (mp3o_encode_pcm_fft_seam),
 * and compared with the device's k_fft outputs (mp3mi_batch_debug_fetch 6 / 7 / 8) by tests/test_fft_seam.py.
 * All members are f32, little-endian. */
#ifndef ORACLE_FFT_SEAM_H
#define ORACLE_FFT_SEAM_H

typedef struct {
    float energy_l[513];
    float phi_l[6], re_l[6], im_l[6];      /* long lines 0..5; im_l[0] = 0 (line 0 is real) */
    float energy_s[3][129];
    float phi_s[3][50], re_s[3][50], im_s[3][50]; /* short lines 2..51 of the three windows */
} fft_seam_t;                               /* one per L3psycho_anal call, in call order: [frame][gr][ch] */

#endif

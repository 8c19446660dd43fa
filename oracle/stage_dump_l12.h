/* TEST INFRASTRUCTURE -- not part of the product path.
 *
 * Binary per-frame record of the observable seams of the Layer I / II frame loop
 * (reference: src/musicin.c:620-704).  Written identically by
 *   - oracle/ref_harness_l12.c (our driver around the UNMODIFIED reference objects), and
 *   - oracle/mp12_oracle.inc   (the CPU restatement),
 * so that tests can compare the two byte for byte.  f64 members first, then i32.
 */
#ifndef ORACLE_STAGE_DUMP_L12_H
#define ORACLE_STAGE_DUMP_L12_H

#include <stdint.h>

#define STAGE_DUMP_L12_MAGIC 0x3270646d /* "mdp2" */

typedef struct {
    /* after the polyphase filterbank: sb_sample[ch][3][12][32] (Layer I fills [ch][0] only), src/musicin.c:622-626, 662-666 */
    double sb_sample[2][3][12][32];
    /* after psycho_anal: ltmin[ch][32] = (double) snr32, src/musicin.c:639-643, 681-686 */
    double ltmin[2][32];
    /* scale factor indices AFTER II_transmission_pattern (Layer I: [ch][0] only), src/encode.c:512-561, 638-691 */
    int32_t scalar[2][3][32];
    int32_t j_scale[3][32];
    int32_t scfsi[2][32];
    int32_t bit_alloc[2][32];
    /* what *_main_bit_allocation left in the header and in adb, src/encode.c:882-948 */
    int32_t mode, mode_ext, jsbound, sblimit, adb_left, crc;
    int32_t magic, frame_index;
} stage_dump_l12_t;

#endif

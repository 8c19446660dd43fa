/* TEST INFRASTRUCTURE -- not part of the product path.
 *
 * Binary per-frame record of the observable seams of the Layer III frame loop
 * (reference: src/musicin.c:708-788).  Written identically by
 *   - oracle/ref_harness.c  (our driver around the UNMODIFIED reference objects), and
 *   - oracle/mp3_oracle.c   (the CPU restatement),
 * so that tests can compare the two byte for byte and commit small golden
 * fixtures under tests/golden/.  Little-endian, no padding surprises: every
 * member is f64 or i32 and f64 members come first.
 */
#ifndef ORACLE_STAGE_DUMP_H
#define ORACLE_STAGE_DUMP_H

#include <stdint.h>

#define STAGE_DUMP_MAGIC 0x33706d64 /* "dmp3" */

typedef struct {
    /* after L3psycho_anal x (gr,ch): src/musicin.c:751-758 */
    double pe[2][2];
    double ratio_l[2][2][21];
    double ratio_s[2][2][12][3];
    /* after the polyphase filterbank, before mdct_sub flips signs: [ch][gr][18][32] */
    double sb_sample[2][2][18][32];
    /* after mdct_sub: xr[gr][ch][576] */
    double xr[2][2][576];
    /* after iteration_loop (values still non-negative) */
    int32_t l3_enc[2][2][576];
    int32_t psy_block_type[2][2];
    int32_t main_data_begin; /* value written in THIS frame's side info */
    int32_t resvDrain;
    int32_t scfsi[2][4];
    struct {
        int32_t part2_3_length, big_values, count1, global_gain, scalefac_compress;
        int32_t window_switching_flag, block_type, mixed_block_flag;
        int32_t table_select[3], subblock_gain[3];
        int32_t region0_count, region1_count, preflag, scalefac_scale, count1table_select;
        int32_t part2_length;
    } gi[2][2];
    int32_t scalefac_l[2][2][22];
    int32_t scalefac_s[2][2][13][3];
    int32_t magic;
    int32_t frame_index;
} stage_dump_t;

#endif

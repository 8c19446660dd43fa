#ifndef MP3MI_MDCT_SHAPE_H
#define MP3MI_MDCT_SHAPE_H
#include <stdint.h>
// The SHAPE of the long-block transform -- which inputs form the operand groups, which groups a row sums -- as
// compile-time constants: k_mdct keeps a band's inputs and groups in registers, so every index must be static.
// (The coefficients stay in the table block.)  mp3mi_build_tables derives the same lists from the reference's
// expressions and refuses to hand out tables that disagree (tables_host.cpp).
#if defined(MP3MI_EMU) || !defined(__HIP_DEVICE_COMPILE__)
#define MP3MI_SHAPE static const
#else
#define MP3MI_SHAPE __device__ static const
#endif
MP3MI_SHAPE uint8_t MDCT_G_OPS[6][6] = {{130, 131, 14, 15, 154, 155}, {129, 132, 13, 16, 153, 156}, {128, 133, 12, 17, 152, 157},
                                        {134, 11, 18, 151, 158, 35}, {135, 10, 19, 150, 159, 34}, {136, 9, 20, 149, 160, 33}};
MP3MI_SHAPE uint8_t MDCT_H_OPS[2][18] = {{0, 129, 132, 5, 8, 137, 140, 13, 16, 145, 148, 21, 24, 153, 156, 29, 32, 161},
                                         {130, 131, 6, 7, 138, 139, 14, 15, 146, 147, 22, 23, 154, 155, 30, 31, 162, 163}};
MP3MI_SHAPE uint8_t MDCT_FULL_ROW[12] = {0, 2, 3, 5, 6, 8, 9, 11, 12, 14, 15, 17};
MP3MI_SHAPE uint8_t MDCT_SMALL_ROW[6] = {1, 4, 7, 10, 13, 16};
MP3MI_SHAPE uint8_t MDCT_SMALL_NT[6] = {6, 2, 6, 6, 2, 6};
MP3MI_SHAPE uint8_t MDCT_SMALL_IDX[6][6] = {{18, 19, 20, 21, 22, 23}, {24, 25, 0, 0, 0, 0}, {20, 19, 18, 21, 22, 23},
                                            {20, 19, 18, 21, 22, 23}, {24, 25, 0, 0, 0, 0}, {20, 19, 18, 21, 22, 23}};

#endif

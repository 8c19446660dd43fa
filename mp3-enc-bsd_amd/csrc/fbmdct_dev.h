// Device-side building blocks shared by k_fbmdct.hip (batched) and k_dropin.hip (the
// reference's per-call surface): the matrixing step of filter_subband (src/encode.c:393-408)
// and the MDCT of one granule (src/mdct.c:57-91, 105-511).
//
// Bit-exactness: all arithmetic is f64 with the reference's association order -- y[i] sums
// its 8 taps left to right, each subband sample accumulates its 31 products in table order
// starting from y[16], the long-block MDCT evaluates the reference's bracketed operand groups
// once per band (they recur in several outputs, up to an exact negation) and then accumulates
// each output's terms in the reference's order; nothing may be contracted to FMA
// (-ffp-contract=off).
#ifndef MP3MI_FBMDCT_DEV_H
#define MP3MI_FBMDCT_DEV_H
#include "mp3mi_host.h"
#include "mdct_shape.h"

// 14.8 KB per wavefront: ten fit a CU, or two beside k_loop's sixteen
struct mdct_lds {
    union {
        double in[36][32]; // [k][band]: 18 slots of the previous granule, then 18 of the current one (sign-flipped
                           // like mdct_sub does); for a long block already multiplied by win[0][k]
        struct {
            double xr[576];    // the result takes the place of the first half of the inputs,
            double V[32][26];  // the long-block operand groups per band that of the second half and beyond: a lane
                               // holds its groups in registers until every lane has read its inputs (mdct_granule)
        };
    };
    double win[4][36];
    double cos_s[6][12];
    double fcoef[12][18];  // the twelve full output rows of the long-block transform: coefficient of pair group t
    uint8_t frow[16];      // and their output row
    double scoef[6][6];    // the six short output rows of the long-block transform (2 or 6 terms): coefficients,
    uint8_t sidx[6][8];    // operand-group indices,
    uint8_t srow[8], snt[8]; // output row and number of terms
    uint8_t g_ops[6][6], h_ops[2][18];
};

// per-lane constants of the long-block MDCT.  Twelve of the 18 output rows are the ordered sum of all 18
// pair groups V[0..17] (T->mdct_full_row): a lane owns ONE BAND (lane & 31) and six of those rows (lane >> 5),
// reads the band's 18 groups once and takes the rows' coefficients from LDS, the same address for all lanes of a
// half (L.fcoef; 36 registers of per-lane coefficients were what made the kernel spill at 168 VGPRs).  The other
// six rows have 2 or 6 terms (T->mdct_small_row) and are taken from LDS.
struct mdct_regs {
    double cs, ca; // alias butterfly coefficients of k = lane & 7 (every step of the alias loop has that k)
};

// s[sub] of filter_subband from the 64 folded window sums y (src/encode.c:398-408)
MP3MI_DEVFN double fbm_matrix(const double *y, const double *frow)
{
    double si = y[16];
    for (int j = 0; j < 16; j++) si = si + frow[j] * (y[j] + y[32 - j]);
    for (int j = 0; j < 15; j++) si = si + frow[16 + j] * (y[33 + j] - y[63 - j]);
    return si;
}

MP3MI_DEVFN void mdct_load_tables(mdct_lds &L, mdct_regs &R, const mp3mi_tables *T)
{
    const int lane = wave_lane();
    for (int i = lane; i < 4 * 36; i += 64) (&L.win[0][0])[i] = (&T->mdct_win[0][0])[i];
    for (int i = lane; i < 72; i += 64) (&L.cos_s[0][0])[i] = (&T->cos_s[0][0])[i];
    if (lane < 36) L.g_ops[lane / 6][lane % 6] = T->mdct_g_ops[lane / 6][lane % 6];
    if (lane < 36) (&L.h_ops[0][0])[lane] = (&T->mdct_h_ops[0][0])[lane];
    if (lane < 36) { // the six short rows
        const int sr = lane / 6, t = lane - 6 * sr, m = T->mdct_small_row[sr];
        L.scoef[sr][t] = T->mdct_vcoef[m][t];
        L.sidx[sr][t] = T->mdct_vidx[m][t];
        if (t == 0) { L.srow[sr] = (uint8_t) m; L.snt[sr] = T->mdct_nterm[m]; }
    }
    for (int i = lane; i < 12 * 18; i += 64) L.fcoef[i / 18][i % 18] = T->mdct_vcoef[T->mdct_full_row[i / 18]][i % 18];
    if (lane < 12) L.frow[lane] = T->mdct_full_row[lane];
    R.cs = T->cs[lane & 7];
    R.ca = T->ca[lane & 7];
}

// L.in <- the 36 inputs of every band from two granules of subband samples held in registers (vp: previous
// granule, vc: current one; element lane + 64 j of the [slot][sub] block each); a long block (bt == 0) gets
// its window applied here.  Ends with a barrier.
MP3MI_DEVFN void mdct_store_inputs(mdct_lds &L, const double (&vp)[9], const double (&vc)[9], int bt)
{
    const int lane = wave_lane();
#pragma unroll
    for (int j = 0; j < 9; j++) {
        const int k = (lane >> 5) + 2 * j;
        (&L.in[0][0])[lane + 64 * j] = (bt == 0) ? L.win[0][k] * vp[j] : vp[j];
        (&L.in[0][0])[576 + lane + 64 * j] = (bt == 0) ? L.win[0][18 + k] * vc[j] : vc[j];
    }
    __syncthreads();
}

MP3MI_DEVFN void mdct_load_inputs(mdct_lds &L, const double *prev, const double *cur, int bt)
{
    const int lane = wave_lane();
    double vp[9], vc[9];
#pragma unroll
    for (int j = 0; j < 9; j++) {
        vp[j] = prev[lane + 64 * j];
        vc[j] = cur[lane + 64 * j];
    }
    mdct_store_inputs(L, vp, vc, bt);
}

// ordered signed sum of windowed inputs: ops[i] = index | 0x80 (subtract / negate).  Six operands are fetched at a
// time (the sum itself is one chain in the reference's order): eighteen at once would hold 36 registers.
template <int N> MP3MI_DEVFN double mdct_group(const mdct_lds &L, int band, const uint8_t *ops)
{
    double acc = 0.0;
#pragma unroll
    for (int i0 = 0; i0 < N; i0 += 6) {
        double f[6];
#pragma unroll
        for (int i = 0; i < 6; i++) f[i] = L.in[ops[i0 + i] & 0x3f][band];
#pragma unroll
        for (int i = 0; i < 6; i++) {
            if (i0 + i == 0) acc = (ops[0] & 0x80) ? -f[0] : f[0];
            else acc = (ops[i0 + i] & 0x80) ? acc - f[i] : acc + f[i];
        }
    }
    return acc;
}

// MDCT + alias reduction of one granule: L.in -> L.xr[band*18 + m].  Ends with a barrier.
MP3MI_DEVFN void mdct_granule(mdct_lds &L, const mdct_regs &R, const mp3mi_tables *T, int bt)
{
    const int lane = wave_lane();
    if (bt == 0) { // long window (src/mdct.c:199-509)
        { // phase A: the 26 operand groups of every band; lane = band*2 + h, h picks the half of the list
            const int band = lane >> 1, h = lane & 1;
            double pq[9], g[4];
#pragma unroll
            for (int j0 = 0; j0 < 9; j0 += 3) { // (three pairs at a time: registers)
                double a[3], b[3];
#pragma unroll
                for (int j = 0; j < 3; j++) {
                    a[j] = L.in[h ? 18 + j0 + j : j0 + j][band];
                    b[j] = L.in[h ? 35 - j0 - j : 17 - j0 - j][band];
                }
#pragma unroll
                for (int j = 0; j < 3; j++) pq[j0 + j] = h ? a[j] + b[j] : a[j] - b[j];
            }
#pragma unroll
            for (int c = 0; c < 3; c++) g[c] = mdct_group<6>(L, band, L.g_ops[3 * h + c]);
            g[3] = mdct_group<18>(L, band, L.h_ops[h]);
            __syncthreads(); // every lane has read its inputs: the groups may take their place
#pragma unroll
            for (int j = 0; j < 9; j++) L.V[band][9 * h + j] = pq[j];
#pragma unroll
            for (int c = 0; c < 3; c++) L.V[band][18 + 3 * h + c] = g[c];
            L.V[band][24 + h] = g[3];
        }
        __syncthreads(); // the inputs are dead from here on
        // phase B: every output is the ordered sum of its terms V * coefficient (src/mdct.c:199-509)
        { // the twelve rows over all 18 pair groups: this lane's band, rows 6 h .. 6 h + 5
            const int band = lane & 31, h = lane >> 5;
            double pr[18];
#pragma unroll
            for (int t = 0; t < 18; t++) pr[t] = L.V[band][t];
#pragma unroll 2
            for (int r = 0; r < 6; r++) {
                const double *cf = L.fcoef[6 * h + r];
                double sum = pr[0] * cf[0]; // (one chain in the reference's order)
#pragma unroll
                for (int t = 1; t < 18; t++) sum = sum + pr[t] * cf[t];
                L.xr[band * 18 + L.frow[6 * h + r]] = sum;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; i++) { // the six short rows: 192 outputs
            const int o = lane + 64 * i, band = o & 31, sr = o >> 5;
            const int nt = L.snt[sr];
            double sum = L.V[band][L.sidx[sr][0]] * L.scoef[sr][0];
#pragma unroll
            for (int t = 1; t < 6; t++) {
                const double q = L.V[band][L.sidx[sr][t]] * L.scoef[sr][t];
                sum = (t < nt) ? sum + q : sum;
            }
            L.xr[band * 18 + L.srow[sr]] = sum;
        }
    } else {
        double out[9];
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const int o = lane + 64 * i, band = o / 18, m = o % 18;
            double sum = 0.0;
            if (bt == 2) { // three short transforms, out[3*mm + l]   (src/mdct.c:173-185)
                const int mm = m / 3, l = m % 3;
                for (int k = 0; k < 12; k++) sum = sum + (L.win[2][k] * L.in[k + 6 * l + 6][band]) * L.cos_s[mm][k];
            } else { // start / stop windows, plain 36-term sum (src/mdct.c:188-198)
                for (int k = 0; k < 36; k++) sum = sum + (L.win[bt][k] * L.in[k][band]) * T->cos_l[m][k];
            }
            out[i] = sum;
        }
        __syncthreads(); // every lane has read its inputs
#pragma unroll
        for (int i = 0; i < 9; i++) L.xr[lane + 64 * i] = out[i];
    }
    __syncthreads();
    if (bt != 2) { // alias reduction butterflies (src/mdct.c:83-91)
        for (int i = lane; i < 31 * 8; i += 64) {
            const int band = i >> 3, k = i & 7;
            double up = L.xr[band * 18 + 17 - k], dn = L.xr[(band + 1) * 18 + k];
            double bu = up * R.cs + dn * R.ca;
            double bd = dn * R.cs - up * R.ca;
            L.xr[band * 18 + 17 - k] = bu;
            L.xr[(band + 1) * 18 + k] = bd;
        }
        __syncthreads();
    }
}

#endif

// Polyphase analysis filterbank, then MDCT + alias reduction.
//
// Replaces, for a whole batch, window_subband/filter_subband (src/encode.c:287-409) and
// mdct_sub/mdct (src/mdct.c:25-511).  Two kernels with the subband samples of the chunk in HBM
// between them (4.6 KB per granule and channel, written once and read twice):
//
//   k_filter  one wavefront per (stream, channel, 64 slots); a lane computes the 32 subband samples of one slot.
//             The filterbank is feed-forward -- a slot is a pure function of 512 PCM samples -- so every
//             slot is independent, including those of the granule BEFORE the chunk, which the reference
//             would still hold in l3_sb_sample[ch][0] and which is recomputed here (granule slot 0).
//   k_mdct    one wavefront per (two tracks, run of 22 granules); a lane transforms one band: 36 inputs from two granules.
//
// Arithmetic and its ordering: fbmdct_dev.h.
#include "fbmdct_dev.h"

// k_filter: one LANE per slot.  A slot's 32 subband samples are a function of 512 PCM samples, and every operand
// of that function but the PCM itself is the same for all slots: the 512 window taps and the 31 x 32 matrixing
// coefficients.  With a lane per slot they are wavefront-uniform -- scalar loads, SGPR operands of the f64
// multiplies -- and the lane's own operands (its 32 folded window sums u) stay in registers: the inner loops read no
// LDS at all.  (A lane per SUBBAND, as before, broadcasts the 32 u's of a slot to all lanes through LDS: one 8-byte
// LDS operand per multiply-add pair, and the LDS pipe, shared by the CU's four SIMDs, was what bounded the kernel.)
// A wavefront takes 64 consecutive slots of one (stream, channel) -- 3.6 granules; the slots of the chunk, the
// granule before it included, are numbered through: slot sigma = 18 gi + q.
//
// The 1/32768 of src/encode.c:306-312 is applied to the finished sample instead of to every PCM value: scaling by a
// power of two commutes with every rounding on the way (products, sums; nothing comes near the subnormals), so
// (sum of (x/32768) w ...) == (sum of x w ...) / 32768 bit for bit; the sign flip of mdct_sub's odd slots of odd
// subbands (src/mdct.c:57-60) rides on the same multiply.
#define FILT_SLOTS 64
#define FILT_WIN (32 * (FILT_SLOTS - 1) + 512)            // PCM samples under a wavefront's slots
#define FILT_PAD(t) ((t) + 2 * ((t) >> 5))                 // a dword of padding per 32 samples: lane stride 17 dwords
struct filter_lds {
    union {
        int16_t pcm[FILT_PAD(FILT_WIN) + 2];
        double tr[FILT_SLOTS][17]; // 16 subbands of every slot on their way to coalesced stores (the PCM is dead by then)
    };
};

// y[i] of filter_subband for this lane's slot, times 32768: sum over k of pcm * enwindow[i + 64 k], in k order
// (src/encode.c:306-312, 393-397); P: this lane's window in the padded LDS copy
template <int I> MP3MI_DEVFN double filt_y(const int16_t *P, const mp3mi_tables *T)
{
    double acc = (double) P[FILT_PAD(511 - I)] * T->enwindow[I];
#pragma unroll
    for (int k = 1; k < 8; k++) acc = acc + (double) P[FILT_PAD(511 - I - 64 * k)] * T->enwindow[I + 64 * k];
    return acc;
}

// u[0] = y[16], u[1+j] = y[j] + y[32-j] (j < 16), u[17+j] = y[33+j] - y[63-j] (j < 15)   (src/encode.c:398-408);
// taken in ascending i so that the window taps are read in address order; a + b == b + a bit for bit
template <int I> MP3MI_DEVFN void filt_fold(double (&u)[32], const int16_t *P, const mp3mi_tables *T)
{
    if constexpr (I < 64) {
        if constexpr (I != 48) { // (y[48] has no partner: its coefficient is cos(pi/2))
            const double y = filt_y<I>(P, T);
            if constexpr (I < 16) u[1 + I] = y;
            else if constexpr (I == 16) u[0] = y;
            else if constexpr (I <= 32) u[1 + 32 - I] = u[1 + 32 - I] + y;
            else if constexpr (I < 48) u[17 + I - 33] = y;
            else u[17 + 63 - I] = u[17 + 63 - I] - y;
        }
        // (keeps the compiler from fetching all 504 taps ahead and spilling them: four slots' worth are in flight)
        if constexpr ((I & 3) == 3) __asm__ volatile("" ::: "memory");
        filt_fold<I + 1>(u, P, T);
    }
}

__global__ void __launch_bounds__(64, 3) k_filter(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                  const int16_t *__restrict__ pcm_all, double *__restrict__ sbs,
                                                  double *__restrict__ sb_dbg)
{
    __shared__ filter_lds L;
    const int lane = wave_lane();
    const int C = geo.channels, G1 = geo.n_gran + 1, NS = G1 * 18, NB = (NS + FILT_SLOTS - 1) / FILT_SLOTS;
    // The two channels of a stereo stream sit interleaved in the same cache lines, and consecutive workgroups go to
    // different XCDs (eight L2 caches): the channel is therefore bit 3 of the workgroup index -- workgroups w and w + 8
    // read the same lines, land on the same XCD a moment apart, and the second finds them in that XCD's L2.
    int bid = (int) blockIdx.x, ch = 0;
    if (C == 2) {
        const int total = (int) gridDim.x, full = total & ~15; // (the last, partial group of 16 falls back to the plain order)
        if (bid < full) { ch = (bid >> 3) & 1; bid = ((bid >> 4) << 3) | (bid & 7); }
        else { ch = (bid - full) & 1; bid = (full >> 1) + ((bid - full) >> 1); }
    }
    const int blk = bid % NB;
    const int s = bid / NB;
    const long n_pitch = geo.pcm_pitch ? geo.pcm_pitch : (long) geo.n_frames * 1152; // row pitch of the PCM buffer
    const long n_per_ch = geo.n_samples ? (long) geo.n_samples[s] : n_pitch; // valid samples: the rest reads as zero (src/encode.c:162-166)
    const int16_t *pcm = pcm_all + (size_t) s * (size_t) n_pitch * (size_t) C;
    const int16_t *hist = geo.hist ? geo.hist + (size_t) s * MP3MI_PCM_HIST * (size_t) C : NULL;
    // slot sigma needs samples [32 sigma - 480, 32 sigma + 32) counted from the first sample of granule slot 0; before
    // the call's first sample the stream's history (zeros at the start of a stream), beyond the stream's last sample 0
    const long t0 = 576 * ((long) geo.g0 - 1) + 32L * FILT_SLOTS * blk - 480;
    if (t0 >= 0 && t0 + FILT_WIN <= n_per_ch) { // all inside the call's PCM: one scalar base, 32-bit lane offsets, no range tests
        const int16_t *p0 = pcm + t0 * C + ch;
        const unsigned lo = (unsigned) lane * (unsigned) C, step = 64u * (unsigned) C;
#pragma unroll
        for (int k0 = 0; k0 < FILT_WIN / 64; k0 += 13) {
            int16_t v[13];
#pragma unroll
            for (int k = 0; k < 13; k++) v[k] = p0[lo + step * (unsigned) (k0 + k)];
#pragma unroll
            for (int k = 0; k < 13; k++) L.pcm[FILT_PAD(lane + 64 * (k0 + k))] = v[k];
        }
        if (lane < FILT_WIN % 64) L.pcm[FILT_PAD(lane + 64 * (FILT_WIN / 64))] = p0[lo + step * (unsigned) (FILT_WIN / 64)];
    } else {
        for (int i = lane; i < FILT_WIN; i += 64) {
            const long t = t0 + i;
            const bool past = hist && t < 0 && t >= -MP3MI_PCM_HIST;
            L.pcm[FILT_PAD(i)] = (t >= 0 && t < n_per_ch) ? pcm[t * C + ch] : (past ? hist[(t + MP3MI_PCM_HIST) * C + ch] : (int16_t) 0);
        }
    }
    __syncthreads();
    double u[32];
    filt_fold<0>(u, &L.pcm[34 * lane], T);
    __syncthreads(); // every lane has read its PCM: the results may take its place

    // where the slots of this wavefront go: lane >> 4 picks the slot of a group of four, the walk below adds four
    const int sg = FILT_SLOTS * blk + (lane >> 4);
    const int gi = sg / 18, q = sg - 18 * gi;
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
        // s[sub] = y[16] + sum_j filt[j] (y[j] + y[32-j]) + sum_j filt[16+j] (y[33+j] - y[63-j])   (src/encode.c:398-408)
#pragma unroll 1
        for (int sb = 0; sb < 16; sb++) {
            const double *f = T->filt[16 * half + sb];
            double si = u[0];
#pragma unroll
            for (int j = 0; j < 31; j++) si = si + f[j] * u[1 + j];
            L.tr[lane][sb] = si;
            __asm__ volatile("" ::: "memory"); // (likewise: one row of coefficients at a time)
        }
        __syncthreads();
        // element lane + 64 m of [slot][16] is slot 4 m + (lane >> 4): the walk adds four slots a step; a slot's parity
        // does not change on the way (4 and 18 are even), so the scale -- 1/32768, negated for the odd slots of odd
        // subbands, which mdct_sub flips before use (src/mdct.c:57-60) -- is one constant per lane
        const int sub = 16 * half + (lane & 15);
        const double scale = ((sub & 1) && (q & 1)) ? -0x1p-15 : 0x1p-15;
        double *const out = sbs + ((size_t) s * G1 * C + ch) * 576 + sub;
        // granule slots below gi_first lie before the stream: the reference's zero-initialised l3_sb_sample
        const int gi_first = (2 * geo.fabs0 + (long) geo.g0 - 1 < 0) ? (int) (1 - (long) geo.g0 - 2 * geo.fabs0) : 0;
        {
            int sgm = sg, gim = gi, qm = q;
            unsigned off = (unsigned) (gi * C * 576 + q * 32);
#pragma unroll 4
            for (int m = 0; m < 16; m++) {
                const double v = L.tr[4 * m + (lane >> 4)][lane & 15] * scale;
                if (sgm < NS) out[off] = gim < gi_first ? 0.0 : v;
                sgm += 4; qm += 4; off += 128;
                if (qm >= 18) { qm -= 18; gim++; off += (unsigned) (C - 1) * 576; }
            }
        }
        if (sb_dbg) { // raw subband samples as filter_subband returns them (parity tests)
            int sgm = sg, gim = gi, qm = q;
            for (int m = 0; m < 16; m++) {
                if (sgm < NS && gim > 0)
                    sb_dbg[(((size_t) s * geo.n_gran + gim - 1) * C + ch) * 576 + qm * 32 + sub] = L.tr[4 * m + (lane >> 4)][lane & 15] * 0x1p-15;
                sgm += 4; qm += 4;
                if (qm >= 18) { qm -= 18; gim++; }
            }
        }
        __syncthreads(); // the other half's results take the same place
    }
}

// k_mdct: one LANE per band.  Lanes 0-31 are the 32 bands of one (stream, channel) -- a track --, lanes 32-63 those of
// the next track; the wavefront walks a run of MDCT_RUN consecutive granules.  A band's 36 inputs, its 26 operand
// groups and its 18 outputs live in the lane's registers; every window value and every transform coefficient is the
// same for all lanes -- scalar loads, SGPR operands of the f64 multiplies --; LDS holds the finished [band][18]
// blocks -- for the alias butterflies' neighbour values and for coalesced stores -- and the previous granule's samples.  (A wavefront per granule with the inputs in LDS, as before, spent two thirds of its
// instructions on LDS traffic and its addresses, and waited on the LDS pipe.)
#define MDCT_RUN 22 // <= 32: lane k of a half holds the block type of the run's granule k
// x: the finished [band][18] blocks of the two tracks.  prev: the band's 18 samples of the granule BEFORE the one being
// transformed, element k of lane l at prev[k][l] (every lane reads and writes its own column only: no barrier, no bank
// conflict).  A granule's samples come from memory ONCE, as "cur", and are parked here on the way for their second use
// as the next granule's "prev" -- until round 3 they were read from memory twice (65 GB of the step's reads instead of
// 36), because 36 more doubles do not fit the registers beside the transform.
struct mdct_out_lds { double x[2][576]; double prev[18][64]; };

// ordered signed sum of windowed inputs: ops[i] = index | 0x80 (subtract / negate)   (src/mdct.c:205-508)
template <int N> MP3MI_DEVFN double mdct_group_reg(const double (&fin)[36], const uint8_t *ops)
{
    double acc = (ops[0] & 0x80) ? -fin[ops[0] & 0x3f] : fin[ops[0] & 0x3f];
#pragma unroll
    for (int i = 1; i < N; i++) acc = (ops[i] & 0x80) ? acc - fin[ops[i] & 0x3f] : acc + fin[ops[i] & 0x3f];
    return acc;
}

// long window (src/mdct.c:199-509): prev = the band's 18 samples of the previous granule (this lane's column of an LDS
// block, stride 64), cur = those of this one (memory, stride 32); park = where cur goes for its second use (the same
// column -- it takes prev's place -- or, for a pass that is not the last one over these inputs, scratch)
MP3MI_DEVFN void mdct_long_reg(const double *prev, const double *cur, double *park, const mp3mi_tables *T, double (&o)[18])
{
    double V[26];
    {
        double fin[36], c[18];
#pragma unroll
        for (int k = 0; k < 18; k++) c[k] = cur[32 * k]; // all in flight together
#pragma unroll
        for (int k = 0; k < 18; k++) {
            fin[k] = T->mdct_win[0][k] * prev[64 * k];
            fin[18 + k] = T->mdct_win[0][18 + k] * c[k];
            park[64 * k] = c[k];
        }
#pragma unroll
        for (int j = 0; j < 9; j++) { V[j] = fin[j] - fin[17 - j]; V[9 + j] = fin[18 + j] + fin[35 - j]; }
#pragma unroll
        for (int c = 0; c < 6; c++) V[18 + c] = mdct_group_reg<6>(fin, MDCT_G_OPS[c]);
#pragma unroll
        for (int c = 0; c < 2; c++) V[24 + c] = mdct_group_reg<18>(fin, MDCT_H_OPS[c]);
    }
    // every output is the ordered sum of its terms V * coefficient
#pragma unroll
    for (int r = 0; r < 12; r++) { // the twelve rows over all 18 pair groups
        const double *cf = T->mdct_vcoef[MDCT_FULL_ROW[r]];
        double sum = V[0] * cf[0];
#pragma unroll
        for (int t = 1; t < 18; t++) sum = sum + V[t] * cf[t];
        o[MDCT_FULL_ROW[r]] = sum;
        if (r & 1) __asm__ volatile("" ::: "memory"); // (two rows of coefficients in flight, not all 216)
    }
#pragma unroll
    for (int r = 0; r < 6; r++) { // the six short rows
        const double *cf = T->mdct_vcoef[MDCT_SMALL_ROW[r]];
        double sum = V[MDCT_SMALL_IDX[r][0]] * cf[0];
#pragma unroll
        for (int t = 1; t < 6; t++)
            if (t < MDCT_SMALL_NT[r]) sum = sum + V[MDCT_SMALL_IDX[r][t]] * cf[t];
        o[MDCT_SMALL_ROW[r]] = sum;
    }
}

// the other block types; bt (1, 2, 3) is the same for all lanes that keep the result
MP3MI_DEVFN void mdct_other_reg(const double *prev, const double *cur, double *park, const mp3mi_tables *T, int bt, double (&o)[18])
{
    double in[36];
#pragma unroll
    for (int k = 0; k < 18; k++) in[18 + k] = cur[32 * k];
#pragma unroll
    for (int k = 0; k < 18; k++) {
        in[k] = prev[64 * k];
        park[64 * k] = in[18 + k];
    }
    if (bt == 2) { // three short transforms, out[3*mm + l]   (src/mdct.c:173-185)
        double w[3][12];
#pragma unroll
        for (int l = 0; l < 3; l++)
#pragma unroll
            for (int k = 0; k < 12; k++) w[l][k] = T->mdct_win[2][k] * in[k + 6 * l + 6];
#pragma unroll
        for (int mm = 0; mm < 6; mm++) {
#pragma unroll
            for (int l = 0; l < 3; l++) {
                double sum = 0.0;
#pragma unroll
                for (int k = 0; k < 12; k++) sum = sum + w[l][k] * T->cos_s[mm][k];
                o[3 * mm + l] = sum;
            }
            __asm__ volatile("" ::: "memory");
        }
    } else { // start / stop windows, plain 36-term sums (src/mdct.c:188-198)
        double w[36];
#pragma unroll
        for (int k = 0; k < 36; k++) w[k] = T->mdct_win[bt][k] * in[k];
#pragma unroll
        for (int m = 0; m < 18; m++) {
            double sum = 0.0;
#pragma unroll
            for (int k = 0; k < 36; k++) sum = sum + w[k] * T->cos_l[m][k];
            o[m] = sum;
            __asm__ volatile("" ::: "memory");
        }
    }
}

__global__ void __launch_bounds__(64, 3) k_mdct(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                const double *__restrict__ sbs, const mp3mi_psy_out *__restrict__ psy,
                                                double *__restrict__ xr_out)
{
    __shared__ mdct_out_lds L;
    const int lane = wave_lane(), band = lane & 31, h = lane >> 5;
    const int C = geo.channels, G = geo.n_gran, NR = (G + MDCT_RUN - 1) / MDCT_RUN, NT = geo.n_streams * C;
    int bid = (int) blockIdx.x;
    const int r = bid % NR;
    const int pair = bid / NR;
    const bool two = 2 * pair + 1 < NT; // (an odd number of tracks: the last wavefront's upper half idles on a copy of the lower one)
    const int tr_lo = 2 * pair, tr_hi = two ? tr_lo + 1 : tr_lo;
    const int g_lo = r * MDCT_RUN, n = G - g_lo < MDCT_RUN ? G - g_lo : MDCT_RUN;
    const size_t rec_lo = ((size_t) (tr_lo / C) * G + g_lo) * C + tr_lo % C, rec_hi = ((size_t) (tr_hi / C) * G + g_lo) * C + tr_hi % C;
    const int tr = h ? tr_hi : tr_lo;
    const int s = tr / C, ch = tr - s * C;
    const size_t rec0 = h ? rec_hi : rec_lo;
    const int btv = band < n ? psy[rec0 + (size_t) band * C].block_type : 0;
    const size_t pitch = (size_t) C * 576;
    const double *blk = sbs + (((size_t) s * (G + 1) + g_lo) * C + ch) * 576 + band; // granule slot g_lo: the one before granule g_lo
    double *prev = &L.prev[0][lane];
#pragma unroll
    for (int k = 0; k < 18; k++) prev[64 * k] = blk[32 * k]; // the run's first "previous granule"
    for (int kk = 0; kk < n; kk++) {
        // (this granule's samples are read where they are used, once, and parked in LDS for the next granule)
        const double *cur = blk + (size_t) (kk + 1) * pitch;
        const int bt0 = wave_readlane_i32(btv, kk), bt1 = wave_readlane_i32(btv, 32 + kk);
        const int bt = h ? bt1 : bt0;
        double o[18];
        const bool mixed = bt0 != bt1; // (rare: the two tracks' block types differ, each half keeps its own transform's result)
        double *own = &L.x[h][band * 18];
#pragma unroll 1
        for (int pass = 0; pass < (mixed ? 2 : 1); pass++) {
            const int v = pass ? bt1 : bt0;
            const mp3mi_tables *Tk = wave_uniform_here(T);
            // the last pass over these inputs parks them in prev's place; the first of two parks into L.x, which is
            // scratch until the pass writes its results there (same layout: [18][64] doubles)
            double *park = (pass == (mixed ? 1 : 0)) ? prev : &L.x[0][0] + lane;
            if (v == 0) mdct_long_reg(prev, cur, park, Tk, o);
            else mdct_other_reg(prev, cur, park, Tk, v, o);
            if (mixed) __syncthreads(); // (every lane is done with L.x as scratch before results go there)
            if (bt == v) {
#pragma unroll
                for (int m = 0; m < 18; m++) own[m] = o[m];
            }
        }
        __syncthreads();
        if (mixed) {
#pragma unroll
            for (int m = 0; m < 18; m++) o[m] = own[m];
        }
        { // alias reduction butterflies between neighbouring bands (src/mdct.c:83-91); not for short blocks
            double dn[8], up[8];
            const mp3mi_tables *Tk = wave_uniform_here(T);
            const double *above = &L.x[h][(band < 31 ? band + 1 : band) * 18], *below = &L.x[h][(band > 0 ? band - 1 : band) * 18];
#pragma unroll
            for (int k = 0; k < 8; k++) { dn[k] = above[k]; up[k] = below[17 - k]; } // xr[band + 1][k], xr[band - 1][17 - k]
            __syncthreads(); // every lane has its neighbours' values: the results may take their place
#pragma unroll
            for (int k = 0; k < 8; k++) {
                const double cs = Tk->cs[k], ca = Tk->ca[k];
                const double bu = o[17 - k] * cs + dn[k] * ca;
                const double bd = o[k] * cs - up[k] * ca;
                if (bt != 2 && band < 31) own[17 - k] = bu;
                if (bt != 2 && band > 0) own[k] = bd;
            }
        }
        __syncthreads();
        // element lane + 64 j of the two [band][18] blocks: j < 9 the lower track's, then the upper one's
        double *out_lo = xr_out + (rec_lo + (size_t) kk * C) * 576, *out_hi = xr_out + (rec_hi + (size_t) kk * C) * 576;
        const double *flat = &L.x[0][0];
#pragma unroll
        for (int j = 0; j < 9; j++) out_lo[lane + 64 * j] = flat[lane + 64 * j];
        if (two) {
#pragma unroll
            for (int j = 0; j < 9; j++) out_hi[lane + 64 * j] = flat[576 + lane + 64 * j];
        }
        __syncthreads(); // the next granule's results take the same place
    }
}

size_t mp3mi_sbs_bytes(const mp3mi_geom &g) { return (size_t) g.n_streams * (size_t) (g.n_gran + 1) * (size_t) g.channels * 576 * sizeof(double); }

void mp3mi_launch_filter(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm, double *sbs, double *sb_dbg, hipStream_t st)
{
    hipLaunchKernelGGL(k_filter, dim3((unsigned) (g.n_streams * g.channels * (((g.n_gran + 1) * 18 + FILT_SLOTS - 1) / FILT_SLOTS))), dim3(64), 0, st, T, g, pcm, sbs, sb_dbg);
}

void mp3mi_launch_mdct(const mp3mi_tables *T, const mp3mi_geom &g, const mp3mi_psy_out *psy, const double *sbs, double *xr, hipStream_t st)
{
    hipLaunchKernelGGL(k_mdct, dim3((unsigned) (((g.n_streams * g.channels + 1) / 2) * ((g.n_gran + MDCT_RUN - 1) / MDCT_RUN))), dim3(64), 0, st, T, g, sbs, psy, xr);
}

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
#include "dmath.h"

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
// (FILT_SAMPLE: a sample is read as ONE sign-extending 16-bit LDS read -- ds_read_i16 --; left to itself the compiler fetches two
// neighbouring samples as a dword and unpacks them with a vector instruction each, 504 per slot in a kernel that is bound by vector
// issue while its LDS pipe idles: profiles/r06_experiments.txt, F9)
#if defined(MP3MI_EMU) || defined(FILT_PAIRED_READS)
typedef const int16_t *filt_pcm_ptr;
#define FILT_SAMPLE(P, i) ((double) (P)[i])
#else
typedef const volatile __attribute__((address_space(3))) int16_t *filt_pcm_ptr; // (an LDS pointer: a volatile generic one would be read through flat loads)
#define FILT_SAMPLE(P, i) ((double) (P)[i])
#endif
template <int I> MP3MI_DEVFN double filt_y(filt_pcm_ptr P, const mp3mi_tables *T)
{
    double acc = FILT_SAMPLE(P, FILT_PAD(511 - I)) * T->enwindow[I];
#pragma unroll
    for (int k = 1; k < 8; k++) acc = acc + FILT_SAMPLE(P, FILT_PAD(511 - I - 64 * k)) * T->enwindow[I + 64 * k];
    return acc;
}

// u[0] = y[16], u[1+j] = y[j] + y[32-j] (j < 16), u[17+j] = y[33+j] - y[63-j] (j < 15)   (src/encode.c:398-408);
// taken in ascending i so that the window taps are read in address order; a + b == b + a bit for bit
template <int I> MP3MI_DEVFN void filt_fold(double (&u)[32], filt_pcm_ptr P, const mp3mi_tables *T)
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
#if defined(MP3MI_EXP_FILT_PRIO) && !defined(MP3MI_EMU) // (experiment F11: the wave priority of the two kernels that end the chain beside k_loop)
    __builtin_amdgcn_s_setprio(MP3MI_EXP_FILT_PRIO);
#endif
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
    filt_fold<0>(u, (filt_pcm_ptr) &L.pcm[34 * lane], T);
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
struct mdct_out_lds { double x[2][576]; double zero[8]; double prev[18][64]; }; // (zero: what an idle chain step of the tail adds)
#define MDCT_ZERO (2 * 576) /* index of zero[0] from x[0][0] */

// ordered signed sum of windowed inputs: ops[i] = index | 0x80 (subtract / negate)   (src/mdct.c:205-508)
template <int N> MP3MI_DEVFN double mdct_group_reg(const double (&fin)[36], const uint8_t *ops)
{
    double acc = (ops[0] & 0x80) ? -fin[ops[0] & 0x3f] : fin[ops[0] & 0x3f];
#pragma unroll
    for (int i = 1; i < N; i++) acc = (ops[i] & 0x80) ? acc - fin[ops[i] & 0x3f] : acc + fin[ops[i] & 0x3f];
    return acc;
}

// long window (src/mdct.c:199-509): prev = the band's 18 samples of the previous granule (this lane's column of an LDS
// block, stride 64), c = those of this one (read by the caller, all in flight together); park = where cur goes for its second use (the same
// column -- it takes prev's place -- or, for a pass that is not the last one over these inputs, scratch)
MP3MI_DEVFN void mdct_long_reg(const double *prev, const double (&c)[18], double *park, const mp3mi_tables *T, double (&o)[18])
{
    double V[26];
    {
        double fin[36];
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
MP3MI_DEVFN void mdct_other_reg(const double *prev, const double (&c)[18], double *park, const mp3mi_tables *T, int bt, double (&o)[18])
{
    double in[36];
#pragma unroll
    for (int k = 0; k < 18; k++) in[18 + k] = c[k];
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

// ---- the stateless head of the iteration loop, computed here while the granule's spectrum is in LDS (k_prep.hip has
//      the reference's walk and the error analysis; this is its parallel form) ----
// calc_xmin (src/loop.c:1085-1118) hands the band energies on as DOUBLES that k_loop compares with noise sums: they are
// summed in the reference's order, every band a chain of its own in a lane of its own (21 long bands, or 12 short
// bands with three windows each; the longest is 102 lines).  Everything else only reaches the loop through an INTEGER:
//   * quantanf_init (src/loop.c:369-402): nint(8 ln sfm) from the total energy -- a 576-term chain in the reference --
//     and the sum of 576 logs.  Here: the lanes' partial sums of the squares added by a butterfly (non-negative terms:
//     any order is within 576 ulp / 2 = 6.4e-14 relative of any other), the product of the mantissas and the sum of the
//     exponents instead of the logs (k_prep.hip), and ln sfm = S / 576 - ln(tot / 576) without the reference's exp:
//     v is within 1e-9 of the reference's, and decided unless it lies within 2e-9 max(1, |v|) of a rounding boundary;
//   * calc_scfsi's (int)(log(x) / log 2) of the total and the band energies and thresholds (src/loop.c:631-667): the
//     exponent field of x, unless x lies within 2^-30 of a power of two (where the reference's quotient of two rounded
//     logarithms decides; an exact power of two is such a case).
// A record with anything undecided is LISTED: k_prep recomputes it the reference's way (probability ~1e-7; all
// records under MP3MI_TEST_PREP_EXACT).
MP3MI_DEVFN int mdct_ilog2_fast(double v, bool &amb)
{
    if (v == 0.0) return 0;
    const long long b = dm_bits(v);
    const int ef = (int) ((b >> 52) & 0x7ff);
    const unsigned long long fr = (unsigned long long) b & 0x000fffffffffffffull;
    if (b < 0 || ef == 0 || ef == 0x7ff || fr < (1ull << 22) || fr > 0x000fffffffffffffull - (1ull << 22)) {
        amb = true;
        return 0;
    }
    const int e = ef - 1023; // log2 v lies strictly between e and e + 1: the cast truncates towards zero
    return e >= 0 ? e : e + 1;
}

#if !defined(MP3MI_EMU)
template <int CTRL> MP3MI_DEVFN double mdct_dpp_f64(double v)
{
    const long long b = dm_bits(v);
    const int lo = (int) b, hi = (int) (b >> 32);
    const unsigned l2 = (unsigned) __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false), h2 = (unsigned) __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return dm_from_bits((long long) (((unsigned long long) h2 << 32) | l2));
}
#endif
#if defined(MP3MI_EMU)
#define MDCT_LDS_PTR(type) type *
#else
#define MDCT_LDS_PTR(type) type __attribute__((address_space(3))) * // (ds_read, not a flat load through a generic pointer)
#endif
// (a function of its own, not inlined: inside k_mdct's loop its presence alone took the kernel from 186 to 222 registers --
// past the 192 that fit beside k_loop)
__device__ __attribute__((noinline)) void mdct_prep_tail(const mp3mi_tables *T, MDCT_LDS_PTR(const double) X, bool wr, int bt, bool any_long, bool any_short, int WL, int WS, bool spoil,
                                                         const mp3mi_psy_out *po, mp3mi_loop_prep *out, mp3mi_prep_fixlist *fix, unsigned rec)
{
    const int band = wave_lane() & 31, h = wave_lane() >> 5;
    // the band-energy chain of this lane: first line and lines, long and short
    int ch_l0 = 0, ch_ln = 0, ch_s0 = 0, ch_sn = 0;
    if (band < 21) { ch_l0 = T->sfb_l[band]; ch_ln = T->sfb_l[band + 1] - ch_l0; }
    if (band < 12) { ch_s0 = 3 * T->sfb_s[band]; ch_sn = T->sfb_s[band + 1] - T->sfb_s[band]; }
    const bool shortb = bt == 2;
    const bool spoiled = spoil && rec % 3 == 0; // MP3MI_TEST_PREP_LIST: this record goes through the list, and what is written here must not survive
    // the psychoacoustic ratios of this lane's band, asked for now and used at the end
    const double *rp = shortb ? &po->ratio_s[band < 12 ? band : 0][0] : &po->ratio_l[band < 21 ? band : 0];
    const double r0 = rp[0], r1 = shortb ? rp[1] : 0.0, r2 = shortb ? rp[2] : 0.0;
    bool amb = false;
    double tot = 0.0, amax = 0.0, prod = 1.0;
    int esum = 0;
    int nzero = 0;
#pragma unroll
    for (int m = 0; m < 18; m++) { // this lane's 18 lines
        const double x = X[band * 18 + m], sq = x * x, ax = __builtin_fabs(x);
        tot = tot + sq;
        amax = __builtin_fmax(amax, ax);
        // sq = m 2^e: the product takes m, the sum e.  A zero line has the mantissa 1 and is counted (its exponent
        // field adds nothing); xr != 0 whose square falls below the normal range is the reference's log's business
        const long long sb = dm_bits(sq);
        const int ef = (int) (sb >> 52);
        amb |= (ef == 0) & (x != 0.0);
        prod = prod * dm_from_bits((sb & 0x000fffffffffffffLL) | 0x3ff0000000000000LL);
        esum += ef;
        nzero += ef == 0;
    }
    esum -= 1023 * (18 - nzero);
    // butterflies within the half (commutative steps: every lane ends with the same values)
#if defined(MP3MI_EMU)
#pragma unroll
    for (int d = 8; d >= 1; d >>= 1) {
        tot = tot + __shfl_xor(tot, d);
        const double oa = __shfl_xor(amax, d);
        amax = oa > amax ? oa : amax;
        prod = prod * __shfl_xor(prod, d);
        esum += __shfl_xor(esum, d);
    }
#else
#define MDCT_TAIL_STEP(CTRL)                                                  \
    {                                                                         \
        tot = tot + mdct_dpp_f64<CTRL>(tot);                                  \
        const double oa = mdct_dpp_f64<CTRL>(amax);                           \
        amax = oa > amax ? oa : amax;                                         \
        prod = prod * mdct_dpp_f64<CTRL>(prod);                               \
        esum += __builtin_amdgcn_update_dpp(esum, esum, CTRL, 0xf, 0xf, false); \
    }
    MDCT_TAIL_STEP(0xB1)  // quad_perm [1, 0, 3, 2]
    MDCT_TAIL_STEP(0x4E)  // quad_perm [2, 3, 0, 1]
    MDCT_TAIL_STEP(0x141) // row_half_mirror
    MDCT_TAIL_STEP(0x140) // row_mirror: the sixteen lanes of a row agree
#undef MDCT_TAIL_STEP
#endif
    { // and the two rows of the half
        tot = tot + __shfl_xor(tot, 16);
        const double oa = __shfl_xor(amax, 16);
        amax = oa > amax ? oa : amax;
        prod = prod * __shfl_xor(prod, 16);
        esum += __shfl_xor(esum, 16);
    }
    // the chains, eight lines at a time: the loads first, all in flight together.  An idle lane -- and a lane past its
    // band's end -- reads the zeros behind the spectrum (MDCT_ZERO) and adds them
    double a0 = 0.0, a1 = 0.0, a2 = 0.0;
    MDCT_LDS_PTR(const double) Z = X + (MDCT_ZERO - 576 * h);
    if (any_long) {
        const int n_l = shortb ? 0 : ch_ln;
        for (int i0 = 0; i0 < WL; i0 += 8) {
            MDCT_LDS_PTR(const double) blk = X + ch_l0 + i0;
            double v[8];
#pragma unroll
            for (int j = 0; j < 8; j++) v[j] = (i0 + j < n_l ? blk : Z)[j];
#pragma unroll
            for (int j = 0; j < 8; j++) a0 = a0 + v[j] * v[j];
        }
    }
    if (any_short) {
        const int n_s = shortb ? ch_sn : 0;
        for (int i0 = 0; i0 < WS; i0 += 2) {
            MDCT_LDS_PTR(const double) blk = X + ch_s0 + 3 * i0;
            double v[6];
#pragma unroll
            for (int j = 0; j < 2; j++) {
                MDCT_LDS_PTR(const double) p = i0 + j < n_s ? blk : Z;
                v[3 * j] = p[3 * j]; v[3 * j + 1] = p[3 * j + 1]; v[3 * j + 2] = p[3 * j + 2];
            }
#pragma unroll
            for (int j = 0; j < 2; j++) {
                a0 = a0 + v[3 * j] * v[3 * j];
                a1 = a1 + v[3 * j + 1] * v[3 * j + 1];
                a2 = a2 + v[3 * j + 2] * v[3 * j + 2];
            }
        }
    }
    if (!shortb) {
        if (band < 21) {
            const double xmin = r0 * a0 / (double) ch_ln;
            const int se = mdct_ilog2_fast(a0, amb), sx = mdct_ilog2_fast(xmin, amb);
            if (wr) { out->xmin[band] = spoiled ? 0.0 : xmin; out->sc_en[band] = se; out->sc_xm[band] = sx; }
        }
    } else if (band < 12) {
        const double cnt = (double) ch_sn;
        if (wr) {
            out->xmin[band * 3 + 0] = spoiled ? 0.0 : r0 * a0 / cnt;
            out->xmin[band * 3 + 1] = r1 * a1 / cnt;
            out->xmin[band * 3 + 2] = r2 * a2 / cnt;
        }
    }
    int tp = 0;
    if (tot != 0.0) {
        // (times 1 / 576 rounded, not divided: one more ulp in a value that is compared with a margin of 2e-9)
        const double A = ((double) esum * 0x1.62e42fefa39efp-1 + dm_log_fast(prod)) * 0x1.c71c71c71c71cp-10, B = tot * 0x1.c71c71c71c71cp-10;
        if (__builtin_fabs(A) < 700.0 && B > 0x1p-1000 && B < 0x1p1000) { // (the reference's exp and quotient stay normal)
            const double v = 8.0 * (A - dm_log_fast(B));
            tp = (v < 0) ? (int) (v - 0.5) : (int) (v + 0.5);
            if (tp < -100) tp = -100;
            const double av = __builtin_fabs(v), fr = av - __builtin_floor(av);
            if (!(__builtin_fabs(fr - 0.5) > 2e-9 * (av > 1.0 ? av : 1.0))) amb = true;
        } else
            amb = true;
    }
    const int en_tot = mdct_ilog2_fast(tot, amb);
    if (spoiled) {
        amb = true;
        tp = 170;
    }
    if (band == 0 && wr) {
        out->q0 = tp - 70;
        out->sc_en_tot = en_tot;
        out->sc_xrmax = (int) amax;
        out->nonzero = (amax != 0.0) ? 1 : 0;
    }
    const unsigned long long am = __ballot(amb);
    if (band == 0 && wr && ((am >> (32 * h)) & 0xffffffffull) != 0) fix->list[atomicAdd(&fix->count, 1u)] = rec;
}

// Diagnostic build only (-DMP3MI_MDCT_PROFILE, tools/mdct_profile.py): a wavefront's cycles per phase, summed over all of them
#if defined(MP3MI_MDCT_PROFILE) && !defined(MP3MI_EMU)
__device__ unsigned long long g_mdct_prof[8];
#define MDCT_PROF_DECL unsigned long long prof_t = __builtin_amdgcn_s_memtime(), prof_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define MDCT_PROF(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); prof_acc[i] += n_ - prof_t; prof_t = n_; } while (0)
#define MDCT_PROF_WAIT asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#define MDCT_PROF_END do { if (wave_lane() == 0) for (int i_ = 0; i_ < 8; i_++) atomicAdd(&g_mdct_prof[i_], prof_acc[i_]); } while (0)
extern "C" void mp3mi_debug_mdct_profile(unsigned long long *out)
{
    unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    hipDeviceSynchronize();
    hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mdct_prof), sizeof(z));
    hipMemcpyToSymbol(HIP_SYMBOL(g_mdct_prof), z, sizeof(z));
}
#else
#define MDCT_PROF_DECL
#define MDCT_PROF(i)
#define MDCT_PROF_WAIT
#define MDCT_PROF_END
#endif
__global__ void __launch_bounds__(64, 3) k_mdct(const mp3mi_tables *__restrict__ T, mp3mi_geom geo,
                                                const double *__restrict__ sbs, const mp3mi_psy_out *__restrict__ psy,
                                                double *__restrict__ xr_out, mp3mi_loop_prep *__restrict__ prep,
                                                mp3mi_prep_fixlist *__restrict__ fix)
{
    __shared__ mdct_out_lds L;
#if defined(MP3MI_EXP_MDCT_PRIO) && !defined(MP3MI_EMU)
    __builtin_amdgcn_s_setprio(MP3MI_EXP_MDCT_PRIO);
#endif
    MDCT_PROF_DECL;
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
    if (lane < 8) L.zero[lane] = 0.0;
    // the longest band-energy chains, long and short (lane & 31 = scalefactor band)
    int WL = 0, WS = 0;
    if (prep) {
        WL = wave_max_i32(band < 21 ? T->sfb_l[band + 1] - T->sfb_l[band] : 0);
        WS = wave_max_i32(band < 12 ? T->sfb_s[band + 1] - T->sfb_s[band] : 0);
    }
    MDCT_PROF(0);
    for (int kk = 0; kk < n; kk++) {
        // (this granule's samples are read where they are used, once, and parked in LDS for the next granule)
        const double *cur = blk + (size_t) (kk + 1) * pitch;
        double c[18];
#pragma unroll
        for (int k = 0; k < 18; k++) c[k] = cur[32 * k]; // all in flight together
        MDCT_PROF_WAIT;
        MDCT_PROF(1);
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
            if (v == 0) mdct_long_reg(prev, c, park, Tk, o);
            else mdct_other_reg(prev, c, park, Tk, v, o);
            if (mixed) __syncthreads(); // (every lane is done with L.x as scratch before results go there)
            if (bt == v) {
#pragma unroll
                for (int m = 0; m < 18; m++) own[m] = o[m];
            }
        }
        MDCT_PROF(2);
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
        MDCT_PROF(3);
        // ---- the loop's stateless head for these two granules (see above); BEFORE the spectrum's stores: a function
        //      begins by waiting for every memory operation in flight ----
        if (prep)
            mdct_prep_tail(T, (MDCT_LDS_PTR(const double)) &L.x[h][0], h == 0 || two, bt, bt0 != 2 || bt1 != 2, bt0 == 2 || bt1 == 2, WL, WS, (geo.test_flags & 32) != 0,
                           &psy[rec0 + (size_t) kk * C], &prep[rec0 + (size_t) kk * C], fix, (unsigned) (rec0 + (size_t) kk * C));
        MDCT_PROF(4);
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
        MDCT_PROF(5);
    }
    MDCT_PROF_END;
}

// The loop's stateless head for a spectrum that does not come out of k_mdct: the drop-in iteration_loop, whose xr is the
// CALLER's (dropin.cpp).  A wavefront per two records: their 576 lines go to LDS in k_mdct's layout and mdct_prep_tail does
// what it does behind a transform -- the band chains in the reference's order, the integers with their margins, the
// undecided records on the list for k_prep.  (Until round 4 the drop-in sent every record through k_prep's 576-step walk
// with a correctly rounded logarithm per line, one LANE per record: 153 us per frame.)
__global__ void __launch_bounds__(64) k_prep_tail(const mp3mi_tables *__restrict__ T, mp3mi_geom geo, const double *__restrict__ xr,
                                                  const mp3mi_psy_out *__restrict__ psy, mp3mi_loop_prep *__restrict__ prep,
                                                  mp3mi_prep_fixlist *__restrict__ fix)
{
    __shared__ mdct_out_lds L;
    const int lane = wave_lane(), band = lane & 31, h = lane >> 5;
    const size_t n_rec = (size_t) geo.n_streams * (size_t) geo.n_gran * (size_t) geo.channels;
    const size_t r0 = 2 * (size_t) blockIdx.x, r1 = r0 + 1 < n_rec ? r0 + 1 : r0;
    const bool two = r0 + 1 < n_rec;
#pragma unroll
    for (int j = 0; j < 9; j++) {
        L.x[0][lane + 64 * j] = xr[r0 * 576 + lane + 64 * j];
        L.x[1][lane + 64 * j] = xr[r1 * 576 + lane + 64 * j];
    }
    if (lane < 8) L.zero[lane] = 0.0;
    const int bt0 = psy[r0].block_type, bt1 = psy[r1].block_type;
    const int WL = wave_max_i32(band < 21 ? T->sfb_l[band + 1] - T->sfb_l[band] : 0);
    const int WS = wave_max_i32(band < 12 ? T->sfb_s[band + 1] - T->sfb_s[band] : 0);
    __syncthreads();
    const size_t rec = h ? r1 : r0;
    mdct_prep_tail(T, (MDCT_LDS_PTR(const double)) &L.x[h][0], h == 0 || two, h ? bt1 : bt0, bt0 != 2 || bt1 != 2, bt0 == 2 || bt1 == 2, WL, WS, false,
                   &psy[rec], &prep[rec], fix, (unsigned) rec);
}

void mp3mi_launch_prep_tail(const mp3mi_tables *T, const mp3mi_geom &g, const double *xr, const mp3mi_psy_out *psy, mp3mi_loop_prep *prep,
                            mp3mi_prep_fixlist *fix, hipStream_t st)
{
    const size_t n_rec = (size_t) g.n_streams * (size_t) g.n_gran * (size_t) g.channels;
    hipLaunchKernelGGL(k_prep_tail, dim3((unsigned) ((n_rec + 1) / 2)), dim3(64), 0, st, T, g, xr, psy, prep, fix);
}

size_t mp3mi_sbs_bytes(const mp3mi_geom &g) { return (size_t) g.n_streams * (size_t) (g.n_gran + 1) * (size_t) g.channels * 576 * sizeof(double); }

void mp3mi_launch_filter(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *pcm, double *sbs, double *sb_dbg, hipStream_t st)
{
    hipLaunchKernelGGL(k_filter, dim3((unsigned) (g.n_streams * g.channels * (((g.n_gran + 1) * 18 + FILT_SLOTS - 1) / FILT_SLOTS))), dim3(64), 0, st, T, g, pcm, sbs, sb_dbg);
}

void mp3mi_launch_mdct(const mp3mi_tables *T, const mp3mi_geom &g, const mp3mi_psy_out *psy, const double *sbs, double *xr,
                       mp3mi_loop_prep *prep, mp3mi_prep_fixlist *fix, hipStream_t st)
{
    hipLaunchKernelGGL(k_mdct, dim3((unsigned) (((g.n_streams * g.channels + 1) / 2) * ((g.n_gran + MDCT_RUN - 1) / MDCT_RUN))), dim3(64), 0, st, T, g, sbs, psy, xr,
                       prep, fix);
}

// Layer III bitstream formatting: header + side information, scalefactors, Huffman code
// words, count1 quadruples, stuffing, and placement of each frame's main data behind its
// back pointer.
//
// Replaces III_format_bitstream / encodeSideInfo / encodeMainData / Huffmancodebits
// (src/l3bitstream.c:67-767), HuffmanCode (src/huffcode.h:16-139), BF_BitstreamFrame /
// WriteMainDataBits (src/formatBitstream.c:52-270) and putbits (src/common.c:1134-1161).
//
// The reference interleaves main data and headers through a queue; because the stream is
// CBR without padding (src/musicin.c:566-581) the result has a closed form: header n sits at
// byte n*frame_bytes, and byte k of the concatenated main data sits in slot k / slot_bytes at
// offset k % slot_bytes behind that slot's header + side info.  Frame n's main data starts at
// main-data offset n*slot_bytes - main_data_begin[n].  Hence every frame formats independently:
// one wavefront per (stream, frame); lanes own code words, a wave prefix sum of the code
// lengths gives each word its bit position, words are OR-ed into an LDS image and the image
// is scattered to the frame's byte positions.
#include "mp3mi_host.h"

struct fmt_lds {
    unsigned words[640];  // main data image, big-endian bit order inside each word
    unsigned si[12];      // header + side info image (<= 36 bytes)
    int sfb_l[23], sfb_s[14];
    unsigned ht_meta[34]; // Huffman table t: first cell | ylen << 16 | linbits << 24 (a code word's cell then is ONE round trip away)
    mp3mi_frame_side side; // the frame's side information: one coalesced read, then every field from here (the kernel reads some
                           // sixty of them, one by one and in dependent steps: from memory that was a round trip each)
};

__device__ static const int FMT_SLEN1[16] = {0, 0, 0, 0, 3, 1, 1, 1, 2, 2, 2, 3, 3, 3, 4, 4};
__device__ static const int FMT_SLEN2[16] = {0, 1, 2, 3, 0, 1, 2, 3, 1, 2, 3, 1, 2, 3, 2, 3};

MP3MI_DEVFN void fmt_or(unsigned *w, unsigned v)
{
#if defined(MP3MI_EMU)
    *w |= v;
#else
    atomicOr(w, v);
#endif
}

// put the low n bits of val at bit position pos (MSB first) of a word image
MP3MI_DEVFN void fmt_put(unsigned *img, int pos, unsigned val, int n)
{
    if (n <= 0) return;
    if (n < 32) val &= (1u << n) - 1u;
    const int wi = pos >> 5, off = pos & 31;
    if (off + n <= 32)
        fmt_or(&img[wi], val << (32 - off - n));
    else {
        const int n2 = off + n - 32;
        fmt_or(&img[wi], val >> n2);
        fmt_or(&img[wi + 1], val << (32 - n2));
    }
}

// exclusive prefix sum over the wave; *total receives the wave sum
MP3MI_DEVFN int fmt_scan(int v, int *total)
{
#if !defined(MP3MI_EMU)
    // inclusive scan inside every row of 16 lanes by four shifts (a lane without a source adds 0), then the totals of row 0
    // into row 1 and of row 2 into row 3 (row_bcast15), then lane 31's -- rows 0 + 1 -- into rows 2 and 3 (row_bcast31): six
    // vector instructions, where six __shfl_up cost an LDS-pipe instruction, an address and a select each
    int incl = v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xf, 0xf, false); // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xf, 0xf, false); // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xf, 0xf, false); // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xf, 0xf, false); // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xa, 0xf, false); // row_bcast15 into rows 1 and 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xc, 0xf, false); // row_bcast31 into rows 2 and 3
    *total = __builtin_amdgcn_readlane(incl, 63);
    return incl - v;
#else
    const int lane = wave_lane();
    int incl = v;
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, (unsigned) d);
        if (lane >= d) incl += o;
    }
    *total = __shfl(incl, 63);
    return incl - v;
#endif
}

// code word(s) of one big-value pair (src/huffcode.h:16-139): code/cbits then ext/xbits
MP3MI_DEVFN void fmt_pair(const mp3mi_tables *T, const unsigned *ht_meta, int t, int x, int y, unsigned *code, int *cbits,
                          unsigned *ext, int *xbits)
{
    *code = 0; *cbits = 0; *ext = 0; *xbits = 0;
    if (t == 0) return;
    unsigned signx = 0, signy = 0;
    if (x < 0) { x = -x; signx = 1; }
    if (y < 0) { y = -y; signy = 1; }
    const unsigned meta = ht_meta[t];
    const int ylen = (int) ((meta >> 16) & 255u), linbits = (int) (meta >> 24), toff = (int) (meta & 0xffffu);
    if (t > 15) {
        unsigned lx = 0, ly = 0, e = 0;
        int xb = 0;
        const int x0 = x, y0 = y;
        if (x > 14) { lx = (unsigned) (x - 15); x = 15; }
        if (y > 14) { ly = (unsigned) (y - 15); y = 15; }
        const int idx = toff + x * ylen + y;
        *code = T->ht_code[idx];
        *cbits = T->ht_len[idx];
        if (x0 > 14) { e |= lx; xb += linbits; }
        if (x0 != 0) { e <<= 1; e |= signx; xb += 1; }
        if (y0 > 14) { e <<= linbits; e |= ly; xb += linbits; }
        if (y0 != 0) { e <<= 1; e |= signy; xb += 1; }
        *ext = e;
        *xbits = xb;
    } else {
        const int idx = toff + x * ylen + y;
        unsigned c = T->ht_code[idx];
        int cb = T->ht_len[idx];
        if (x != 0) { c = (c << 1) | signx; cb += 1; }
        if (y != 0) { c = (c << 1) | signy; cb += 1; }
        *code = c;
        *cbits = cb;
    }
}

// BF_FlushBitstream (src/formatBitstream.c:87-105) writes sum(frameLength - SILength) zero bits over the queued headers in
// words of 32 and ends with WriteMainDataBits(0, bits % 32).  When the main data written so far ends exactly on a slot
// boundary (nothing left in the current slot) with k >= 1 headers still queued and k * slot bits are a multiple of 32, that
// last call has nothing to write but finds BitCount == ThisFrameSize, asks for another header and get_side_info's
// assert( l ) fails (src/formatBitstream.c:225-230, 381-390): the reference dies in its flush.  n_done frames encoded,
// m_end bytes of main data written, slot = main-data bytes per frame.
MP3MI_DEVFN bool fmt_flush_dies(long n_done, long m_end, int slot)
{
    if (n_done <= 0) return false;
    const long written = (m_end + slot - 1) / slot; // headers written so far: one per slot the main data has entered
    const long queued = n_done - written;
    return queued >= 1 && written * slot == m_end && ((queued * (long) slot * 8) % 32) == 0;
}

#define FMT_KERNEL_ARGS const mp3mi_tables *__restrict__ T, mp3mi_geom geo, const int16_t *__restrict__ ix_all,                          \
                        const mp3mi_frame_side *__restrict__ side_all, const int32_t *__restrict__ bits_per_frame,                 \
                        const int32_t *__restrict__ bitrate_index, uint8_t *__restrict__ out, size_t out_stride,                   \
                        uint32_t *__restrict__ out_len, int32_t *__restrict__ loop_state, int loop_state_words,                    \
                        unsigned *__restrict__ voided

// one frame of one stream, by one wavefront (the body of k_format and of k_format_marked)
MP3MI_DEVFN void fmt_frame(fmt_lds &L, FMT_KERNEL_ARGS)
{
    const int lane = wave_lane();
    const int C = geo.channels, G = geo.n_gran;
    const int fl = (int) blockIdx.x % geo.nf, s = (int) blockIdx.x / geo.nf;
    const long n_call = (long) geo.f0 + fl;     // frame index within this call
    const long n_abs = geo.fabs0 + n_call;      // and within the stream: what places the frame in the file
    // ragged batch: this stream's frame count; frames beyond it were not encoded and emit nothing
    const long n_frames_s = geo.n_samples ? ((long) geo.n_samples[s] + 1151) / 1152 : (long) geo.n_frames;
    if (n_call >= n_frames_s) {
        if (n_frames_s == 0 && n_call == 0 && geo.whole_file && wave_lane() == 0) out_len[s] = 0; // no samples, no file body
        return;
    }
    {
        const int32_t *src = (const int32_t *) &side_all[(size_t) s * geo.nf + fl];
        for (int i = lane; i < (int) (sizeof(mp3mi_frame_side) / 4); i += 64) ((int32_t *) &L.side)[i] = src[i];
    }
    const mp3mi_frame_side *sd = &L.side; // (valid behind the barrier below)
    // ... and the frame's quantised values as pairs, a word a lane and step, for all four (granule, channel) records at once:
    // where they are does not depend on the side information, so they travel together with it (pair e = lines 2 e, 2 e + 1)
    unsigned pw[4][5];
#pragma unroll
    for (int gc = 0; gc < 4; gc++) {
        const size_t rec = ((size_t) s * G + (2 * fl + (gc >= C ? 1 : 0))) * C + (gc >= C ? gc - C : gc);
        const unsigned *ixw = (const unsigned *) (ix_all + rec * 576);
#pragma unroll
        for (int j = 0; j < 5; j++) pw[gc][j] = (gc < 2 * C && 64 * j + lane < 288) ? ixw[64 * j + lane] : 0u;
    }
    const int frame_bytes = bits_per_frame[s] / 8;
    const int crc_bits = geo.crc ? 16 : 0; // the reference's CRC word for Layer III is always 0 (src/l3bitstream.c:312, 338-342)
    const int si_bytes = (32 + crc_bits + (C == 2 ? 256 : 136)) / 8;
    const int slot = frame_bytes - si_bytes;
    // byte 0 of the stream's output row is file position out_base[s] (streaming: what earlier calls delivered)
    uint8_t *dst = out + (size_t) s * out_stride - (geo.out_base ? (size_t) geo.out_base[s] : (size_t) 0);

    for (int i = lane; i < 640; i += 64) L.words[i] = 0;
    if (lane < 12) L.si[lane] = 0;
    if (lane < 23) L.sfb_l[lane] = T->sfb_l[lane];
    if (lane < 14) L.sfb_s[lane] = T->sfb_s[lane];
    if (lane < 34) L.ht_meta[lane] = (unsigned) T->ht_off[lane] | ((unsigned) T->ht_ylen[lane] << 16) | ((unsigned) T->ht_linbits[lane] << 24);
    __syncthreads();

    // ---- header and side information (src/l3bitstream.c:314-458) ----
    if (lane == 4) {
        int pos = 0;
        fmt_put(L.si, pos, 0xfff, 12); pos += 12;
        fmt_put(L.si, pos, 1, 1); pos += 1;                       // MPEG-1
        fmt_put(L.si, pos, 1, 2); pos += 2;                       // 4 - layer
        fmt_put(L.si, pos, geo.crc ? 0u : 1u, 1); pos += 1;       // protection bit: set = no CRC
        fmt_put(L.si, pos, (unsigned) bitrate_index[s], 4); pos += 4;
        fmt_put(L.si, pos, (unsigned) T->rate_idx, 2); pos += 2;
        pos += 2;                                                 // padding 0, extension 0
        fmt_put(L.si, pos, (unsigned) geo.hdr_mode, 2); pos += 2; // mode
        fmt_put(L.si, pos, (unsigned) geo.hdr_flags, 6); pos += 6; // mode_ext, copyright, original, emphasis
        pos += crc_bits;                                          // CRC word: zeros
        fmt_put(L.si, pos, (unsigned) sd->main_data_begin, 9); pos += 9;
        pos += (C == 2) ? 3 : 5;                                  // private_bits 0
        for (int ch = 0; ch < C; ch++)
            for (int b = 0; b < 4; b++) { fmt_put(L.si, pos, (unsigned) sd->scfsi[ch][b], 1); pos += 1; }
    }
    if (lane < 2 * C) {
        const int gr = lane / C, ch = lane % C;
        const mp3mi_gr_side *g = &sd->gr[gr][ch];
        int pos = 32 + crc_bits + 9 + ((C == 2) ? 3 : 5) + 4 * C + 59 * lane;
        fmt_put(L.si, pos, (unsigned) g->part2_3_length, 12); pos += 12;
        fmt_put(L.si, pos, (unsigned) g->big_values, 9); pos += 9;
        fmt_put(L.si, pos, (unsigned) g->global_gain, 8); pos += 8;
        fmt_put(L.si, pos, (unsigned) g->scalefac_compress, 4); pos += 4;
        fmt_put(L.si, pos, (unsigned) g->window_switching_flag, 1); pos += 1;
        if (g->window_switching_flag) {
            fmt_put(L.si, pos, (unsigned) g->block_type, 2); pos += 2;
            pos += 1; // mixed_block_flag 0
            fmt_put(L.si, pos, (unsigned) g->table_select[0], 5); pos += 5;
            fmt_put(L.si, pos, (unsigned) g->table_select[1], 5); pos += 5;
            pos += 9; // subblock_gain 0
        } else {
            for (int r = 0; r < 3; r++) { fmt_put(L.si, pos, (unsigned) g->table_select[r], 5); pos += 5; }
            fmt_put(L.si, pos, (unsigned) g->region0_count, 4); pos += 4;
            fmt_put(L.si, pos, (unsigned) g->region1_count, 3); pos += 3;
        }
        fmt_put(L.si, pos, (unsigned) g->preflag, 1); pos += 1;
        pos += 1; // scalefac_scale 0
        fmt_put(L.si, pos, (unsigned) g->count1table_select, 1);
    }

    // ---- main data (src/l3bitstream.c:174-310, 516-716) ----
    int gpos = 0; // bit position of the current granule-channel in the image
#pragma unroll
    for (int gc = 0; gc < 4; gc++) {
        if (gc < 2 * C) {
            const int gr = gc >= C ? 1 : 0, ch = gc >= C ? gc - C : gc;
            const mp3mi_gr_side *g = &sd->gr[gr][ch];
            const size_t rec = ((size_t) s * G + (2 * fl + gr)) * C + ch;
            const int16_t *ix = ix_all + rec * 576;
            const bool shortb = g->window_switching_flag && g->block_type == 2;
            const int slen1 = FMT_SLEN1[g->scalefac_compress], slen2 = FMT_SLEN2[g->scalefac_compress];
            int pos = gpos, total;
            { // scalefactors
                int n = 0;
                if (shortb) { if (lane < 36) n = (lane < 18) ? slen1 : slen2; }
                else if (lane < 21) {
                    const int band = (lane < 6) ? 0 : (lane < 11 ? 1 : (lane < 16 ? 2 : 3));
                    if (gr == 0 || sd->scfsi[ch][band] == 0) n = (lane < 11) ? slen1 : slen2;
                }
                const int ofs = fmt_scan(n, &total);
                if (n) fmt_put(L.words, pos + ofs, (unsigned) g->scalefac[lane], n);
                pos += total;
            }
            // Code words: EVERYTHING THAT COMES OUT OF MEMORY FIRST -- the pairs and quadruples of every step of the granule and the
            // table entries they select (two dependent round trips a step) are independent of the bit positions, so all of the
            // granule's (at most five + three) steps are asked for together; the positions -- a scan per step -- and the puts follow.
            // (One step at a time, load - look up - scan - put, the kernel spent 77 % of its wavefronts' time waiting, and its
            // 315 000 one-wavefront workgroups kept the transforms' and k_mdct's out of the CUs for 3.4 ms between two k_loop launches.)
            const int bigvalues = g->big_values * 2;
            const int r1s = (shortb || !bigvalues) ? 0 : L.sfb_l[g->region0_count + 1];
            const int r2s = (shortb || !bigvalues) ? 0 : L.sfb_l[g->region0_count + g->region1_count + 2];
            // (a granule has 576 lines: at most 288 pairs -- five steps of 64 -- and 144 quadruples -- three)
            const int npairs = bigvalues ? (shortb ? 288 : (bigvalues / 2 < 288 ? bigvalues / 2 : 288)) : 0;
            const int nquad = g->count1 < 144 ? g->count1 : 144;
            const int ts0 = g->table_select[0], ts1 = g->table_select[1], ts2 = g->table_select[2];
            const int toff = (int) (L.ht_meta[32 + g->count1table_select] & 0xffffu);
            unsigned code[5], ext[5], qval[3];
            int cb[5], xb[5], qnb[3];
#pragma unroll
            for (int j = 0; j < 5; j++) {
                code[j] = 0; ext[j] = 0; cb[j] = 0; xb[j] = 0;
                if (64 * j < npairs) {
                    const int e = 64 * j + lane;
                    if (e < npairs) {
                        int x, y, t;
                        if (shortb) { // sfb -> window -> line order (src/l3bitstream.c:556-579)
                            int sfb = 0;
                            while (3 * L.sfb_s[sfb + 1] / 2 <= e) sfb++;
                            const int start = L.sfb_s[sfb], half = (L.sfb_s[sfb + 1] - start) / 2;
                            const int r = e - 3 * start / 2, w = r / half, line = start + 2 * (r - w * half);
                            x = ix[line * 3 + w];
                            y = ix[(line + 1) * 3 + w];
                            t = (start < 12) ? ts0 : ts1;
                        } else {
                            const int i = 2 * e;
                            x = (int) (int16_t) (pw[gc][j] & 0xffffu);
                            y = (int) (int16_t) (pw[gc][j] >> 16);
                            t = (i < r1s) ? ts0 : (i < r2s ? ts1 : ts2);
                        }
                        fmt_pair(T, L.ht_meta, t, x, y, &code[j], &cb[j], &ext[j], &xb[j]);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 3; j++) { // count1 quadruples (src/l3bitstream.c:727-767): at most 144
                qval[j] = 0; qnb[j] = 0;
                if (64 * j < nquad) {
                    const int qd = 64 * j + lane;
                    if (qd < nquad) {
                        const int i = bigvalues + 4 * qd;
                        int q[4];
                        unsigned sg[4];
                        for (int k = 0; k < 4; k++) {
                            q[k] = ix[i + k];
                            if (q[k] > 0) sg[k] = 0; else { q[k] = -q[k]; sg[k] = 1; }
                        }
                        const int p = q[0] + (q[1] << 1) + (q[2] << 2) + (q[3] << 3);
                        unsigned val = T->ht_code[toff + p];
                        int nb = T->ht_len[toff + p];
                        for (int k = 0; k < 4; k++)
                            if (q[k]) { val = (val << 1) | sg[k]; nb += 1; }
                        qval[j] = val;
                        qnb[j] = nb;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 5; j++) {
                if (64 * j < npairs) {
                    const int ofs = fmt_scan(cb[j] + xb[j], &total);
                    if (cb[j]) fmt_put(L.words, pos + ofs, code[j], cb[j]);
                    if (xb[j]) fmt_put(L.words, pos + ofs + cb[j], ext[j], xb[j]);
                    pos += total;
                }
            }
#pragma unroll
            for (int j = 0; j < 3; j++) {
                if (64 * j < nquad) {
                    const int ofs = fmt_scan(qnb[j], &total);
                    if (qnb[j]) fmt_put(L.words, pos + ofs, qval[j], qnb[j]);
                    pos += total;
                }
            }
            { // stuffing with ones up to part2_3_length (src/l3bitstream.c:695-710)
                const int endpos = gpos + g->part2_3_length;
                for (int p = pos + lane * 32; p < endpos; p += 64 * 32) {
                    const int n = (endpos - p < 32) ? endpos - p : 32;
                    fmt_put(L.words, p, 0xffffffffu, n);
                }
                gpos = endpos;
            }
        }
    }
    gpos += sd->resvDrain; // zeros (src/l3bitstream.c:492-509)
    __syncthreads();

    // ---- scatter: header + side info at n*frame_bytes, main data behind the back pointer ----
    const size_t hdr = (size_t) n_abs * (size_t) frame_bytes;
    if (lane < si_bytes) dst[hdr + lane] = (uint8_t) (L.si[lane >> 2] >> (24 - 8 * (lane & 3)));
    // main-data byte m = n_abs * slot - main_data_begin + k lives in frame m / slot at offset m % slot behind that
    // frame's side info.  main_data_begin < 512, so the quotient and remainder of the first byte come from small
    // 32-bit numbers, and every further byte needs one 32-bit division instead of two 64-bit ones.
    const unsigned mdb = (unsigned) sd->main_data_begin, uslot = (unsigned) slot;
    const unsigned back = (mdb + uslot - 1u) / uslot;          // frames the data reaches back
    const long q0 = (long) n_abs - (long) back;                // frame of the first byte (>= 0: the reservoir starts empty)
    const unsigned r0 = back * uslot - mdb;                    // its offset
    const int nbytes = gpos / 8;
    for (int k = lane; k < nbytes; k += 64) {
        const unsigned r = r0 + (unsigned) k, dq = r / uslot, rem = r - dq * uslot;
        const size_t phys = (size_t) (q0 + (long) dq) * (size_t) frame_bytes + (size_t) si_bytes + (size_t) rem;
        dst[phys] = (uint8_t) (L.words[k >> 2] >> (24 - 8 * (k & 3)));
    }
    if (geo.whole_file && n_call == n_frames_s - 1 && lane == 0) {
        // file length (src/formatBitstream.c:87-120 + src/common.c:843-868, 968): the flush stops
        // short of the last slot by what the current slot still has free, and close writes the
        // byte under construction as well
        const long mend = (long) n_abs * (long) slot - (long) mdb + nbytes; // once per stream
        const long rem = ((mend + slot - 1) / slot) * slot - mend;
        uint32_t len = (uint32_t) (n_frames_s * frame_bytes - rem + 1);
        if (loop_state) {
            int32_t *st = &loop_state[(size_t) s * loop_state_words + (loop_state_words - 1)];
            if (*st == 0 && fmt_flush_dies(n_frames_s, mend, slot)) *st = MP3MI_DEV_STATUS(MP3MI_DEV_ABORT_FLUSH_SLOT, n_frames_s);
            if (*st != 0) { // the reference died on this stream: there is no file
                len = 0;
                if (voided) atomicAdd(voided, 1u);
            }
        }
        out_len[s] = len;
    }
}


__global__ void __launch_bounds__(64) k_format(FMT_KERNEL_ARGS)
{
    __shared__ fmt_lds L;
    fmt_frame(L, T, geo, ix_all, side_all, bits_per_frame, bitrate_index, out, out_stride, out_len, loop_state, loop_state_words, voided);
}

// The drop-in symbols' formatter (dropin.cpp: one frame, one workgroup): it tells the spinning host that the kernel BEFORE it on
// the stream has finished (k_loop: a kernel starts when the one before it has ended and its stores are visible) and, at its own
// end, that the frame's bytes are in place -- two stores to host-mapped memory instead of two one-thread kernels between and behind
// the two (4-5 us of dispatch latency each).
__global__ void __launch_bounds__(64) k_format_marked(FMT_KERNEL_ARGS, volatile unsigned *flag, unsigned seq_before, unsigned seq_done,
                                                      unsigned *__restrict__ ix_host, unsigned *__restrict__ side_host)
{
    __shared__ fmt_lds L;
    // k_loop left the frame's values and side information in device memory (this kernel reads them many times over, in dependent
    // steps: from host-mapped memory that was most of its time); the host's copies are written here, before the host is told
    for (int i = (int) threadIdx.x; i < geo.channels * 2 * 576 / 2; i += 64) ix_host[i] = ((const unsigned *) ix_all)[i];
    for (int i = (int) threadIdx.x; i < (int) (sizeof(mp3mi_frame_side) / 4); i += 64) side_host[i] = ((const unsigned *) side_all)[i];
    __threadfence_system();
    if (threadIdx.x == 0) *flag = seq_before;
    fmt_frame(L, T, geo, ix_all, side_all, bits_per_frame, bitrate_index, out, out_stride, out_len, loop_state, loop_state_words, voided);
    __threadfence_system(); // (every lane: its stores to the host's window first)
    if (threadIdx.x == 0) *flag = seq_done;
}

// ---- streaming (mp3mi_batch_encode_next / mp3mi_batch_flush) ----
// The file bytes of a stream become final in order: once m bytes of main data have been written, everything up
// to the physical position of main-data byte m - 1 is final (the reference emits exactly these, src/formatBitstream.c
// :218-247); the rest of that slot and the headers behind it still wait for later frames' main data.  A call
// therefore delivers the bytes [final_before, final_after) of every stream's file and keeps [final_after, end of
// the call's last frame) in the stream's carry buffer; the next call starts its output row with them.
MP3MI_DEVFN long fmt_final_upto(long m, int slot, int frame_bytes, int si_bytes)
{
    if (m == 0) return 0;
    return ((m - 1) / slot) * (long) frame_bytes + si_bytes + ((m - 1) % slot) + 1;
}

__global__ void __launch_bounds__(64) k_carry_in(const uint8_t *__restrict__ carry, const int32_t *__restrict__ carry_len,
                                                 uint8_t *__restrict__ out, size_t out_stride)
{
    const int s = (int) blockIdx.x, n = carry_len[s];
    for (int i = (int) threadIdx.x; i < n; i += 64) out[(size_t) s * out_stride + i] = carry[(size_t) s * MP3MI_CARRY_BYTES + i];
}

// After the call's frames are formatted (flush = 0): how much of the row is final, what stays in the carry.
// flush = 1 (III_FlushBitstream + close_bit_stream_w, src/formatBitstream.c:87-120, src/common.c:843-868, 968):
// the carry goes out up to where the last main data ends, plus the byte under construction.
__global__ void __launch_bounds__(64) k_stream_tail(mp3mi_geom geo, int flush, int32_t *__restrict__ loop_state, int loop_state_words,
                                                    const int32_t *__restrict__ bits_per_frame, uint8_t *__restrict__ out, size_t out_stride,
                                                    int64_t *__restrict__ out_base, uint8_t *__restrict__ carry,
                                                    int32_t *__restrict__ carry_len, uint32_t *__restrict__ out_len,
                                                    unsigned *__restrict__ voided)
{
    const int s = (int) blockIdx.x, lane = (int) threadIdx.x;
    const int C = geo.channels;
    const int frame_bytes = bits_per_frame[s] / 8, si_bytes = (32 + (geo.crc ? 16 : 0) + (C == 2 ? 256 : 136)) / 8, slot = frame_bytes - si_bytes;
    const long n_done = geo.fabs0 + (flush ? 0 : geo.n_frames); // frames of the stream encoded so far
    const long resv_bytes = loop_state[(size_t) s * loop_state_words] / 8; // ResvSize / 8 = the next frame's main_data_begin
    const long m_end = n_done * slot - resv_bytes;                 // main data written so far
    const long base = out_base[s];
    uint8_t *row = out + (size_t) s * out_stride;
    uint8_t *cr = carry + (size_t) s * MP3MI_CARRY_BYTES;
    if (!flush) {
        const long fin = fmt_final_upto(m_end, slot, frame_bytes, si_bytes), end = n_done * frame_bytes;
        const int keep = (int) (end - fin); // <= 511 bytes of open slots plus the headers in between
        for (int i = lane; i < keep && i < MP3MI_CARRY_BYTES; i += 64) cr[i] = row[fin - base + i];
        if (lane == 0) {
            // (a stream the reference died on delivers nothing more; mp3mi_batch_stream_status says why.  The sync after
            // the call in which it happened reports it -- once: MP3MI_DEV_ABORT_REPORTED marks the word)
            int32_t *st = &loop_state[(size_t) s * loop_state_words + (loop_state_words - 1)];
            if (*st != 0 && !(*st & MP3MI_DEV_ABORT_REPORTED)) {
                *st |= MP3MI_DEV_ABORT_REPORTED;
                if (voided) atomicAdd(voided, 1u);
            }
            out_len[s] = *st ? 0u : (uint32_t) (fin - base);
            carry_len[s] = keep < MP3MI_CARRY_BYTES ? keep : MP3MI_CARRY_BYTES;
            out_base[s] = fin;
        }
    } else {
        long total = 0;
        if (n_done > 0) {
            const long rem = ((m_end + slot - 1) / slot) * slot - m_end;
            total = n_done * frame_bytes - rem + 1;
        }
        const int n = (int) (total - base), have = carry_len[s];
        for (int i = lane; i < n; i += 64) row[i] = i < have ? cr[i] : (uint8_t) 0;
        if (lane == 0) {
            int32_t *st = &loop_state[(size_t) s * loop_state_words + (loop_state_words - 1)];
            if (*st == 0 && fmt_flush_dies(n_done, m_end, slot)) *st = MP3MI_DEV_STATUS(MP3MI_DEV_ABORT_FLUSH_SLOT, n_done);
            if (*st != 0 && !(*st & MP3MI_DEV_ABORT_REPORTED)) { // (an earlier call of the stream may have reported it already)
                *st |= MP3MI_DEV_ABORT_REPORTED;
                if (voided) atomicAdd(voided, 1u);
            }
            out_len[s] = *st ? 0u : (uint32_t) (n > 0 ? n : 0);
            carry_len[s] = 0;
            out_base[s] = total;
        }
    }
}

__global__ void __launch_bounds__(256) k_status_gather(int n_streams, const int32_t *__restrict__ loop_state, int loop_state_words, int32_t *__restrict__ status)
{
    const int s = (int) (blockIdx.x * 256 + threadIdx.x);
    if (s < n_streams) status[s] = loop_state[(size_t) s * loop_state_words + (loop_state_words - 1)] & ~MP3MI_DEV_ABORT_REPORTED;
}

void mp3mi_launch_status_gather(int n_streams, const int32_t *loop_state, int loop_state_words, int32_t *status, hipStream_t st)
{
    hipLaunchKernelGGL(k_status_gather, dim3((unsigned) ((n_streams + 255) / 256)), dim3(256), 0, st, n_streams, loop_state, loop_state_words, status);
}

// the last MP3MI_PCM_HIST samples of the call (a frame has 1152 > MP3MI_PCM_HIST) are the next call's history
__global__ void __launch_bounds__(256) k_hist_save(mp3mi_geom geo, const int16_t *__restrict__ pcm, int16_t *__restrict__ hist)
{
    const int s = (int) blockIdx.x, C = geo.channels;
    const size_t n_call = (size_t) geo.n_frames * 1152;
    const int16_t *src = pcm + ((size_t) s * n_call + (n_call - MP3MI_PCM_HIST)) * C;
    int16_t *dst = hist + (size_t) s * MP3MI_PCM_HIST * C;
    for (int i = (int) threadIdx.x; i < MP3MI_PCM_HIST * C; i += 256) dst[i] = src[i];
}

void mp3mi_launch_carry_in(int n_streams, const uint8_t *carry, const int32_t *carry_len, uint8_t *out, size_t out_stride, hipStream_t st)
{
    hipLaunchKernelGGL(k_carry_in, dim3((unsigned) n_streams), dim3(64), 0, st, carry, carry_len, out, out_stride);
}

void mp3mi_launch_stream_tail(const mp3mi_geom &g, int flush, int32_t *loop_state, int loop_state_words, const int32_t *bits_per_frame,
                              uint8_t *out, size_t out_stride, int64_t *out_base, uint8_t *carry, int32_t *carry_len, uint32_t *out_len,
                              unsigned *voided, hipStream_t st)
{
    hipLaunchKernelGGL(k_stream_tail, dim3((unsigned) g.n_streams), dim3(64), 0, st, g, flush, loop_state, loop_state_words, bits_per_frame,
                       out, out_stride, out_base, carry, carry_len, out_len, voided);
}

void mp3mi_launch_hist_save(const mp3mi_geom &g, const int16_t *pcm, int16_t *hist, hipStream_t st)
{
    hipLaunchKernelGGL(k_hist_save, dim3((unsigned) g.n_streams), dim3(256), 0, st, g, pcm, hist);
}

void mp3mi_launch_format(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *ix, const mp3mi_frame_side *side,
                         const int32_t *bits_per_frame, const int32_t *bitrate_index, uint8_t *out,
                         size_t out_stride, uint32_t *out_len, int32_t *loop_state, int loop_state_words, unsigned *voided, hipStream_t st)
{
    const unsigned grid = (unsigned) (g.n_streams * g.nf);
    hipLaunchKernelGGL(k_format, dim3(grid), dim3(64), 0, st, T, g, ix, side, bits_per_frame, bitrate_index, out,
                       out_stride, out_len, loop_state, loop_state_words, voided);
}

void mp3mi_launch_format_marked(const mp3mi_tables *T, const mp3mi_geom &g, const int16_t *ix, const mp3mi_frame_side *side,
                                const int32_t *bits_per_frame, const int32_t *bitrate_index, uint8_t *out, size_t out_stride, uint32_t *out_len,
                                unsigned *flag, unsigned seq_before, unsigned seq_done, int16_t *ix_host, mp3mi_frame_side *side_host, hipStream_t st)
{
    hipLaunchKernelGGL(k_format_marked, dim3(1), dim3(64), 0, st, T, g, ix, side, bits_per_frame, bitrate_index, out, out_stride, out_len,
                       (int32_t *) NULL, 0, (unsigned *) NULL, (volatile unsigned *) flag, seq_before, seq_done, (unsigned *) ix_host, (unsigned *) side_host);
}

"""Compare the stage seams of a batch run (product or emulated) with oracle stage dumps."""
import ctypes

import numpy as np

from mp3common import GI, PREP_DT, PSY_DT, SIDE_DT


def run_batch_with_stages(lib, pcm, rate, channels, kbps, n_frames, mode=None, expect_abort=False, flags=0):
    """pcm: int16 [S, n_frames*1152*channels]; mode: None or the driver's option string as oracle/ref_harness.c
    takes it (the -m letter s / d / m, then e / c / o for -e / -c / -o).  Returns (bytes per stream, stages dict);
    with expect_abort (inputs the reference dies on) mp3mi_batch_sync must say so and the stages dict carries the
    per-stream status under "status"."""
    L = lib.lib
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    S = pcm.shape[0]
    b = ctypes.c_void_p()
    karr = None if np.isscalar(kbps) else np.ascontiguousarray(kbps, dtype=np.int32)
    rc = L.mp3mi_batch_create(ctypes.byref(b), S, rate, channels, karr.ctypes.data if karr is not None else None,
                              int(kbps) if karr is None else 0, n_frames)
    assert rc == 0, "mp3mi_batch_create -> %d" % rc
    want_sync = -6 if expect_abort else 0  # MP3MI_ERR_REFERENCE_ABORT
    try:
        if mode:
            assert L.mp3mi_batch_set_mode(b, {"s": 0, "d": 2, "m": 3}[mode[0]]) == 0
            assert L.mp3mi_batch_set_error_protection(b, 1 if "e" in mode[1:] else 0) == 0
            assert L.mp3mi_batch_set_header(b, 1 if "c" in mode[1:] else 0, 1 if "o" in mode[1:] else 0, 0) == 0
        L.mp3mi_batch_debug_enable(b, 1)
        assert L.mp3mi_batch_set_test_flags(b, flags) == 0
        stride = L.mp3mi_batch_out_stride(b, n_frames)
        is_emu = b"emulator" in L.mp3mi_version()
        if is_emu:
            out = np.zeros((S, stride), np.uint8)
            lens = np.zeros(S, np.uint32)
            rc = L.mp3mi_batch_encode(b, pcm.ctypes.data, n_frames, out.ctypes.data, stride, lens.ctypes.data)
            assert rc == 0
            rc = L.mp3mi_batch_sync(b)
            assert rc == want_sync, "mp3mi_batch_sync -> %d" % rc
        else:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
            hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            hip.hipFree.argtypes = [ctypes.c_void_p]
            dp, do, dl = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_void_p()
            assert hip.hipMalloc(ctypes.byref(dp), pcm.nbytes) == 0
            assert hip.hipMalloc(ctypes.byref(do), S * stride) == 0
            assert hip.hipMalloc(ctypes.byref(dl), 4 * S) == 0
            assert hip.hipMemcpy(dp, pcm.ctypes.data, pcm.nbytes, 1) == 0
            rc = L.mp3mi_batch_encode(b, dp, n_frames, do, stride, dl)
            assert rc == 0, rc
            rc = L.mp3mi_batch_sync(b)
            assert rc == want_sync, "mp3mi_batch_sync -> %d" % rc
            out = np.zeros((S, stride), np.uint8)
            lens = np.zeros(S, np.uint32)
            assert hip.hipMemcpy(out.ctypes.data, do, S * stride, 2) == 0
            assert hip.hipMemcpy(lens.ctypes.data, dl, 4 * S, 2) == 0
            for p in (dp, do, dl):
                hip.hipFree(p)
        G = 2 * n_frames
        st = {}
        st["psy"] = np.zeros((S, G, channels), PSY_DT)
        st["xr"] = np.zeros((S, G, channels, 576))
        st["ix"] = np.zeros((S, G, channels, 576), np.int16)
        st["side"] = np.zeros((S, n_frames), SIDE_DT)
        st["sb"] = np.zeros((S, G, channels, 18, 32))
        st["prep"] = np.zeros((S, G, channels), PREP_DT)
        # the psychoacoustic transforms' outputs (oracle/fft_seam.h): long energies in rows of 544, short energies, raw lines
        st["energy_l"] = np.zeros((S, G, channels, 544), np.float32)
        st["energy_s"] = np.zeros((S, G, channels, 3, 129), np.float32)
        st["fft_bins"] = np.zeros((S, G, channels, 312), np.float32)
        for what, key in ((0, "psy"), (1, "xr"), (2, "ix"), (3, "side"), (4, "sb"), (5, "prep"), (6, "energy_l"), (7, "energy_s"), (8, "fft_bins")):
            n = L.mp3mi_batch_debug_fetch(b, what, st[key].ctypes.data, st[key].nbytes)
            assert n == st[key].nbytes, (key, n, st[key].nbytes)
        st["status"] = np.zeros(S, np.int32)
        assert L.mp3mi_batch_stream_status(b, st["status"].ctypes.data) >= 0
        return [out[s, :lens[s]].tobytes() for s in range(S)], st
    finally:
        L.mp3mi_batch_destroy(b)


def compare_stages(st, s, dumps, channels, frames=None):
    """dumps[k] is the reference's / the oracle's record of frame frames[k] (default: k).  Returns a list of mismatch
    descriptions (empty = identical)."""
    bad = []

    def chk(where, name, a, b):
        if not np.array_equal(np.asarray(a), np.asarray(b)):
            bad.append("%s %s" % (where, name))

    for k, f in enumerate(frames if frames is not None else range(len(dumps))):
        d = dumps[k]
        for gr in range(2):
            for c in range(channels):
                g = 2 * f + gr
                w = "stream %d frame %d gr %d ch %d" % (s, f, gr, c)
                p = st["psy"][s, g, c]
                chk(w, "pe", p["pe"], d["pe"][gr, c])
                chk(w, "block_type", p["block_type"], d["psy_bt"][gr, c])
                chk(w, "ratio_l", p["ratio_l"], d["ratio_l"][gr, c])
                chk(w, "ratio_s", p["ratio_s"], d["ratio_s"][gr, c])
                chk(w, "subband samples", st["sb"][s, g, c], d["sb"][c, gr])
                chk(w, "xr", st["xr"][s, g, c], d["xr"][gr, c])
                sg, gi = st["side"][s, f]["gr"][gr, c], d["gi"][gr, c]
                # the lines the side information declares (2 * big_values + 4 * count1): an all-zero granule skips the
                # search in the reference and leaves the PREVIOUS frame's values in its static l3_enc (src/loop.c:346-349),
                # which nothing reads; the product hands zeros on
                coded = 2 * int(gi[GI["big_values"]]) + 4 * int(gi[GI["count1"]])
                chk(w, "|ix|", np.abs(st["ix"][s, g, c].astype(np.int32))[:coded], d["l3_enc"][gr, c][:coded])
                for nm in ("part2_3_length", "big_values", "count1", "global_gain", "scalefac_compress",
                           "window_switching_flag", "block_type", "region0_count", "region1_count", "preflag",
                           "count1table_select", "part2_length"):
                    chk(w, nm, sg[nm], gi[GI[nm]])
                chk(w, "table_select", sg["table_select"], gi[8:11])
                if sg["window_switching_flag"] and sg["block_type"] == 2:
                    chk(w, "scalefac_s", sg["scalefac"][:36].reshape(12, 3), d["scalefac_s"][gr, c][:12])
                else:
                    chk(w, "scalefac_l", sg["scalefac"][:21], d["scalefac_l"][gr, c][:21])
        w = "stream %d frame %d" % (s, f)
        chk(w, "main_data_begin", st["side"][s, f]["main_data_begin"], d["main_data_begin"])
        chk(w, "resvDrain", st["side"][s, f]["resvDrain"], d["resvDrain"])
        chk(w, "scfsi", st["side"][s, f]["scfsi"][:channels], d["scfsi"][:channels])
    return bad


def compare_prep_records(a, b, psy):
    """The loop's stateless head (mp3mi_loop_prep) of two runs, field by field and bit for bit, over the fields a
    record of its block type defines: long blocks xmin[0..20], sc_en, sc_xm; short blocks xmin[0..35]; all the
    four scalars.  Returns the number of records compared."""
    assert a.shape == b.shape == psy.shape
    short = psy["block_type"] == 2
    for name in ("q0", "sc_en_tot", "sc_xrmax", "nonzero"):
        bad = np.argwhere(a[name] != b[name])
        assert bad.size == 0, (name, bad[:4], a[name][tuple(bad[0])], b[name][tuple(bad[0])])
    xa, xb = a["xmin"].view(np.uint64), b["xmin"].view(np.uint64)
    bad = np.argwhere((xa != xb) & (short[..., None] | (np.arange(36) < 21)))
    assert bad.size == 0, ("xmin", bad[:4])
    for name in ("sc_en", "sc_xm"):
        bad = np.argwhere((a[name] != b[name]) & ~short[..., None])
        assert bad.size == 0, (name, bad[:4])
    return int(a.size)


def compare_fft_seam(st, s, seam, channels, frames=None):
    """seam[k] (FFT_SEAM_DT records [gr][ch], oracle/fft_seam.h) is the reference's / the oracle's record of frame
    frames[k] (default: k): what fft() / enphinew() returned to L3psycho_anal (src/subs.c:38-123).  Compared bit for
    bit with what k_fft handed on: the 513 long and 3 x 129 short energies (floor applied, src/subs.c:70-74) and the raw
    lines the unpredictability is computed from (long 0..5, short 2..51: src/l3psy.c:496-549).  Returns mismatches."""
    bad = []

    def chk(where, name, a, b):
        a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
        if not np.array_equal(a.view(np.uint32), b.view(np.uint32)):
            bad.append("%s %s (%d of %d differ)" % (where, name, int(np.sum(a.view(np.uint32) != b.view(np.uint32))), a.size))

    seam = np.asarray(seam).reshape(-1, 2, channels)
    for k, f in enumerate(frames if frames is not None else range(len(seam))):
        for gr in range(2):
            for c in range(channels):
                g, r = 2 * f + gr, seam[k, gr, c]
                w = "stream %d frame %d gr %d ch %d" % (s, f, gr, c)
                chk(w, "energy_l", st["energy_l"][s, g, c, :513], r["energy_l"])
                chk(w, "energy_s", st["energy_s"][s, g, c], r["energy_s"])
                bins = st["fft_bins"][s, g, c]
                sh = bins[:300].reshape(3, 50, 2)
                chk(w, "short lines re", sh[:, :, 0], r["re_s"])
                chk(w, "short lines im", sh[:, :, 1], r["im_s"])
                chk(w, "long lines re", bins[300:306], r["re_l"])
                chk(w, "long lines im", bins[307:312], r["im_l"][1:])  # (line 0 is real: the kernel stores -0 so that atan2(-im, re) is atan2(0, re))
                if bins[306] != 0.0:
                    bad.append(w + " long line 0 im")
    return bad

"""Shared helpers for the test-suite: ctypes bindings of the product library (libmp3mi.so),
its CPU-emulated test build (tests/hipemu/_build/libmp3mi_emu.so) and the oracle
(oracle/_build/liboracle.so), plus numpy views of the records they exchange."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# MP3MI_LIB: a diagnostic build of the same library (tools/gpu_ulp_census*.sh), as mp3-enc-bsd_amd/__init__.py honours it
PRODUCT_SO = os.environ.get("MP3MI_LIB") or os.path.join(ROOT, "mp3-enc-bsd_amd", "libmp3mi.so")
EMU_SO = os.path.join(ROOT, "tests", "hipemu", "_build", "libmp3mi_emu.so")
ORACLE_SO = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
REF_HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
SEED = 0x6D70336D

# oracle/stage_dump.h
STAGE_DT = np.dtype([
    ("pe", "<f8", (2, 2)), ("ratio_l", "<f8", (2, 2, 21)), ("ratio_s", "<f8", (2, 2, 12, 3)),
    ("sb", "<f8", (2, 2, 18, 32)), ("xr", "<f8", (2, 2, 576)), ("l3_enc", "<i4", (2, 2, 576)),
    ("psy_bt", "<i4", (2, 2)), ("main_data_begin", "<i4"), ("resvDrain", "<i4"), ("scfsi", "<i4", (2, 4)),
    ("gi", "<i4", (2, 2, 20)), ("scalefac_l", "<i4", (2, 2, 22)), ("scalefac_s", "<i4", (2, 2, 13, 3)),
    ("magic", "<i4"), ("frame_index", "<i4")])
GI = {n: i for i, n in enumerate([
    "part2_3_length", "big_values", "count1", "global_gain", "scalefac_compress", "window_switching_flag",
    "block_type", "mixed_block_flag", "table_select0", "table_select1", "table_select2", "subblock_gain0",
    "subblock_gain1", "subblock_gain2", "region0_count", "region1_count", "preflag", "scalefac_scale",
    "count1table_select", "part2_length"])}

# oracle/fft_seam.h: one record per L3psycho_anal call, in call order [frame][gr][ch]
FFT_SEAM_DT = np.dtype([("energy_l", "<f4", (513,)), ("phi_l", "<f4", (6,)), ("re_l", "<f4", (6,)), ("im_l", "<f4", (6,)),
                        ("energy_s", "<f4", (3, 129)), ("phi_s", "<f4", (3, 50)), ("re_s", "<f4", (3, 50)), ("im_s", "<f4", (3, 50))])

# mp3-enc-bsd_amd/csrc/mp3mi_dev.h
PSY_DT = np.dtype([("pe", "<f8"), ("ratio_l", "<f8", (21,)), ("ratio_s", "<f8", (12, 3)), ("block_type", "<i4"), ("pad", "<i4")])
PREP_DT = np.dtype([("xmin", "<f8", (36,)), ("sc_en", "<i4", (21,)), ("sc_xm", "<i4", (21,)), ("sc_en_tot", "<i4"), ("sc_xrmax", "<i4"),
                    ("q0", "<i4"), ("nonzero", "<i4")])  # mp3mi_loop_prep, csrc/mp3mi_dev.h
GRSIDE_DT = np.dtype([
    ("part2_3_length", "<i4"), ("big_values", "<i4"), ("count1", "<i4"), ("global_gain", "<i4"),
    ("scalefac_compress", "<i4"), ("window_switching_flag", "<i4"), ("block_type", "<i4"),
    ("table_select", "<i4", (3,)), ("region0_count", "<i4"), ("region1_count", "<i4"), ("preflag", "<i4"),
    ("count1table_select", "<i4"), ("part2_length", "<i4"), ("scalefac", "<i4", (39,))])
SIDE_DT = np.dtype([("main_data_begin", "<i4"), ("resvDrain", "<i4"), ("scfsi", "<i4", (2, 4)), ("gr", GRSIDE_DT, (2, 2))])


def build_oracle():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "oracle"], check=True)


def build_emu():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "tests", "hipemu")], check=True)


class ReferenceAborts(Exception):
    """The reference dies on this input (an assertion fails): .status is the MP3O_ABORT_* / MP3MI_STREAM_* code,
    .frame the index of the frame it died in, the number of frames for the final flush (oracle/mp3_oracle.h)."""

    def __init__(self, code):
        Exception.__init__(self, "the reference aborts on this input: status %d in frame %d" % (code & 255, code >> 8))
        self.status, self.frame = code & 255, code >> 8


class Oracle:
    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        self.lib = ctypes.CDLL(ORACLE_SO)
        self.lib.mp3o_encode_pcm_ex.restype = ctypes.c_size_t
        self.lib.mp3o_encode_pcm_ex.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_char_p, ctypes.c_void_p,
                                                ctypes.c_size_t, ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int,
                                                ctypes.POINTER(ctypes.c_int)]
        self.lib.mp3o_encode_pcm_fft_seam.restype = ctypes.c_long
        self.lib.mp3o_encode_pcm_fft_seam.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t,
                                                      ctypes.c_void_p, ctypes.c_int]
        self.libc = ctypes.CDLL("libc.so.6")
        self.libc.free.argtypes = [ctypes.c_void_p]

    def fft_seam(self, pcm, rate, kbps, channels, frames):
        """the transforms' outputs of every L3psycho_anal call of the first `frames` frames (oracle/fft_seam.h):
        records [frames][2][channels] of FFT_SEAM_DT"""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        assert FFT_SEAM_DT.itemsize == 4 * (513 + 18 + 3 * 129 + 9 * 50)
        seam = np.zeros((frames, 2, channels), dtype=FFT_SEAM_DT)
        n = self.lib.mp3o_encode_pcm_fft_seam(rate, kbps, channels, pcm.ctypes.data, pcm.size, seam.ctypes.data, frames)
        assert n == frames * 2 * channels, (n, frames)
        return seam

    def encode(self, pcm, rate, kbps, channels, dumps=0, mode=None):
        """pcm: int16 array, interleaved; mode: None or the driver's -m letter (s / d / m) followed by e / c / o for
        its -e / -c / -o options.  Returns (mp3 bytes, stage dumps or None); raises ReferenceAborts where the reference
        dies, ValueError where it refuses the configuration."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        out = ctypes.c_void_p()
        ab = ctypes.c_int(0)
        d = np.zeros(dumps, dtype=STAGE_DT) if dumps else None
        n = self.lib.mp3o_encode_pcm_ex(rate, kbps, channels, mode.encode() if mode else None, pcm.ctypes.data, pcm.size,
                                        ctypes.byref(out), d.ctypes.data if dumps else None, dumps, ctypes.byref(ab))
        if not out.value:
            raise ValueError("oracle refused configuration")
        data = ctypes.string_at(out.value, n)
        self.libc.free(out)
        if ab.value:
            raise ReferenceAborts(ab.value)
        return data, d


ERR_REFERENCE_ABORT = -6  # include/mp3mi.h


class BatchOptions(ctypes.Structure):
    """include/mp3mi.h: mp3mi_batch_options"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("scratch_mb", ctypes.c_uint32), ("chunk_frames", ctypes.c_int32),
                ("test_flags", ctypes.c_uint32), ("call_overlap", ctypes.c_int32), ("gate", ctypes.c_int32),
                ("placement", ctypes.c_int32), ("loop_part_streams", ctypes.c_int32),
                ("y_after_loop", ctypes.c_int32), ("psy_beside", ctypes.c_int32), ("dropin_lookahead", ctypes.c_int32), ("call_hold", ctypes.c_int32), ("dropin_stats", ctypes.c_int32),
                ("abi", ctypes.c_uint32)]


class Mp3mi:
    """The product library (emu=False) or its emulated CPU test build (emu=True)."""

    def options(self, **kw):
        """mp3mi_batch_options with the defaults, fields overridden by keyword"""
        o = BatchOptions()
        self.lib.mp3mi_batch_options_default(ctypes.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def __init__(self, emu=False):
        path = EMU_SO if emu else PRODUCT_SO
        if emu and not os.path.exists(path):
            build_emu()
        self.lib = ctypes.CDLL(path)
        L = self.lib
        L.mp3mi_batch_create.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        L.mp3mi_batch_create_ex.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
        L.mp3mi_batch_options_default.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_options_from_env.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_stream_status.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.mp3mi_batch_destroy.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_out_stride.restype = ctypes.c_size_t
        L.mp3mi_batch_out_stride.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_batch_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_batch_sync.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_debug_fetch.restype = ctypes.c_long
        L.mp3mi_batch_debug_fetch.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t]
        L.mp3mi_batch_debug_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_batch_last_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
        L.mp3mi_encode_host.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_encode_host_ex.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_synth_pcm.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
        L.mp3mi_version.restype = ctypes.c_char_p
        L.mp3mi_batch_set_test_flags.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.mp3mi_synth_pcm_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
        L.mp3mi_debug_fastmath_bounds.argtypes = [ctypes.c_void_p]
        L.mp3mi_debug_pknorm_bound.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_encode_next.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_batch_encode_host_async.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_batch_host_io_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.mp3mi_host_alloc.argtypes = [ctypes.c_size_t]
        L.mp3mi_host_alloc.restype = ctypes.c_void_p
        L.mp3mi_host_free.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_flush.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_batch_reset.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_debug_cw_fixups.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.mp3mi_batch_debug_prep_fixups.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
        L.mp3mi_batch_set_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_batch_set_error_protection.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_batch_set_header.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]

    def synth(self, n_per_ch, channels, rate, stream, seed=SEED):
        out = np.zeros(n_per_ch * channels, dtype=np.int16)
        self.lib.mp3mi_synth_pcm(out.ctypes.data, n_per_ch, channels, rate, stream, seed)
        return out

    def encode_host(self, pcm, rate, channels, kbps, n_frames):
        """pcm: int16 [n_streams, n_frames*1152*channels]; kbps: int or per-stream list.
        Returns list of bytes per stream."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        S = pcm.shape[0]
        assert pcm.shape[1] == n_frames * 1152 * channels
        if np.isscalar(kbps):
            karr, kall, kmax = None, int(kbps), int(kbps)
        else:
            karr = np.ascontiguousarray(kbps, dtype=np.int32)
            kall, kmax = 0, int(karr.max())
        stride = (n_frames * (int(1152 / (rate / 1000.0) * kmax / 8) + 1) + 1 + 255) // 256 * 256
        out = np.zeros((S, stride), dtype=np.uint8)
        lens = np.zeros(S, dtype=np.uint32)
        rc = self.lib.mp3mi_encode_host(S, rate, channels, karr.ctypes.data if karr is not None else None, kall,
                                        pcm.ctypes.data, n_frames, out.ctypes.data, stride, lens.ctypes.data)
        if rc != 0:
            raise RuntimeError("mp3mi_encode_host failed: %d" % rc)
        return [out[s, :lens[s]].tobytes() for s in range(S)]


def encode_host_ex(mp, pcm, n_samples, rate, channels, kbps, n_frames, copyright=0, original=0, emphasis=0):
    """Ragged batch through the host wrapper: pcm int16 [S, n_frames*1152*channels], n_samples per channel
    per stream (or None).  Returns list of bytes per stream."""
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    S = pcm.shape[0]
    ns = None if n_samples is None else np.ascontiguousarray(n_samples, dtype=np.int32)
    stride = (n_frames * (int(1152 / (rate / 1000.0) * kbps / 8) + 1) + 1 + 255) // 256 * 256
    out = np.zeros((S, stride), dtype=np.uint8)
    lens = np.zeros(S, dtype=np.uint32)
    rc = mp.lib.mp3mi_encode_host_ex(S, rate, channels, None, int(kbps), pcm.ctypes.data, ns.ctypes.data if ns is not None else None,
                                     n_frames, copyright, original, emphasis, out.ctypes.data, stride, lens.ctypes.data)
    if rc != 0:
        raise RuntimeError("mp3mi_encode_host_ex failed: %d" % rc)
    return [out[s, :lens[s]].tobytes() for s in range(S)]


def pad_frames(pcm, channels):
    """zero-fill interleaved PCM to whole frames (src/encode.c:162-166)"""
    per = 1152 * channels
    n = (len(pcm) + per - 1) // per
    out = np.zeros(n * per, dtype=np.int16)
    out[:len(pcm)] = pcm
    return out, n


class DevMem:
    """Memory the library under test computes on: HBM through libamdhip64 for the product, plain host memory
    for the emulated test build (whose 'device pointers' are host pointers)."""

    def __init__(self, mp):
        self.emu = b"emulator" in mp.lib.mp3mi_version()
        self.bufs = []
        if not self.emu:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
            hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
            hip.hipFree.argtypes = [ctypes.c_void_p]
            self.hip = hip

    def alloc(self, nbytes):
        if self.emu:
            a = np.zeros(max(nbytes, 1), np.uint8)
            self.bufs.append(a)
            return a.ctypes.data
        p = ctypes.c_void_p()
        assert self.hip.hipMalloc(ctypes.byref(p), max(nbytes, 1)) == 0, "hipMalloc(%d)" % nbytes
        self.bufs.append(p)
        return p.value

    def upload(self, dptr, arr):
        arr = np.ascontiguousarray(arr)
        if self.emu:
            ctypes.memmove(dptr, arr.ctypes.data, arr.nbytes)
        else:
            assert self.hip.hipMemcpy(dptr, arr.ctypes.data, arr.nbytes, 1) == 0

    def download(self, dptr, shape, dtype):
        out = np.zeros(shape, dtype)
        if self.emu:
            ctypes.memmove(out.ctypes.data, dptr, out.nbytes)
        else:
            assert self.hip.hipMemcpy(out.ctypes.data, dptr, out.nbytes, 2) == 0
        return out

    def free(self):
        if not self.emu:
            for p in self.bufs:
                self.hip.hipFree(p)
        self.bufs = []


class BatchRun:
    """One batch of S streams on device memory through the C ABI: PCM synthesised on the device (or uploaded),
    encoded as often as wanted (e.g. once per test-flag setting), outputs fetched per stream."""

    def __init__(self, mp, S, rate, channels, kbps, n_frames, stream0=0, pcm=None, seed=SEED, mode=None, crc=0, options=None,
                 copyright=0, original=0):
        """mode: the header mode field (0 stereo, 2 dual channel, 3 mono) or the driver's option string as
        oracle/ref_harness.c takes it (-m letter, then e / c / o); options: a BatchOptions (mp3mi_batch_create_ex)"""
        self.mp, self.S, self.rate, self.ch, self.nf = mp, S, rate, channels, n_frames
        L = mp.lib
        self.mem = DevMem(mp)
        self.b = ctypes.c_void_p()
        karr = None if np.isscalar(kbps) else np.ascontiguousarray(kbps, dtype=np.int32)
        if options is None:
            rc = L.mp3mi_batch_create(ctypes.byref(self.b), S, rate, channels, karr.ctypes.data if karr is not None else None,
                                      int(kbps) if karr is None else 0, n_frames)
        else:
            rc = L.mp3mi_batch_create_ex(ctypes.byref(self.b), S, rate, channels, karr.ctypes.data if karr is not None else None,
                                         int(kbps) if karr is None else 0, n_frames, ctypes.byref(options))
        assert rc == 0, "mp3mi_batch_create -> %d" % rc
        if isinstance(mode, str):
            crc = crc or ("e" in mode[1:])
            copyright = copyright or ("c" in mode[1:])
            original = original or ("o" in mode[1:])
            mode = {"s": 0, "d": 2, "m": 3}[mode[0]]
        if mode is not None:
            assert L.mp3mi_batch_set_mode(self.b, mode) == 0
        if crc:
            assert L.mp3mi_batch_set_error_protection(self.b, 1) == 0
        if copyright or original:
            assert L.mp3mi_batch_set_header(self.b, int(bool(copyright)), int(bool(original)), 0) == 0
        self.stride = L.mp3mi_batch_out_stride(self.b, n_frames)
        self.n_per_ch = n_frames * 1152
        self.pcm_bytes = S * self.n_per_ch * channels * 2
        self.d_pcm = self.mem.alloc(self.pcm_bytes)
        self.d_out = self.mem.alloc(S * self.stride)
        self.d_len = self.mem.alloc(4 * S)
        if pcm is None:
            rc = L.mp3mi_synth_pcm_device(self.d_pcm, S, self.n_per_ch, channels, rate, stream0, seed)
            assert rc == 0, "mp3mi_synth_pcm_device -> %d" % rc
        else:
            pcm = np.ascontiguousarray(pcm, dtype=np.int16)
            assert pcm.nbytes == self.pcm_bytes
            self.mem.upload(self.d_pcm, pcm)

    def pcm_of(self, s):
        """stream s's PCM as the device holds it"""
        n = self.n_per_ch * self.ch
        return self.mem.download(self.d_pcm + s * n * 2, (n,), np.int16)

    def encode(self, flags=0, expect_abort=False):
        L = self.mp.lib
        assert L.mp3mi_batch_set_test_flags(self.b, flags) == 0
        rc = L.mp3mi_batch_encode(self.b, self.d_pcm, self.nf, self.d_out, self.stride, self.d_len)
        assert rc == 0, "mp3mi_batch_encode -> %d" % rc
        rc = L.mp3mi_batch_sync(self.b)
        assert rc == (ERR_REFERENCE_ABORT if expect_abort else 0), "mp3mi_batch_sync -> %d" % rc
        out = self.mem.download(self.d_out, (self.S, self.stride), np.uint8)
        lens = self.mem.download(self.d_len, (self.S,), np.uint32)
        return out, lens

    def status(self):
        """mp3mi_batch_stream_status: per stream 0 or code | frame << 8"""
        st = np.zeros(self.S, np.int32)
        rc = self.mp.lib.mp3mi_batch_stream_status(self.b, st.ctypes.data)
        assert rc >= 0, "mp3mi_batch_stream_status -> %d" % rc
        return st

    def fetch(self, what, dtype):
        """mp3mi_batch_debug_fetch of the last call (one chunk): records [S, 2 * n_frames, channels] of dtype"""
        a = np.zeros((self.S, 2 * self.nf, self.ch), dtype)
        n = self.mp.lib.mp3mi_batch_debug_fetch(self.b, what, a.ctypes.data, a.nbytes)
        assert n == a.nbytes, (n, a.nbytes)
        return a

    def prep_fixups(self):
        """records of the last item that k_mdct's tail listed for k_prep"""
        a = ctypes.c_int()
        assert self.mp.lib.mp3mi_batch_debug_prep_fixups(self.b, ctypes.byref(a)) == 0
        return a.value

    def cw_fixups(self):
        """(records listed for the second tier of the unpredictability, records) of the last call's last chunk"""
        a, n = ctypes.c_int(), ctypes.c_int()
        assert self.mp.lib.mp3mi_batch_debug_cw_fixups(self.b, ctypes.byref(a), ctypes.byref(n)) == 0
        return a.value, n.value

    def encode_streaming(self, pieces):
        """The same PCM fed piece by piece (frames per call) through mp3mi_batch_encode_next, then flushed.
        Returns the concatenated bytes per stream."""
        L = self.mp.lib
        assert sum(pieces) == self.nf
        got = [b""] * self.S
        row = self.n_per_ch * self.ch  # int16 per stream in the resident buffer
        f0 = 0
        for nfp in pieces:
            n = nfp * 1152 * self.ch
            # the call's frames of every stream, packed [S][nfp*1152][C]
            whole = self.mem.download(self.d_pcm, (self.S, row), np.int16)
            piece = np.ascontiguousarray(whole[:, f0 * 1152 * self.ch: f0 * 1152 * self.ch + n])
            d_piece = self.mem.alloc(piece.nbytes)
            self.mem.upload(d_piece, piece)
            rc = L.mp3mi_batch_encode_next(self.b, d_piece, nfp, self.d_out, self.stride, self.d_len)
            assert rc == 0, "mp3mi_batch_encode_next -> %d" % rc
            assert L.mp3mi_batch_sync(self.b) == 0
            out = self.mem.download(self.d_out, (self.S, self.stride), np.uint8)
            lens = self.mem.download(self.d_len, (self.S,), np.uint32)
            for s in range(self.S):
                got[s] += out[s, :lens[s]].tobytes()
            f0 += nfp
        assert L.mp3mi_batch_flush(self.b, self.d_out, self.stride, self.d_len) == 0
        assert L.mp3mi_batch_sync(self.b) == 0
        out = self.mem.download(self.d_out, (self.S, self.stride), np.uint8)
        lens = self.mem.download(self.d_len, (self.S,), np.uint32)
        for s in range(self.S):
            got[s] += out[s, :lens[s]].tobytes()
        return got

    def encode_streaming_unsynced(self, pieces, whole_first=False):
        """encode_streaming with every call (and the flush) issued back to back, each into an output buffer of its
        own, and ONE mp3mi_batch_sync at the end: a call's feed-forward kernels start while the loop kernels of the
        call before still run (batch.cpp, encode_impl).  whole_first: a whole-file call of the same PCM is issued
        first, unsynchronised too; its bytes per stream are returned as a second list."""
        L = self.mp.lib
        assert sum(pieces) == self.nf
        row = self.n_per_ch * self.ch
        whole = self.mem.download(self.d_pcm, (self.S, row), np.int16)
        calls, f0 = [], 0
        for nfp in pieces:
            n = nfp * 1152 * self.ch
            piece = np.ascontiguousarray(whole[:, f0 * 1152 * self.ch: f0 * 1152 * self.ch + n])
            d_piece = self.mem.alloc(piece.nbytes)
            self.mem.upload(d_piece, piece)
            calls.append((d_piece, nfp, self.mem.alloc(self.S * self.stride), self.mem.alloc(4 * self.S)))
            f0 += nfp
        d_fout, d_flen = self.mem.alloc(self.S * self.stride), self.mem.alloc(4 * self.S)
        d_wout, d_wlen = self.mem.alloc(self.S * self.stride), self.mem.alloc(4 * self.S)
        if whole_first:
            assert L.mp3mi_batch_encode(self.b, self.d_pcm, self.nf, d_wout, self.stride, d_wlen) == 0
        for d_piece, nfp, d_o, d_l in calls:
            rc = L.mp3mi_batch_encode_next(self.b, d_piece, nfp, d_o, self.stride, d_l)
            assert rc == 0, "mp3mi_batch_encode_next -> %d" % rc
        assert L.mp3mi_batch_flush(self.b, d_fout, self.stride, d_flen) == 0
        assert L.mp3mi_batch_sync(self.b) == 0
        got = [b""] * self.S
        for d_o, d_l in [(c[2], c[3]) for c in calls] + [(d_fout, d_flen)]:
            out = self.mem.download(d_o, (self.S, self.stride), np.uint8)
            lens = self.mem.download(d_l, (self.S,), np.uint32)
            for s in range(self.S):
                got[s] += out[s, :lens[s]].tobytes()
        if not whole_first:
            return got
        out = self.mem.download(d_wout, (self.S, self.stride), np.uint8)
        lens = self.mem.download(d_wlen, (self.S,), np.uint32)
        return got, [out[s, :lens[s]].tobytes() for s in range(self.S)]

    def close(self):
        if self.b:
            self.mp.lib.mp3mi_batch_destroy(self.b)
            self.b = ctypes.c_void_p()
        self.mem.free()


# ---------------------------------------------------------------------------------------------------------------------
# Layers I and II: oracle/mp12_oracle.inc, oracle/ref_harness_l12.c, include/mp3mi_l12.h
# ---------------------------------------------------------------------------------------------------------------------
REF_HARNESS_L12 = os.path.join(ROOT, "oracle", "_ref", "ref_harness_l12")
REF_ENCODE = os.path.join(ROOT, "oracle", "_ref", "encode")
# oracle/stage_dump_l12.h
L12_DT = np.dtype([("sb", "<f8", (2, 3, 12, 32)), ("ltmin", "<f8", (2, 32)), ("scalar", "<i4", (2, 3, 32)), ("j_scale", "<i4", (3, 32)),
                   ("scfsi", "<i4", (2, 32)), ("bit_alloc", "<i4", (2, 32)), ("mode", "<i4"), ("mode_ext", "<i4"), ("jsbound", "<i4"),
                   ("sblimit", "<i4"), ("adb_left", "<i4"), ("crc", "<i4"), ("magic", "<i4"), ("frame_index", "<i4")])
# include/mp3mi_l12.h: mp3mi_l12_frame_seams
L12_SEAM_DT = np.dtype([("ltmin", "<f8", (2, 32)), ("scalar", "<i4", (2, 3, 32)), ("j_scale", "<i4", (3, 32)), ("scfsi", "<i4", (2, 32)),
                        ("bit_alloc", "<i4", (2, 32)), ("mode", "<i4"), ("mode_ext", "<i4"), ("jsbound", "<i4"), ("sblimit", "<i4"),
                        ("adb_left", "<i4"), ("crc", "<i4"), ("pad", "<i4", (2,))])
L12_SEAMS = ["ltmin", "scalar", "j_scale", "scfsi", "bit_alloc", "mode", "mode_ext", "jsbound", "sblimit", "adb_left", "crc"]
L12_BITRATES = {1: [32, 64, 96, 128, 160, 192, 224, 256, 288, 320, 352, 384, 416, 448],
                2: [32, 48, 56, 64, 80, 96, 112, 128, 160, 192, 224, 256, 320, 384]}
L12_MODES = {"s": 0, "j": 1, "d": 2, "m": 3}


def l12_spf(layer):
    return 384 if layer == 1 else 1152


def oracle_l12(orc, layer, rate, kbps, mode, pcm, dumps=0):
    """orc: an Oracle; mode: the driver's -m letter (s / d / j / m) followed by e / c / o.  Returns (bytes, dumps or None)."""
    L = orc.lib
    L.mp3o_encode_pcm_l12.restype = ctypes.c_size_t
    L.mp3o_encode_pcm_l12.argtypes = [ctypes.c_int] * 4 + [ctypes.c_char_p, ctypes.c_void_p, ctypes.c_size_t,
                                                            ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, ctypes.c_int]
    pcm = np.ascontiguousarray(pcm, dtype=np.int16)
    out = ctypes.c_void_p()
    d = np.zeros(dumps, L12_DT) if dumps else None
    ch = 1 if mode[0] == "m" else 2
    n = L.mp3o_encode_pcm_l12(layer, rate, kbps, ch, mode.encode(), pcm.ctypes.data, pcm.size, ctypes.byref(out),
                              d.ctypes.data if dumps else None, dumps)
    if not out.value:
        raise ValueError("oracle refused configuration")
    data = ctypes.string_at(out.value, n)
    orc.libc.free(out)
    return data, d


def ref_l12(layer, rate, kbps, mode, pcm, workdir, dumps=True):
    """the UNMODIFIED reference through oracle/_ref/ref_harness_l12 (bytes, stage_dump_l12 records)"""
    wav, mpg, dmp = (os.path.join(workdir, n) for n in ("in.wav", "out.mpg", "dump.bin"))
    with open(wav, "wb") as f:
        f.write(b"\0" * 44)
        f.write(np.ascontiguousarray(pcm, dtype="<i2").tobytes())
    r = subprocess.run([REF_HARNESS_L12, wav, mpg, str(layer), str(rate), str(kbps), mode] + ([dmp] if dumps else []),
                       capture_output=True, cwd=workdir)
    assert r.returncode == 0, (r.returncode, r.stderr[-300:])
    return open(mpg, "rb").read(), (np.fromfile(dmp, L12_DT) if dumps else None)


def l12_signal(n_per_ch, ch, seed, rate=44100):
    """a sweep, noise and a burst per channel (test input of the Layer I / II parity tests)"""
    rng = np.random.default_rng(seed)
    t = np.arange(n_per_ch)
    x = np.zeros((n_per_ch, ch))
    for c in range(ch):
        f0 = rng.uniform(100, 4000)
        x[:, c] = rng.uniform(2000, 12000) * np.sin(2 * np.pi * f0 * t / rate * (1 + t / max(n_per_ch, 1))) + rng.uniform(100, 3000) * rng.standard_normal(n_per_ch)
        a = n_per_ch // 3
        x[a:a + 200, c] += rng.uniform(0, 15000) * rng.standard_normal(len(x[a:a + 200, c]))
    return np.clip(x, -32768, 32767).astype(np.int16).reshape(-1)


class L12Run:
    """One Layer I / II batch through the C ABI (include/mp3mi_l12.h) on device memory (or the emulated test build's
    host memory): pcm_list = interleaved int16 arrays, one per stream (ragged lengths allowed)."""

    def __init__(self, mp, layer, rate, kbps, mode, pcm_list=None, n_frames=None, scratch_mb=0, flags=0, seams=False,
                 synth=None):
        """synth = (S, stream0): PCM synthesised on the device by mp3mi_synth_pcm_device instead of pcm_list"""
        L = mp.lib
        self.mp, self.layer, self.rate, self.mode = mp, layer, rate, mode
        self.ch = ch = 1 if mode[0] == "m" else 2
        spf = l12_spf(layer)
        if synth:
            S = synth[0]
        else:
            S = len(pcm_list)
            n_frames = n_frames or max((len(p) // ch + spf - 1) // spf for p in pcm_list)
        self.S, self.nf = S, n_frames
        L.mp3mi_l12_batch_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
        L.mp3mi_l12_batch_out_stride.restype = ctypes.c_size_t
        L.mp3mi_l12_batch_out_stride.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_l12_batch_encode.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_l12_batch_sync.argtypes = [ctypes.c_void_p]
        L.mp3mi_l12_batch_destroy.argtypes = [ctypes.c_void_p]
        L.mp3mi_l12_batch_set_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_l12_batch_set_error_protection.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_l12_batch_set_header.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.mp3mi_l12_batch_set_test_flags.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.mp3mi_l12_batch_debug_enable.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_l12_batch_debug_fetch.restype = ctypes.c_long
        L.mp3mi_l12_batch_debug_fetch.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p]
        L.mp3mi_l12_batch_total_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
        self.b = ctypes.c_void_p()
        karr = None if np.isscalar(kbps) else np.ascontiguousarray(kbps, dtype=np.int32)
        rc = L.mp3mi_l12_batch_create(ctypes.byref(self.b), layer, S, rate, ch, karr.ctypes.data if karr is not None else None,
                                      int(kbps) if karr is None else 0, n_frames, scratch_mb)
        assert rc == 0, "mp3mi_l12_batch_create -> %d" % rc
        assert L.mp3mi_l12_batch_set_mode(self.b, L12_MODES[mode[0]]) == 0
        assert L.mp3mi_l12_batch_set_error_protection(self.b, int("e" in mode[1:])) == 0
        assert L.mp3mi_l12_batch_set_header(self.b, int("c" in mode[1:]), int("o" in mode[1:]), 0) == 0
        assert L.mp3mi_l12_batch_set_test_flags(self.b, flags) == 0
        if seams:
            L.mp3mi_l12_batch_debug_enable(self.b, 1)
        self.mem = DevMem(mp)
        self.stride = L.mp3mi_l12_batch_out_stride(self.b, n_frames)
        row = n_frames * spf * ch
        self.row = row
        self.d_pcm = self.mem.alloc(S * row * 2)
        self.d_out = self.mem.alloc(S * self.stride)
        self.d_len = self.mem.alloc(4 * S)
        self.d_ns = None
        if synth:
            rc = L.mp3mi_synth_pcm_device(self.d_pcm, S, n_frames * spf, ch, rate, synth[1], SEED)
            assert rc == 0, "mp3mi_synth_pcm_device -> %d" % rc
        else:
            pcm = np.zeros((S, row), np.int16)
            ns = np.zeros(S, np.int32)
            for i, p in enumerate(pcm_list):
                pcm[i, :len(p)] = p
                ns[i] = len(p) // ch
            self.mem.upload(self.d_pcm, pcm)
            self.d_ns = self.mem.alloc(4 * S)
            self.mem.upload(self.d_ns, ns)

    def pcm_of(self, s):
        return self.mem.download(self.d_pcm + s * self.row * 2, (self.row,), np.int16)

    def set_flags(self, flags):
        assert self.mp.lib.mp3mi_l12_batch_set_test_flags(self.b, flags) == 0

    def encode(self):
        L = self.mp.lib
        rc = L.mp3mi_l12_batch_encode(self.b, self.d_pcm, self.d_ns, self.nf, self.d_out, self.stride, self.d_len)
        assert rc == 0, "mp3mi_l12_batch_encode -> %d" % rc
        assert L.mp3mi_l12_batch_sync(self.b) == 0
        out = self.mem.download(self.d_out, (self.S, self.stride), np.uint8)
        lens = self.mem.download(self.d_len, (self.S,), np.uint32)
        return [out[i, :lens[i]].tobytes() for i in range(self.S)]

    def encode_streaming(self, pieces):
        """the same PCM fed piece by piece (frames per call) through mp3mi_l12_batch_encode_next, then flushed; returns the
        concatenated bytes per stream.  Whole-length streams only (ragged batches and streaming do not combine)."""
        L = self.mp.lib
        L.mp3mi_l12_batch_encode_next.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_l12_batch_flush.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        assert sum(pieces) == self.nf
        spf = l12_spf(self.layer)
        whole = self.mem.download(self.d_pcm, (self.S, self.row), np.int16)
        got = [b""] * self.S
        f0 = 0
        for nfp in pieces:
            n = nfp * spf * self.ch
            piece = np.ascontiguousarray(whole[:, f0 * spf * self.ch: f0 * spf * self.ch + n])
            d_piece = self.mem.alloc(piece.nbytes)
            self.mem.upload(d_piece, piece)
            rc = L.mp3mi_l12_batch_encode_next(self.b, d_piece, nfp, self.d_out, self.stride, self.d_len)
            assert rc == 0, "mp3mi_l12_batch_encode_next -> %d" % rc
            assert L.mp3mi_l12_batch_sync(self.b) == 0
            out = self.mem.download(self.d_out, (self.S, self.stride), np.uint8)
            lens = self.mem.download(self.d_len, (self.S,), np.uint32)
            for s in range(self.S):
                got[s] += out[s, :lens[s]].tobytes()
            f0 += nfp
        assert L.mp3mi_l12_batch_flush(self.b, self.d_out, self.stride, self.d_len) == 0
        assert L.mp3mi_l12_batch_sync(self.b) == 0
        out = self.mem.download(self.d_out, (self.S, self.stride), np.uint8)
        lens = self.mem.download(self.d_len, (self.S,), np.uint32)
        for s in range(self.S):
            got[s] += out[s, :lens[s]].tobytes()
        return got

    def seams(self):
        """(records [S][frames of the last chunk], first frame of that chunk)"""
        d = np.zeros((self.S * self.nf,), L12_SEAM_DT)
        f0, nf = ctypes.c_int(), ctypes.c_int()
        n = self.mp.lib.mp3mi_l12_batch_debug_fetch(self.b, d.ctypes.data, d.nbytes, ctypes.byref(f0), ctypes.byref(nf))
        assert n > 0, n
        return d[:self.S * nf.value].reshape(self.S, nf.value), f0.value

    def kernel_ms(self):
        ms, calls = ctypes.c_double(), ctypes.c_long()
        assert self.mp.lib.mp3mi_l12_batch_total_timing(self.b, ctypes.byref(ms), ctypes.byref(calls)) == 0
        return ms.value, calls.value

    def close(self):
        if self.b:
            self.mp.lib.mp3mi_l12_batch_destroy(self.b)
            self.b = ctypes.c_void_p()
        self.mem.free()


def l12_compare_seams(ref_dumps, got, f0):
    """names of the seams of `got` ([frames of a chunk] L12_SEAM_DT, first frame f0) that differ from the stage_dump_l12
    records of the same frames"""
    bad = []
    for k in range(len(got)):
        if f0 + k >= len(ref_dumps):
            break
        for name in L12_SEAMS:
            if not np.array_equal(ref_dumps[f0 + k][name], got[k][name]):
                bad.append((f0 + k, name))
    return bad

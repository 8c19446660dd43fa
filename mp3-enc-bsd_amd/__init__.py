"""mp3-enc-bsd_amd: MI355X-native MPEG-1 Layer III encoding hot path.

The product is the C-ABI shared library next to this file (libmp3mi.so, built from csrc/ by
`make -C csrc` or `__graft_entry__.build()`); this module is only a thin ctypes binding used by
bench.py, smoke() and the tests.  There is no Python or CPU implementation of the path here:
if the library or a GPU is missing, everything fails loudly.

Import with importlib (the directory name carries a hyphen):
    mp3 = importlib.import_module("mp3-enc-bsd_amd")
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MP3MI_LIB: another build of the same library (tools/gpu_ab.sh: A/B of two builds on one device; tools/gpu_loop_profile.sh:
# a diagnostic build) -- so that measurements never overwrite the product library
LIB_PATH = os.environ.get("MP3MI_LIB") or os.path.join(_HERE, "libmp3mi.so")
FRAME_SAMPLES = 1152


class Mp3miError(RuntimeError):
    pass


class ReferenceAborts(Mp3miError):
    """mp3mi_batch_sync returned MP3MI_ERR_REFERENCE_ABORT: the work is done, but the reference would have died on
    at least one stream's input (include/mp3mi.h); that stream's out_len is 0, Batch.stream_status() says why."""


ERR_REFERENCE_ABORT = -6


class BatchOptions(ctypes.Structure):
    """include/mp3mi.h: mp3mi_batch_options"""
    _fields_ = [("struct_size", ctypes.c_uint32), ("scratch_mb", ctypes.c_uint32), ("chunk_frames", ctypes.c_int32),
                ("test_flags", ctypes.c_uint32), ("call_overlap", ctypes.c_int32), ("gate", ctypes.c_int32),
                ("placement", ctypes.c_int32), ("loop_part_streams", ctypes.c_int32),
                ("y_after_loop", ctypes.c_int32), ("psy_beside", ctypes.c_int32), ("dropin_lookahead", ctypes.c_int32), ("call_hold", ctypes.c_int32), ("dropin_stats", ctypes.c_int32),
                ("abi", ctypes.c_uint32)]


def default_options(**kw):
    o = BatchOptions()
    lib().mp3mi_batch_options_default(ctypes.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


_lib = None


def lib():
    """Load libmp3mi.so (once).  Raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Mp3miError("libmp3mi.so not found at %s -- run __graft_entry__.build()" % LIB_PATH)
        # PyTorch bundles its own HIP runtime.  Load torch first (when it is installed) so that this
        # process has ONE runtime whatever the caller's import order: libmp3mi.so then binds to the
        # already loaded libamdhip64 instead of bringing in a second one that sees no device.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = ctypes.CDLL(LIB_PATH)
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
        L.mp3mi_batch_encode.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p,
                                         ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_batch_sync.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_encode_host_async.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_batch_host_io_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.mp3mi_host_alloc.argtypes = [ctypes.c_size_t]
        L.mp3mi_host_alloc.restype = ctypes.c_void_p
        L.mp3mi_host_free.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_last_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float),
                                              ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_int)]
        L.mp3mi_batch_total_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double),
                                               ctypes.POINTER(ctypes.c_long), ctypes.POINTER(ctypes.c_long)]
        L.mp3mi_synth_pcm.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
        L.mp3mi_synth_pcm_device.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_long, ctypes.c_int, ctypes.c_int, ctypes.c_uint32, ctypes.c_uint32]
        L.mp3mi_batch_set_test_flags.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.mp3mi_batch_encode_next.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_batch_flush.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_batch_reset.argtypes = [ctypes.c_void_p]
        L.mp3mi_batch_set_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_batch_set_error_protection.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_batch_set_header.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        # Layers I and II (include/mp3mi_l12.h)
        L.mp3mi_l12_batch_create.argtypes = [ctypes.POINTER(ctypes.c_void_p)] + [ctypes.c_int] * 4 + [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint]
        L.mp3mi_l12_batch_destroy.argtypes = [ctypes.c_void_p]
        L.mp3mi_l12_batch_set_mode.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_l12_batch_set_error_protection.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_l12_batch_set_header.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
        L.mp3mi_l12_batch_set_test_flags.argtypes = [ctypes.c_void_p, ctypes.c_uint]
        L.mp3mi_l12_batch_out_stride.restype = ctypes.c_size_t
        L.mp3mi_l12_batch_out_stride.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.mp3mi_l12_batch_encode.argtypes = [ctypes.c_void_p] * 3 + [ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
        L.mp3mi_l12_batch_sync.argtypes = [ctypes.c_void_p]
        L.mp3mi_l12_batch_total_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_long)]
        L.mp3mi_l12_batch_kernel_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        L.mp3mi_version.restype = ctypes.c_char_p
        L.mp3mi_source_hash.restype = ctypes.c_char_p
        _lib = L
    return _lib


def frame_bytes(rate_hz, kbps):
    """slots per frame, never padded (reference: src/musicin.c:562-581)"""
    return int((1152.0 / (rate_hz / 1000.0)) * (kbps / 8.0))


def synth_pcm_device(pcm, n_per_ch, channels, rate_hz, stream0=0, seed=0x6D70336D):
    """Fill the int16 cuda tensor pcm [S, n_per_ch*channels] with the deterministic synthetic PCM of streams
    stream0 .. stream0+S-1 (csrc/pcm_synth_core.h: the same bytes as the host's mp3mi_synth_pcm)."""
    assert pcm.is_cuda and pcm.is_contiguous() and pcm.shape[1] == n_per_ch * channels
    rc = lib().mp3mi_synth_pcm_device(pcm.data_ptr(), pcm.shape[0], n_per_ch, channels, rate_hz, stream0, seed)
    if rc != 0:
        raise Mp3miError("mp3mi_synth_pcm_device failed with %d" % rc)


class Batch:
    """Batched encoder over device memory handed in as torch tensors (device pointers)."""

    def __init__(self, n_streams, rate_hz, channels, kbps, max_frames, options=None):
        """options: a BatchOptions (mp3mi_batch_create_ex: no environment variable is read), or None
        (mp3mi_batch_create: the defaults, overridden by the MP3MI_* variables of tools/ and tests/)"""
        import numpy as np
        self.L = lib()
        self.n_streams, self.rate_hz, self.channels, self.max_frames = n_streams, rate_hz, channels, max_frames
        self.h = ctypes.c_void_p()
        arr = None if isinstance(kbps, int) else np.ascontiguousarray(kbps, dtype=np.int32)
        karg, kall = (None, kbps) if arr is None else (arr.ctypes.data, 0)
        if options is None:
            rc = self.L.mp3mi_batch_create(ctypes.byref(self.h), n_streams, rate_hz, channels, karg, kall, max_frames)
        else:
            rc = self.L.mp3mi_batch_create_ex(ctypes.byref(self.h), n_streams, rate_hz, channels, karg, kall, max_frames, ctypes.byref(options))
        if rc != 0:
            raise Mp3miError("mp3mi_batch_create failed with %d" % rc)

    def set_test_flags(self, flags):
        """force the exact tier of the two-tier decisions (include/mp3mi.h, MP3MI_TEST_*)"""
        rc = self.L.mp3mi_batch_set_test_flags(self.h, flags)
        if rc != 0:
            raise Mp3miError("mp3mi_batch_set_test_flags failed with %d" % rc)

    def out_stride(self, n_frames):
        return self.L.mp3mi_batch_out_stride(self.h, n_frames)

    def encode(self, pcm, n_frames, out, out_len):
        """pcm: int16 cuda tensor [S, n_frames*1152*C]; out: uint8 cuda [S, stride]; out_len: int32 cuda [S]."""
        assert pcm.is_cuda and out.is_cuda and out_len.is_cuda and pcm.is_contiguous() and out.is_contiguous()
        rc = self.L.mp3mi_batch_encode(self.h, pcm.data_ptr(), n_frames, out.data_ptr(), out.shape[1], out_len.data_ptr())
        if rc != 0:
            raise Mp3miError("mp3mi_batch_encode failed with %d" % rc)

    def _check(self, rc, what):
        if rc != 0:
            raise Mp3miError("%s failed with %d" % (what, rc))

    def encode_host_async(self, pcm, n_frames, out, out_len):
        """Host tensors in and out (pinned: torch's pin_memory), PCM up and file bytes down chunk by chunk beside the
        kernels (mp3mi_batch_encode_host_async); the results are there after sync()."""
        assert not pcm.is_cuda and not out.is_cuda and not out_len.is_cuda and pcm.is_contiguous() and out.is_contiguous()
        self._check(self.L.mp3mi_batch_encode_host_async(self.h, pcm.data_ptr(), n_frames, out.data_ptr(), out.shape[1], out_len.data_ptr()),
                    "mp3mi_batch_encode_host_async")

    def host_io_stats(self):
        """{h2d_bytes, d2h_bytes, h2d_ms, d2h_ms, calls} over the host-buffer calls so far (waits for them)"""
        class S(ctypes.Structure):
            _fields_ = [("h2d_bytes", ctypes.c_double), ("d2h_bytes", ctypes.c_double), ("h2d_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double), ("calls", ctypes.c_long)]
        st = S()
        self._check(self.L.mp3mi_batch_host_io_stats(self.h, ctypes.byref(st)), "mp3mi_batch_host_io_stats")
        return {k: getattr(st, k) for k, _ in S._fields_}

    def encode_next(self, pcm, n_frames, out, out_len):
        """Streaming: the NEXT n_frames frames of every stream (pcm holds only these); out / out_len receive the file
        bytes that became final with this call.  Concatenate them call after call, then flush()."""
        assert pcm.is_cuda and out.is_cuda and out_len.is_cuda and pcm.is_contiguous() and out.is_contiguous()
        self._check(self.L.mp3mi_batch_encode_next(self.h, pcm.data_ptr(), n_frames, out.data_ptr(), out.shape[1], out_len.data_ptr()),
                    "mp3mi_batch_encode_next")

    def flush(self, out, out_len):
        """III_FlushBitstream + close_bit_stream_w: the remaining bytes of every stream; ends the streams."""
        self._check(self.L.mp3mi_batch_flush(self.h, out.data_ptr(), out.shape[1], out_len.data_ptr()), "mp3mi_batch_flush")

    def reset(self):
        self._check(self.L.mp3mi_batch_reset(self.h), "mp3mi_batch_reset")

    def set_mode(self, mode):
        """0 stereo, 2 dual channel, 3 mono (the reference's -m s|d|m); joint stereo is refused as in the reference"""
        self._check(self.L.mp3mi_batch_set_mode(self.h, mode), "mp3mi_batch_set_mode")

    def set_error_protection(self, on):
        """the reference's -e: protection bit cleared, zero CRC word after the header"""
        self._check(self.L.mp3mi_batch_set_error_protection(self.h, 1 if on else 0), "mp3mi_batch_set_error_protection")

    def set_header(self, copyright=0, original=0, emphasis=0):
        self._check(self.L.mp3mi_batch_set_header(self.h, copyright, original, emphasis), "mp3mi_batch_set_header")

    def sync(self):
        rc = self.L.mp3mi_batch_sync(self.h)
        if rc == ERR_REFERENCE_ABORT:
            raise ReferenceAborts("the reference would have died on at least one stream's input: out_len 0 there (stream_status())")
        if rc != 0:
            raise Mp3miError("mp3mi_batch_sync failed with %d" % rc)

    def stream_status(self):
        """per stream 0, or code | frame << 8 (include/mp3mi.h, MP3MI_STREAM_ABORT_*); waits for the work issued"""
        import numpy as np
        st = np.zeros(self.n_streams, np.int32)
        rc = self.L.mp3mi_batch_stream_status(self.h, st.ctypes.data)
        if rc < 0:
            raise Mp3miError("mp3mi_batch_stream_status failed with %d" % rc)
        return st

    def last_timing(self):
        a, b, n = ctypes.c_float(), ctypes.c_float(), ctypes.c_int()
        rc = self.L.mp3mi_batch_last_timing(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n))
        if rc != 0:
            raise Mp3miError("mp3mi_batch_last_timing failed with %d (no encode call yet?)" % rc)
        return a.value, b.value, n.value

    def total_timing(self):
        """(ms inside the loop kernel, ms inside all kernels, loop kernel launches, calls) summed over every encode
        call of this batch so far; waits for them"""
        a, b, n, k = ctypes.c_double(), ctypes.c_double(), ctypes.c_long(), ctypes.c_long()
        rc = self.L.mp3mi_batch_total_timing(self.h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(n), ctypes.byref(k))
        if rc != 0:
            raise Mp3miError("mp3mi_batch_total_timing failed with %d" % rc)
        return a.value, b.value, n.value, k.value

    def close(self):
        if self.h:
            self.L.mp3mi_batch_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


L12_KERNELS = ("k_fft12", "k12_psy", "k_filter", "k12_alloc")
L12_FRAME_SAMPLES = {1: 384, 2: 1152}


def frame_bytes_l12(layer, rate_hz, kbps):
    """slots per frame times the slot size, never padded (reference: src/musicin.c:562-581)"""
    if layer == 1:
        return 4 * int((384.0 / (rate_hz / 1000.0)) * (kbps / 32.0))
    return int((1152.0 / (rate_hz / 1000.0)) * (kbps / 8.0))


class BatchL12:
    """Layer I / II batched encoder (include/mp3mi_l12.h) over device memory handed in as torch tensors."""

    def __init__(self, layer, n_streams, rate_hz, channels, kbps, max_frames, mode=None, error_protection=False, scratch_mb=0):
        import numpy as np
        self.L = lib()
        self.layer, self.n_streams, self.rate_hz, self.channels, self.max_frames = layer, n_streams, rate_hz, channels, max_frames
        self.h = ctypes.c_void_p()
        arr = None if isinstance(kbps, int) else np.ascontiguousarray(kbps, dtype=np.int32)
        karg, kall = (None, kbps) if arr is None else (arr.ctypes.data, 0)
        rc = self.L.mp3mi_l12_batch_create(ctypes.byref(self.h), layer, n_streams, rate_hz, channels, karg, kall, max_frames, scratch_mb)
        if rc != 0:
            raise Mp3miError("mp3mi_l12_batch_create failed with %d" % rc)
        if mode is not None:
            self._check(self.L.mp3mi_l12_batch_set_mode(self.h, mode), "mp3mi_l12_batch_set_mode")
        if error_protection:
            self._check(self.L.mp3mi_l12_batch_set_error_protection(self.h, 1), "mp3mi_l12_batch_set_error_protection")

    def _check(self, rc, what):
        if rc != 0:
            raise Mp3miError("%s failed with %d" % (what, rc))

    def out_stride(self, n_frames):
        return self.L.mp3mi_l12_batch_out_stride(self.h, n_frames)

    def encode(self, pcm, n_frames, out, out_len, n_samples=None):
        assert pcm.is_cuda and out.is_cuda and out_len.is_cuda and pcm.is_contiguous() and out.is_contiguous()
        self._check(self.L.mp3mi_l12_batch_encode(self.h, pcm.data_ptr(), n_samples.data_ptr() if n_samples is not None else None,
                                                  n_frames, out.data_ptr(), out.stride(0), out_len.data_ptr()), "mp3mi_l12_batch_encode")

    def sync(self):
        self._check(self.L.mp3mi_l12_batch_sync(self.h), "mp3mi_l12_batch_sync")

    def total_timing(self):
        """(ms inside all kernels, calls) since create; waits for the work issued so far"""
        a, k = ctypes.c_double(), ctypes.c_long()
        self._check(self.L.mp3mi_l12_batch_total_timing(self.h, ctypes.byref(a), ctypes.byref(k)), "mp3mi_l12_batch_total_timing")
        return a.value, k.value

    def kernel_timing(self):
        """{kernel: (ms, launches)} since create (HIP events around every launch)"""
        ms, n = (ctypes.c_double * 4)(), (ctypes.c_long * 4)()
        self._check(self.L.mp3mi_l12_batch_kernel_timing(self.h, ms, n), "mp3mi_l12_batch_kernel_timing")
        return {L12_KERNELS[i]: (ms[i], n[i]) for i in range(4)}

    def close(self):
        if self.h:
            self.L.mp3mi_l12_batch_destroy(self.h)
            self.h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

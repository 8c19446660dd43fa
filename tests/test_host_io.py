"""mp3mi_batch_encode_host_async (include/mp3mi.h): host buffers in and out, the PCM uploaded and the file bytes downloaded
chunk by chunk beside the kernels -- the batched counterpart of the reference driver's get_audio / fwrite
(/root/reference/src/encode.c:123-269).  The bytes must be those of the device-pointer path, i.e. the oracle's."""
import ctypes

import numpy as np
import pytest

from mp3common import BatchRun


def host_call(mp, run, pcm, nf):
    L = mp.lib
    L.mp3mi_batch_encode_host_async.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    out = np.full((run.S, run.stride), 0x5A, np.uint8)
    lens = np.full(run.S, 0xDEADBEEF, np.uint32)
    rc = L.mp3mi_batch_encode_host_async(run.b, pcm.ctypes.data, nf, out.ctypes.data, run.stride, lens.ctypes.data)
    assert rc == 0, rc
    return out, lens


def test_host_calls_back_to_back_emulated(emu, oracle, monkeypatch):
    """three host calls on one batch without a sync in between (the third reuses the first one's device buffers), a
    different PCM each, three chunks per call, mixed bitrates: every call's bytes are the oracle's; the statistics count
    what crossed"""
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "2")
    S, nf, rate, ch, kbps = 3, 5, 48000, 2, [64, 320, 128]
    run = BatchRun(emu, S, rate, ch, kbps, nf, stream0=7)
    try:
        pcms = [np.ascontiguousarray(np.stack([emu.synth(nf * 1152, ch, rate, 100 * k + s) for s in range(S)]), dtype=np.int16) for k in range(3)]
        res = [host_call(emu, run, p, nf) for p in pcms]
        assert emu.lib.mp3mi_batch_sync(run.b) == 0
        for k in range(3):
            out, lens = res[k]
            for s in range(S):
                assert out[s, :lens[s]].tobytes() == oracle.encode(pcms[k][s], rate, kbps[s], ch)[0], (k, s)

        class St(ctypes.Structure):
            _fields_ = [("h2d_bytes", ctypes.c_double), ("d2h_bytes", ctypes.c_double), ("h2d_ms", ctypes.c_double), ("d2h_ms", ctypes.c_double), ("calls", ctypes.c_long)]
        st = St()
        assert emu.lib.mp3mi_batch_host_io_stats(run.b, ctypes.byref(st)) == 0
        assert st.calls == 3 and st.h2d_bytes == 3 * pcms[0].nbytes and st.d2h_bytes >= sum(int(r[1].sum()) for r in res)
        # and a device-pointer call on the same batch afterwards is undisturbed
        out, lens = run.encode()
        for s in range(S):
            assert out[s, :lens[s]].tobytes() == oracle.encode(run.pcm_of(s), rate, kbps[s], ch)[0]
    finally:
        run.close()


@pytest.mark.gpu
@pytest.mark.parametrize("S,nf,rate,ch,kbps", [(4096, 96, 44100, 2, 128), (1536, 60, 48000, 2, "mix"), (2048, 40, 32000, 1, 64)])
def test_host_path_equals_device_path_gpu(product, oracle, S, nf, rate, ch, kbps):
    """a full-width batch through the host path (page-locked buffers from mp3mi_host_alloc, two calls in flight) gives, for
    EVERY stream, the bytes of the device-pointer path; a sample of them is compared with the oracle as well"""
    L = product.lib
    L.mp3mi_host_alloc.restype = ctypes.c_void_p
    L.mp3mi_host_alloc.argtypes = [ctypes.c_size_t]
    L.mp3mi_host_free.argtypes = [ctypes.c_void_p]
    kb = [[64, 96, 128, 192, 256, 320][s % 6] for s in range(S)] if kbps == "mix" else kbps
    run = BatchRun(product, S, rate, ch, kb, nf, stream0=11)
    bufs = []
    try:
        dev_out, dev_lens = run.encode()
        pcm = run.mem.download(run.d_pcm, (S, nf * 1152 * ch), np.int16)

        def pinned(nbytes, dtype, shape):
            p = L.mp3mi_host_alloc(nbytes)
            assert p
            bufs.append(p)
            return np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(p)).view(dtype).reshape(shape)

        h_pcm = pinned(pcm.nbytes, np.int16, pcm.shape)
        h_pcm[:] = pcm
        outs = []
        for k in range(2):  # two calls in flight
            h_out = pinned(S * run.stride, np.uint8, (S, run.stride))
            h_len = pinned(4 * S, np.uint32, (S,))
            h_out[:] = 0xA5
            assert L.mp3mi_batch_encode_host_async(run.b, h_pcm.ctypes.data, nf, h_out.ctypes.data, run.stride, h_len.ctypes.data) == 0
            outs.append((h_out, h_len))
        assert L.mp3mi_batch_sync(run.b) == 0
        for h_out, h_len in outs:
            assert np.array_equal(h_len, dev_lens)
            for s in range(S):
                n = int(dev_lens[s])
                assert np.array_equal(h_out[s, :n], dev_out[s, :n]), "stream %d differs between the host and the device path" % s
        kl = [kb] * S if np.isscalar(kb) else kb
        for s in sorted(set(np.linspace(0, S - 1, 12).astype(int).tolist())):
            assert outs[0][0][s, :int(dev_lens[s])].tobytes() == oracle.encode(pcm[s], rate, kl[s], ch)[0]
    finally:
        run.close()
        for p in bufs:
            L.mp3mi_host_free(p)

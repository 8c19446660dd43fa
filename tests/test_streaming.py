"""Streaming continuation: a stream encoded in several mp3mi_batch_encode_next calls and flushed
must give, concatenated, the bytes of the same stream encoded in one call -- which are the oracle's.  What is
carried from call to call: psychoacoustic state, PCM history of the filterbank and the FFT window, MDCT overlap
(recomputed from that history), the bit reservoir, and the formatted bytes whose slots are still open
(/root/reference/src/musicin.c:585-805, src/formatBitstream.c:52-120)."""
import numpy as np
import pytest

from mp3common import BatchRun


def streaming_case(mp, oracle, rate, ch, kbps, S, pieces, stream0=0, chunk=None, monkeypatch=None):
    nf = sum(pieces)
    if chunk and monkeypatch:
        monkeypatch.setenv("MP3MI_CHUNK_FRAMES", str(chunk))
    run = BatchRun(mp, S, rate, ch, kbps, nf, stream0=stream0)
    try:
        kb = [kbps] * S if np.isscalar(kbps) else list(kbps)
        ref = [oracle.encode(run.pcm_of(s), rate, kb[s], ch)[0] for s in range(S)]
        out, lens = run.encode()
        whole = [out[s, :lens[s]].tobytes() for s in range(S)]
        assert whole == ref
        got = run.encode_streaming(pieces)
        for s in range(S):
            assert got[s] == ref[s], "stream %d: streamed bytes differ (len %d vs %d)" % (s, len(got[s]), len(ref[s]))
        # and once more on the same batch: flush left fresh streams behind
        got = run.encode_streaming(pieces[::-1])
        for s in range(S):
            assert got[s] == ref[s], "second pass, stream %d" % s
    finally:
        run.close()


def test_streaming_equals_one_call_emulated(emu, oracle, monkeypatch):
    streaming_case(emu, oracle, 44100, 2, 128, 2, [3, 1, 4, 2], stream0=30, chunk=3, monkeypatch=monkeypatch)


def test_streaming_low_bitrate_long_back_pointer_emulated(emu, oracle):
    """32 kbps at 48 kHz: 60-byte slots, the reservoir reaches back over up to nine frames, so the carried bytes
    span many headers; mixed with a 320 kbps stream in the same batch"""
    streaming_case(emu, oracle, 48000, 2, [32, 320], 2, [2, 5, 1, 1], stream0=44)


def test_streaming_mono_emulated(emu, oracle):
    streaming_case(emu, oracle, 32000, 1, 64, 1, [1, 1, 3], stream0=8)


@pytest.mark.gpu
@pytest.mark.parametrize("rate,ch,kbps,S,pieces", [
    (44100, 2, 128, 8, [100, 1, 150, 7, 125]),      # BASELINE configs[0] length: 383 frames in five calls
    (48000, 2, [32, 64, 128, 320], 4, [9, 30, 2, 40]),
    (32000, 1, 64, 6, [50, 50, 1, 20]),
])
def test_streaming_equals_one_call_gpu(product, oracle, rate, ch, kbps, S, pieces):
    streaming_case(product, oracle, rate, ch, kbps, S, pieces, stream0=0)


@pytest.mark.gpu
def test_streaming_full_chip_batch_gpu(product, oracle):
    """4096 streams in three calls: placement, pacing and chunk overlap active in every call"""
    rate, ch, kbps, S, pieces = 44100, 2, 128, 4096, [10, 6, 8]
    run = BatchRun(product, S, rate, ch, kbps, sum(pieces))
    try:
        out, lens = run.encode()
        got = run.encode_streaming(pieces)
        bad = [s for s in range(S) if got[s] != out[s, :lens[s]].tobytes()]
        assert not bad, "%d streams differ between streamed and whole-file encoding (first %d)" % (len(bad), bad[0])
        for s in (0, 1777, 4095):
            assert got[s] == oracle.encode(run.pcm_of(s), rate, kbps, ch)[0]
    finally:
        run.close()


def unsynced_case(mp, oracle, rate, ch, kbps, S, pieces, n_oracle, chunk, monkeypatch, hold=None):
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", str(chunk))
    if hold is not None:
        monkeypatch.setenv("MP3MI_CALL_HOLD", str(hold))
    run = BatchRun(mp, S, rate, ch, kbps, sum(pieces))
    try:
        kb = [kbps] * S if np.isscalar(kbps) else list(kbps)
        got, whole = run.encode_streaming_unsynced(pieces, whole_first=True)
        bad = [s for s in range(S) if got[s] != whole[s]]
        assert not bad, "%d streams differ between the streamed and the whole-file call (first %d)" % (len(bad), bad[0])
        for s in sorted(set(int(x) for x in np.linspace(0, S - 1, n_oracle))):
            assert whole[s] == oracle.encode(run.pcm_of(s), rate, kb[s], ch)[0], "stream %d" % s
    finally:
        run.close()


def test_back_to_back_calls_without_sync_emulated(emu, oracle, monkeypatch):
    unsynced_case(emu, oracle, 44100, 2, 128, 2, [3, 2, 4], 2, 2, monkeypatch)


@pytest.mark.gpu
@pytest.mark.parametrize("S,pieces,chunk,hold", [
    (4096, [5, 3, 6, 1, 4], 2, None),   # odd and even numbers of chunks per call: the double-buffer slots change sides between calls
    (600, [40, 7, 33], 9, None),
    (4096, [5, 3, 6, 1, 4], 2, 0),      # options.call_hold off: a call's last k_loop does not wait for the next call's transforms
    (6000, [4, 1, 3], 2, 1),            # two parts per chunk: the held launch is the last part's, the joining item the first part's
])
def test_back_to_back_calls_without_sync_gpu(product, oracle, monkeypatch, S, pieces, chunk, hold):
    """A whole-file call, streaming calls and the flush issued without a sync between them: every call's feed-forward
    kernels overlap with the loop kernels of the call before, and a call's LAST k_loop is held on the device until the
    next call's first transforms are through (encode_impl, k_hold; calls of one chunk chain hold to hold).  The bytes must
    be those of synchronised calls -- the oracle's."""
    unsynced_case(product, oracle, 44100, 2, [(96, 128, 160, 128)[s % 4] for s in range(S)], S, pieces, 12, chunk, monkeypatch, hold)


@pytest.mark.gpu
def test_a_held_call_ends_without_a_successor(product, oracle):
    """the hold in front of a call's last k_loop is let go by whatever waits for the call (sync here, then status and
    destroy with another call in flight) -- and runs out by itself (20 ms) for a caller who polls a HIP stream of his own"""
    import time
    S, nf = 64, 6
    run = BatchRun(product, S, 44100, 2, 128, nf)
    try:
        L = product.lib
        t0 = time.perf_counter()
        assert L.mp3mi_batch_encode(run.b, run.d_pcm, nf, run.d_out, run.stride, run.d_len) == 0
        assert L.mp3mi_batch_sync(run.b) == 0
        assert time.perf_counter() - t0 < 5.0
        lens = run.mem.download(run.d_len, (S,), np.uint32)
        out = run.mem.download(run.d_out, (S, run.stride), np.uint8)
        assert out[3, :lens[3]].tobytes() == oracle.encode(run.pcm_of(3), 44100, 128, 2)[0]
        assert L.mp3mi_batch_encode(run.b, run.d_pcm, nf, run.d_out, run.stride, run.d_len) == 0
        assert all(x == 0 for x in run.status())  # (waits for the call)
        assert L.mp3mi_batch_encode(run.b, run.d_pcm, nf, run.d_out, run.stride, run.d_len) == 0
    finally:
        run.close()  # destroy with a held call in flight


@pytest.mark.gpu
def test_a_foreign_device_wide_sync_waits_out_the_hold_and_no_more(product, oracle):
    """An integrator who ends a call with hipDeviceSynchronize (torch.cuda.synchronize) instead of mp3mi_batch_sync does not
    let go of the hold in front of the call's last k_loop (mp3mi.h, options.call_hold): that launch then starts when the hold's
    bound runs out -- 0.4 ms per frame of a chunk, at least 20 ms, at most 200 -- and everything is still correct.  Measured
    here: the same call ended by the library's own sync and by the device-wide one; the difference stays below the bound
    (plus a margin for the box), and the bytes are the oracle's both times."""
    import ctypes
    import time
    S, nf = 1024, 48  # one chunk of 48 frames: the bound is 20 ms
    run = BatchRun(product, S, 44100, 2, 128, nf)
    hip = ctypes.CDLL("libamdhip64.so")
    try:
        L = product.lib
        assert L.mp3mi_batch_encode(run.b, run.d_pcm, nf, run.d_out, run.stride, run.d_len) == 0 and L.mp3mi_batch_sync(run.b) == 0  # warm
        t = []
        for how in ("library", "device-wide") * 3:
            t0 = time.perf_counter()
            assert L.mp3mi_batch_encode(run.b, run.d_pcm, nf, run.d_out, run.stride, run.d_len) == 0
            if how == "library":
                assert L.mp3mi_batch_sync(run.b) == 0
            else:
                assert hip.hipDeviceSynchronize() == 0
            t.append((how, (time.perf_counter() - t0) * 1e3))
            if how != "library":
                assert L.mp3mi_batch_sync(run.b) == 0  # (the call is over: nothing left to wait for; keeps the batch's bookkeeping in step)
            lens = run.mem.download(run.d_len, (S,), np.uint32)
            out = run.mem.download(run.d_out, (S, run.stride), np.uint8)
            for s in (0, 511, 1023):
                assert out[s, :lens[s]].tobytes() == oracle.encode(run.pcm_of(s), 44100, 128, 2)[0], (how, s)
        lib_ms = min(ms for how, ms in t if how == "library")
        dev_ms = min(ms for how, ms in t if how != "library")
        print("one call of %d x %d frames ended by mp3mi_batch_sync: %.1f ms, by hipDeviceSynchronize: %.1f ms (the hold's bound: 20 ms)" % (S, nf, lib_ms, dev_ms))
        assert dev_ms < lib_ms + 20.0 + 10.0
    finally:
        run.close()


def test_flush_without_frames_and_argument_errors_emulated(emu):
    """a flush right after create or after a whole-file encode delivers nothing; streaming calls check their arguments"""
    import ctypes
    from mp3common import BatchRun
    run = BatchRun(emu, 2, 44100, 2, 128, 3, stream0=1)
    try:
        L = emu.lib
        assert L.mp3mi_batch_flush(run.b, run.d_out, run.stride, run.d_len) == 0 and L.mp3mi_batch_sync(run.b) == 0
        assert (run.mem.download(run.d_len, (2,), np.uint32) == 0).all()
        out, lens = run.encode()
        assert L.mp3mi_batch_flush(run.b, run.d_out, run.stride, run.d_len) == 0 and L.mp3mi_batch_sync(run.b) == 0
        assert (run.mem.download(run.d_len, (2,), np.uint32) == 0).all()
        assert L.mp3mi_batch_encode_next(run.b, run.d_pcm, 0, run.d_out, run.stride, run.d_len) == -1      # no frames
        assert L.mp3mi_batch_encode_next(run.b, run.d_pcm, 4, run.d_out, run.stride, run.d_len) == -1      # more than max_frames
        assert L.mp3mi_batch_encode_next(run.b, run.d_pcm, 3, run.d_out, 100, run.d_len) == -1             # row too short
        assert L.mp3mi_batch_flush(run.b, run.d_out, 100, run.d_len) == -1
        assert L.mp3mi_batch_reset(None) == -1
        # abandon a stream half way, start again: the one-call bytes
        assert L.mp3mi_batch_encode_next(run.b, run.d_pcm, 2, run.d_out, run.stride, run.d_len) == 0
        assert L.mp3mi_batch_reset(run.b) == 0
        got = run.encode_streaming([2, 1])
        assert [got[s] for s in range(2)] == [out[s, :lens[s]].tobytes() for s in range(2)]
    finally:
        run.close()

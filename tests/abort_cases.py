"""Inputs the reference dies on (an assertion fails: tests/golden/coverage_notes.json), inside a batch: what the library
must do with them.  The bodies are shared by the emulator tests (tests/test_emu_parity.py) and the device tests
(tests/test_gpu_golden.py): `mp` is the emulated test build or the product library."""
import numpy as np

from golden_util import aborting_cases, case_pcm
from mp3common import pad_frames


def neighbours_case(mp, oracle):
    """one stream of a batch is an input the reference dies on: its file is voided, the other streams' bytes are the
    oracle's, and the status survives until the next reset"""
    from mp3common import BatchRun
    case = [c for c in aborting_cases() if c["name"] == "abort_global_gain"][0]
    bad, nf = pad_frames(case_pcm(case, mp.synth), 2)
    good = mp.synth(nf * 1152, 2, 44100, 77)
    run = BatchRun(mp, 3, 44100, 2, 128, nf, pcm=np.stack([good, bad, good]))
    try:
        out, lens = run.encode(expect_abort=True)
        ref = oracle.encode(good, 44100, 128, 2)[0]
        assert lens[1] == 0 and out[0, :lens[0]].tobytes() == ref and out[2, :lens[2]].tobytes() == ref
        st = run.status()
        assert st[0] == 0 and st[2] == 0 and st[1] & 255 == 1
        assert mp.lib.mp3mi_batch_sync(run.b) == 0  # reported once
    finally:
        run.close()


def host_wrapper_case(mp, oracle):
    """mp3mi_encode_host with an input the reference dies on between two good ones: the call returns
    MP3MI_ERR_REFERENCE_ABORT AND the outputs -- that stream's out_len 0, the others' bytes the oracle's (mp3mi.h)"""
    from mp3common import ERR_REFERENCE_ABORT
    case = [c for c in aborting_cases() if c["name"] == "abort_global_gain"][0]
    bad, nf = pad_frames(case_pcm(case, mp.synth), 2)
    good = mp.synth(nf * 1152, 2, 44100, 78)
    pcm = np.ascontiguousarray(np.stack([good, bad, good]), dtype=np.int16)
    stride = (nf * 418 + 1 + 255) // 256 * 256
    out = np.full((3, stride), 0xAA, np.uint8)
    lens = np.full(3, 0xDEADBEEF, np.uint32)
    rc = mp.lib.mp3mi_encode_host(3, 44100, 2, None, 128, pcm.ctypes.data, nf, out.ctypes.data, stride, lens.ctypes.data)
    assert rc == ERR_REFERENCE_ABORT
    ref = oracle.encode(good, 44100, 128, 2)[0]
    assert lens[1] == 0 and out[0, :lens[0]].tobytes() == ref and out[2, :lens[2]].tobytes() == ref


def streaming_case(mp, oracle):
    """streaming: the sync after the call in which a stream dies returns MP3MI_ERR_REFERENCE_ABORT -- once -- while the
    status says which frame; later calls and the flush deliver nothing for it and do not report it again; the status is
    still readable after the flush"""
    from mp3common import BatchRun, ERR_REFERENCE_ABORT
    case = [c for c in aborting_cases() if c["name"] == "abort_global_gain"][0]
    bad, nf = pad_frames(case_pcm(case, mp.synth), 2)
    good = mp.synth(nf * 1152, 2, 44100, 79)
    run = BatchRun(mp, 2, 44100, 2, 128, nf, pcm=np.stack([good, bad]))
    L = mp.lib
    try:
        whole = np.stack([good, bad])
        got, seen, f0 = b"", [], 0
        frame = case["reference_aborts"]["frame"]
        for nfp in (2, frame - 2 + 1, nf - frame - 1):  # the second call holds the fatal frame
            piece = np.ascontiguousarray(whole[:, f0 * 2304:(f0 + nfp) * 2304])
            d_piece = run.mem.alloc(piece.nbytes)
            run.mem.upload(d_piece, piece)
            assert L.mp3mi_batch_encode_next(run.b, d_piece, nfp, run.d_out, run.stride, run.d_len) == 0
            seen.append(L.mp3mi_batch_sync(run.b))
            st = run.status()
            lens = run.mem.download(run.d_len, (2,), np.uint32)
            out = run.mem.download(run.d_out, (2, run.stride), np.uint8)
            got += out[0, :lens[0]].tobytes()
            if seen[-1] == ERR_REFERENCE_ABORT:
                assert (st[1] & 255, st[1] >> 8) == (1, frame) and st[0] == 0
            if len(seen) >= 2:
                assert lens[1] == 0
            f0 += nfp
        assert seen == [0, ERR_REFERENCE_ABORT, 0], seen
        assert L.mp3mi_batch_flush(run.b, run.d_out, run.stride, run.d_len) == 0
        assert L.mp3mi_batch_sync(run.b) == 0  # (reported already)
        lens = run.mem.download(run.d_len, (2,), np.uint32)
        out = run.mem.download(run.d_out, (2, run.stride), np.uint8)
        got += out[0, :lens[0]].tobytes()
        assert lens[1] == 0 and got == oracle.encode(good, 44100, 128, 2)[0]
        st = run.status()  # the flush reset the streams, their status is kept until the next encode
        assert (st[1] & 255, st[1] >> 8) == (1, frame) and st[0] == 0
        assert L.mp3mi_batch_reset(run.b) == 0  # ... or an explicit reset: "since the last reset" (mp3mi.h)
        st = run.status()
        assert st[0] == 0 and st[1] == 0
    finally:
        run.close()

"""The concurrency contract of include/mp3mi.h ("Threads"): a batch object belongs to one thread at a time, DIFFERENT
batch objects are independent -- created, used and destroyed concurrently, on one device or several.  The only
library-wide state is the construction of the constant tables (csrc/tables_host.cpp), serialised by the library."""
import ctypes
import threading

import numpy as np
import pytest

from mp3common import BatchRun


def test_tables_are_built_consistently_from_many_threads(product):
    """mp3mi_tables_digest (the table build without the device upload) from eight threads at once, all three rates:
    every thread sees the same member hashes as a lone call -- the generator's statics have one user at a time"""
    L = product.lib
    L.mp3mi_tables_digest.restype = ctypes.c_int
    cap = 256

    def digest(ri):
        h = (ctypes.c_uint64 * cap)()
        n = L.mp3mi_tables_digest(ri, h, None, cap)
        assert n > 0, n
        return tuple(h[:n])

    alone = [digest(ri) for ri in range(3)]
    got, errs = {}, []

    def work(t):
        try:
            for k in range(6):
                ri = (t + k) % 3
                got[(t, k)] = (ri, digest(ri))
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e))

    th = [threading.Thread(target=work, args=(t,)) for t in range(8)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs
    assert len(got) == 48 and all(h == alone[ri] for ri, h in got.values())


@pytest.mark.gpu
def test_two_threads_two_batches_one_device(product, oracle):
    """two host threads, each with a batch of its own on the SAME device (different rate, bitrate and size, created
    inside the threads at the same time), encoding several times with the calls of the two interleaving freely --
    whole-file calls and streaming pieces -- : both bit-exact against the oracle every time"""
    L = product.lib
    barrier = threading.Barrier(2)
    errs = []

    def work(t):
        try:
            rate, ch, kbps, S, nf, stream0 = [(44100, 2, 128, 96, 24, 500), (48000, 2, 192, 64, 30, 900)][t]
            barrier.wait()
            run = BatchRun(product, S, rate, ch, kbps, nf, stream0=stream0)  # table build + allocation, concurrently
            try:
                sample = [0, S // 3, S - 1]
                ref = {s: oracle.encode(run.pcm_of(s), rate, kbps, ch)[0] for s in sample}
                for rep in range(3):
                    barrier.wait()  # start every round together: the kernels of the two batches share the device
                    out, lens = run.encode()
                    for s in sample:
                        assert out[s, :lens[s]].tobytes() == ref[s], "thread %d, round %d, stream %d" % (t, rep, s)
                    got = run.encode_streaming([nf // 3, 1, nf - nf // 3 - 1])
                    for s in sample:
                        assert got[s] == ref[s], "thread %d, round %d, stream %d (streamed)" % (t, rep, s)
            finally:
                run.close()
        except Exception as e:  # noqa: BLE001
            errs.append("thread %d: %r" % (t, e))
            try:
                barrier.abort()
            except Exception:  # noqa: BLE001
                pass

    th = [threading.Thread(target=work, args=(t,)) for t in range(2)]
    for x in th:
        x.start()
    for x in th:
        x.join()
    assert not errs, errs

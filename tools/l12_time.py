#!/usr/bin/env python3
"""Times the Layer I / II batch on the device (bench.py --layer N is the reported line; this prints per-call figures).
usage: l12_time.py layer rate kbps mode streams frames [reps] [scratch_mb]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from mp3common import L12Run, Mp3mi  # noqa: E402

layer, rate, kbps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode, S, nf = sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
reps = int(sys.argv[7]) if len(sys.argv) > 7 else 3
scratch = int(sys.argv[8]) if len(sys.argv) > 8 else 0
mp = Mp3mi()
run = L12Run(mp, layer, rate, kbps, mode, n_frames=nf, synth=(S, 0), scratch_mb=scratch)
L = mp.lib
for r in range(reps + 1):
    t = time.time()
    assert L.mp3mi_l12_batch_encode(run.b, run.d_pcm, None, nf, run.d_out, run.stride, run.d_len) == 0
    assert L.mp3mi_l12_batch_sync(run.b) == 0
    dt = time.time() - t
    print("call %d: %.1f ms wall, %.2f M frames/s" % (r, dt * 1e3, S * nf / dt / 1e6), flush=True)
ms, calls = run.kernel_ms()
print("kernel time %.1f ms over %d calls" % (ms, calls))
run.close()

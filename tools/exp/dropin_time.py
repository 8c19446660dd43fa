import os, sys, time, subprocess, tempfile
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from mp3common import Mp3mi, SEED, ROOT
from test_dropin import write_wav, run_cli
mp = Mp3mi()
d = tempfile.mkdtemp()
for n in (44100 // 10, 441000, 441000 * 3):
    pcm = mp.synth(n, 2, 44100, 0, SEED)
    write_wav(os.path.join(d, "a.wav"), pcm, 2, 44100)
    for b in ("encode_dropin", "encode"):
        t0 = time.perf_counter(); run_cli(b, os.path.join(d, "a.wav"), os.path.join(d, "a.mp3"), 44100, 128, False); dt = time.perf_counter() - t0
        print(b, "samples", n, "frames", (n + 1151) // 1152, "seconds %.3f" % dt)

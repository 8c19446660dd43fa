"""Diagnostic: where the wavefronts of the two transform kernels spend their cycles, from a -DMP3MI_FFT_PROFILE build of the
library (make -C mp3-enc-bsd_amd/csrc EXTRA=-DMP3MI_FFT_PROFILE; never the product build).  MP3MI_LIB names the library."""
import ctypes, importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
mp3 = importlib.import_module("mp3-enc-bsd_amd")
sys.argv = [sys.argv[0]]
import bench
torch = bench.load_torch()
S, nf = 4096, 77
w = bench.Workload(mp3, {"streams": S, "frames": nf, "channels": 2, "rate": 44100, "kbps": 128}, torch.device("cuda:0"), 0)
L = mp3.lib()
prof = (ctypes.c_ulonglong * 8)()
w.step(); L.mp3mi_debug_fft_profile(prof)
w.step(); L.mp3mi_debug_fft_profile(prof)
v = np.array(list(prof), dtype=np.float64)
names = ["long: program load", "long: window + register rounds", "long: program", "long: read-out",
         "short: program load", "short: window + register rounds", "short: program", "short: read-out"]
for k in (0, 4):
    tot = v[k:k + 4].sum()
    for n, x in zip(names[k:k + 4], v[k:k + 4]): print("%-34s %6.2f %%   %8.0f cycles per task" % (n, 100 * x / tot, x / (S * nf * 2)))
    print("   total cycles per task per wave: %.0f" % (tot / (S * nf * 2)))

"""Diagnostic: where the wavefronts of k_mdct spend their cycles, from a -DMP3MI_MDCT_PROFILE build of the library
(make -C mp3-enc-bsd_amd/csrc EXTRA=-DMP3MI_MDCT_PROFILE; never the product build).  MP3MI_LIB names the library."""
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
w.step(); L.mp3mi_debug_mdct_profile(prof)
w.step(); L.mp3mi_debug_mdct_profile(prof)
v = np.array(list(prof), dtype=np.float64)
names = ["run set-up (the first previous granule)", "waiting for the granule's samples", "transform", "barrier + alias butterflies", "the loop's stateless head", "stores + barrier"]
tot = v.sum()
for n, x in zip(names, v): print("%-44s %6.2f %%   %8.0f cycles per granule pair" % (n, 100 * x / tot, x / (S * nf * 2)))
print("   total cycles per granule pair per wave: %.0f" % (tot / (S * nf * 2)))

"""Diagnostic: phase shares of k_fft from a -DMP3MI_FFT_PROFILE build (never the product build)."""
import ctypes, importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mp3 = importlib.import_module("mp3-enc-bsd_amd")
sys.argv = [sys.argv[0]]
import bench
S, nf = 4096, 48
dev = torch.device("cuda:0")
b = mp3.Batch(S, 44100, 2, 128, nf)
pcm = bench.synth_on_device(dev, S, nf * 1152, 2, 44100, 0)
out = torch.zeros((S, b.out_stride(nf)), dtype=torch.uint8, device=dev); ln = torch.zeros(S, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
L = mp3.lib()
prof = (ctypes.c_ulonglong * 8)()
b.encode(pcm, nf, out, ln); b.sync(); L.mp3mi_debug_fft_profile(prof)
b.encode(pcm, nf, out, ln); b.sync(); L.mp3mi_debug_fft_profile(prof)
v = np.array(list(prof), dtype=np.float64)
names = ["window", "long fft", "long bins", "short fft", "short energies", "short phases (atan2)", "cw (sincos)", "-"]
for n, x in zip(names, v): print("%-26s %6.2f %%   %.3g cycles/granule" % (n, 100 * x / v.sum(), x / (S * nf * 2)))
print("total cycles per granule per wave: %.3g" % (v.sum() / (S * nf * 2)), b.last_timing())

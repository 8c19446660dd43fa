import ctypes, importlib, sys, os
sys.path.insert(0, "/root/repo")
import torch
mp3 = importlib.import_module("mp3-enc-bsd_amd")
L = mp3.lib()
S, nf = 4096, 77
dev = torch.device("cuda:0")
b = mp3.Batch(S, 44100, 2, 128, nf)
pcm = torch.empty((S, nf * 1152 * 2), dtype=torch.int16, device=dev)
mp3.synth_pcm_device(pcm, nf * 1152, 2, 44100, stream0=0)
out = torch.zeros((S, b.out_stride(nf)), dtype=torch.uint8, device=dev); ln = torch.zeros(S, dtype=torch.int32, device=dev)
b.encode(pcm, nf, out, ln); b.sync()
a, n = ctypes.c_int(), ctypes.c_int()
L.mp3mi_batch_debug_cw_fixups.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
print("rc", L.mp3mi_batch_debug_cw_fixups(b.h, ctypes.byref(a), ctypes.byref(n)), "records listed for the second tier:", a.value, "of", n.value)

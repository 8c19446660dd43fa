"""Diagnostic: phase shares of k_loop from a -DMP3MI_LOOP_PROFILE build (never the product build)."""
import ctypes, importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
mp3 = importlib.import_module("mp3-enc-bsd_amd")
S, nf = 4096, (int(sys.argv[1]) if len(sys.argv) > 1 else 48)
dev = torch.device("cuda:0")
b = mp3.Batch(S, 44100, 2, 128, nf)
pcm = torch.empty((S, nf * 1152 * 2), dtype=torch.int16, device=dev)
mp3.synth_pcm_device(pcm, nf * 1152, 2, 44100, stream0=0)
out = torch.zeros((S, b.out_stride(nf)), dtype=torch.uint8, device=dev); ln = torch.zeros(S, dtype=torch.int32, device=dev)
torch.cuda.synchronize()
L = mp3.lib()
prof = (ctypes.c_ulonglong * 8)()
cbp = (ctypes.c_ulonglong * 8)()
b.encode(pcm, nf, out, ln); b.sync(); L.mp3mi_debug_loop_profile(prof); L.mp3mi_debug_cb_profile(cbp)
b.encode(pcm, nf, out, ln); b.sync(); L.mp3mi_debug_loop_profile(prof); L.mp3mi_debug_cb_profile(cbp)
v = np.array(list(prof), dtype=np.float64)
names = ["setup/load", "other(part2,search ctl)", "quantize", "count_bits", "calc_noise", "preemph+amp", "loop_break+scale", "tail"]
for n, x in zip(names, v): print("%-26s %6.2f %%   %.3g cycles/(gr,ch)" % (n, 100 * x / v.sum(), x / (S * nf * 4)))
cv = np.array(list(cbp), dtype=np.float64)
for n, x in zip(["cb: runlen+count1", "cb: subdivide", "cb: region max+reduce", "cb: desc+walks", "cb: reduce+pick"], cv):
    print("  %-24s %6.2f %% of count_bits   %.3g cycles/(gr,ch)" % (n, 100 * x / max(cv.sum(), 1), x / (S * nf * 4)))
print("total cycles per (gr,ch) per wave: %.3g" % (v.sum() / (S * nf * 4)), b.last_timing())

w = (ctypes.c_ulonglong * (2 * S))()
L.mp3mi_debug_loop_waves(w, S)
a = np.array(list(w), dtype=np.uint64).reshape(S, 2)
cyc = a[:, 0].astype(np.float64)
hw = a[:, 1] & np.uint64(0xffffffff)
xcc = (a[:, 1] >> np.uint64(32)).astype(np.int64)
# HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64); cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
print("per-stream cycles: mean %.4g  p50 %.4g  p90 %.4g  p99 %.4g  max %.4g  (max/mean %.3f)" % (cyc.mean(), np.percentile(cyc, 50), np.percentile(cyc, 90), np.percentile(cyc, 99), cyc.max(), cyc.max() / cyc.mean()))
u, inv = np.unique(key, return_inverse=True)
per = np.bincount(inv, weights=cyc); cnt = np.bincount(inv)
print("SIMDs seen %d, waves/SIMD min %d max %d; per-SIMD summed cycles mean %.4g max %.4g (max/mean %.3f)" % (len(u), cnt.min(), cnt.max(), per.mean(), per.max(), per.max() / per.mean()))
by = [cyc[(np.arange(S) // 3) % 4 == k].mean() for k in range(4)]
print("mean cycles by noise class:", ["%.4g" % x for x in by])

st = (ctypes.c_ulonglong * S)()
L.mp3mi_debug_loop_starts(st, S)
st = np.array(list(st), dtype=np.float64); st -= st.min()
print("wave start spread (s_memrealtime ticks, 100 MHz): p50 %.0f p90 %.0f p99 %.0f max %.0f" % (np.percentile(st, 50), np.percentile(st, 90), np.percentile(st, 99), st.max()))

wk = (ctypes.c_ulonglong * S)()
L.mp3mi_debug_loop_work(wk, S)
wk = np.array(list(wk), dtype=np.uint64)
work = (wk & np.uint64((1 << 20) - 1)).astype(np.float64); tend = (wk >> np.uint64(20)).astype(np.float64)
tend -= tend.min()
perw = np.bincount(inv, weights=work)
print("work units per stream: mean %.0f max %.0f (max/mean %.3f); per-SIMD work sum max/mean %.3f min/mean %.3f" % (work.mean(), work.max(), work.max() / work.mean(), perw.max() / perw.mean(), perw.min() / perw.mean()))
print("stream end times (100 MHz ticks after the first end): p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(tend, [10, 50, 90, 100])))
print("corr(work, cycles) = %.3f" % np.corrcoef(work, cyc)[0, 1])

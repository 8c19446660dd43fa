#!/usr/bin/env python3
"""Scheduling-independence check on the GPU: the full 4096-stream batch encoded several times (and once in
a child process with the gate and the placement switched off, i.e. under a different schedule) must give
bit-identical output buffers -- a race between the two streams' kernels would show up here."""
import hashlib, importlib, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

def run(S=4096, nf=96):
    mp3 = importlib.import_module("mp3-enc-bsd_amd")
    dev = torch.device("cuda:0")
    b = mp3.Batch(S, 44100, 2, 128, nf)
    pcm = bench.synth_on_device(dev, S, nf * 1152, 2, 44100, 0)
    digests = []
    for rep in range(3):
        out = torch.zeros((S, b.out_stride(nf)), dtype=torch.uint8, device=dev)
        ln = torch.zeros(S, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        b.encode(pcm, nf, out, ln); b.sync()
        digests.append(hashlib.sha256(out.cpu().numpy().tobytes() + ln.cpu().numpy().tobytes()).hexdigest())
    b.close()
    return digests

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print(run()[0])
    else:
        d = run()
        env = dict(os.environ, MP3MI_NO_GATE="1", MP3MI_NO_PLACE="1", MP3MI_CHUNK_FRAMES="17")
        other = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True).stdout.strip().split("\n")[-1]
        print("repeat digests equal:", len(set(d)) == 1, "| different schedule equal:", other == d[0])
        sys.exit(0 if (len(set(d)) == 1 and other == d[0]) else 1)

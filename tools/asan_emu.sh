#!/bin/bash
# CPU sanitizer run of the kernel sources (GPU AddressSanitizer is not available on this pool): builds the wave-emulated
# test library with -fsanitize=address,undefined and runs three small configurations through whole-file encode,
# streaming and all exact tiers.  Test infrastructure only.  Usage: tools/asan_emu.sh   (about five minutes)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/tests/hipemu
mkdir -p _asan
FL="-O1 -g -mfma -ffp-contract=off -fPIC -std=c++17 -DMP3MI_EMU -fsanitize=address,undefined -fno-omit-frame-pointer -I. -I../../mp3-enc-bsd_amd/csrc -I../../include -Wno-unused"
for f in k_fft k_psy k_fbmdct k_prep k_loop k_format k_debug k_dropin k_synth; do g++ $FL -x c++ -c ../../mp3-enc-bsd_amd/csrc/$f.hip -o _asan/$f.o & done
for f in batch dropin tables_host pcm_synth_host; do g++ $FL -c ../../mp3-enc-bsd_amd/csrc/$f.cpp -o _asan/$f.o & done
g++ $FL -c hipemu.cpp -o _asan/hipemu.o
wait
g++ -shared -fsanitize=address,undefined -o _asan/libmp3mi_emu_asan.so _asan/*.o -lm -ldl
cat > _asan/run.py <<PY
import sys
sys.path.insert(0, "$ROOT/tests")
import mp3common
mp3common.EMU_SO = "$ROOT/tests/hipemu/_asan/libmp3mi_emu_asan.so"
from mp3common import Mp3mi, BatchRun
mp = Mp3mi(emu=True)
for S, rate, ch, kbps, nf, s0 in [(2, 44100, 2, 128, 5, 5), (2, 48000, 2, 32, 4, 3), (2, 32000, 1, 64, 4, 8), (3, 32000, 1, 96, 3, 11)]: # (the last: an odd number of tracks)
    run = BatchRun(mp, S, rate, ch, kbps, nf, stream0=s0)
    out, lens = run.encode()
    got = run.encode_streaming([1, nf - 1])
    assert all(got[s] == out[s, :lens[s]].tobytes() for s in range(S))
    out2, _ = run.encode(63)
    assert (out2 == out).all()
    run.close()
    print("sanitizer run ok:", rate, ch, kbps)
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python3 _asan/run.py 2>&1 | grep -v "doesn't fully support makecontext"

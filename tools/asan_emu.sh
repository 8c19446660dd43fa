#!/bin/bash
# CPU sanitizer run of the kernel sources (GPU AddressSanitizer is not available on this pool): builds the wave-emulated
# test library with -fsanitize=address,undefined and runs three small configurations through whole-file encode,
# streaming and all exact tiers.  Test infrastructure only.  Usage: tools/asan_emu.sh   (about five minutes)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/tests/hipemu
mkdir -p _asan
FL="-O1 -g -mfma -ffp-contract=off -fPIC -std=c++17 -DMP3MI_EMU -fsanitize=address,undefined -fno-omit-frame-pointer -I. -I../../mp3-enc-bsd_amd/csrc -I../../include -Wno-unused"
for f in k_fft k_psy k_fbmdct k_prep k_loop k_format k_debug k_dropin k_synth k_l12; do g++ $FL -x c++ -c ../../mp3-enc-bsd_amd/csrc/$f.hip -o _asan/$f.o & done
for f in batch l12_batch dropin tables_host pcm_synth_host; do g++ $FL -c ../../mp3-enc-bsd_amd/csrc/$f.cpp -o _asan/$f.o & done
g++ $FL -c hipemu.cpp -o _asan/hipemu.o
wait
(cd ../../mp3-enc-bsd_amd/csrc && ld -r -b binary -z noexecstack -o $ROOT/tests/hipemu/_asan/tables_blob.o tables_blob.bin)
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
    out3, _ = run.encode(64)  # the records k_mdct's tail lists, through k_prep
    assert (out3 == out).all()
    run.close()
    print("sanitizer run ok:", rate, ch, kbps)
# Layers I and II: whole-file (ragged, several chunks), streaming, every exact tier
from mp3common import L12Run, l12_signal, l12_spf
for layer, rate, kbps, mode, nfr in [(2, 44100, 128, "j", 5), (1, 32000, 96, "se", 11), (2, 48000, 56, "m", 4), (1, 44100, 448, "d", 7)]:
    ch = 1 if mode[0] == "m" else 2
    pcms = [l12_signal(l12_spf(layer) * nfr - 77 * i, ch, 9 + i, rate) for i in range(3)]
    run = L12Run(mp, layer, rate, kbps, mode, pcms, scratch_mb=1)
    a = run.encode()
    run.set_flags(2 | 4 | 32)
    assert run.encode() == a
    run.close()
    full = [l12_signal(l12_spf(layer) * nfr, ch, 9 + i, rate) for i in range(2)]
    run = L12Run(mp, layer, rate, kbps, mode, full)
    assert run.encode_streaming([1, nfr - 2, 1]) == run.encode()
    run.close()
    print("sanitizer run ok: layer", layer, rate, kbps, mode)
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 python3 _asan/run.py 2>&1 | grep -v "doesn't fully support makecontext"
# The drop-in surface (dropin.cpp: host-mapped buffers, the filterbank's result cache): the reference's unchanged driver objects
# (oracle/_ref/obj, where the reference was compiled) linked against the sanitized library, twelve frames, bytes compared
# with the reference binary's.
if [ -f $ROOT/oracle/_ref/obj/musicin.o ] && [ -x $ROOT/oracle/_ref/encode ]; then
  (cd $ROOT/oracle && gcc -rdynamic -o $ROOT/tests/hipemu/_asan/encode_dropin_asan $(for f in musicin common ieeefloat portableio psy subs tables tonal; do echo _ref/obj/$f.o; done) \
      _ref/obj/encode_nofb.o $ROOT/tests/hipemu/_asan/libmp3mi_emu_asan.so -lm -lstdc++ 2>/dev/null)
  T=$(mktemp -d)
  python3 - $ROOT $T <<'PY'
import sys
sys.path.insert(0, sys.argv[1] + "/tests")
from mp3common import Mp3mi, SEED
from test_dropin import write_wav
write_wav(sys.argv[2] + "/a.wav", Mp3mi(emu=True).synth(1152 * 12, 2, 44100, 0, SEED), 2, 44100)
PY
  (cd $T && LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0:detect_stack_use_after_return=0 \
      $ROOT/tests/hipemu/_asan/encode_dropin_asan -s 44.1 -b 128 a.wav a.mp3 > drop.log 2>&1; $ROOT/oracle/_ref/encode -s 44.1 -b 128 a.wav r.mp3 > /dev/null 2>&1
   if grep -q "ERROR: AddressSanitizer\|runtime error" drop.log; then grep -m5 "ERROR\|runtime error" drop.log; exit 1; fi
   cmp a.mp3 r.mp3 && echo "sanitizer run ok: drop-in driver, 12 frames, bytes identical to the reference binary's")
  rm -rf $T
fi

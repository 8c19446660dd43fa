"""Drop-in boundary: the reference's UNCHANGED driver objects (musicin.o, common.o, ... compiled
from /root/reference/src into oracle/_ref/obj here) linked against the product library instead
of the reference's Layer III objects must produce the reference's MP3 byte for byte.

 * encode_dropin_emu: linked against the emulated CPU test build  -> runs in the CPU suite
 * encode_dropin:     linked against libmp3mi.so (HIP)            -> runs with -m gpu
Both binaries are built by `make -C oracle _ref/encode_dropin[_emu]` where the reference exists
and travel with the repo; the tests skip where they are absent."""
import os
import struct
import subprocess

import numpy as np
import pytest

from mp3common import ROOT, SEED

REF = os.path.join(ROOT, "oracle", "_ref")


def write_wav(path, pcm, ch, rate):
    data = pcm.astype("<i2").tobytes()
    with open(path, "wb") as f:
        f.write(b"RIFF" + struct.pack("<I", 36 + len(data)) + b"WAVEfmt " +
                struct.pack("<IHHIIHH", 16, 1, ch, rate, rate * ch * 2, ch * 2, 16) + b"data" + struct.pack("<I", len(data)) + data)


def run_cli(binary, wav, mp3, rate, kbps, mono, extra=()):
    args = [os.path.join(REF, binary), "-s", "%g" % (rate / 1000.0), "-b", str(kbps)] + list(extra)
    if mono:
        args += ["-m", "m"]
    subprocess.run(args + [str(wav), str(mp3)], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                   cwd=os.path.dirname(str(mp3)))  # (the reference's psycho_anal writes "out.dat" where it runs)
    return open(mp3, "rb").read()


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "encode_dropin_emu")), reason="oracle/_ref/encode_dropin_emu not built")
def test_reference_driver_over_emulated_library(emu, oracle, tmp_path):
    rate, ch, kbps = 44100, 2, 128
    pcm = emu.synth(1152 * 6 + 500, ch, rate, 5, SEED)
    write_wav(tmp_path / "a.wav", pcm, ch, rate)
    got = run_cli("encode_dropin_emu", tmp_path / "a.wav", tmp_path / "a.mp3", rate, kbps, False)
    ref, _ = oracle.encode(pcm, rate, kbps, ch)
    assert got == ref
    if os.path.exists(os.path.join(REF, "encode")):
        assert got == run_cli("encode", tmp_path / "a.wav", tmp_path / "r.mp3", rate, kbps, False)


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "encode_dropin")), reason="oracle/_ref/encode_dropin not built")
@pytest.mark.parametrize("rate,ch,kbps,frames", [(44100, 2, 128, 40), (32000, 1, 64, 30), (48000, 2, 320, 25)])
def test_reference_driver_over_hip_library(product, oracle, tmp_path, rate, ch, kbps, frames):
    pcm = product.synth(1152 * frames - 300, ch, rate, 21, SEED)
    write_wav(tmp_path / "a.wav", pcm, ch, rate)
    got = run_cli("encode_dropin", tmp_path / "a.wav", tmp_path / "a.mp3", rate, kbps, ch == 1)
    ref, _ = oracle.encode(pcm, rate, kbps, ch)
    assert got == ref
    if os.path.exists(os.path.join(REF, "encode")):
        assert got == run_cli("encode", tmp_path / "a.wav", tmp_path / "r.mp3", rate, kbps, ch == 1)


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "encode_dropin_emu")) and os.path.exists(os.path.join(REF, "encode"))), reason="oracle/_ref binaries not built")
def test_reference_driver_options_over_emulated_library(emu, tmp_path):
    """-m d and -e of the reference's own driver reach the library through the unchanged frame_params"""
    rate, ch, kbps = 44100, 2, 128
    pcm = emu.synth(1152 * 4, ch, rate, 6, SEED)
    write_wav(tmp_path / "a.wav", pcm, ch, rate)
    for extra in (["-m", "d"], ["-e"], ["-m", "d", "-e", "-c"]):
        got = run_cli("encode_dropin_emu", tmp_path / "a.wav", tmp_path / "a.mp3", rate, kbps, False, extra)
        assert got == run_cli("encode", tmp_path / "a.wav", tmp_path / "r.mp3", rate, kbps, False, extra), extra


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "encode_dropin")) and os.path.exists(os.path.join(REF, "encode"))), reason="oracle/_ref binaries not built")
def test_baseline_config0_through_the_reference_driver(product, tmp_path):
    """BASELINE configs[0] at full length: the 10 s 44.1 kHz stereo stream (383 frames) through the reference's own
    main() linked against libmp3mi.so, with -m d -e on a second pass.  The frames/s of this per-call plumbing path
    (a kernel launch and a host round trip per reference call -- not the throughput path) goes to gpurun_out/."""
    import json
    import time
    rate, ch, kbps = 44100, 2, 128
    pcm = product.synth(441000, ch, rate, 0, SEED)
    write_wav(tmp_path / "a.wav", pcm, ch, rate)
    ref = run_cli("encode", tmp_path / "a.wav", tmp_path / "r.mp3", rate, kbps, False)
    t0 = time.perf_counter()
    got = run_cli("encode_dropin", tmp_path / "a.wav", tmp_path / "a.mp3", rate, kbps, False)
    dt = time.perf_counter() - t0
    assert got == ref and len(got) == 159704
    write_wav(tmp_path / "short.wav", pcm[: 4410 * ch], ch, rate)  # four frames: what a process costs before its first frame
    t0 = time.perf_counter()
    run_cli("encode_dropin", tmp_path / "short.wav", tmp_path / "short.mp3", rate, kbps, False)
    dt_start = time.perf_counter() - t0
    # the marginal rate: three times the length, the difference of the two runs over the 766 frames in between (start-up --
    # process, HIP, tables: 0.2-0.3 s, and not the same twice -- drops out)
    write_wav(tmp_path / "long.wav", np.concatenate([pcm, pcm, pcm]), ch, rate)
    t0 = time.perf_counter()
    run_cli("encode_dropin", tmp_path / "long.wav", tmp_path / "long.mp3", rate, kbps, False)
    dt_long = time.perf_counter() - t0
    # ... and the library's own clock: from the first frame's first call to the flush (options.dropin_stats: MP3MI_DROPIN_STATS)
    r = subprocess.run([os.path.join(REF, "encode_dropin"), "-s", "44.1", "-b", str(kbps), str(tmp_path / "long.wav"), str(tmp_path / "long2.mp3")],
                       capture_output=True, text=True, env=dict(os.environ, MP3MI_DROPIN_STATS="1"), cwd=str(tmp_path))
    line = [x for x in r.stderr.replace("\r", "\n").splitlines() if "mp3mi drop-in:" in x][-1]
    in_process = {"frames": int(line.split("drop-in:")[1].split()[0]), "seconds": float(line.split(" in ")[1].split()[0]),
                  "frames_per_s": float(line.split("= ")[1].split()[0]), "waits_for_the_device": int(line.split("frames/s, ")[1].split()[0])}
    assert in_process["frames"] == 1149
    t0 = time.perf_counter()
    run_cli("encode", tmp_path / "long.wav", tmp_path / "long_ref.mp3", rate, kbps, False)
    dt_long_ref = time.perf_counter() - t0
    t0 = time.perf_counter()
    run_cli("encode", tmp_path / "a.wav", tmp_path / "r2.mp3", rate, kbps, False)
    dt_ref = time.perf_counter() - t0
    extra = ["-m", "d", "-e"]
    assert run_cli("encode_dropin", tmp_path / "a.wav", tmp_path / "b.mp3", rate, kbps, False, extra) == \
        run_cli("encode", tmp_path / "a.wav", tmp_path / "s.mp3", rate, kbps, False, extra)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    json.dump({"what": "BASELINE configs[0]: 383 frames through oracle/_ref/encode_dropin (reference main() + libmp3mi.so drop-in symbols), process start to exit",
               "frames": 383, "seconds": round(dt, 3), "frames_per_s": round(383 / dt, 1), "bit_exact": True,
               "seconds_of_a_four_frame_run": round(dt_start, 3), "seconds_1149_frames": round(dt_long, 3),
               "frames_per_s_marginal": round(766 / max(dt_long - dt, 1e-9), 1),
               "inside_the_process_1149_frames": in_process,
               "reference_binary_on_one_host_core": {"seconds_383_frames": round(dt_ref, 3), "seconds_1149_frames": round(dt_long_ref, 3),
                                                     "frames_per_s_marginal": round(766 / max(dt_long_ref - dt_ref, 1e-9), 1)},
               "note": "the per-call surface over ONE hidden stream: with the look-ahead of mp3mi_dropin.h a frame is two launches "
                       "(one for the four L3psycho_anal calls, one for the 72 window_subband / filter_subband calls, mdct_sub, iteration_loop and "
                       "III_format_bitstream, each served from it if its arguments are what the launch read) "
                       "instead of 79; 60 % of what is left is k_loop's one wavefront; throughput comes from the batched API"},
              open(os.path.join(out, "dropin_config0.json"), "w"), indent=1)



LAYER12 = [(2, 160, ["-l", "2"]), (1, 192, ["-l", "1"]), (2, 64, ["-l", "2", "-m", "j"])]


def without_private_bit(data, layer, kbps, rate=44100):
    """The reference's driver never initialises info.extension (the header's private bit): `layer info` is an automatic
    variable of main() (src/musicin.c:470) and parse_args sets every field but that one, so the bit is whatever the stack
    held -- it differs between two LINKS of the same driver.  Masked in every frame header before two binaries are compared."""
    fb = 4 * int(384 / (rate / 1000.0) * kbps / 32) if layer == 1 else int(1152 / (rate / 1000.0) * kbps / 8)
    b = bytearray(data)
    for p in range(0, len(b) - 3, fb):
        assert b[p] == 0xff and (b[p + 1] & 0xf0) == 0xf0
        b[p + 2] &= 0xfe
    return bytes(b)


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "encode_dropin_emu")) and os.path.exists(os.path.join(REF, "encode"))), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("layer,kbps,extra", LAYER12)
def test_reference_driver_layers_1_2_over_emulated_library(emu, tmp_path, layer, kbps, extra):
    """-l 1 / -l 2 of the reference's own driver over the drop-in link: the reference's Layer I / II code with the library's
    window_subband / filter_subband under it (the batched counterpart is include/mp3mi_l12.h)"""
    pcm = emu.synth(1152 * 4 + 200, 2, 44100, 7, SEED)
    write_wav(tmp_path / "a.wav", pcm, 2, 44100)
    got = run_cli("encode_dropin_emu", tmp_path / "a.wav", tmp_path / "a.mpg", 44100, kbps, False, extra)
    ref = run_cli("encode", tmp_path / "a.wav", tmp_path / "r.mpg", 44100, kbps, False, extra)
    assert without_private_bit(got, layer, kbps) == without_private_bit(ref, layer, kbps)


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "encode_dropin")) and os.path.exists(os.path.join(REF, "encode"))), reason="oracle/_ref binaries not built")
@pytest.mark.parametrize("layer,kbps,extra", LAYER12)
def test_reference_driver_layers_1_2_over_hip_library(product, tmp_path, layer, kbps, extra):
    pcm = product.synth(1152 * 20 + 200, 2, 44100, 7, SEED)
    write_wav(tmp_path / "a.wav", pcm, 2, 44100)
    got = run_cli("encode_dropin", tmp_path / "a.wav", tmp_path / "a.mpg", 44100, kbps, False, extra)
    ref = run_cli("encode", tmp_path / "a.wav", tmp_path / "r.mpg", 44100, kbps, False, extra)
    assert without_private_bit(got, layer, kbps) == without_private_bit(ref, layer, kbps)


def run_probe(binary, dump, frames, lookahead=None):
    env = dict(os.environ)
    if lookahead is not None:
        env["MP3MI_DROPIN_LOOKAHEAD"] = str(lookahead)
    r = subprocess.run([os.path.join(REF, binary), str(dump), str(frames)], check=True, capture_output=True, text=True, env=env, cwd=os.path.dirname(str(dump)))
    waits = [int(x.split()[1]) for x in r.stdout.splitlines() if x.startswith("waits")]
    return open(dump, "rb").read(), (waits[0] if waits else None)


def probe_case(binary, tmp_path, frames=11):
    """oracle/dropin_probe.c: a caller that rewrites samples the look-ahead has read, moves its pointer back in the middle of a
    frame and changes buffers between frames -- every value the library returns equals what the reference's own functions
    return to the same caller, with the look-ahead on, off and half on; and the look-ahead did save waits where the caller
    behaved, and did fall back where it did not"""
    ref, _ = run_probe("dropin_probe_ref", tmp_path / "ref.bin", frames)
    waits = {}
    for mode in (0, 1, 2, 3):
        got, waits[mode] = run_probe(binary, tmp_path / ("m%d.bin" % mode), frames, lookahead=mode)
        assert got == ref, "look-ahead mode %d: the library's values differ from the reference's" % mode
    assert waits[0] == frames * (4 + 72)  # every call on its own
    # with both: 2 + 1 waits per well-behaved frame; the misbehaving frames fall back to call-by-call service part of the way
    assert frames * 3 < waits[1] < waits[0] // 2, waits
    assert waits[1] < waits[2] < waits[0] and waits[1] < waits[3] < waits[0], waits


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "dropin_probe_emu")) and os.path.exists(os.path.join(REF, "dropin_probe_ref"))), reason="oracle/_ref/dropin_probe* not built")
def test_lookahead_survives_a_caller_that_breaks_its_assumptions_emulated(tmp_path):
    probe_case("dropin_probe_emu", tmp_path)


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "dropin_probe")) and os.path.exists(os.path.join(REF, "dropin_probe_ref"))), reason="oracle/_ref/dropin_probe* not built")
def test_lookahead_survives_a_caller_that_breaks_its_assumptions_gpu(tmp_path):
    probe_case("dropin_probe", tmp_path)


def run_frame_probe(binary, tmp_path, tag, frames, lookahead=None):
    env = dict(os.environ, MP3MI_DROPIN_STATS="1")
    if lookahead is not None:
        env["MP3MI_DROPIN_LOOKAHEAD"] = str(lookahead)
    dump, mp3 = tmp_path / (tag + ".bin"), tmp_path / (tag + ".mp3")
    r = subprocess.run([os.path.join(REF, binary), str(dump), str(mp3), str(frames)], check=True, capture_output=True, text=True, env=env, cwd=str(tmp_path))
    ahead = None
    for line in r.stderr.splitlines():
        if line.startswith("mp3mi drop-in:"):  # "... N waits for the device, A / B frames' loops / formatters launched ahead of their calls"
            a, b = line.split("waits for the device, ")[1].split(" frames'")[0].split(" / ")
            ahead = (int(a), int(b))
    return open(dump, "rb").read(), open(mp3, "rb").read(), ahead


def frame_probe_case(binary, tmp_path, frames=12):
    """oracle/dropin_probe_frame.c: all seven functions, frame by frame, from a caller that changes a subband sample before
    mdct_sub, lines of the spectrum or a perceptual entropy before iteration_loop, the signs' spectrum or a header bit before
    III_format_bitstream, and the buffer it hands over: every returned value and the bitstream equal what the reference's own
    functions give that caller -- with mdct_sub / iteration_loop / III_format_bitstream launched ahead of their calls and
    without; the launches ahead were used where the caller behaved and dropped where it did not"""
    ref_dump, ref_mp3, _ = run_frame_probe("dropin_probe_frame_ref", tmp_path, "ref", frames)
    ahead = {}
    for mode in (0, 1, 4):
        dump, mp3, ahead[mode] = run_frame_probe(binary, tmp_path, "m%d" % mode, frames, lookahead=mode)
        assert dump == ref_dump, "look-ahead mode %d: the library's values differ from the reference's" % mode
        assert mp3 == ref_mp3, "look-ahead mode %d: the bitstream differs from the reference's" % mode
    hostile_fmt = sum(1 for f in range(frames) if f % 7 in (4, 5))
    assert ahead[0] == (0, 0) and ahead[4] == (0, 0), ahead
    assert 0 < ahead[1][0] < frames and ahead[1][1] == frames - hostile_fmt, ahead


@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "dropin_probe_frame_emu")) and os.path.exists(os.path.join(REF, "dropin_probe_frame_ref"))),
                    reason="oracle/_ref/dropin_probe_frame* not built")
def test_launches_ahead_survive_a_caller_that_changes_what_it_was_handed_emulated(tmp_path):
    frame_probe_case("dropin_probe_frame_emu", tmp_path)


@pytest.mark.gpu
@pytest.mark.skipif(not (os.path.exists(os.path.join(REF, "dropin_probe_frame")) and os.path.exists(os.path.join(REF, "dropin_probe_frame_ref"))),
                    reason="oracle/_ref/dropin_probe_frame* not built")
def test_launches_ahead_survive_a_caller_that_changes_what_it_was_handed_gpu(tmp_path):
    frame_probe_case("dropin_probe_frame", tmp_path, frames=40)

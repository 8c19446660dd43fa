"""GPU parity tests proper: the HIP path (through the C ABI of libmp3mi.so) against the oracle
on the same seeded inputs, stage by stage and byte by byte.  Run with -m gpu on an MI355X."""
import hashlib

import numpy as np
import pytest

from mp3common import SEED
from stage_check import compare_stages, run_batch_with_stages

pytestmark = pytest.mark.gpu

CONFIGS = [
    # rate, channels, kbps, streams, frames
    (44100, 2, 128, 6, 24),
    (48000, 2, 64, 3, 16),
    (48000, 2, 320, 3, 16),
    (32000, 1, 64, 4, 20),
    (44100, 1, 96, 2, 12),
    (44100, 2, 32, 2, 12),
]


@pytest.mark.parametrize("rate,channels,kbps,S,nf", CONFIGS)
def test_stages_and_bytes_match_oracle(product, oracle, rate, channels, kbps, S, nf):
    pcm = np.stack([product.synth(nf * 1152, channels, rate, 100 + s) for s in range(S)])
    got, st = run_batch_with_stages(product, pcm, rate, channels, kbps, nf)
    for s in range(S):
        ref, dumps = oracle.encode(pcm[s], rate, kbps, channels, dumps=nf)
        bad = compare_stages(st, s, dumps, channels)
        assert not bad, bad[:8]
        assert got[s] == ref, "stream %d: bytes differ (len %d vs %d)" % (s, len(got[s]), len(ref))


def test_mixed_bitrates_in_one_batch(product, oracle):
    """config 4 of BASELINE.json: 48 kHz, stream s uses bitrate s mod 6 of {64..320}."""
    rates = [64, 96, 128, 192, 256, 320]
    S, nf, rate, ch = 12, 10, 48000, 2
    kb = [rates[s % 6] for s in range(S)]
    pcm = np.stack([product.synth(nf * 1152, ch, rate, 300 + s) for s in range(S)])
    got = product.encode_host(pcm, rate, ch, kb, nf)
    for s in range(S):
        ref, _ = oracle.encode(pcm[s], rate, kb[s], ch)
        assert got[s] == ref, "stream %d at %d kbps" % (s, kb[s])


def test_silence_and_full_scale(product, oracle):
    """edge inputs: digital silence, a full-scale square wave, a single impulse"""
    nf, rate, ch = 8, 44100, 2
    sil = np.zeros(nf * 1152 * ch, np.int16)
    sq = np.where((np.arange(nf * 1152 * ch) // 200) % 2 == 0, 32767, -32768).astype(np.int16)
    imp = np.zeros(nf * 1152 * ch, np.int16)
    imp[5000] = 30000
    pcm = np.stack([sil, sq, imp])
    got = product.encode_host(pcm, rate, ch, 128, nf)
    for s in range(3):
        ref, _ = oracle.encode(pcm[s], rate, 128, ch)
        assert got[s] == ref, "edge stream %d" % s
    # silence must not take the second tier of the unpredictability (every c_w is an exact zero, marked by k_cw)
    from mp3common import BatchRun
    run = BatchRun(product, 64, rate, ch, 128, nf, pcm=np.zeros((64, nf * 1152 * ch), np.int16))
    try:
        run.encode()
        listed, records = run.cw_fixups()
        assert listed == 0, "%d of %d records of silent streams listed for the second tier" % (listed, records)
    finally:
        run.close()


def test_tonal_inputs(product, oracle):
    """Stationary tones: the unpredictability of a well-predicted line is tiny or exactly zero (the lines above the
    tones sit at the energy floor with phase 0 in all three short windows).  Exact zeros are marked by k_cw and need
    no second tier (until round 3 every such record took it: a third of this batch); what is left of tiny non-zero
    values is decided by the first tier or goes through k_cw_fix -- the bytes are the reference's either way, and
    the same with the second tier forced for every record."""
    from mp3common import BatchRun
    nf, rate, ch, S = 40, 44100, 2, 48
    t = np.arange(nf * 1152, dtype=np.float64) / rate
    rng = np.random.default_rng(5)
    pcm = np.zeros((S, nf * 1152, ch), np.int16)
    for s in range(S):
        f1, f2 = 110.0 * 2 ** (s / 8.0), 997.0 + 371.0 * s
        a = [3000.0, 12000.0, 30000.0][s % 3]
        left = a * np.sin(2 * np.pi * f1 * t)
        right = a * np.sin(2 * np.pi * f1 * t + 0.5) if s % 2 else 0.5 * a * (np.sin(2 * np.pi * f1 * t) + np.sin(2 * np.pi * f2 * t))
        if s % 4 == 3:
            left = left + rng.normal(0.0, 2.0, left.shape)  # a little noise under the tone
        pcm[s, :, 0] = np.clip(np.rint(left), -32768, 32767)
        pcm[s, :, 1] = np.clip(np.rint(right), -32768, 32767)
    run = BatchRun(product, S, rate, ch, 128, nf, pcm=pcm.reshape(S, -1))
    try:
        out, lens = run.encode(0)
        listed, records = run.cw_fixups()
        print("tonal input: %d of %d records took the second tier of the unpredictability" % (listed, records))
        assert listed < records // 20
        for s in range(S):
            ref, _ = oracle.encode(pcm[s].reshape(-1), rate, 128, ch)
            assert out[s, :lens[s]].tobytes() == ref, "tonal stream %d" % s
        out2, lens2 = run.encode(32)  # and with the second tier for every record
        assert np.array_equal(lens, lens2) and np.array_equal(out, out2)
    finally:
        run.close()


def test_chunked_equals_unchunked(product, oracle, monkeypatch):
    """the internal frame chunking must not change a byte (state carried across chunks)"""
    nf, rate, ch, S = 23, 44100, 2, 3
    pcm = np.stack([product.synth(nf * 1152, ch, rate, 700 + s) for s in range(S)])
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "5")
    got = product.encode_host(pcm, rate, ch, 128, nf)
    monkeypatch.delenv("MP3MI_CHUNK_FRAMES")
    for s in range(S):
        ref, _ = oracle.encode(pcm[s], rate, 128, ch)
        assert got[s] == ref


def test_full_length_stream_matches_golden_md5(product, oracle):
    """BASELINE config 1: 10 s 44.1 kHz stereo -> 383 frames, whole file against the oracle."""
    rate, ch, nf = 44100, 2, 383
    pcm = product.synth(441000, ch, rate, 0)
    padded = np.zeros(nf * 1152 * ch, np.int16)
    padded[:len(pcm)] = pcm
    got = product.encode_host(padded[None, :], rate, ch, 128, nf)[0]
    ref, _ = oracle.encode(pcm, rate, 128, ch)
    assert len(got) == len(ref)
    assert hashlib.md5(got).hexdigest() == hashlib.md5(ref).hexdigest()


def test_full_chip_batch_with_placement_and_overlap(product, oracle, monkeypatch):
    """More streams than the chip has SIMDs x 2: k_loop places streams on SIMDs by the previous
    chunk's cost (every stream must be taken exactly once), paces them by wave priority, and the
    next chunk's feed-forward kernels overlap with it.  Three chunks, mixed bitrates; EVERY stream
    is checked against the oracle."""
    from concurrent.futures import ThreadPoolExecutor
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "2")
    S, nf, rate, ch = 2304, 6, 44100, 2
    base = np.stack([product.synth(nf * 1152, ch, rate, 500 + s) for s in range(48)])
    # 48 distinct signals at 48 gains each: all streams differ, the oracle work stays small
    gains = (np.arange(S) // 48 + 1).astype(np.float64) / 48.0
    pcm = np.round(base[np.arange(S) % 48].astype(np.float64) * gains[:, None]).astype(np.int16)
    kb = [(96, 128, 160)[s % 3] for s in range(S)]
    got = product.encode_host(pcm, rate, ch, kb, nf)
    with ThreadPoolExecutor(max_workers=16) as ex:
        refs = list(ex.map(lambda s: oracle.encode(pcm[s], rate, kb[s], ch)[0], range(S)))
    bad = [s for s in range(S) if got[s] != refs[s]]
    assert not bad, "streams differ: %s" % bad[:10]


def test_more_streams_than_wave_slots(product, oracle, monkeypatch):
    """5000 mono streams: more wavefronts than the chip holds at once, so k_loop runs in rounds, late
    wavefronts take their streams through the placement's scan path, and the census gate times out.
    Every stream is checked."""
    from concurrent.futures import ThreadPoolExecutor
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "2")
    S, nf, rate, ch, kbps = 5000, 4, 32000, 1, 64
    base = np.stack([product.synth(nf * 1152, ch, rate, 900 + s) for s in range(50)])
    gains = (np.arange(S) // 50 + 1).astype(np.float64) / 100.0
    pcm = np.round(base[np.arange(S) % 50].astype(np.float64) * gains[:, None]).astype(np.int16)
    got = product.encode_host(pcm, rate, ch, kbps, nf)
    with ThreadPoolExecutor(max_workers=16) as ex:
        refs = list(ex.map(lambda s: oracle.encode(pcm[s], rate, kbps, ch)[0], range(S)))
    bad = [s for s in range(S) if got[s] != refs[s]]
    assert not bad, "streams differ: %s" % bad[:10]


def test_output_independent_of_schedule(product, monkeypatch):
    """The whole chip's worth of streams under two different schedules (chunking, gate and placement on /
    off): identical bytes.  A race between the front stream's and the loop stream's kernels, or a stream
    taken twice by the placement, would show up as a difference (tools/determinism_check.py does the same
    at 4096 x 96)."""
    S, nf, rate, ch = 4096, 10, 44100, 2
    base = np.stack([product.synth(nf * 1152, ch, rate, 1200 + s) for s in range(64)])
    gains = (np.arange(S) // 64 + 1).astype(np.float64) / 64.0
    pcm = np.round(base[np.arange(S) % 64].astype(np.float64) * gains[:, None]).astype(np.int16)
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "3")
    a = product.encode_host(pcm, rate, ch, 128, nf)
    b = product.encode_host(pcm, rate, ch, 128, nf)
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "4")
    monkeypatch.setenv("MP3MI_NO_GATE", "1")
    monkeypatch.setenv("MP3MI_NO_PLACE", "1")
    c = product.encode_host(pcm, rate, ch, 128, nf)
    assert a == b and a == c
    # ... and with k_cw / k_part / k_psy between the k_loop launches, or k_psy alone beside them (batch.cpp, stage X)
    monkeypatch.delenv("MP3MI_NO_GATE")
    monkeypatch.delenv("MP3MI_NO_PLACE")
    for mode in ("0", "2"):
        monkeypatch.setenv("MP3MI_PSY_BESIDE", mode)
        assert product.encode_host(pcm, rate, ch, 128, nf) == a, "MP3MI_PSY_BESIDE=%s" % mode


@pytest.mark.parametrize("rate,ch,kbps,S,nf,stream0", [(44100, 2, 128, 8192, 12, 0), (32000, 1, 64, 16384, 10, 0)])
def test_batches_larger_than_the_chip(product, oracle, rate, ch, kbps, S, nf, stream0):
    """BASELINE configs[2] and [4] at their stream counts: 8192 stereo streams are two residency rounds of k_loop (one
    wavefront per stream, 4096 resident) with placement in both, 16 384 mono streams four; three chunks each.  A
    256-stream sample spread over the batch is compared with the oracle (tools/full_parity.py compares all of them
    at full length: profiles/r02_parity_config2.json, _config4.json)."""
    import os
    from mp3common import BatchRun
    os.environ["MP3MI_CHUNK_FRAMES"] = "4"
    try:
        run = BatchRun(product, S, rate, ch, kbps, nf, stream0=stream0)
    finally:
        del os.environ["MP3MI_CHUNK_FRAMES"]
    try:
        out, lens = run.encode()
        for s in sorted(set(np.linspace(0, S - 1, 256).astype(int).tolist())):
            ref, _ = oracle.encode(run.pcm_of(s), rate, kbps, ch)
            assert out[s, :lens[s]].tobytes() == ref, "stream %d of %d differs from the oracle" % (s, S)
    finally:
        run.close()

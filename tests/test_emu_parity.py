"""Kernel LOGIC on the CPU: the product's kernel sources compiled against the test-only wave
emulator (tests/hipemu) must reproduce the oracle stage by stage and byte by byte.  Small sizes:
the emulator runs every lane as a fiber.  (The real HIP build is covered by the -m gpu tests.)"""
import numpy as np
import pytest

import hashlib

import abort_cases

from golden_util import aborting_cases, case_pcm, case_stages, encoding_cases, manifest
from mp3common import pad_frames
from stage_check import compare_stages, run_batch_with_stages


@pytest.mark.parametrize("rate,channels,kbps,stream,nf", [(44100, 2, 128, 5, 5), (48000, 2, 320, 3, 3), (32000, 1, 64, 8, 4)])
def test_emulated_kernels_match_oracle(emu, oracle, rate, channels, kbps, stream, nf):
    pcm = emu.synth(nf * 1152, channels, rate, stream)
    got, st = run_batch_with_stages(emu, pcm[None, :], rate, channels, kbps, nf)
    ref, dumps = oracle.encode(pcm, rate, kbps, channels, dumps=nf)
    bad = compare_stages(st, 0, dumps, channels)
    assert not bad, bad[:8]
    assert got[0] == ref


def test_emulated_kernels_match_reference_golden(emu):
    """straight against the reference's own stage dumps (no oracle in between)"""
    case = [c for c in manifest() if c["name"] == "s44_128_bursty"][0]
    gold = case_stages(case)[0][:6]
    pcm, nf = pad_frames(case_pcm(case, emu.synth), case["channels"])
    nf = 6
    got, st = run_batch_with_stages(emu, pcm[None, :nf * 1152 * case["channels"]], case["rate"], case["channels"], case["kbps"], nf)
    bad = compare_stages(st, 0, gold, case["channels"])
    assert not bad, bad[:8]


CRAFTED = [c for c in encoding_cases() if c["name"].startswith("x")]


@pytest.mark.parametrize("case", CRAFTED, ids=[c["name"] for c in CRAFTED])
def test_emulated_kernels_reproduce_the_crafted_goldens(emu, case):
    """the fixtures made for the reference's rarely taken branches (tests/golden/coverage_notes.json,
    oracle/crafted_inputs.py): scfsi, reservoir caps and the drain into ancillary data, queued headers, exact-zero
    lines, error protection -- md5 of the whole file and the stage seams of the dumped frames"""
    pcm, nf = pad_frames(case_pcm(case, emu.synth), case["channels"])
    got, st = run_batch_with_stages(emu, pcm[None, :], case["rate"], case["channels"], case["kbps"], nf, mode=case.get("mode"))
    assert len(got[0]) == case["mp3_len"] and hashlib.md5(got[0]).hexdigest() == case["mp3_md5"]
    gold, frames = case_stages(case)
    bad = compare_stages(st, 0, gold, case["channels"], frames)
    assert not bad, bad[:8]


@pytest.mark.parametrize("case", aborting_cases(), ids=[c["name"] for c in aborting_cases()])
def test_emulated_kernels_report_where_the_reference_dies(emu, case):
    pcm, nf = pad_frames(case_pcm(case, emu.synth), case["channels"])
    got, st = run_batch_with_stages(emu, pcm[None, :], case["rate"], case["channels"], case["kbps"], nf, mode=case.get("mode"),
                                    expect_abort=True)
    assert got[0] == b""
    assert (st["status"][0] & 255, st["status"][0] >> 8) == (case["reference_aborts"]["status"], case["reference_aborts"]["frame"])


def test_an_aborting_stream_leaves_its_neighbours_alone(emu, oracle):
    abort_cases.neighbours_case(emu, oracle)


def test_host_wrapper_delivers_the_neighbours_of_an_aborting_stream(emu, oracle):
    abort_cases.host_wrapper_case(emu, oracle)


def test_a_streaming_call_reports_the_abort_when_it_happens(emu, oracle):
    abort_cases.streaming_case(emu, oracle)


def test_options_out_of_range_are_refused(emu):
    import ctypes
    from mp3common import BatchOptions
    L = emu.lib
    for field, bad in (("gate", 2), ("gate", -2), ("placement", 5), ("call_overlap", -3), ("y_after_loop", 2), ("psy_beside", -2),
                       ("psy_beside", 3), ("loop_part_streams", 100), ("loop_part_streams", -64), ("call_hold", 2)):
        o = BatchOptions()
        L.mp3mi_batch_options_default(ctypes.byref(o))
        setattr(o, field, bad)
        b = ctypes.c_void_p()
        assert L.mp3mi_batch_create_ex(ctypes.byref(b), 1, 44100, 2, None, 128, 2, ctypes.byref(o)) == -1, (field, bad)


def test_silence_does_not_take_the_second_tier(emu, oracle):
    """digital silence: every unpredictability is an exact zero in the reference too (k_cw marks it), so no record is
    listed for the correctly rounded sines -- it used to be all of them"""
    from mp3common import BatchRun
    nf = 3
    pcm = np.zeros((2, nf * 1152 * 2), np.int16)
    pcm[1, 2 * 1152 * 2:] = emu.synth(1152, 2, 44100, 9)  # the second stream wakes up in its last frame
    run = BatchRun(emu, 2, 44100, 2, 128, nf, pcm=pcm)
    try:
        out, lens = run.encode()
        listed, records = run.cw_fixups()
        assert records == 2 * nf * 2 * 2 and listed <= 2, (listed, records)
        for s in range(2):
            assert out[s, :lens[s]].tobytes() == oracle.encode(pcm[s], 44100, 128, 2)[0]
    finally:
        run.close()


def test_two_streams_mixed_bitrate_and_chunking(emu, oracle, monkeypatch):
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "2")
    nf, rate, ch = 5, 48000, 2
    pcm = np.stack([emu.synth(nf * 1152, ch, rate, 40 + s) for s in range(2)])
    got = emu.encode_host(pcm, rate, ch, [64, 192], nf)
    for s, kb in enumerate([64, 192]):
        assert got[s] == oracle.encode(pcm[s], rate, kb, ch)[0]


@pytest.mark.parametrize("mode", ["0", "1", "2"])
def test_stage_x_schedules(emu, oracle, monkeypatch, mode):
    """what of stage X is launched beside k_loop (MP3MI_PSY_BESIDE: nothing / k_cw, k_part, k_psy / k_psy only) is a
    matter of launch order only: four chunks under each order, against the oracle"""
    monkeypatch.setenv("MP3MI_PSY_BESIDE", mode)
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "1")
    nf, rate, ch = 4, 44100, 2
    pcm = np.stack([emu.synth(nf * 1152, ch, rate, 60 + s) for s in range(2)])
    got = emu.encode_host(pcm, rate, ch, [128, 96], nf)
    for s, kb in enumerate([128, 96]):
        assert got[s] == oracle.encode(pcm[s], rate, kb, ch)[0]


def test_loop_in_parts(emu, oracle, monkeypatch):
    """a batch of more streams than k_loop holds resident goes through it in parts (batch.cpp): 70 mono streams in
    parts of 64 and 6, two chunks; every stream against the oracle"""
    monkeypatch.setenv("MP3MI_LOOP_PART_STREAMS", "64")
    monkeypatch.setenv("MP3MI_CHUNK_FRAMES", "1")
    S, nf, rate, ch = 70, 2, 32000, 1
    base = np.stack([emu.synth(nf * 1152, ch, rate, 40 + s) for s in range(7)])
    pcm = np.stack([np.round(base[s % 7].astype(np.float64) * (1 + s // 7) / 10.0).astype(np.int16) for s in range(S)])
    kb = [(64, 96)[s % 2] for s in range(S)]
    got = emu.encode_host(pcm, rate, ch, kb, nf)
    for s in range(S):
        assert got[s] == oracle.encode(pcm[s], rate, kb[s], ch)[0], "stream %d" % s


def test_prep_exact_tier_matches_fast_tier(emu, oracle, monkeypatch):
    """k_prep decides quantanf_init's integer from plain-double logs and repeats the walk with the
    correctly rounded log only near a rounding boundary; forcing the second tier must give the
    same bytes (and more than 64 granules, so that a wavefront carries a ragged tail)."""
    nf, rate, ch = 9, 44100, 2
    pcm = np.stack([emu.synth(nf * 1152, ch, rate, 70 + s) for s in range(2)])
    fast = emu.encode_host(pcm, rate, ch, 128, nf)
    monkeypatch.setenv("MP3MI_PREP_EXACT", "1")
    exact = emu.encode_host(pcm, rate, ch, 128, nf)
    for s in range(2):
        ref = oracle.encode(pcm[s], rate, 128, ch)[0]
        assert fast[s] == ref and exact[s] == ref


@pytest.mark.parametrize("rate,ch", [(44100, 2), (48000, 1), (32000, 2)])
def test_loop_prep_of_mdct_tail_equals_k_prep(emu, rate, ch):
    """The loop's stateless head comes from k_mdct's tail (band energies in the reference's order, the integers from
    order-free sums with a margin) and from k_prep -- the reference's 576-line walk -- only for the records the tail
    lists.  Both paths over every record: the records must agree bit for bit, long and short blocks."""
    from stage_check import run_batch_with_stages, compare_prep_records
    nf = 6
    pcm = np.stack([emu.synth(nf * 1152, ch, rate, 90 + s) for s in range(2)])  # (bursts: short blocks)
    pcm[1, : pcm.shape[1] // 3] = 0  # digital silence: records without energy
    got, st = run_batch_with_stages(emu, pcm, rate, ch, 128, nf)
    got_k, st_k = run_batch_with_stages(emu, pcm, rate, ch, 128, nf, flags=16)  # MP3MI_TEST_PREP_EXACT
    assert got == got_k
    assert (st["psy"]["block_type"] == 2).any() and (st["psy"]["block_type"] == 0).any()
    assert compare_prep_records(st["prep"], st_k["prep"], st["psy"]) == 2 * 2 * nf * ch


def test_noise_exact_tier_matches_partial_sums(emu, oracle, monkeypatch):
    """k_loop decides `noise > xmin` from partial sums spread over the lanes and takes the reference's
    sequential order only when a band lands within 1e-12 of its threshold; forcing the sequential
    sums must give the same bytes.  The input has bursts, so short blocks are covered too."""
    nf, rate, ch = 8, 44100, 2
    pcm = np.stack([emu.synth(nf * 1152, ch, rate, 90 + s) for s in range(2)])
    ref = [oracle.encode(pcm[s], rate, 128, ch)[0] for s in range(2)]
    assert emu.encode_host(pcm, rate, ch, 128, nf) == ref
    monkeypatch.setenv("MP3MI_NOISE_EXACT", "1")
    assert emu.encode_host(pcm, rate, ch, 128, nf) == ref


def test_phase_exact_tier_matches_fast_tier(emu, oracle, monkeypatch):
    """k_cw takes the phases (needed as floats) from a plain-double atan2 unless a value is within 2^-46
    of a float midpoint; forcing the correctly rounded atan2 must give the same bytes."""
    nf, rate, ch = 6, 44100, 2
    pcm = np.stack([emu.synth(nf * 1152, ch, rate, 95 + s) for s in range(2)])
    ref = [oracle.encode(pcm[s], rate, 128, ch)[0] for s in range(2)]
    assert emu.encode_host(pcm, rate, ch, 128, nf) == ref
    monkeypatch.setenv("MP3MI_PHASE_EXACT", "1")
    assert emu.encode_host(pcm, rate, ch, 128, nf) == ref


def test_psy_exact_tier_matches_fast_tier(emu, oracle, monkeypatch):
    """k_psy takes the masking threshold nb (a float) from plain-double log / exp unless the product lies within
    2^-44 of a float midpoint; forcing dm_log / dm_exp must give the same bytes."""
    nf, rate, ch = 6, 48000, 2
    pcm = np.stack([emu.synth(nf * 1152, ch, rate, 97 + s) for s in range(2)])
    ref = [oracle.encode(pcm[s], rate, 128, ch)[0] for s in range(2)]
    assert emu.encode_host(pcm, rate, ch, 128, nf) == ref
    monkeypatch.setenv("MP3MI_PSY_EXACT", "1")
    assert emu.encode_host(pcm, rate, ch, 128, nf) == ref


def test_quant_exact_tier_matches_estimate(emu, oracle):
    """k_loop's quantiser settles a line by a float estimate of x^(3/4) unless it lies within the guard band of a
    table boundary; forcing the table search for every line (MP3MI_TEST_QUANT_EXACT) -- and all five exact tiers
    together -- must give the same bytes.  (The device's raw sqrt / exp2 are covered by tests/test_gpu_tiers.py.)"""
    from mp3common import BatchRun
    nf, rate, ch = 6, 44100, 2
    run = BatchRun(emu, 2, rate, ch, 128, nf, stream0=60)
    try:
        ref = [oracle.encode(run.pcm_of(s), rate, 128, ch)[0] for s in range(2)]
        for flags in (0, 8, 32, 63):
            out, lens = run.encode(flags)
            assert [out[s, :lens[s]].tobytes() for s in range(2)] == ref, "flags %d" % flags
    finally:
        run.close()


def test_records_listed_by_the_mdct_tail_are_redone_by_k_prep(emu, oracle):
    """The tail lists a record it cannot decide (probability ~1e-7: no input of the corpus does it) and k_prep works
    through the list.  MP3MI_TEST_PREP_LIST makes the tail list every third record and spoil what it wrote for it:
    the bytes are the oracle's only if the list path delivers."""
    from stage_check import run_batch_with_stages, compare_prep_records
    nf, rate, ch = 6, 44100, 2
    pcm = np.stack([emu.synth(nf * 1152, ch, rate, 90 + s) for s in range(2)])
    got, st = run_batch_with_stages(emu, pcm, rate, ch, 128, nf)
    got_l, st_l = run_batch_with_stages(emu, pcm, rate, ch, 128, nf, flags=64)
    assert got_l == got == [oracle.encode(pcm[s], rate, 128, ch)[0] for s in range(2)]
    compare_prep_records(st["prep"], st_l["prep"], st["psy"])

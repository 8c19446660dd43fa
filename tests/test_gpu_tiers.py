"""The six two-tier decisions on the DEVICE: forcing the exact tier of each -- and of all
of them -- must not change one emitted byte on a full-chip batch, and the raw hardware square root / exp2 that
the quantiser's first tier is built from must stay inside the error its guard band budgets."""
import ctypes

import numpy as np
import pytest

from mp3common import BatchRun, PREP_DT, PSY_DT
from stage_check import compare_prep_records

pytestmark = pytest.mark.gpu

FLAGS = {"noise": 1, "phase": 2, "psy": 4, "quant": 8, "prep": 16, "cw": 32, "all": 63}


def test_raw_sqrt_and_exp2_stay_inside_the_guard_band_budget(product):
    """k_loop.hip loop_quantize: the two estimates of x^(3/4) + 0.4054 - 0.5 are
         fma(y34, exp2_raw(-3q/16) * (1 +- 3.5e-6) / 65535, (0.4054 - 0.5 +- 2e-6) / 65535),  y34 = sqrt_raw(a * sqrt_raw(a)) [* up to 17 rescalings]
    and the band between them budgets 7e-7 relative for everything but the rescalings.  Measured on this device over
    all 2^24 floats of [1, 4) (the error repeats exactly every factor of 4), every step size the search can ask
    for, and 2^24 more exp2 arguments.  2^-24 each: the roundings of the product y34 * cq and of the sum."""
    out = (ctypes.c_double * 3)()
    assert product.lib.mp3mi_debug_fastmath_bounds(out) == 0
    e_y34, e_exp_used, e_exp_any = out[0], out[1], out[2]
    print("max relative error: y34 %.3e, exp2 (801 step sizes) %.3e, exp2 (2^24 arguments) %.3e" % (e_y34, e_exp_used, e_exp_any))
    assert 0.0 < e_y34 < 3.0e-7      # two 1-ulp roots, the inner one halved, and the rounding of a * sqrt(a)
    assert e_exp_used < 1.5e-7 and 0.0 < e_exp_any < 2.0e-7
    assert e_y34 + max(e_exp_used, e_exp_any) + 2 * 2.0 ** -24 < 7e-7


def test_the_quantisers_rounding_instruction_rounds_to_nearest(product):
    """k_loop.hip loop_quant_pair: v_cvt_pknorm_u16_f32 turns the upper and the lower estimate of x^(3/4) + 0.4054 - 0.5 (scaled
    by 1 / 65535) into integers, a pair per instruction; the proof that a line whose two roundings agree is settled needs
    |n - a * 65535| <= 0.5 and a monotone conversion.  Every float of [2^-31, 2048.5 / 65535] (218 M arguments) against double
    arithmetic, both halves of the instruction, both clamps."""
    out = (ctypes.c_double * 3)()
    assert product.lib.mp3mi_debug_pknorm_bound(out) == 0
    print("v_cvt_pknorm_u16_f32: max |n - a * 65535| = %.9f, non-monotone neighbours %d, half / clamp mismatches %d" % (out[0], out[1], out[2]))
    assert 0.25 < out[0] <= 0.5 and out[1] == 0 and out[2] == 0


@pytest.mark.parametrize("rate,ch,kbps,S,nf,stream0", [(44100, 2, 128, 4096, 24, 0), (48000, 2, 320, 1024, 20, 5000), (32000, 1, 64, 2048, 20, 9000)])
def test_every_exact_tier_gives_identical_bytes(product, oracle, rate, ch, kbps, S, nf, stream0):
    run = BatchRun(product, S, rate, ch, kbps, nf, stream0=stream0)
    try:
        base, base_len = run.encode(0)
        # the second tier of the unpredictability was exercised, and only for a few records (k_part's rounding check)
        listed, records = run.cw_fixups()
        assert 0 < listed < records // 10, "%d of %d records listed" % (listed, records)
        for name, fl in FLAGS.items():
            out, lens = run.encode(fl)
            assert np.array_equal(lens, base_len), "lengths differ with the %s tier forced" % name
            bad = np.nonzero((out != base).any(axis=1))[0]
            assert bad.size == 0, "%d streams differ with the %s tier forced (first: %d)" % (bad.size, name, bad[0])
        for s in sorted(set(np.linspace(0, S - 1, 24).astype(int).tolist())):  # and they are the oracle's bytes
            ref, _ = oracle.encode(run.pcm_of(s), rate, kbps, ch)
            assert base[s, :base_len[s]].tobytes() == ref, "stream %d differs from the oracle" % s
    finally:
        run.close()


@pytest.mark.parametrize("rate,ch,kbps,S,nf,stream0", [(44100, 2, 128, 4096, 24, 0), (48000, 2, 320, 1024, 20, 5000), (32000, 1, 64, 2048, 20, 9000)])
def test_loop_prep_of_mdct_tail_equals_k_prep(product, rate, ch, kbps, S, nf, stream0):
    """The loop's stateless head (allowed distortion, calc_scfsi's integers, the first quantiser step) comes from
    k_mdct's tail -- band energies in the reference's order, the integers from order-free sums with a margin -- and
    from k_prep, the reference's 576-line walk, only for the records the tail lists.  Both over EVERY record of a
    full-chip batch: the records agree bit for bit, and the tail lists next to nothing."""
    run = BatchRun(product, S, rate, ch, kbps, nf, stream0=stream0)
    try:
        run.encode(0)
        listed = run.prep_fixups()
        tail, psy = run.fetch(5, PREP_DT), run.fetch(0, PSY_DT)
        run.encode(FLAGS["prep"])
        walk = run.fetch(5, PREP_DT)
        n = compare_prep_records(tail, walk, psy)
        short = int((psy["block_type"] == 2).sum())
        print("%d records (%d short blocks) agree; %d listed for k_prep" % (n, short, listed))
        assert n == S * 2 * nf * ch and short > 0 and listed < n // 1000
    finally:
        run.close()


def test_records_listed_by_the_mdct_tail_are_redone_by_k_prep(product):
    """MP3MI_TEST_PREP_LIST: the tail lists every third record and spoils what it wrote for it; k_prep's walk through
    the list must restore every one of them (a third of 393 216 records), bit for bit."""
    run = BatchRun(product, 4096, 44100, 2, 128, 24)
    try:
        base, base_len = run.encode(0)
        tail, psy = run.fetch(5, PREP_DT), run.fetch(0, PSY_DT)
        out, lens = run.encode(64)
        listed = run.prep_fixups()
        assert listed == (tail.size + 2) // 3, listed
        assert np.array_equal(lens, base_len) and np.array_equal(out, base)
        compare_prep_records(tail, run.fetch(5, PREP_DT), psy)
    finally:
        run.close()


def test_listed_records_in_a_batch_that_goes_through_k_loop_in_parts(product):
    """The list of undecided records is one per batch and the items of a larger batch (two parts of 4096 streams,
    two chunks) use it in turn: with every third record listed and spoiled, the bytes stay what they are."""
    run = BatchRun(product, 8192, 44100, 2, 128, 12, options=product.options(chunk_frames=6))
    try:
        base, base_len = run.encode(0)
        out, lens = run.encode(64)
        assert np.array_equal(lens, base_len) and np.array_equal(out, base)
    finally:
        run.close()

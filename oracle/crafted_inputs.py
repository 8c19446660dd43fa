"""TEST INFRASTRUCTURE -- hand-made PCM inputs that steer the reference into the branches the synthetic bench family
never takes (tools/ref_coverage.py measures which).  oracle/gen_golden.py encodes them with the unmodified reference
and commits PCM + md5 + stage dumps under tests/golden/; the generators are kept so the fixtures can be remade, but
the tests read the committed PCM, never this module (floating-point sines differ between numpy builds).

Every generator returns int16 interleaved PCM.  `rate`, `ch` are the case's sampling rate and channel count.
"""
import numpy as np


def _t(n, rate):
    return np.arange(n, dtype=np.float64) / rate


def _i16(x):
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def _stereo(left, right):
    out = np.zeros((len(left), 2), np.int16)
    out[:, 0] = _i16(left)
    out[:, 1] = _i16(right)
    return out.reshape(-1)


def stationary_tones(rate, ch, frames=14, seed=11, amp=9000.0, noise=1.5, f=(440.0, 1370.0, 3100.0)):
    """Stationary chords with a whisper of noise: both granules of a frame look alike, so calc_scfsi sets scfsi bands
    (src/loop.c:676-711) and granule 1 copies granule 0's scalefactors (:1177-1180, :1264-1311)."""
    n = frames * 1152
    t = _t(n, rate)
    rng = np.random.default_rng(seed)
    sig = sum(amp / (k + 1) * np.sin(2 * np.pi * fk * t + 0.7 * k) for k, fk in enumerate(f))
    left = sig + rng.normal(0.0, noise, n)
    if ch == 1:
        return _i16(left)
    right = 0.8 * sig + rng.normal(0.0, noise, n)
    return _stereo(left, right)


def silence_then_noise(rate, ch, silent_frames=6, loud_frames=6, seed=3, amp=14000.0, tail_silent=0, lowpass=0):
    """Digital silence fills the bit reservoir to its limit, then full-band noise asks for everything at once:
    ResvMaxBits' second 4095 cap (src/reservoir.c:131-132), stuffing plan b and the drain into ancillary data
    (:197-216) at high bitrates, queued headers in the formatter at low ones (src/formatBitstream.c:366-374)."""
    rng = np.random.default_rng(seed)
    n0, n1, n2 = silent_frames * 1152, loud_frames * 1152, tail_silent * 1152
    x = rng.normal(0.0, amp, (n1, ch))
    if lowpass:
        k = np.ones(lowpass) / lowpass
        for c in range(ch):
            x[:, c] = np.convolve(x[:, c], k, mode="same")
    out = np.zeros((n0 + n1 + n2, ch))
    out[n0:n0 + n1] = x
    return _i16(out).reshape(-1)


def silence_then_tones(rate, ch, silent_frames=3, tone_frames=6, amp=2500.0, f=(523.0, 1046.0, 2093.0), seed=4, noise=1.0):
    """A full reservoir meets a signal of modest perceptual entropy: ResvMaxBits grants exactly the extra bits asked
    for, not the 60 % share (`frac < more_bits` false, src/reservoir.c:121-124)."""
    n0, n1 = silent_frames * 1152, tone_frames * 1152
    t = _t(n1, rate)
    rng = np.random.default_rng(seed)
    out = np.zeros((n0 + n1, ch))
    for c in range(ch):
        out[n0:, c] = sum(amp / (k + 1) * np.sin(2 * np.pi * fk * (1.0 + 0.003 * c) * t) for k, fk in enumerate(f)) + rng.normal(0.0, noise, n1)
    return _i16(out).reshape(-1)


def click_after_silence(rate, ch, frames=6, at=(3 * 1152 + 500,), width=40, amp=30000.0, seed=9):
    """Silence, then a short click late inside a granule: the granule turns into short blocks whose first windows see
    nothing but zeros -- exact-zero spectral lines beside non-zero ones (quantanf_init's `xr[i] != 0`, src/loop.c:381)."""
    rng = np.random.default_rng(seed)
    out = np.zeros((frames * 1152, ch))
    for a in at:
        out[a:a + width] = rng.uniform(-amp, amp, (width, ch))
    return _i16(out).reshape(-1)


def bursts(rate, ch, frames=14, period=2 * 576 + 300, width=120, amp=28000.0, floor=60.0, seed=21, first=700):
    """Loud bursts a little more than two granules apart over a quiet floor: attack, stop, and an attack again right
    behind the stop block (STOP -> SHORT, src/l3psy.c:689-694)."""
    rng = np.random.default_rng(seed)
    n = frames * 1152
    out = rng.normal(0.0, floor, (n, ch))
    p = first
    while p + width < n:
        out[p:p + width] += rng.uniform(-amp, amp, (width, ch))
        p += period
    return _i16(out).reshape(-1)


def full_scale_transients(rate, ch, frames=10, seed=5, period=900, width=200):
    """Full-scale square bursts between near-silence: short blocks quantised to values >= 1024 at a high bitrate
    (calc_noise's direct pow() path, src/loop.c:1061-1062; ESC codes with linbits in the short regions)."""
    rng = np.random.default_rng(seed)
    n = frames * 1152
    out = rng.normal(0.0, 2.0, (n, ch))
    p = 300
    while p + width < n:
        sign = np.where((np.arange(width) // 3) % 2 == 0, 1.0, -1.0)
        for c in range(ch):
            out[p:p + width, c] = 32767.0 * sign * (1.0 if c == 0 else -1.0)
        p += period
    return _i16(out).reshape(-1)


def faint_tone(rate, ch, frames=8, amp=3.0, f=1000.0, dc=0.0):
    """A tone of a few LSB: the flatness measure of quantanf_init runs into its floor (src/loop.c:392-393), almost
    every line quantises to zero and the big-value region shrinks to a few bands."""
    n = frames * 1152
    t = _t(n, rate)
    x = amp * np.sin(2 * np.pi * f * t) + dc
    if ch == 1:
        return _i16(x)
    return _stereo(x, amp * np.sin(2 * np.pi * f * 1.5 * t))


def loud_tone(rate, ch, frames=8, amp=32000.0, f=(1000.0,), seed=1, noise=0.0):
    """A full-scale pure tone: a handful of huge lines over nothing (linbits tables, large global_gain span,
    preemphasis off / on, scalefactor limits)."""
    n = frames * 1152
    t = _t(n, rate)
    rng = np.random.default_rng(seed)
    x = sum(amp / len(f) * np.sin(2 * np.pi * fk * t) for fk in f) + rng.normal(0.0, noise, n)
    if ch == 1:
        return _i16(x)
    return _stereo(x, x[::-1])


def pink_swell(rate, ch, frames=12, seed=17, amp=6000.0):
    """Low-passed noise whose level swells and fades: amplification of many bands, preflag, scalefac limits."""
    rng = np.random.default_rng(seed)
    n = frames * 1152
    out = np.zeros((n, ch))
    env = 0.02 + np.abs(np.sin(np.pi * np.arange(n) / n * 3))
    for c in range(ch):
        w = rng.normal(0.0, 1.0, n)
        y = np.zeros(n)
        acc = 0.0
        a = 0.93
        # one-pole low-pass, vectorised through lfilter-free cumulative form
        for i in range(n):
            acc = a * acc + w[i]
            y[i] = acc
        out[:, c] = amp * env * y / 3.0
    return _i16(out).reshape(-1)


def phase_pair(rate, ch, frames=12, f=3836.0, amp=30000.0, phase=0.5, skip=0):
    """One full-scale sine, the right channel the same sine half a radian later (tests/test_gpu_parity.py's tonal
    family): the scfsi decision finds the energies of the two granules alike but not the allowed distortions
    (`sum0 < krit && sum1 < krit` false through its second operand, src/loop.c:704)."""
    n = frames * 1152
    t = _t(n + skip * 1152, rate)[skip * 1152:]
    left = amp * np.sin(2 * np.pi * f * t)
    if ch == 1:
        return _i16(left)
    return _stereo(left, amp * np.sin(2 * np.pi * f * t + phase))


def random_blocks(rate, ch, frames=5, seed=1100, levels=(3.0, 30.0, 300.0, 3000.0), p_on=0.5):
    """Every (granule, channel) block is silent or noise of a random level: the reservoir wanders over its whole range,
    and a stream ends now and then with its last main data exactly on a slot boundary."""
    rng = np.random.default_rng(seed)
    out = np.zeros((frames * 1152, 2))
    for g in range(frames * 2):
        for c in range(2):
            if rng.random() < p_on:
                out[g * 576:(g + 1) * 576, c] = rng.normal(0, rng.choice(levels), 576)
    out = out[:, :ch]
    return _i16(out).reshape(-1)


GENERATORS = {f.__name__: f for f in (stationary_tones, silence_then_noise, silence_then_tones, click_after_silence, bursts, full_scale_transients,
                                      faint_tone, loud_tone, pink_swell, random_blocks, phase_pair)}


def make(spec, rate, ch):
    """spec: {"gen": name, **kwargs}"""
    kw = dict(spec)
    return GENERATORS[kw.pop("gen")](rate, ch, **kw)

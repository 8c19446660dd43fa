#!/usr/bin/env python3
"""Development aid for the coverage-guided corpus (TEST INFRASTRUCTURE): runs candidate inputs from
oracle/crafted_inputs.py through the instrumented reference one by one and prints, per candidate, which of the lines /
branch outcomes the golden fixtures leave unexecuted it reaches; then a greedy cover.  The chosen candidates become
CASES of oracle/gen_golden.py.

    python3 tools/cov_search.py [--list candidates.json]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_coverage as rc  # noqa: E402
import crafted_inputs as ci  # noqa: E402

CANDIDATES = [
    # name, rate, ch, kbps, spec
    ("tones44", 44100, 2, 128, {"gen": "stationary_tones"}),
    ("tones44_quiet", 44100, 2, 128, {"gen": "stationary_tones", "amp": 900.0, "noise": 0.5}),
    ("tones44_64", 44100, 2, 64, {"gen": "stationary_tones", "frames": 10}),
    ("tones48_mono", 48000, 1, 96, {"gen": "stationary_tones", "frames": 10}),
    ("tones32_192", 32000, 2, 192, {"gen": "stationary_tones", "frames": 10, "f": (250.0, 800.0, 5000.0, 9000.0)}),
    ("resv_m48_160", 48000, 1, 160, {"gen": "silence_then_noise"}),
    ("resv_m44_160", 44100, 1, 160, {"gen": "silence_then_noise"}),
    ("resv_m32_112", 32000, 1, 112, {"gen": "silence_then_noise"}),
    ("resv_m32_320", 32000, 1, 320, {"gen": "silence_then_noise", "silent_frames": 3, "loud_frames": 3, "tail_silent": 2}),
    ("resv_m32_256", 32000, 1, 256, {"gen": "silence_then_noise", "silent_frames": 3, "loud_frames": 3, "tail_silent": 2}),
    ("resv_s44_320", 44100, 2, 320, {"gen": "silence_then_noise", "silent_frames": 2, "loud_frames": 3, "tail_silent": 2}),
    ("resv_s48_256", 48000, 2, 256, {"gen": "silence_then_noise", "silent_frames": 3, "loud_frames": 3, "tail_silent": 2}),
    ("lag_s48_32", 48000, 2, 32, {"gen": "silence_then_noise", "silent_frames": 8, "loud_frames": 4, "tail_silent": 6, "amp": 3000.0, "lowpass": 8}),
    ("lag_s48_128", 48000, 2, 128, {"gen": "silence_then_noise", "silent_frames": 4, "loud_frames": 3, "tail_silent": 4, "amp": 3000.0}),
    ("lag_s44_128", 44100, 2, 128, {"gen": "silence_then_noise", "silent_frames": 4, "loud_frames": 3, "tail_silent": 5, "amp": 3000.0}),
    ("lag_m32_32", 32000, 1, 32, {"gen": "silence_then_noise", "silent_frames": 8, "loud_frames": 4, "tail_silent": 6, "amp": 3000.0, "lowpass": 8}),
    ("click", 44100, 2, 128, {"gen": "click_after_silence"}),
    ("click_b", 44100, 2, 128, {"gen": "click_after_silence", "at": (3 * 1152 + 400, 4 * 1152 + 900), "width": 20}),
    ("click_c", 44100, 1, 64, {"gen": "click_after_silence", "at": (2 * 1152 + 1000,), "width": 8}),
    ("click_d", 48000, 2, 192, {"gen": "click_after_silence", "at": (2 * 1152 + 200, 3 * 1152 + 800), "width": 60}),
    ("bursts", 44100, 2, 128, {"gen": "bursts"}),
    ("bursts_b", 44100, 2, 128, {"gen": "bursts", "period": 3 * 576 + 100}),
    ("bursts_c", 48000, 2, 96, {"gen": "bursts", "period": 2 * 576 + 40, "width": 60}),
    ("bursts_d", 32000, 1, 64, {"gen": "bursts", "period": 2 * 576 + 500, "width": 200}),
    ("transients_320", 44100, 2, 320, {"gen": "full_scale_transients"}),
    ("transients_48_320", 48000, 2, 320, {"gen": "full_scale_transients", "period": 1300, "width": 300}),
    ("transients_m_256", 44100, 1, 256, {"gen": "full_scale_transients", "period": 700, "width": 100}),
    ("faint", 44100, 2, 128, {"gen": "faint_tone"}),
    ("faint_1", 44100, 2, 128, {"gen": "faint_tone", "amp": 1.2}),
    ("faint_dc", 44100, 2, 128, {"gen": "faint_tone", "amp": 2.0, "dc": 500.0}),
    ("faint_m", 32000, 1, 64, {"gen": "faint_tone", "amp": 5.0, "f": 300.0}),
    ("loud", 44100, 2, 128, {"gen": "loud_tone"}),
    ("loud_32", 44100, 2, 32, {"gen": "loud_tone", "f": (200.0, 3000.0, 9000.0)}),
    ("loud_320", 48000, 2, 320, {"gen": "loud_tone", "f": (5000.0, 11000.0), "noise": 300.0}),
    ("crc_dual", 44100, 2, 128, {"gen": "bursts", "frames": 6, "mode": "de"}),
    ("crc_mono", 32000, 1, 64, {"gen": "bursts", "frames": 6, "mode": "me"}),
    ("resv_tone_a", 44100, 2, 128, {"gen": "silence_then_tones"}),
    ("resv_tone_b", 44100, 2, 128, {"gen": "silence_then_tones", "amp": 9000.0, "noise": 20.0}),
    ("resv_tone_c", 48000, 1, 96, {"gen": "silence_then_tones", "amp": 600.0}),
    ("resv_tone_d", 44100, 2, 96, {"gen": "silence_then_tones", "amp": 20000.0, "noise": 100.0, "f": (300.0, 2500.0, 7000.0)}),
    ("resv_faint_a", 44100, 2, 128, {"gen": "silence_then_tones", "amp": 3.0, "noise": 0.0, "f": (1000.0,)}),
    ("resv_faint_b", 44100, 2, 128, {"gen": "silence_then_tones", "amp": 8.0, "noise": 0.0, "f": (700.0, 1500.0)}),
    ("resv_faint_c", 48000, 1, 64, {"gen": "silence_then_tones", "amp": 4.0, "noise": 0.0, "f": (1000.0,)}),
    ("resv_mid_a", 44100, 2, 128, {"gen": "silence_then_noise", "silent_frames": 3, "loud_frames": 5, "amp": 60.0}),
    ("resv_mid_b", 44100, 2, 128, {"gen": "silence_then_noise", "silent_frames": 3, "loud_frames": 5, "amp": 300.0}),
    ("resv_mid_c", 44100, 2, 192, {"gen": "silence_then_noise", "silent_frames": 3, "loud_frames": 5, "amp": 1200.0}),
    ("resv_mid_d", 48000, 1, 96, {"gen": "silence_then_noise", "silent_frames": 3, "loud_frames": 5, "amp": 300.0, "lowpass": 6}),
]


def main():
    cands = CANDIDATES
    if len(sys.argv) > 2 and sys.argv[1] == "--list":
        cands = [tuple(c) for c in json.load(open(sys.argv[2]))]
    synth = rc.load_synth()
    notes_path = os.path.join(rc.GOLD, "coverage_notes.json")
    notes = json.load(open(notes_path)) if os.path.exists(notes_path) else {}
    base = rc.CovRun()
    base.run(list(rc.golden_items(synth)))
    _, open0 = rc.summarise(*base.collect(), notes)
    base.close()
    open0 = set(open0)
    print("open after the golden fixtures:", len(open0))
    reach = {}
    for name, rate, ch, kbps, spec in cands:
        spec = dict(spec)
        opts = spec.pop("mode", None)
        pcm = ci.make(spec, rate, ch)
        r = rc.CovRun()
        try:
            r.run([(name, pcm, rate, ch, kbps, opts, None)])
        except Exception as e:  # the reference aborts on some inputs: worth knowing
            print("%-18s REFERENCE FAILED: %s" % (name, e))
            r.close()
            continue
        lines, branches, funcs = r.collect()
        r.close()
        hit = set()
        for it in open0:
            f, rest = it.split(":")
            if "#" in rest:
                l, k = rest.split("#")
                if branches.get((f, int(l), int(k)), 0) > 0:
                    hit.add(it)
            elif lines.get((f, int(rest)), [0])[0] > 0:
                hit.add(it)
        reach[name] = hit
        print("%-18s frames %3d reaches %3d: %s" % (name, len(pcm) // ch // 1152, len(hit), " ".join(sorted(hit))[:400]))
    # greedy cover
    left, chosen = set().union(*reach.values()) if reach else set(), []
    while left:
        best = max(reach, key=lambda n: len(reach[n] & left))
        if not reach[best] & left:
            break
        chosen.append((best, len(reach[best] & left)))
        left -= reach[best]
    print("greedy cover:", chosen)
    still = sorted(open0 - set().union(*reach.values())) if reach else sorted(open0)
    print("still open: %d" % len(still))
    for it in still:
        print("   ", it, "|", rc.src_line(it.split(":")[0], int(it.split(":")[1].split("#")[0]))[:100])


if __name__ == "__main__":
    main()

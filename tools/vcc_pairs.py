#!/usr/bin/env python3
"""Counts, per kernel of a gfx950 assembly listing (hipcc -S --cuda-device-only), the VALU instructions that read VCC
implicitly (VOP2 v_cndmask_b32_e32 / v_addc / v_subb ... with a trailing `vcc` source) directly after another VALU
instruction that read VCC without writing it: measured on MI355X such a pair stalls the wavefront ~40 cycles, a third
and further one ~90 each (tools/exp/cndmask2.hip).  usage: vcc_pairs.py file.s [-v]"""
import re
import sys

def reads_vcc(op, args):
    if not op.startswith("v_"):
        return False
    if op.endswith("_e64") or "sdwa" in op or "dpp" in op:
        # explicit SGPR-pair operand forms named vcc behave like any SGPR pair (no penalty measured)
        return False
    a = [x.strip() for x in args.split(",")]
    if op.startswith("v_cndmask_b32"):
        return len(a) >= 4 and a[3].startswith("vcc")
    if op.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co")):
        return True
    if op.startswith("v_div_fmas"):
        return True
    return False

def writes_vcc(op, args):
    a = [x.strip() for x in args.split(",")]
    if op.startswith("v_cmp") and (a[0].startswith("vcc") or op.endswith("_e32")):
        return True
    if op.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co", "v_add_co", "v_sub_co", "v_subrev_co")) and len(a) > 1 and a[1].startswith("vcc"):
        return True
    if op.startswith("v_div_scale") and len(a) > 1 and a[1].startswith("vcc"):
        return True
    return False

def main():
    verbose = "-v" in sys.argv
    kernel = None
    stats = {}
    prev_read = False  # the previous VALU instruction read vcc and did not write it
    run = 0
    for n, line in enumerate(open(sys.argv[1]), 1):
        m = re.match(r"^(_Z\w+|k_\w+):", line)
        if m:
            kernel = m.group(1)
            prev_read = False
            continue
        t = line.strip()
        if not t or t.startswith((";", ".", "//")) or kernel is None:
            continue
        parts = t.split(None, 1)
        op, args = parts[0], (parts[1] if len(parts) > 1 else "")
        args = args.split(";")[0]
        if op.startswith("s_") and ("vcc" in args.split(",")[0]):
            prev_read = False  # SALU writes vcc: (s_and measured fast, s_mov slow; counted as a reset)
            continue
        if not op.startswith("v_"):
            continue
        r, w = reads_vcc(op, args), writes_vcc(op, args)
        st = stats.setdefault(kernel, [0, 0, 0])
        st[0] += 1
        if r and prev_read:
            run += 1
            st[1 if run == 1 else 2] += 1
            if verbose:
                print("%s:%d: %s" % (kernel[:30], n, t))
        else:
            run = 0
        prev_read = r and not w
    for k, (nv, p2, p3) in stats.items():
        if p2 or p3:
            print("%-60s VALU %6d  second-in-a-row %4d  third-or-later %4d" % (k[:60], nv, p2, p3))

main()

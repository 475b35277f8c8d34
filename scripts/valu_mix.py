#!/usr/bin/env python3
"""Static VALU instruction-class mix of a compiled kernel (from `hipcc -save-temps` assembly).

Classes by measured issue cost (scripts/microbench/valu_rate.hip, profiles/r01_h_microbench/valu_rate.txt, 8 waves per
SIMD, wall clock): "full" ~2.5-2.9 cycles per wave64 instruction per SIMD (architecturally 2: SIMD-32), "half" ~4.5
(architecturally 4), "quarter" ~8.8 (v_rcp & co).  Usage: valu_mix.py file.s <substring of the kernel symbol>"""
import collections
import re
import sys

FULL = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_fmamk_f32", "v_fmaak_f32",
        "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_write_b32", "v_accvgpr_read_b32"}
QUARTER = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}


def main(path, sym):
    text = open(path).read()
    m = re.search(r"^(%s[^\n:]*):" % ("_Z[^\n:]*" + re.escape(sym)), text, re.M)
    start = m.end()
    end = text.index("s_endpgm", start)
    body = text[start:end]
    ops = collections.Counter()
    for line in body.splitlines():
        t = line.strip().split()
        if t and t[0].startswith("v_"):
            ops[re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])] += 1
    n = sum(ops.values())
    full = sum(c for o, c in ops.items() if o in FULL)
    quarter = sum(c for o, c in ops.items() if o in QUARTER)
    half = n - full - quarter
    print("%s: %d VALU instructions: full-rate %d (%.1f %%), half-rate %d (%.1f %%), quarter-rate %d (%.1f %%)" % (
        m.group(1)[:60], n, full, 100.0 * full / n, half, 100.0 * half / n, quarter, 100.0 * quarter / n))
    arch = (2 * full + 4 * half + 8 * quarter) / n
    meas = (2.7 * full + 4.5 * half + 8.8 * quarter) / n
    print("  mean issue cost per instruction: %.2f cycles (architectural 2 / 4 / 8), %.2f cycles (measured 2.7 / 4.5 / 8.8)" % (arch, meas))
    print("  class-weighted issue ceiling at 2.4 GHz, 1024 SIMDs: %.0f G instr/s (architectural), %.0f (measured costs)" % (
        1024 * 2.4 / arch, 1024 * 2.4 / meas))
    print("  most frequent:", ", ".join("%s %d" % oc for oc in ops.most_common(14)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

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


def mix(text, sym):
    """-> (symbol, Counter of VALU opcodes, n, full, half, quarter) of the first kernel whose mangled name contains `sym`."""
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
    return m.group(1), ops, n, full, n - full - quarter, quarter


# the three render kernels as bench.py names them -> substring of the mangled symbol (non-STATS, non-TEX instantiations)
RENDER_KERNELS = {"k_trace_extend": "k_traceILb0ELb0ENS_12_GLOBAL__N_18ExtendIO", "k_trace_connect": "k_traceILb1ELb0ENS_12_GLOBAL__N_19ConnectIO",
                  "k_shade": "k_shadeILb0E"}


def write_json(path, out, digest):
    """csrc/Makefile: the static class mix of the render kernels next to the library, stamped with its source digest;
    bench.py turns it into the class-weighted VALU issue ceiling of `roofline`."""
    import json

    text = open(path).read()
    rec = {"library_digest": digest, "costs_cycles": {"full": 2, "half": 4, "quarter": 8}, "kernels": {}}
    for name, sym in RENDER_KERNELS.items():
        _, _, n, full, half, quarter = mix(text, sym)
        arch = (2 * full + 4 * half + 8 * quarter) / n
        rec["kernels"][name] = {"valu_instructions_static": n, "full": full, "half": half, "quarter": quarter, "mean_issue_cycles": arch}
    json.dump(rec, open(out, "w"), indent=1)


def main(path, sym):
    text = open(path).read()
    name, ops, n, full, half, quarter = mix(text, sym)
    m = type("M", (), {"group": lambda self, i: name})()
    print("%s: %d VALU instructions: full-rate %d (%.1f %%), half-rate %d (%.1f %%), quarter-rate %d (%.1f %%)" % (
        m.group(1)[:60], n, full, 100.0 * full / n, half, 100.0 * half / n, quarter, 100.0 * quarter / n))
    arch = (2 * full + 4 * half + 8 * quarter) / n
    meas = (2.7 * full + 4.5 * half + 8.8 * quarter) / n
    print("  mean issue cost per instruction: %.2f cycles (architectural 2 / 4 / 8), %.2f cycles (measured 2.7 / 4.5 / 8.8)" % (arch, meas))
    print("  class-weighted issue ceiling at 2.4 GHz, 1024 SIMDs: %.0f G instr/s (architectural), %.0f (measured costs)" % (
        1024 * 2.4 / arch, 1024 * 2.4 / meas))
    print("  most frequent:", ", ".join("%s %d" % oc for oc in ops.most_common(14)))


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[2] == "--json":
        write_json(sys.argv[1], sys.argv[3], sys.argv[4])
    else:
        main(sys.argv[1], sys.argv[2])

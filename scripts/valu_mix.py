#!/usr/bin/env python3
"""VALU instruction mix of the compiled render kernels (device assembly of pt_render.hip, `hipcc -S`).

Two classifications of every VALU opcode:

* ISSUE RATE, measured (scripts/microbench/valu_rate.hip -> profiles/r03_valu_rate.txt, r05_valu_rate.txt; 8 waves per SIMD):
  "full" ~2.5-2.9 cycles per wave64 instruction per SIMD (architecturally 2: SIMD-32), "half" ~4.5 (architecturally 4),
  "quarter" ~8.1-8.8 (v_rcp / v_sqrt & co).
* HARDWARE CLASS = which of the SQ_INSTS_VALU_{ADD_F32, MUL_F32, FMA_F32, TRANS_F32, INT32, INT64, CVT} counters an
  instruction ticks, or none ("OTHER"), calibrated with one rocprofv3 --pmc pass over the same microbenchmark
  (scripts/valu_class_calib.sh -> profiles/r05_valu_class_calib.txt): e.g. v_div_fmas_f32 counts as FMA_F32 (half rate),
  v_cmp_*_u32 as INT32, v_cndmask / v_cmp_*_f32 / v_max_f32 / v_div_scale / v_div_fixup / bit ops as nothing.

The rocprofv3 class counters give the DYNAMIC class mix of a kernel (what the inner loops execute, not what the binary
holds); inside a class the static listing says what it costs to issue.  Written per kernel: static count and mean issue
cost of every hardware class, so that

    dynamic mean issue cost = sum_c dyn_count[c] * static_cost[c] / SQ_INSTS_VALU        (c over the 7 classes + OTHER)

(bench.py, scripts/pmc_summary.py).  Also written: `kernel_digest` = sha256 of the three render kernels' instruction
streams (comments and label numbers stripped): the committed counter files belong to THIS device code; a change of host
code alone re-stamps the library but leaves the counters valid.

Usage: valu_mix.py file.s <substring of the kernel symbol>          (prints)
       valu_mix.py file.s --json out.json <library digest>          (csrc/Makefile)"""
import collections
import hashlib
import re
import sys

FULL = {"v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_and_b32", "v_or_b32", "v_xor_b32",
        "v_mov_b32", "v_lshrrev_b32", "v_ashrrev_i32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_fmamk_f32", "v_fmaak_f32",
        "v_not_b32", "v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_accvgpr_write_b32", "v_accvgpr_read_b32",
        "v_bitop3_b32"}
QUARTER = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"}
COST = {"full": 2, "half": 4, "quarter": 8}
HW_CLASSES = ("ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32", "INT32", "INT64", "CVT", "OTHER")
# opcodes the calibration pass ran (profiles/r05_valu_class_calib.txt); everything else is classified by its name
CALIBRATED = {
    "v_add_f32": "ADD_F32", "v_sub_f32": "ADD_F32",
    "v_mul_f32": "MUL_F32", "v_pk_mul_f32": "MUL_F32",
    "v_fma_f32": "FMA_F32", "v_fmac_f32": "FMA_F32", "v_fmamk_f32": "FMA_F32", "v_fma_mix_f32": "FMA_F32", "v_pk_fma_f32": "FMA_F32",
    "v_div_fmas_f32": "FMA_F32",
    "v_rcp_f32": "TRANS_F32", "v_rsq_f32": "TRANS_F32", "v_sqrt_f32": "TRANS_F32",
    "v_add3_u32": "INT32", "v_add_lshl_u32": "INT32", "v_add_u32": "INT32", "v_ashrrev_i32": "INT32", "v_bcnt_u32_b32": "INT32",
    "v_bfe_u32": "INT32", "v_cmp_lt_u32": "INT32", "v_ffbl_b32": "INT32", "v_lshl_add_u32": "INT32", "v_mad_u32_u24": "INT32",
    "v_max_i32": "INT32", "v_max_u32": "INT32", "v_mbcnt_lo_u32_b32": "INT32", "v_min3_u32": "INT32", "v_min_i32": "INT32",
    "v_min_u32": "INT32", "v_mul_u32_u24": "INT32", "v_mul_hi_u32": "INT32", "v_mul_lo_u32": "INT32",
    "v_cmp_lt_u64": "INT64", "v_lshl_add_u64": "INT64",
    "v_cvt_f32_u32": "CVT", "v_cvt_f32_f16": "CVT", "v_cvt_i32_f32": "CVT", "v_cvt_pk_f32_fp8": "CVT", "v_cvt_scalef32_pk_f32_fp8": "CVT",
    "v_cvt_u32_f32": "CVT", "v_cvt_f32_ubyte0": "CVT", "v_cvt_f32_ubyte1": "CVT",
    "v_alignbit_b32": "OTHER", "v_and_b32": "OTHER", "v_and_or_b32": "OTHER", "v_bfm_b32": "OTHER", "v_bitop3_b32": "OTHER",
    "v_cmp_le_f32": "OTHER", "v_cmp_class_f32": "OTHER", "v_cndmask_b32": "OTHER", "v_div_fixup_f32": "OTHER", "v_div_scale_f32": "OTHER",
    "v_dot2_f32_f16": "OTHER", "v_ldexp_f32": "OTHER", "v_lshl_or_b32": "OTHER", "v_max3_f32": "OTHER", "v_max_f32": "OTHER",
    "v_med3_f32": "OTHER", "v_min_f32": "OTHER", "v_min3_f32": "OTHER", "v_mov_b32": "OTHER", "v_not_b32": "OTHER", "v_or_b32": "OTHER",
    "v_perm_b32": "OTHER", "v_rndne_f32": "OTHER", "v_lshlrev_b32": "OTHER", "v_lshrrev_b32": "OTHER", "v_xor_b32": "OTHER",
}


def hw_class(op):
    """-> (class, calibrated?)"""
    if op in CALIBRATED:
        return CALIBRATED[op], True
    if re.match(r"v_cmpx?_\w+_(u32|i32)$", op):
        return "INT32", True   # (v_cmp_lt_u32 calibrated; the other predicates are the same instruction family)
    if re.match(r"v_cmpx?_\w+_(u64|i64)$", op):
        return "INT64", True
    if re.match(r"v_cmpx?_\w+_f32$", op):
        return "OTHER", True
    if op.startswith("v_cvt_"):
        return "CVT", True
    # not in the calibration pass: by name
    if op in ("v_subrev_f32",):
        return "ADD_F32", False
    if op in ("v_fmaak_f32", "v_mad_f32", "v_mac_f32"):
        return "FMA_F32", False
    if op in ("v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32", "v_rcp_iflag_f32"):
        return "TRANS_F32", False
    if re.search(r"_(u64|i64|b64)$", op) or op == "v_mad_u64_u32":
        return "INT64", False
    if re.search(r"_(u32|i32|u24|i24)(_b32)?$", op) or op in ("v_add_co_u32", "v_addc_co_u32", "v_sub_co_u32", "v_subb_co_u32", "v_ffbh_u32",
                                                                "v_mbcnt_hi_u32_b32", "v_bfe_i32"):
        return "INT32", False
    return "OTHER", False


def rate(op):
    return "full" if op in FULL else ("quarter" if op in QUARTER else "half")


def kernel_body(text, sym):
    m = re.search(r"^(%s[^\n:]*):" % ("_Z[^\n:]*" + re.escape(sym)), text, re.M)
    start = m.end()
    # to the END of the function (.Lfunc_endN), not to the first s_endpgm: a kernel with an early exit has code behind it
    # (r05 ADVICE: the digest must cover the whole instruction stream)
    end = re.compile(r"^\.Lfunc_end\d+:", re.M).search(text, start).start()
    return m.group(1), text[start:end]


def mix(text, sym):
    """-> (symbol, Counter of VALU opcodes, n, full, half, quarter) of the first kernel whose mangled name contains `sym`."""
    name, body = kernel_body(text, sym)
    ops = collections.Counter()
    for line in body.splitlines():
        t = line.strip().split()
        if t and t[0].startswith("v_"):
            ops[re.sub(r"_(e32|e64|sdwa|dpp)$", "", t[0])] += 1
    n = sum(ops.values())
    full = sum(c for o, c in ops.items() if o in FULL)
    quarter = sum(c for o, c in ops.items() if o in QUARTER)
    return name, ops, n, full, n - full - quarter, quarter


def class_table(ops):
    """-> {class: {"static": count, "mean_issue_cycles": cost, "uncalibrated": count}}"""
    tab = {c: {"static": 0, "cycles": 0, "uncalibrated": 0} for c in HW_CLASSES}
    for op, n in ops.items():
        c, cal = hw_class(op)
        tab[c]["static"] += n
        tab[c]["cycles"] += n * COST[rate(op)]
        if not cal:
            tab[c]["uncalibrated"] += n
    return {c: {"static": v["static"], "mean_issue_cycles": (v["cycles"] / v["static"]) if v["static"] else None, "uncalibrated": v["uncalibrated"]}
            for c, v in tab.items()}


def instruction_stream(body):
    """the kernel's instructions without comments, blank lines, assembler directives and label NUMBERS"""
    out = []
    for line in body.splitlines():
        line = line.split(";")[0].strip()
        if not line or line.startswith("."):
            if not re.match(r"\.LBB\d+_\d+:", line):
                continue
        out.append(re.sub(r"\.LBB\d+_", ".LBB_", line))
    return "\n".join(out)


# the three render kernels as bench.py names them -> substring of the mangled symbol (non-STATS, non-TEX instantiations)
RENDER_KERNELS = {"k_trace_extend": "k_traceILb0ELb0ENS_12_GLOBAL__N_18ExtendIO", "k_trace_connect": "k_traceILb1ELb0ENS_12_GLOBAL__N_19ConnectIO",
                  "k_shade": "k_shadeILb0ELb0E"}


def dynamic_mean_cost(classes, dyn, total):
    """classes: class_table() of the kernel; dyn: {hw class name: dynamic count} for the 7 counted classes; total: SQ_INSTS_VALU.
    -> (mean issue cycles, {class: dynamic share})"""
    counted = sum(dyn.get(c, 0.0) for c in HW_CLASSES[:-1])
    d = dict(dyn)
    d["OTHER"] = max(0.0, total - counted)
    cyc = 0.0
    for c in HW_CLASSES:
        cost = classes[c]["mean_issue_cycles"] if classes[c]["mean_issue_cycles"] is not None else 4.0
        cyc += d.get(c, 0.0) * cost
    return cyc / max(1.0, total), {c: d.get(c, 0.0) / max(1.0, total) for c in HW_CLASSES}


def write_json(path, out, digest):
    """csrc/Makefile: the class tables of the render kernels next to the library, stamped with its source digest and the
    digest of the kernels' instruction streams."""
    import json

    text = open(path).read()
    rec = {"library_digest": digest, "costs_cycles": COST, "kernels": {}}
    h = hashlib.sha256()
    for name, sym in RENDER_KERNELS.items():
        _, ops, n, full, half, quarter = mix(text, sym)
        arch = (2 * full + 4 * half + 8 * quarter) / n
        rec["kernels"][name] = {"valu_instructions_static": n, "full": full, "half": half, "quarter": quarter, "mean_issue_cycles": arch,
                                "classes": class_table(ops)}
        h.update(instruction_stream(kernel_body(text, sym)[1]).encode())
    rec["kernel_digest"] = h.hexdigest()[:16]
    json.dump(rec, open(out, "w"), indent=1)


def main(path, sym):
    text = open(path).read()
    name, ops, n, full, half, quarter = mix(text, sym)
    print("%s: %d VALU instructions: full-rate %d (%.1f %%), half-rate %d (%.1f %%), quarter-rate %d (%.1f %%)" % (
        name[:60], n, full, 100.0 * full / n, half, 100.0 * half / n, quarter, 100.0 * quarter / n))
    arch = (2 * full + 4 * half + 8 * quarter) / n
    meas = (2.7 * full + 4.5 * half + 8.8 * quarter) / n
    print("  mean issue cost per instruction (static mix): %.2f cycles (architectural 2 / 4 / 8), %.2f cycles (measured 2.7 / 4.5 / 8.8)" % (arch, meas))
    for c, v in class_table(ops).items():
        if v["static"]:
            print("  %-9s %5d static (%.1f %%), %.2f cycles each%s" % (c, v["static"], 100.0 * v["static"] / n, v["mean_issue_cycles"],
                                                                       ", %d classified by name only" % v["uncalibrated"] if v["uncalibrated"] else ""))
    print("  most frequent:", ", ".join("%s %d" % oc for oc in ops.most_common(14)))


if __name__ == "__main__":
    if len(sys.argv) == 5 and sys.argv[2] == "--json":
        write_json(sys.argv[1], sys.argv[3], sys.argv[4])
    else:
        main(sys.argv[1], sys.argv[2])

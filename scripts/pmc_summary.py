#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass, scripts/pmc_bench.sh) by kernel.

Writes, next to the passes:
  summary.txt (stdout)   per-kernel per-launch averages of every counter, L2 hit rate, wave-cycle split, lane use
  pmc_bench.json         for the three render kernels the PER-DISPATCH counter values in dispatch order -- bench.py takes
                         the last K dispatches (K = its own timed launches: the timed region is the tail of the run) so
                         that counters and HIP-event times cover exactly the same launches -- plus the FETCH_SIZE
                         calibration of scripts/microbench/fetch_calib.hip
FETCH_SIZE / WRITE_SIZE are in KiB.  gfx950: FETCH_SIZE tallies 64 B per 128-B request for wide coalesced reads
(MI355X_MICROARCH.md, HBM section); whether that also holds for the traversal's divergent 16-B gathers is what the
calibration measures (known bytes / counter bytes).
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    if "k_trace" in name:
        io = "extend" if "ExtendIO" in name else "connect" if "ConnectIO" in name else "test"
        stats = "_stats" if re.search(r"k_trace<(false|true|\d), true", name) else ""
        return "k_trace_" + io + stats
    m = re.search(r"(k_[a-z_0-9]+)(<[a-z]+>)?", name)
    if m and m.group(1) == "k_shade":  # <false> = the reference's path (what bench.py runs), <true> = textured scenes
        return "k_shade" if (m.group(2) or "<false>") == "<false>" else "k_shade_tex"
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


KEEP = ("k_trace_extend", "k_trace_connect", "k_shade")


def read_pass(root, name):
    """-> {kernel: {counter: [value per dispatch, in dispatch order]}}"""
    per = defaultdict(lambda: defaultdict(dict))
    for f in glob.glob(os.path.join(root, name, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            d = int(r["Dispatch_Id"])
            per[k][r["Counter_Name"]][d] = per[k][r["Counter_Name"]].get(d, 0.0) + float(r["Counter_Value"])
    return {k: {c: [v for _, v in sorted(dd.items())] for c, dd in cs.items()} for k, cs in per.items()}


def read_trace(root, name):
    dur = defaultdict(list)
    for f in glob.glob(os.path.join(root, name, "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted(csv.DictReader(open(f)), key=lambda r: float(r["Start_Timestamp"]))
        for r in rows:
            dur[short(r["Kernel_Name"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    return dur


def main(root):
    counters = defaultdict(dict)  # kernel -> counter -> list
    for p in ("fetch", "write", "l2", "sq1", "sq2", "grbm", "cls"):
        for k, cs in read_pass(root, p).items():
            counters[k].update(cs)
    dur = read_trace(root, "trace") or read_trace(root, "grbm")
    out = {"kernels": {}, "source": os.path.basename(root)}
    for k in sorted(counters, key=lambda k: -sum(dur.get(k, [0.0]))):
        if not k.startswith("k_"):
            continue
        c = counters[k]
        n = len(dur.get(k, [])) or 1
        avg = {name: sum(v) / max(1, len(v)) for name, v in c.items()}
        avg_us = sum(dur.get(k, [0.0])) / n
        print("== %s: %d launches, avg %.1f us (kernel trace without counters)" % (k, len(dur.get(k, [])), avg_us))
        for name in sorted(c):
            print("   %-28s %16.1f per launch (%d dispatches)" % (name, avg[name], len(c[name])))
        if "FETCH_SIZE" in avg:
            print("   memory side: FETCH_SIZE %.1f MB raw per launch, WRITE_SIZE %.1f MB" % (avg["FETCH_SIZE"] * 1024 / 1e6, avg.get("WRITE_SIZE", 0.0) * 1024 / 1e6))
        if "TCC_HIT_sum" in avg:
            print("   L2 hit rate %.1f %%" % (100 * avg["TCC_HIT_sum"] / max(1.0, avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])))
        if "SQ_WAVE_CYCLES" in avg:
            wc = avg["SQ_WAVE_CYCLES"]
            print("   wave cycles: wait_any %.1f %%  wait_inst %.1f %%  active_inst %.1f %%  (valu %.1f %%)" % (
                100 * avg.get("SQ_WAIT_ANY", 0) / wc, 100 * avg.get("SQ_WAIT_INST_ANY", 0) / wc,
                100 * avg.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * avg.get("SQ_ACTIVE_INST_VALU", 0) / wc))
        if "SQ_THREAD_CYCLES_VALU" in avg and "SQ_INSTS_VALU" in avg:
            print("   lanes enabled per VALU instruction: %.3f of 64" % (avg["SQ_THREAD_CYCLES_VALU"] / max(1.0, avg["SQ_INSTS_VALU"]) / 64.0))
        if "SQ_LDS_BANK_CONFLICT" in avg and "SQ_ACTIVE_INST_LDS" in avg:
            print("   LDS: bank-conflict cycles %.1f M vs LDS-instruction cycles %.1f M" % (avg["SQ_LDS_BANK_CONFLICT"] / 1e6, avg["SQ_ACTIVE_INST_LDS"] / 1e6))
        if k in KEEP:
            out["kernels"][k] = {"trace_us": dur.get(k, []), "counters": c}
    # ---- FETCH_SIZE calibration
    cal = read_pass(root, "calib")
    for k, cs in read_pass(root, "calibw").items():  # WRITE_SIZE pass over the same microbenchmark
        cal.setdefault(k, {}).update(cs)
    txt = os.path.join(root, "fetch_calib.txt")
    if cal and os.path.exists(txt):
        t = open(txt).read()
        print("== FETCH_SIZE calibration (scripts/microbench/fetch_calib.hip)")
        print(t.strip())
        calib = {}
        m = re.search(r"k_gather64: (\d+) lanes x (\d+) records x 64 B", t)
        mw = re.search(r"k_wscatter16: (\d+) lanes x (\d+) quads x 16 B", t)
        for k, cs in cal.items():
            if "k_gather64" in k and m and "FETCH_SIZE" in cs:
                known, counter, cname = float(m.group(1)) * float(m.group(2)) * 64.0, "FETCH_SIZE", "fetch_size_bytes"
            elif "k_stream" in k and "FETCH_SIZE" in cs:
                known, counter, cname = float(1 << 26) * 64.0, "FETCH_SIZE", "fetch_size_bytes"
            elif "k_wstream" in k and "WRITE_SIZE" in cs:
                known, counter, cname = float(1 << 26) * 64.0, "WRITE_SIZE", "write_size_bytes"
            elif "k_wscatter16" in k and mw and "WRITE_SIZE" in cs:
                known, counter, cname = float(mw.group(1)) * float(mw.group(2)) * 16.0, "WRITE_SIZE", "write_size_bytes"
            else:
                continue
            counter_bytes = sum(cs[counter]) * 1024.0
            calib[k] = {"known_bytes": known, cname: counter_bytes, "known_over_counter": known / max(1.0, counter_bytes)}
            print("   %-12s known %.1f MB, %s %.1f MB -> known / counter = %.3f" % (k, known / 1e6, counter, counter_bytes / 1e6, known / max(1.0, counter_bytes)))
        out["fetch_calibration"] = calib
    # ---- which bench run this was: the JSON line bench.py printed under the kernel-trace pass
    for log in ("trace.log", "grbm.log", "sq1.log"):
        try:
            lines = [l for l in open(os.path.join(root, log)) if l.startswith('{"metric"')]
        except OSError:
            continue
        if lines:
            b = json.loads(lines[-1])
            out["bench_config"] = {k: b["config"][k] for k in ("workload", "triangles", "resolution", "spp_per_step", "primary_memo") if k in b["config"]}
            out["steps"], out["warmup"] = b["steps"], b["warmup"]
            out["timed_launches"] = b["roofline"]["launches"]
            out["library_digest"] = b["config"].get("library_digest")
            out["kernel_digest"] = b["config"].get("kernel_digest")  # bench.py refuses the file for any other device code
            out["bench_value_under_profiler"] = b["value"]
            # what the host made of this workload: launch count and work per launch of the timed region.  bench.py takes the file as
            # an EXACT match only when its own run agrees (a host-side change of the pool / chunk / grid heuristics moves these while
            # the kernels' instruction streams -- kernel_digest -- stay the same: r05 ADVICE)
            out["launch_shape"] = {"extension_rays": b["config"].get("extension_rays"), "shadow_rays": b["config"].get("shadow_rays"),
                                   "launches": b["roofline"]["launches"]}
            print("== bench under the kernel-trace pass: %.1f %s, %d timed k_trace<ExtendIO> launches of %.3f ms (HIP events)" % (
                b["value"], b["unit"], b["roofline"]["launches"], b["roofline"]["avg_launch_ms"]))
            k = out["kernels"].get("k_trace_extend")
            if k and len(k["trace_us"]) >= out["timed_launches"]:
                t = k["trace_us"][-out["timed_launches"]:]
                print("   rocprofv3 kernel trace, the same %d launches: avg %.3f ms" % (len(t), sum(t) / len(t) / 1e3))
            break
    json.dump(out, open(os.path.join(root, "pmc_bench.json"), "w"))
    # ---- dynamic VALU class mix of the timed launches: hardware class counters x issue cost per class
    mixf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpuspectral_amd", "lib", "valu_mix.json")
    if os.path.exists(mixf) and out.get("timed_launches"):
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        import valu_mix as vmx

        vm = json.load(open(mixf))
        take = out["timed_launches"]
        rec = {"source": out["source"], "library_digest": vm.get("library_digest"), "kernel_digest": vm.get("kernel_digest"),
               "timed_launches": take, "costs_cycles": vm.get("costs_cycles"),
               "method": "dynamic count per hardware class (rocprofv3 SQ_INSTS_VALU_* summed over the timed launches; OTHER = SQ_INSTS_VALU - the seven) "
                         "x mean issue cost of the class's instructions in the kernel's binary (scripts/valu_mix.py)", "kernels": {}}
        print("== dynamic VALU class mix of the timed launches")
        for kname, k in out["kernels"].items():
            c = k["counters"]
            tab = (vm["kernels"].get(kname) or {}).get("classes")
            if not tab or "SQ_INSTS_VALU_FMA_F32" not in c or len(c.get("SQ_INSTS_VALU", [])) < take:
                continue
            total = float(sum(c["SQ_INSTS_VALU"][-take:]))
            dyn = {cl: float(sum(c.get("SQ_INSTS_VALU_" + cl, [0.0])[-take:])) for cl in vmx.HW_CLASSES[:-1]}
            cyc, shares = vmx.dynamic_mean_cost(tab, dyn, total)
            n_static = vm["kernels"][kname]["valu_instructions_static"]
            rec["kernels"][kname] = {"SQ_INSTS_VALU": total, "mean_issue_cycles_dynamic": cyc, "mean_issue_cycles_static": vm["kernels"][kname]["mean_issue_cycles"],
                                     "classes": {cl: {"dynamic_share": shares[cl], "static_share": tab[cl]["static"] / n_static,
                                                      "issue_cycles": tab[cl]["mean_issue_cycles"]} for cl in vmx.HW_CLASSES}}
            print("   %-16s mean issue cost %.3f cycles (dynamic mix) vs %.3f (static mix of the binary): " % (kname, cyc, vm["kernels"][kname]["mean_issue_cycles"]) +
                  "  ".join("%s %.1f %% (static %.1f %%)" % (cl, 100 * shares[cl], 100 * tab[cl]["static"] / n_static) for cl in vmx.HW_CLASSES))
        json.dump(rec, open(os.path.join(root, "valu_mix_dynamic.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])

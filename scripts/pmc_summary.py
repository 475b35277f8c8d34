#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSVs (one directory per pass) by kernel: per-launch averages.

FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB... (gfx950: FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads -- MI355X_MICROARCH.md, HBM section; the raw and the
doubled figure are both printed)."""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict


def short(name):
    if "k_trace" in name:
        io = "fused" if "FusedIO" in name else "extend" if "ExtendIO" in name else "connect" if "ConnectIO" in name else "test"
        stats = "_stats" if re.search(r"k_trace<(false|true|\d), true", name) else ""
        return "k_trace_" + io + stats
    m = re.search(r"(k_[a-z_0-9]+)(<[a-z]+>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main(root):
    per = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))  # kernel -> counter -> [sum, dispatches]
    dur = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, "*", "**", "*counter_collection.csv"), recursive=True):
        seen = defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            c = r["Counter_Name"]
            per[k][c][0] += float(r["Counter_Value"])
            seen[(k, c)].add(r["Dispatch_Id"])
        for (k, c), ids in seen.items():
            per[k][c][1] += len(ids)
    for f in glob.glob(os.path.join(root, "grbm", "**", "*kernel_trace.csv"), recursive=True) or glob.glob(os.path.join(root, "*", "**", "*kernel_trace.csv"), recursive=True)[:1]:
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            dur[k][0] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
            dur[k][1] += 1
    out = {}
    for k in sorted(per, key=lambda k: -dur[k][0]):
        if not k.startswith("k_"):
            continue
        c = per[k]
        n = max(1, dur[k][1])
        line = {"launches": dur[k][1], "avg_us": dur[k][0] / n / 1e3}
        for name, (s, d) in c.items():
            line[name] = s / max(1, d)
        out[k] = line
        print("== %s: %d launches, avg %.1f us (profiled)" % (k, dur[k][1], line["avg_us"]))
        for name in sorted(c):
            print("   %-28s %14.1f per launch" % (name, line[name]))
        if "FETCH_SIZE" in line:
            fb = line["FETCH_SIZE"] * 1024.0
            wb = line.get("WRITE_SIZE", 0.0) * 1024.0
            print("   HBM read  %.1f MB/launch raw (x2 = %.1f MB if all wide-coalesced), write %.1f MB" % (fb / 1e6, 2 * fb / 1e6, wb / 1e6))
            line["hbm_read_bytes_raw"] = fb
            line["hbm_write_bytes"] = wb
        if "TCC_HIT_sum" in line:
            h, m = line["TCC_HIT_sum"], line["TCC_MISS_sum"]
            print("   L2 hit rate %.1f %%" % (100 * h / max(1.0, h + m)))
        if "SQ_WAVE_CYCLES" in line:
            wc = line["SQ_WAVE_CYCLES"]
            print("   wave cycles: wait_any %.1f %%  wait_inst %.1f %%  active_inst %.1f %%  (valu %.1f %%)" % (
                100 * line.get("SQ_WAIT_ANY", 0) / wc, 100 * line.get("SQ_WAIT_INST_ANY", 0) / wc,
                100 * line.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * line.get("SQ_ACTIVE_INST_VALU", 0) / wc))
        if "SQ_THREAD_CYCLES_VALU" in line and "SQ_ACTIVE_INST_VALU" in per[k]:
            pass
    json.dump(out, open(os.path.join(root, "summary.json"), "w"), indent=1)
    ext = out.get("k_trace_fused") or out.get("k_trace_extend")
    ext_name = "k_trace<2, FusedIO>" if out.get("k_trace_fused") else "k_trace<ExtendIO>"
    if ext and "hbm_read_bytes_raw" in ext:
        # gfx950: FETCH_SIZE tallies 64 B per 128-B request for 16-B-per-lane loads (MI355X_MICROARCH.md,
        # HBM section) -> reads doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
        json.dump({"kernel": ext_name, "launches": ext["launches"], "avg_us_profiled": ext["avg_us"],
                   "fetch_size_kib_per_launch": ext["FETCH_SIZE"], "write_size_kib_per_launch": ext.get("WRITE_SIZE", 0.0),
                   "hbm_bytes_per_launch": 2.0 * ext["hbm_read_bytes_raw"] + ext.get("hbm_write_bytes", 0.0),
                   "note": "2 x FETCH_SIZE + WRITE_SIZE (gfx950 correction for 16-B-per-lane loads); memory-side L2 requests, Infinity-Cache hits included"},
                  open(os.path.join(root, "pmc_extend.json"), "w"), indent=1)


if __name__ == "__main__":
    main(sys.argv[1])

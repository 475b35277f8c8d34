"""Fills DESIGN.md's table of current numbers from the final evidence of a round:
python scripts/design_numbers.py profiles/r04_bench_n1_s20_w5.json profiles/r04_bench_n1.json  -> markdown on stdout"""
import json, sys
rows = []
for f in sys.argv[1:]:
    d = json.load(open(f)); r = d["roofline"]; c = d["config"]
    fr = r["fractions"]
    def pct(x): return "—" if x is None else "%.2f" % x
    rows.append("| `%s` (`--steps %d --warmup %d`) | **%.0f Mrays/s, %.1f Msamples/s**, %.1f ms per 64-spp step | closest-hit %.1f ms / shade %.1f / any-hit %.1f per step; `bound` = `%s` %.2f; fractions: VALU issue %s (class-weighted with the dynamic mix %s; %s at the %s GHz the chip held), L1 request %s, HBM %s; %.1f VALU instr per ray at %.2f lanes, %.1f lane loads per ray (%.2f of %.2f node visits from LDS), %.0f %% of wave cycles waiting, L2 hit %.0f %% | CPU oracle %.1f Mrays/s on %d threads (%.2f on one) |" % (
        f.split("/")[-1], d["steps"], d["warmup"], d["value"], d["msamples_per_s"], d["ms_per_step"],
        r["extend_ms"] / d["steps"], r["shade_ms"] / d["steps"], r["connect_ms"] / d["steps"], r["bound"], r["frac"] or 0,
        pct(fr["valu_issue"]), pct(fr["valu_issue_weighted"]), pct(fr.get("valu_issue_weighted_at_clock")), pct(r.get("shader_clock_ghz")), pct(fr["l1_request"]), pct(fr["hbm"]),
        r["valu_instr_per_ray"] or 0, r["lanes_per_instr"] or 0, r["lane_loads_per_ray"], r["lds_nodes_per_ray"], r["nodes_per_ray"],
        100 * (r["wait_share"] or 0), 100 * (r["l2_hit_rate"] or 0),
        d.get("cpu_baseline", {}).get("value", 0), d.get("cpu_baseline", {}).get("cores", 0), d.get("cpu_baseline", {}).get("single_thread_value", 0)))
    for k, v in r["other_kernels"].items():
        rows.append("| &nbsp;&nbsp;`%s` | | class-weighted issue %s (%s at %s GHz; plain %s), L1 request %s, HBM %s, %.0f %% waiting, lanes %.2f | |" % (
            k, pct(v.get("issue_frac_weighted")), pct(v.get("issue_frac_weighted_at_clock")), pct(v.get("shader_clock_ghz")), pct(v.get("issue_frac")), pct(v.get("l1_request_frac")), pct(v.get("hbm_frac")),
            100 * (v.get("wait_share") or 0), v.get("lanes_per_instr") or 0))
print("| bench line | throughput | dominant kernel `k_trace<ExtendIO>` and the other two | CPU baseline |\n|---|---|---|---|")
print("\n".join(rows))

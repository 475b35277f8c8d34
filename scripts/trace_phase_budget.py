#!/usr/bin/env python3
"""Phase budget of the traversal kernel k_trace (pt_wavetrace.h), both hot instantiations (ExtendIO = closest hit, ConnectIO =
any hit): wave64 VALU instructions and enabled lane-instructions per ray in each phase of its loop.

    python scripts/trace_phase_budget.py static  OUT.json       # here: no GPU.  The device assembly with line tables
                                                                 # (-gline-tables-only: same instruction stream, checked) ->
                                                                 # instructions per phase, by the pt_wavetrace.h line each one
                                                                 # is inlined into
    GSP_LIB_PATH=<-DGSP_WAVE_PROFILE build> python scripts/trace_phase_budget.py gpu OUT.json
                                                                 # GPU box: how often each phase runs and with how many lanes
                                                                 # (bench workload, 64 spp behind a warm-up)
    python scripts/trace_phase_budget.py report STATIC.json DYN.json [BENCH.json [AB_PADDING.txt]] > profiles/r06_trace_phase_budget.txt

Phases = the sections of the loop, found by their marker comments in pt_wavetrace.h (so the budget follows edits):
  commit.test   ballots + threshold test at the top of every loop pass
  commit.store  io.store of the finished rays (runs when the batch threshold is met)
  refill.test   idle ballot + threshold, every pass
  refill.body   the hand-out loop: rank, io.load, make_raybox / make_shear_rot (the ray's set-up), stack reset
  schedule      "what can run": node / leaf ballots, the step decision
  node          one node step: group_next, record fetch (LDS copy or HBM), node_step, stack push / pop, triangle-group hand-over
  leaf          one leaf step: packet fetch, intersect_tri_rot, hit update
An instruction belongs to the phase whose source lines it was inlined into (outermost pt_wavetrace.h location of its .loc chain);
`glue` = instructions the compiler gives line 0 (moves at joins, loop rotation), charged to the phase of the previous instruction.
Dynamic count of a phase = static count x executions (GPU counters); node-step record fetches are weighted by the path taken.
Cross-check: the sum over phases must land on SQ_INSTS_VALU / ray of the PMC passes (same kernels, no instrumentation).
"""
import ctypes
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpuspectral_amd", "csrc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-slp-vectorize"]
KERNELS = {"extend": "k_traceILb0ELb0ENS_12_GLOBAL__N_18ExtendIOE", "connect": "k_traceILb1ELb0ENS_12_GLOBAL__N_19ConnectIOE"}
MARKS = [("loop", "  for (;;) {"), ("commit", "// ---- commit finished rays"), ("refill", "// ---- refill idle lanes"),
         ("schedule", "// ---- what can run?"), ("node", "// ---- one node step"), ("leaf", "// ---- leaf step"), ("end", "#undef tmin")]


def phase_lines():
    """-> [(first line, phase)] of pt_wavetrace.h inside k_trace, plus the line sets of the conditional sub-blocks"""
    src = open(os.path.join(CSRC, "pt_wavetrace.h")).read().split("\n")
    k0 = next(i for i, l in enumerate(src) if "void k_trace(" in l)
    at = {}
    for name, text in MARKS:
        at[name] = next(i for i in range(k0, len(src)) if src[i].lstrip().startswith(text.strip()) or src[i].startswith(text)) + 1
    bounds = [(k0 + 1, "prologue"), (at["loop"], "prologue"), (at["commit"], "commit.test"), (at["refill"], "refill.test"),
              (at["schedule"], "schedule"), (at["node"], "node"), (at["leaf"], "leaf"), (at["end"], "epilogue")]
    # sub-blocks: the commit's store block, the refill's hand-out loop
    store0 = next(i for i in range(at["commit"], at["refill"]) if "if (pending) {" in src[i]) + 1
    store1 = next(i for i in range(store0, at["refill"]) if src[i].startswith("          }")) + 1
    body0 = next(i for i in range(at["refill"], at["schedule"]) if "while (idle_m)" in src[i]) + 1
    body1 = at["schedule"] - 1
    idle0 = next(i for i in range(at["schedule"], at["node"]) if "if ((node_m | leaf_m) == 0) {" in src[i]) + 1
    return bounds, (store0, store1), (body0, body1), (idle0, idle0 + 3)


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("s_waitcnt", "s_nop", "s_barrier")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def kernel_text(asm, sym):
    m = re.search(r"^(_Z[^\n:]*%s[^\n:]*):" % re.escape(sym), asm, re.M)
    end = re.compile(r"^\.Lfunc_end\d+:", re.M).search(asm, m.end())
    return asm[m.end():end.start()]


def stream(body):
    """instructions only (no directives / labels / comments), label numbers removed: to compare two builds"""
    out = []
    for line in body.split("\n"):
        line = line.split(";")[0].strip()
        if not line or line.startswith(".") or line.endswith(":"):
            continue
        out.append(re.sub(r"\.L\w+", ".L", line))
    return out


def static(out_path):
    bounds, store, body, idle_path = phase_lines()
    with tempfile.TemporaryDirectory() as tmp:
        g = os.path.join(tmp, "g.s")
        plain = os.path.join(tmp, "p.s")
        for path, extra in ((g, ["-gline-tables-only"]), (plain, [])):
            subprocess.check_call(["hipcc"] + FLAGS + extra + ["--cuda-device-only", "-S", "pt_render.hip", "-o", path], cwd=CSRC,
                                  stderr=subprocess.DEVNULL)
        asm_g, asm_p = open(g).read(), open(plain).read()
    res = {"flags": FLAGS, "kernels": {}}
    for kname, sym in KERNELS.items():
        tg, tp = kernel_text(asm_g, sym), kernel_text(asm_p, sym)
        same = stream(tg) == stream(tp)
        # pass 1: (phase or None for line-0 glue, op) per instruction; glue takes the phase of the NEXT located instruction (the moves
        # the compiler puts in front of a block belong to that block), at the end of the kernel the previous one
        seq, cur = [], None
        for line in tg.split("\n"):
            ls = line.strip()
            if ls.startswith(".loc"):
                locs = re.findall(r"([\w./+-]+):(\d+):\d+", ls.split(";", 1)[1] if ";" in ls else "")
                wt = [(f, int(n)) for f, n in locs if f.endswith("pt_wavetrace.h")]
                if wt and wt[-1][1] > 0:
                    ln = wt[-1][1]
                    ph = [p_ for first, p_ in bounds if first <= ln][-1]
                    if ph == "commit.test" and store[0] <= ln <= store[1]:
                        ph = "commit.store"
                    if ph == "refill.test" and body[0] <= ln <= body[1]:
                        ph = "refill.body"
                    if ph == "schedule" and idle_path[0] <= ln <= idle_path[1]:
                        ph = "schedule.idle"  # `continue` with every lane idle: a dozen loop-carried moves, run a few times per wave
                    cur = ph
                else:
                    cur = None
                continue
            code = ls.split(";")[0].strip()
            if not code or code.startswith(".") or code.endswith(":"):
                continue
            seq.append([cur, code.split()[0]])
        nxt = "epilogue"
        for item in reversed(seq):
            if item[0] is None:
                item[0] = nxt
            else:
                nxt = item[0]
        counts = {}
        for ph, op in seq:
            cls = classify(op)
            d = counts.setdefault(ph, {"valu": 0, "salu": 0, "vmem": 0, "lds": 0, "wait": 0, "branch": 0, "other": 0, "ops": {}})
            d[cls] += 1
            if cls in ("vmem", "lds"):
                d["ops"][op] = d["ops"].get(op, 0) + 1
            if ph == "node" and op in ("global_store_dword", "global_load_dword") and kname == "extend":
                d["spill_ops"] = d.get("spill_ops", 0) + 1
        # the node step exists in `copies` copies (rep 0 peeled + the loop over rep 1..): ds_read_b128 of the LDS node fetch / 4 quads
        counts.setdefault("node", {}).setdefault("ops", {})
        counts["node"]["copies"] = max(1, counts["node"]["ops"].get("ds_read_b128", 4) // 4)
        res["kernels"][kname] = {"symbol": sym, "same_instruction_stream_as_plain_build": same, "instructions": len(stream(tp)), "phases": counts}
    json.dump(res, open(out_path, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(k, "same stream as the plain build:", v["same_instruction_stream_as_plain_build"], "| VALU per phase:",
              {p: c.get("valu", 0) for p, c in sorted(v["phases"].items())}, "| node copies", v["phases"]["node"]["copies"])


def gpu(out_path):
    sys.path.insert(0, ROOT)
    import gpuspectral_amd as g
    from gpuspectral_amd import scenes

    L = ctypes.CDLL(os.environ["GSP_LIB_PATH"])
    out = (ctypes.c_ulonglong * 48)()
    res = {}
    with g.Context(0) as ctx:
        ctx.upload_scene(scenes.interior(1_000_000, seed=7))
        ctx.frame_begin(1920, 1080)
        ctx.render(spp=64)
        ctx.sync()
        L.gsp_debug_wave_profile(out)  # (read + clear: the warm-up does not count)
        ctx.reset_stats()
        ctx.render(spp=64, first_timestamp=64, collect_kernel_times=1)
        ctx.sync()
        L.gsp_debug_wave_profile(out)
        st = ctx.stats()
    names = ["node_steps", "node_lanes", "leaf_steps", "leaf_lanes", "loop_passes", "refill_passes", "refill_idle_lanes", "node_idle_lanes",
             "node_stalled_lanes", "endgame_node_steps", "endgame_node_lanes", "commit_blocks", "commit_lanes", "refill_inner", "node_steps_hbm",
             "node_lanes_hbm", "node_steps_lds", "node_lanes_lds", "rays_refilled"]
    for kind, k in (("extend", 0), ("connect", 1)):
        res[kind] = {n: int(out[24 * k + i]) for i, n in enumerate(names)}
    res["stats"] = {k: st[k] for k in ("extension_rays", "shadow_rays", "memoised_rays", "memo_build_rays", "extend_kernel_ms", "connect_kernel_ms",
                                       "shade_kernel_ms", "extend_launches", "samples")}
    res["workload"] = "scenes.interior(1_000_000, seed=7), 1920x1080, 64 spp behind a 64-spp warm-up, -DGSP_WAVE_PROFILE build"
    json.dump(res, open(out_path, "w"), indent=1)
    print(json.dumps(res))


def report(static_path, dyn_path, bench_path=None, pad_path=None):
    S, D = json.load(open(static_path)), json.load(open(dyn_path))
    bench = json.loads(open(bench_path).read().strip().splitlines()[-1]) if bench_path else None
    st = D["stats"]
    rays = {"extend": st["extension_rays"] - st["memoised_rays"] + st["memo_build_rays"], "connect": st["shadow_rays"]}
    print("# k_trace phase budget (scripts/trace_phase_budget.py).")
    print("# Static: VALU instructions per phase of the SHIPPED kernels, by the pt_wavetrace.h line each instruction is inlined into (device")
    print("# assembly with line tables; instruction stream identical to the plain build: %s).  The node step exists in two copies (rep 0 peeled +" % all(k["same_instruction_stream_as_plain_build"] for k in S["kernels"].values()))
    print("# the loop over rep 1..): its count is per copy.  Dynamic: executions and enabled lanes of each phase, -DGSP_WAVE_PROFILE build,")
    print("# %s." % D["workload"])
    print("# instr/ray = static x executions / rays (what SQ_INSTS_VALU / rays measures); lane-instr/ray = x enabled lanes (SQ_THREAD_CYCLES_VALU).")
    model = {}
    for kind in ("extend", "connect"):
        ph, d, n = S["kernels"][kind]["phases"], D[kind], float(rays[kind])
        loops, nsteps, lsteps = d["loop_passes"], d["node_steps"], d["leaf_steps"]
        copies = ph["node"].get("copies", 1)
        node_lanes, leaf_lanes = d["node_lanes"] / max(1, nsteps), d["leaf_lanes"] / max(1, lsteps)
        rows = [("commit.test", loops, 64.0, 1), ("commit.store", d["commit_blocks"], d["commit_lanes"] / max(1, d["commit_blocks"]), 1),
                ("refill.test", loops, 64.0, 1), ("refill.body", d["refill_inner"], d["rays_refilled"] / max(1, d["refill_inner"]), 1),
                ("schedule", loops, 64.0, 1), ("node", nsteps, node_lanes, copies), ("leaf", lsteps, leaf_lanes, 1)]
        print("\n== k_trace<%s>: %.0f M rays; per ray (x 1/64 = per wave): %.2f loop passes, %.2f node steps at %.1f lanes, %.2f leaf steps at %.1f lanes, "
              "%.2f commit blocks at %.1f lanes, %.2f refill passes at %.1f rays" % (
                  "ExtendIO" if kind == "extend" else "ConnectIO", n / 1e6, loops * 64 / n, nsteps * 64 / n, node_lanes, lsteps * 64 / n, leaf_lanes,
                  d["commit_blocks"] * 64 / n, d["commit_lanes"] / max(1, d["commit_blocks"]), d["refill_inner"] * 64 / n, d["rays_refilled"] / max(1, d["refill_inner"])))
        print("   node steps: %.1f lanes read the record from HBM / L2, %.1f from the LDS copy; %.1f lanes have no ray, %.1f wait for a leaf step (both triangle groups taken);"
              " end game (hand-out empty) %.1f %% of the steps at %.1f lanes" % (
                  d["node_lanes_hbm"] / max(1, nsteps), d["node_lanes_lds"] / max(1, nsteps), d["node_idle_lanes"] / max(1, nsteps), d["node_stalled_lanes"] / max(1, nsteps),
                  100.0 * d["endgame_node_steps"] / max(1, nsteps), d["endgame_node_lanes"] / max(1, d["endgame_node_steps"])))
        print("   %-13s %5s %5s %5s %4s | %11s %6s | %9s %6s | %14s" % ("phase", "VALU", "SALU", "VMEM", "LDS", "executions", "lanes", "instr/ray", "share", "lane-instr/ray"))
        tot_i = tot_l = 0.0
        table = []
        for name, execs, lanes, div in rows:
            c = ph.get(name, {})
            valu = c.get("valu", 0) / div
            ipr = valu * execs / n
            table.append((name, valu, c.get("salu", 0) / div, c.get("vmem", 0) / div, c.get("lds", 0) / div, execs, lanes, ipr, ipr * lanes))
            tot_i += ipr
            tot_l += ipr * lanes
        for name, valu, salu, vmem, lds, execs, lanes, ipr, lpr in table:
            print("   %-13s %5.0f %5.0f %5.0f %4.0f | %11d %6.1f | %9.2f %5.1f%% | %14.1f" % (name, valu, salu, vmem, lds, execs, lanes, ipr, 100.0 * ipr / tot_i, lpr))
        print("   %-13s %46s | %9.2f        | %14.1f" % ("sum (model)", "", tot_i, tot_l))
        model[kind] = {r[0]: r[7] for r in table}
        model[kind]["sum"] = tot_i
        model[kind]["measured"] = None
        if bench:
            rf = bench["roofline"]
            if kind == "extend":
                meas_i, meas_l = rf.get("valu_instr_per_ray"), rf.get("valu_lane_instr_per_ray")
            else:
                k = rf["other_kernels"]["k_trace<ConnectIO>"]
                meas_i = k["valu_ginstr_s"] * 1e9 * rf["connect_ms"] * 1e-3 / bench["config"]["shadow_rays"] if k.get("valu_ginstr_s") else None
                meas_l = meas_i * k["lanes_per_instr"] * 64 if (meas_i and k.get("lanes_per_instr")) else None
            if meas_i:
                model[kind]["measured"] = meas_i
                print("   hardware counters of the shipped kernels (bench.py + %s): %.2f instr/ray, %.1f lane-instr/ray -> the model is %+.1f %% / %+.1f %% off"
                      % (rf.get("pmc", "?").split(" ")[0], meas_i, meas_l, 100.0 * (tot_i / meas_i - 1), 100.0 * (tot_l / meas_l - 1)))
                print("   (the model charges every instruction of a phase once per execution: the stack-spill path, the hand-out atomic and blocks a wave-uniform branch skips are over-counted)")
        # floors
        tris = d["leaf_lanes"] / n
        visits = d["node_lanes"] / n
        leaf_i, node_i = model[kind]["leaf"], model[kind]["node"]
        leaf_floor = tris * (ph["leaf"]["valu"]) / 64.0
        node_floor = visits * (ph["node"]["valu"] / copies) / 64.0
        book = model[kind]["commit.test"] + model[kind]["refill.test"] + model[kind]["schedule"]
        print("   buckets >= 15 %% of the instructions and their floors (every issued instruction with all 64 lanes on a ray):")
        print("     node steps  %5.2f instr/ray (%4.1f %%): %.2f node visits/ray x %.0f / 64 = %5.2f  -> floor %4.1f %% below" % (
            node_i, 100 * node_i / tot_i, visits, ph["node"]["valu"] / copies, node_floor, 100 * (1 - node_floor / node_i)))
        print("     leaf steps  %5.2f instr/ray (%4.1f %%): %.2f triangle tests/ray x %.0f / 64 = %5.2f  -> floor %4.1f %% below" % (
            leaf_i, 100 * leaf_i / tot_i, tris, ph["leaf"]["valu"], leaf_floor, 100 * (1 - leaf_floor / leaf_i)))
        print("     (bookkeeping -- commit.test + refill.test + schedule, wave-uniform, all lanes enabled: %.2f instr/ray, %.1f %%; ray set-up + commit %.2f, %.1f %%)" % (
            book, 100 * book / tot_i, model[kind]["refill.body"] + model[kind]["commit.store"], 100 * (model[kind]["refill.body"] + model[kind]["commit.store"]) / tot_i))
    if pad_path:
        print("\n== what an issue slot is worth: ablation by padding (N full-rate VALU instructions added to ONE phase, results unchanged, no spill;")
        print("   scripts/ab_probe.py: bench scene, 48 spp, kernel ms; same box, three rounds, medians)")
        import statistics
        runs = {}
        for line in open(pad_path):
            m = re.match(r"(\w+): ([\d.]+) Mrays/s \| extend ([\d.]+) shade ([\d.]+) connect ([\d.]+) ms", line)
            if m:
                runs.setdefault(m.group(1), []).append((float(m.group(3)), float(m.group(5)), float(m.group(2))))
        base = {k: statistics.median(r[i] for r in runs["current"]) for i, k in ((0, "extend"), (1, "connect"), (2, "mrays"))}
        print("   current: extend %.1f ms, connect %.1f ms, %.0f Mrays/s" % (base["extend"], base["connect"], base["mrays"]))
        pads = {"padnode32": ("node", 32, "node_steps"), "padleaf32": ("leaf", 32, "leaf_steps"), "padbook16": ("bookkeeping", 16, "loop_passes")}
        for v, (what, n_pad, key) in pads.items():
            if v not in runs:
                continue
            for kind, idx in (("extend", 0), ("connect", 1)):
                t = statistics.median(r[idx] for r in runs[v])
                add = n_pad * D[kind][key] / float(rays[kind])
                di, dt = add / (model[kind]["measured"] or model[kind]["sum"]), t / base[kind] - 1
                print("   +%2d VALU per %-11s %-7s: +%.2f instr/ray (+%4.1f %% of the kernel's instructions) -> %.1f ms (%+.1f %%): elasticity %.2f" % (
                    n_pad, what + " step" if what != "bookkeeping" else "loop pass", kind, add, 100 * di, t, 100 * dt, dt / di))
        print("   -> a VALU issue slot is worth about HALF its share of the instruction count in every phase (the kernel waits ~50 % of its wave cycles")
        print("      with 7 waves per SIMD: latency of the dependent fetches is what the other half of the time is).")


if __name__ == "__main__":
    mode = sys.argv[1] if len(sys.argv) > 1 else ""
    if mode == "static":
        static(sys.argv[2])
    elif mode == "gpu":
        gpu(sys.argv[2])
    elif mode == "report":
        report(*sys.argv[2:6])
    else:
        raise SystemExit(__doc__)

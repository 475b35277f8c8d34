#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X path-tracing integrator.

Metric (BASELINE.json): Mrays/s (+ Msamples/s) on a ~1M-triangle Mitsuba-style
scene at 1920x1080, on 1/2/4/8 MI355X.

  python bench.py --gpus N --steps K --warmup W

* One "step" = `--spp-per-step` (default 64) samples per pixel over the whole 1920x1080
  frame of the generated bathroom2 stand-in (gpuspectral_amd.scenes.interior,
  ~1M triangles, seeded).  Throughput is linear in spp, so K steps of S spp is
  a K*S-spp slice of the 4096-spp target render; the timestamps continue across
  steps exactly as in one long render.
* N > 1 (launched by torch.distributed.run, one rank per GPU): the frame is
  partitioned by interleaved 32x32 tiles, every rank holds the whole scene+BVH,
  renders its own pixels and, after the K steps, ONE RCCL gather brings the HDR
  tiles to rank 0 (inside the timed region).  Total work is fixed -> "strong".
* Timed region: barrier + device sync on both sides, max over ranks.
* Scene upload and BVH build are outside the timed region (reported separately).

Rank 0 prints ONE JSON line.  `roofline` describes the dominant kernel, k_trace<ExtendIO> (closest-hit BVH traversal).
No dense contraction: MFMA unused; HBM at about a third of its peak.  `bound` / `achieved` / `peak` / `frac` name the LARGEST of the three
datasheet fractions this kernel can run against, picked from the data (r06; r05 hard-coded VALU issue):
  valu_issue   wave64 VALU instructions per second (rocprofv3 SQ_INSTS_VALU of exactly the timed launches, from the committed PMC
               passes over this same command line, profiles/pmc_bench_*.json -- `counters_from` says which file -- over THIS run's
               HIP-event kernel time) / 1228.8 G/s (1024 SIMD-32 x 2.4 GHz / 2 cycles, MI355X_MICROARCH.md)
  l1_request   16-B lane loads per second (live statistics pass) against one per cycle and CU (614 G/s, measured)
  hbm          counter bytes (FETCH_SIZE + WRITE_SIZE, corrected as the guide prescribes) per second / 8 TB/s
`fractions` carries all three plus the class-weighted issue figures, each defined in one line in DESIGN.md 5:
  valu_issue_weighted            the same rate against 1024 x 2.4 GHz / the kernel's mean issue cost, where the class mix is
                                 DYNAMIC (rocprofv3 SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F32 / INT32 / INT64 / CVT of the same
                                 launches; cost of a class = static mix inside the class, lib/valu_mix.json; class membership
                                 calibrated by profiles/r05_valu_class_calib.txt); `valu_issue_weighted_static` = r04's figure
                                 (static mix of the binary), kept for comparison
  valu_issue_weighted_at_clock   ... at the clock the chip HELD under the kernel (GRBM_GUI_ACTIVE / 8 XCDs / kernel time)
`traffic` = those counter bytes per launch.  SURVEY 8(d)'s byte model (every node / triangle record the traversal reads,
charged as if it came from HBM) is reported as `survey_byte_model`: it is NOT a bound for a 31-MB tree -- half of the node
visits are served by the LDS copy of the top of the tree and L2 hits 56 % -- so it is printed with
`traffic_over_survey_model` (~0.25) instead of as a rate against the HBM peak.  The fractions are UTILISATIONS (a build
that executes more instructions per ray scores higher), so the work-normalised figures stand next to it: `effective` =
frac x lanes enabled per instruction, `valu_instr_per_ray`, `valu_lane_instr_per_ray`; `other_kernels` carries the same
figures for k_trace<ConnectIO> and k_shade.  A PMC file must carry the digest of the render kernels' instruction streams
(`config.kernel_digest`, lib/valu_mix.json) of the loaded library: a file from other device code is refused and the fields
stay null; a file whose launch shape (rays of the timed region) differs from this run's is used scaled, and the line says so.
`cpu_baseline` times the scalar CPU oracle on this box's host cores on a bounded sample of the same workload (N = 1 only).
N = 1: the read-back of the frame into host memory is INSIDE the timed region (config.download_ms), like the gather of N > 1;
BASELINE's other single-GPU configs (2, 3, 5) run behind it, outside `value`: config.other_workloads.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
# Vector-memory request path: one divergent 16-B lane load per cycle and CU (scripts/microbench/lane_fetch.hip measures 614 G/s
# = 256 CUs x 2.4 GHz for the traversal's access pattern, profiles/r03_trace_bound.txt)
L1_REQUEST_PEAK_G = 256 * 2.4
# VALU issue peak: 256 CUs x 4 SIMD-32, one wave64 VALU instruction per 2 cycles per SIMD, 2.4 GHz max clock (same guide)
VALU_PEAK_GINST = 256 * 4 * 2.4 / 2.0
PMC_GLOB = os.path.join(ROOT, "profiles", "pmc_bench_*.json")  # one file per profiled command line (steps / warmup)


def pmc_for_run(config, timed_launches, digest, kernel_digest=None):
    """Counters of the timed launches of the three render kernels from the committed rocprofv3 --pmc passes
    (scripts/pmc_bench.sh) over THIS command with THIS build: the passes ran the same workload, so their last
    `timed_launches` dispatches of a kernel are the launches of the timed region (it is the tail of the run).
    -> ({kernel: {counter: sum over the timed launches}}, source string) or (None, reason): a file taken on another
    workload, or with other DEVICE CODE, is refused -- the fields stay null rather than describe another binary.  "Same
    device code" = the file's `kernel_digest` (sha256 of the three render kernels' instruction streams, lib/valu_mix.json)
    equals the loaded library's; files from before r05 carry only the library's source digest and must match that."""
    import glob

    want = {k: config[k] for k in ("workload", "triangles", "resolution", "spp_per_step")}  # (launch_shape: compared below, not here)
    if "primary_memo" in config:  # (the launches of a run without the primary-hit memo trace other ray sets)
        want["primary_memo"] = config["primary_memo"]
    cands, stale = [], 0
    for path in sorted(glob.glob(PMC_GLOB)):
        try:
            cand = json.load(open(path))
        except (OSError, ValueError):
            continue
        have = cand.get("bench_config", {})
        if not all(have.get(k) == v for k, v in want.items()):
            continue
        same = (cand.get("kernel_digest") == kernel_digest) if (kernel_digest and cand.get("kernel_digest")) else (cand.get("library_digest") == digest)
        if not same:
            stale += 1
            continue
        cands.append((path, cand))
    if not cands:
        return None, ("stale build: %d PMC file(s) of this workload were taken with other device code (kernel / library digest)" % stale) if stale else "no PMC file for this workload"
    def same_shape(c):  # (files from before r06 carry no launch_shape: launch count + command line decide, as they did)
        ls, mine = c.get("launch_shape"), config.get("launch_shape")
        if not ls or not mine:
            return True
        return all(ls.get(k) and mine.get(k) and abs(ls[k] / mine[k] - 1.0) < 0.005 for k in ("extension_rays", "shadow_rays"))

    exact = [(p, c) for p, c in cands if c.get("timed_launches") == timed_launches and c.get("steps") == config.get("steps")
             and c.get("warmup") == config.get("warmup") and same_shape(c)]
    if exact:
        pm_path, pm = exact[0]
        scale = None
    else:
        # no pass over exactly this command line (other --steps / --warmup, or a pool the free memory sized differently):
        # the per-launch averages of the longest pass over the same workload, scaled to this run's launch count --
        # steady-state launches trace the same ~52 M rays, so the fractions hold to a few per cent; the line says so
        pm_path, pm = max(cands, key=lambda pc: pc[1].get("timed_launches", 0))
        scale = float(timed_launches) / float(max(1, pm["timed_launches"]))
    take = timed_launches if scale is None else int(pm["timed_launches"])
    out = {}
    for kname, k in pm["kernels"].items():
        vals_out = {}
        for name, vals in k["counters"].items():
            # connect / shade launch once per extend launch in the timed region
            if len(vals) < take:
                return None, "PMC file has fewer dispatches than the timed region"
            vals_out[name] = float(sum(vals[-take:])) * (1.0 if scale is None else scale)
        if len(k.get("trace_us", [])) >= take:  # durations of the same dispatches in the kernel-trace pass (no counters)
            vals_out["trace_us_sum"] = float(sum(k["trace_us"][-take:])) * (1.0 if scale is None else scale)
        out[kname] = vals_out
    src = "profiles/%s (%s, kernels %s)%s" % (os.path.basename(pm_path), pm.get("source", "?"), pm.get("kernel_digest") or ("library " + str(pm.get("library_digest"))),
                                              "" if scale is None else ", per-launch averages of its %d timed launches scaled" % take)
    return out, src


def kernel_rates(c, ms, launches, coalesced_read_bytes):
    """Issue rate, lane use, waiting share and memory-side traffic of one kernel from its counters (sums over the timed
    launches) and this run's HIP-event time.  FETCH_SIZE (KiB) reads 1.000 x the bytes of divergent 16-B gathers and 0.5 x
    those of coalesced 16-B-per-lane reads (scripts/microbench/fetch_calib.hip, calibration stored in the PMC file), so
    the missing half of the kernel's coalesced reads is added back; WRITE_SIZE is exact for streaming writes and an
    upper bound (x 2) for scattered 16-B ones."""
    r = {"valu_ginstr_s": None, "issue_frac": None, "lanes_per_instr": None, "effective": None, "wait_share": None,
         "traffic": None, "hbm_gbs": None, "hbm_frac": None, "l2_hit_rate": None, "shader_clock_ghz": None}
    if not c or ms <= 0:
        return r
    # the clock the chip held during this kernel: GRBM_GUI_ACTIVE (busy cycles, summed over the 8 XCDs) of the profiled
    # dispatches over their durations in the kernel-trace pass -- 2.1-2.3 GHz under these kernels, not the 2.4 GHz the peaks
    # above are quoted at (the chip regulates its clock by power)
    if c.get("GRBM_GUI_ACTIVE", 0) > 0 and c.get("trace_us_sum", 0) > 0:
        r["shader_clock_ghz"] = c["GRBM_GUI_ACTIVE"] / 8.0 / (c["trace_us_sum"] * 1e-6) / 1e9
    insts = c.get("SQ_INSTS_VALU", 0.0)
    if insts > 0:
        r["valu_ginstr_s"] = insts / (ms * 1e-3) / 1e9
        r["issue_frac"] = r["valu_ginstr_s"] / VALU_PEAK_GINST
        if "SQ_THREAD_CYCLES_VALU" in c:
            r["lanes_per_instr"] = c["SQ_THREAD_CYCLES_VALU"] / insts / 64.0
            r["effective"] = r["issue_frac"] * r["lanes_per_instr"]  # share of the chip's lane-slots doing enabled work
    if c.get("SQ_WAVE_CYCLES", 0) > 0 and "SQ_WAIT_ANY" in c:
        r["wait_share"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rd = c["FETCH_SIZE"] * 1024.0 + 0.5 * coalesced_read_bytes
        wr = c["WRITE_SIZE"] * 1024.0
        r["traffic"] = (rd + wr) / max(1, launches)
        r["hbm_gbs"] = (rd + wr) / (ms * 1e-3) / 1e9
        r["hbm_frac"] = r["hbm_gbs"] / HBM_PEAK_GBS
    if c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0) > 0:
        r["l2_hit_rate"] = c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    return r


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks of the job (default: WORLD_SIZE, or 1 without a launcher)")
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp-per-step", type=int, default=64)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tris", type=int, default=1_000_000)
    ap.add_argument("--scene", default="interior", choices=["interior", "materials", "caustics", "cornell"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-workloads", action="store_true", help="skip BASELINE's other configs after the timed region (N = 1)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--dump", default="", help="write the gathered frame as .npy (rank 0)")
    ap.add_argument("--force-dist", action="store_true",
                    help="with one rank and no launcher: still create the process group and run the gather through it "
                         "(RCCL initialisation + torch / HIP-runtime coexistence smoke run on a one-GPU box)")
    ap.add_argument("--dist-timeout", type=float, default=120.0,
                    help="seconds a rank waits in a collective of the process group before it gives up (N > 1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend for N > 1 (nccl = RCCL over xGMI; gloo = host tensors, for dry runs)")
    return ap.parse_args()


def make_scene(args):
    from gpuspectral_amd import scenes

    if args.scene == "interior":
        return scenes.interior(args.tris, seed=7), "bathroom2 stand-in: scenes.interior(target_tris=%d, seed=7)" % args.tris
    if args.scene == "materials":
        return scenes.cornell_materials(96), "Cornell + full BSDF set: scenes.cornell_materials(96)"
    if args.scene == "caustics":
        return scenes.caustics(args.tris, seed=11), "dielectric caustics: scenes.caustics(%d, seed=11)" % args.tris
    # the reference's Cornell box through the product's own C++ loader (the oracle package is imported by the
    # cpu_baseline leg only)
    from gpuspectral_amd import host

    xml = os.path.join(ROOT, "tests", "golden", "cornell-box", "scene.xml")
    return host.Scene(xml, os.path.dirname(os.path.dirname(xml))).arrays(), "Cornell box (reference scene.xml)"


def cpu_baseline(sc, args, scene_name):
    """Scalar CPU oracle on a bounded, image-spread sample of the same workload."""
    from gpuspectral_amd import scenes
    from oracle import oracle as orc

    threads = orc.usable_cpus()
    o = orc.Oracle(sc)
    W, H = args.width, args.height
    # probe 1/64 of the frame at 1 spp to size the sample for ~cpu_seconds
    ids = scenes.tile_pixel_ids(W, H, 0, 64, tile=16)
    _, st = o.render(W, H, spp=1, pixel_ids=ids, threads=threads)
    rate = (st["extension_rays"] + st["shadow_rays"]) / max(st["seconds"], 1e-9)
    per_px = (st["extension_rays"] + st["shadow_rays"]) / max(len(ids), 1)
    want_px = max(len(ids), args.cpu_seconds * rate / max(per_px, 1e-9))
    if want_px >= W * H:  # the hosts are fast enough for whole frames: several samples per pixel
        spp_c, parts = int(min(64, max(1, round(want_px / (W * H))))), 1
    else:
        spp_c, parts = 1, max(1, int(round(W * H / want_px)))
    ids = scenes.tile_pixel_ids(W, H, 0, parts, tile=16)
    _, st = o.render(W, H, spp=spp_c, pixel_ids=ids, threads=threads)
    rays = st["extension_rays"] + st["shadow_rays"]
    # one thread on ~2 s of the same work (SURVEY 8d asks for the 1-thread figure next to the all-cores one)
    ids1 = scenes.tile_pixel_ids(W, H, 0, max(1, int(round(W * H * per_px / max(2.0 * rate / max(threads, 1), 1.0)))), tile=16)
    _, st1 = o.render(W, H, spp=1, pixel_ids=ids1, threads=1)
    rate1 = (st1["extension_rays"] + st1["shadow_rays"]) / max(st1["seconds"], 1e-9) / 1e6
    cpu_model = None
    try:
        cpu_model = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except (OSError, StopIteration):
        pass
    return {
        "value": rays / st["seconds"] / 1e6,
        "unit": "Mrays/s",
        "cores": threads,
        "cpu_model": cpu_model,  # core count and CPU model next to the figure
        "kind": "port",
        "single_thread_value": rate1,
        "msamples_per_s": st["samples"] / st["seconds"] / 1e6,
        "sample": "%s, %dx%d, %d spp on %d pixels (every %d-th 16x16 tile), %.1f s, scalar C++ oracle with %d std::threads (= the CPUs the cgroup grants); BVH build %.2f s excluded"
        % (scene_name, W, H, spp_c, len(ids), parts, st["seconds"], threads, o.build_seconds),
    }


def other_workloads(g, seconds=1.0):
    """BASELINE.json's other single-GPU configs, after the timed region and OUTSIDE `value`: ~`seconds` of rendering
    each, so that the driver's record carries them too (r05 review, item 3).  Config 2 = Cornell + full BSDF set at 1024 x
    1024; Config 3 = the ~600 k-triangle interior at 1080p; Config 5 = the dielectric caustics scene at 4096 x 4096,
    max depth 32 -- the tile share rank 0 of the 8-GPU job owns.  Same call sequence as the headline (gsp_upload_scene,
    gsp_frame_begin, gsp_render with HIP-event kernel times, gsp_sync), wall-clock around render + sync."""
    from gpuspectral_amd import abi, multigpu, scenes

    todo = [
        ("config2: Cornell + full BSDF set: scenes.cornell_materials(96)", lambda: scenes.cornell_materials(96), 1024, 1024, None, {}),
        ("config3: bathroom2 stand-in: scenes.interior(target_tris=600000, seed=7)", lambda: scenes.interior(600_000, seed=7), 1920, 1080, None, {}),
        ("config5: dielectric caustics: scenes.caustics(1000000, seed=11), rank 0's share of 8", lambda: scenes.caustics(1_000_000, seed=11),
         4096, 4096, (0, 8), {"max_depth": 32}),
    ]
    out = []
    for name, make, W, H, share, over in todo:
        with g.Context(0) as ctx:
            ctx.upload_scene(make())
            ctx.frame_begin(W, H, multigpu.partition(W, H, *share) if share else None)
            p = abi.default_render_params()
            for k, v in over.items():
                setattr(p, k, v)
            ctx.render(spp=8, first_timestamp=0, params=p)  # pool sized, queues touched
            ctx.sync()
            t = time.perf_counter()
            ctx.render(spp=8, first_timestamp=8, params=p)
            ctx.sync()
            probe = time.perf_counter() - t
            spp = int(max(8, min(4096, 8 * seconds / max(probe, 1e-4))))
            ctx.reset_stats()
            t = time.perf_counter()
            ctx.render(spp=spp, first_timestamp=16, params=p, collect_kernel_times=1)
            ctx.sync()
            dt = time.perf_counter() - t
            st = ctx.stats()
            out.append({"workload": name, "triangles": int(st["num_triangles"]), "resolution": "%dx%d" % (W, H), "pixels": int(ctx.num_pixels),
                        "max_depth": int(p.max_depth), "spp": spp, "seconds": dt, "mrays_per_s": st["traced_rays"] / dt / 1e6,
                        "msamples_per_s": st["samples"] / dt / 1e6, "extend_ms": st["extend_kernel_ms"], "shade_ms": st["shade_kernel_ms"],
                        "connect_ms": st["connect_kernel_ms"], "launches": int(st["extend_launches"]), "bvh_build_ms": st["bvh_build_ms"]})
    return out


def _launch_ranks(args):
    """`python bench.py --gpus N` without a launcher: start torch.distributed.run as a CHILD process (this process has
    not touched the GPU and never replaces itself) and return its exit code."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    print("bench.py: --gpus %d without WORLD_SIZE: launching\n  %s" % (args.gpus, " ".join(cmd)), file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def main():
    """Entry point: any exception of a rank is printed WITH its rank and ends the process with status 1 -- a plain exit,
    never a re-exec -- so that the launcher tears the other ranks down instead of leaving them in a barrier."""
    try:
        _main()
    except SystemExit:
        raise
    except BaseException as e:  # noqa: BLE001 (the rank must die loudly whatever it was)
        import traceback

        traceback.print_exc()
        print("bench.py: rank %s of %s failed: %s: %s" % (os.environ.get("RANK", "0"), os.environ.get("WORLD_SIZE", "1"),
                                                          type(e).__name__, e), file=sys.stderr, flush=True)
        sys.stderr.flush()
        os._exit(1)  # (not sys.exit: a process group that is half torn down may hang in its destructors)


def _main():
    args = parse()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:  # (before any process group exists: nothing to tear down)
        if rank == 0:
            print("bench.py: --gpus %d but the launcher started WORLD_SIZE %d ranks" % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    dist = None
    torch = None
    if world == 1 and args.force_dist:  # a one-rank group, rendezvous on this host
        import socket

        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(sk.getsockname()[1]))
        sk.close()
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
    if world > 1 or args.force_dist:
        from datetime import timedelta

        # torch FIRST: the tracer's library then binds to the HIP runtime torch has loaded (the other order leaves torch's
        # bundled runtime without a device: "No HIP GPUs are available")
        import torch
        import torch.distributed as dist

        # RCCL wants one GPU per rank: refuse an over-subscribed launch BEFORE any GPU call (torch.cuda.device_count only
        # counts devices, it initialises none)
        if args.backend == "nccl" and world > max(1, torch.cuda.device_count()):
            if rank == 0:
                print("bench.py: --gpus %d over the nccl backend needs %d visible GPUs, this node shows %d "
                      "(use --backend gloo for a dry run that shares GPUs)" % (world, world, torch.cuda.device_count()), file=sys.stderr)
            sys.exit(2)
        # a rank that dies must not leave the others in a collective until the driver's limit: every wait of this group
        # gives up after --dist-timeout seconds
        tmo = timedelta(seconds=args.dist_timeout)
        if args.backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), timeout=tmo)
        else:
            dist.init_process_group("gloo", timeout=tmo)

    import gpuspectral_amd as g
    from gpuspectral_amd import multigpu, scenes  # noqa: F401

    sc, scene_name = make_scene(args)
    W, H = args.width, args.height
    ids = multigpu.partition(W, H, rank, world)

    # (gloo dry runs may put several ranks on one GPU; RCCL runs use one GPU per rank)
    ctx = g.Context(local_rank if args.backend == "nccl" else local_rank % max(1, g.device_count()))
    t0 = time.time()
    ctx.upload_scene(sc)
    upload_s = time.time() - t0
    ctx.frame_begin(W, H, ids)
    npix_local = ctx.num_pixels
    S = args.spp_per_step

    on_gpu = args.backend == "nccl"
    tdev = "cuda" if on_gpu else "cpu"

    def barrier():
        ctx.sync()  # this rank's tracer has nothing queued or in flight (its own stream + the path pool)
        if dist is not None:
            if on_gpu:
                torch.cuda.synchronize()
            dist.barrier()
            if on_gpu:
                torch.cuda.synchronize()

    ts = 0
    for _ in range(args.warmup):
        ctx.render(spp=S, first_timestamp=ts)
        ts += S
    # untimed statistics pass: BVH node / triangle reads per extension ray (for the roofline)
    ctx.reset_stats()
    ctx.render(spp=1, first_timestamp=ts, collect_traversal_stats=1)
    ts += 1
    st = ctx.stats()
    nodes_per_ray = st["nodes_visited"] / max(1, st["stat_rays"])
    tris_per_ray = st["tris_tested"] / max(1, st["stat_rays"])
    # requests on the vector-memory path per ray ("lane loads": one lane's 16-B load = one request): a node record read from
    # HBM / L2 is 4 of them, one read from the block's LDS copy of the top of the tree is none, a triangle packet 3, the
    # ray's own record 2
    lds_nodes_per_ray = st["nodes_from_lds"] / max(1, st["stat_rays"])
    ext_lane_loads = 4.0 * (nodes_per_ray - lds_nodes_per_ray) + 3.0 * tris_per_ray + 2.0
    sh_n = max(1, st["shadow_stat_rays"])
    sh_lane_loads = 4.0 * (st["shadow_nodes_visited"] - st["shadow_nodes_from_lds"]) / sh_n + 3.0 * st["shadow_tris_tested"] / sh_n + 2.0 + st["shadow_stat_occluded"] / sh_n  # (+ the occluded ray's commit load)
    ctx.reset_stats()

    local_t = None
    if dist is not None:
        local_t = torch.zeros((npix_local, 4), dtype=torch.float32, device=tdev)
        # untimed rehearsal of the job's one collective: RCCL sets its point-to-point channels up on first use
        multigpu.gather_frame(local_t, W, H, rank, world, dist, force=args.force_dist)

    # the host's framebuffer exists before the job starts (a viewer owns one): allocated and touched outside the timed region
    host_frame = np.ones((H, W, 4), np.float32) if dist is None else None
    barrier()
    t_begin = time.perf_counter()
    for _ in range(args.steps):
        ctx.render(spp=S, first_timestamp=ts, collect_kernel_times=1)
        ts += S
    frame = None
    gather_s = 0.0
    download_s = 0.0
    if dist is None:
        # north_star's entry point is render(scene) -> HOST framebuffer: the read-back of the HDR frame (gsp_download,
        # S/renderer/PathTracer.cpp's blit source) belongs to the job and is INSIDE the timed region, like the gather of N > 1
        ctx.sync()  # (the tracer drained first, as in front of the gather: download_ms is the read-back alone)
        t_d = time.perf_counter()
        ctx.download(out=host_frame)
        download_s = time.perf_counter() - t_d
    if dist is not None:
        # the single collective of the job: HDR tiles -> rank 0 over xGMI.  Its own time (this rank's tracer drained first, so
        # that the clock does not charge the gather with the tail of the render; device-synchronised on both sides) goes into
        # the line as config.gather_ms (max over ranks) -- still INSIDE the timed region
        ctx.sync()
        t_g = time.perf_counter()
        if on_gpu:
            ctx.copy_accum_to_device(local_t.data_ptr(), npix_local * 16)
        else:
            local_t.copy_(torch.from_numpy(ctx.download_compact()))
        frame = multigpu.gather_frame(local_t, W, H, rank, world, dist, force=args.force_dist)
        if on_gpu:
            torch.cuda.synchronize()
        gather_s = time.perf_counter() - t_g
    barrier()
    elapsed = time.perf_counter() - t_begin

    st = ctx.stats()
    local = np.array(
        [elapsed, st["extension_rays"], st["shadow_rays"], st["samples"], st["extend_kernel_ms"], st["extend_launches"],
         st["shade_kernel_ms"], st["connect_kernel_ms"], st["shaded_vertices"], st["memoised_rays"], st["memo_build_rays"], gather_s],
        np.float64,
    )
    # what a first 8-GPU run needs in order to explain its own imbalance: every rank's own clock, work and pool
    mine = np.array([elapsed, st["traced_rays"], st["samples"], npix_local, st["extend_kernel_ms"] + st["shade_kernel_ms"] + st["connect_kernel_ms"],
                     st["extend_launches"], st["device_bytes"], st["render_seconds"], upload_s, gather_s], np.float64)
    per_rank = [mine]
    if dist is not None:
        tm = torch.from_numpy(mine).to(tdev)
        got = [torch.zeros_like(tm) for _ in range(world)] if rank == 0 else None
        dist.gather(tm, got, dst=0)
        if rank == 0:
            per_rank = [x.cpu().numpy() for x in got]
    if dist is not None:
        t = torch.from_numpy(local).to(tdev)
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        tsum = t.clone()
        dist.all_reduce(tsum, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0].item())
        gather_s = float(tmax[11].item())
        tot = tsum.cpu().numpy()
    else:
        tot = local
    if rank == 0 and args.dump:
        np.save(args.dump, frame.cpu().numpy().reshape(H, W, 4) if frame is not None else host_frame)

    if rank == 0:
        ext_rays, sh_rays, samples = tot[1], tot[2], tot[3]
        memoised, memo_build = tot[9], tot[10]
        # `value` counts the rays that were TRACED: the depth-0 segments answered from the primary-hit memo (the reference
        # shoots the same camera ray for every sample of a pixel) are left out, the one trace of each camera ray is in
        traced_ext = ext_rays - memoised + memo_build
        rays = traced_ext + sh_rays
        # ---- roofline of the dominant kernel (rank 0's launches): this run's HIP-event time, the committed counters
        b_ray = 32.0 + 16.0 + 64.0 * nodes_per_ray + 48.0 * tris_per_ray  # ray + hit + node / triangle records
        ext_ms = st["extend_kernel_ms"]
        launches = max(1, st["extend_launches"])
        # rays rank 0's closest-hit launches traced in the timed region (the one trace of each camera ray for the memo included)
        ext_traced0 = st["extension_rays"] - st["memoised_rays"] + st["memo_build_rays"]
        alg_bytes = ext_traced0 * b_ray
        cfg_key = {"workload": scene_name, "triangles": int(st["num_triangles"]), "resolution": "%dx%d" % (W, H), "spp_per_step": S,
                   "steps": args.steps, "warmup": args.warmup, "primary_memo": bool(st["memoised_rays"] > 0),
                   "launch_shape": {"extension_rays": int(traced_ext), "shadow_rays": int(sh_rays)}}
        digest = g.pt.build_info()["digest"]
        # class tables + the digest of the render kernels' instruction streams, written next to the library by csrc/Makefile
        mixf = os.path.join(os.path.dirname(g.lib_path()), "valu_mix.json")
        vmix, kernel_digest = {}, None
        try:
            vm = json.load(open(mixf))
            if vm.get("library_digest") == digest:
                vmix, kernel_digest = vm["kernels"], vm.get("kernel_digest")
        except (OSError, ValueError):
            pass
        pmc, pmc_src = pmc_for_run(cfg_key, int(st["extend_launches"]), digest, kernel_digest) if world == 1 else (None, "N > 1: counters are a single-GPU measurement")
        vertices = float(tot[8])
        ext = kernel_rates((pmc or {}).get("k_trace_extend"), ext_ms, launches, 32.0 * ext_traced0)
        # k_trace<ConnectIO> streams the 32-B shadow ray in (an occluded one loads 16 B more at its commit: a gather);
        # k_shade streams the hit + the path record in (80 B per vertex)
        con = kernel_rates((pmc or {}).get("k_trace_connect"), st["connect_kernel_ms"], launches, 32.0 * st["shadow_rays"])
        shd = kernel_rates((pmc or {}).get("k_shade"), st["shade_kernel_ms"], launches, 80.0 * vertices)
        # k_shade's compulsory queue traffic: hit + path record in, the survivor's record and the 48-B shadow record out
        shade_bytes = 80.0 * vertices + 64.0 * max(0.0, st["extension_rays"] - samples) + 48.0 * st["shadow_rays"]
        shd["queue_bytes_per_vertex"] = shade_bytes / max(1.0, vertices)
        shd["queue_gbs"] = shade_bytes / (st["shade_kernel_ms"] * 1e-3) / 1e9 if st["shade_kernel_ms"] > 0 else None
        shd["queue_frac_of_hbm_peak"] = shd["queue_gbs"] / HBM_PEAK_GBS if shd["queue_gbs"] else None

        def ceilings(k, kname, lane_loads_per_ray, rays, ms):
            """adds the request-path and class-weighted issue figures to a kernel_rates() dict"""
            c = (pmc or {}).get(kname) or {}
            k["lane_loads_per_ray"] = lane_loads_per_ray
            k["lane_loads_g_s"] = lane_loads_per_ray * rays / (ms * 1e-3) / 1e9 if ms > 0 else None
            k["l1_request_frac"] = k["lane_loads_g_s"] / L1_REQUEST_PEAK_G if k["lane_loads_g_s"] else None
            tab = vmix.get(kname) or {}
            cyc_static = tab.get("mean_issue_cycles")
            # DYNAMIC class mix: the hardware's class counters of the same launches x the issue cost of each class
            cyc, shares = None, None
            if tab.get("classes") and c.get("SQ_INSTS_VALU", 0) > 0 and "SQ_INSTS_VALU_FMA_F32" in c:
                from scripts import valu_mix as _vm

                dyn = {cl: c.get("SQ_INSTS_VALU_" + cl, 0.0) for cl in _vm.HW_CLASSES[:-1]}
                cyc, shares = _vm.dynamic_mean_cost(tab["classes"], dyn, c["SQ_INSTS_VALU"])
            k["valu_mean_issue_cycles_static"] = cyc_static
            k["valu_mean_issue_cycles"] = cyc
            k["valu_class_shares"] = shares
            k["issue_frac_weighted_static"] = k["valu_ginstr_s"] / (1024 * 2.4 / cyc_static) if (cyc_static and k["valu_ginstr_s"]) else None
            k["valu_weighted_peak_ginstr_s"] = 1024 * 2.4 / cyc if cyc else None
            k["issue_frac_weighted"] = k["valu_ginstr_s"] / k["valu_weighted_peak_ginstr_s"] if (cyc and k["valu_ginstr_s"]) else None
            # ... and against the same ceiling at the clock the chip actually held (informational: `frac` stays on the nominal peak)
            k["issue_frac_weighted_at_clock"] = (k["valu_ginstr_s"] / (1024 * k["shader_clock_ghz"] / cyc)
                                                 if (cyc and k["valu_ginstr_s"] and k.get("shader_clock_ghz")) else None)
            fr = {"valu_issue": k["issue_frac"], "l1_request": k["l1_request_frac"], "hbm": k["hbm_frac"]}
            fr = {a_: b_ for a_, b_ in fr.items() if b_ is not None}
            k["bound"] = max(fr, key=fr.get) if fr else None  # (of the three datasheet ceilings; the weighted figures are secondaries)
            return k

        ceilings(ext, "k_trace_extend", ext_lane_loads, ext_traced0, ext_ms)
        ceilings(con, "k_trace_connect", sh_lane_loads, st["shadow_rays"], st["connect_kernel_ms"])
        ceilings(shd, "k_shade", 9.0, vertices, st["shade_kernel_ms"])  # 5 coalesced record quads + the 4 quads of the shading packet
        survey_gbs = alg_bytes / (ext_ms * 1e-3) / 1e9 if ext_ms > 0 else None
        roof = {
            "kernel": "k_trace<ExtendIO> (closest-hit traversal of the wide BVH)",
            # bound / achieved / peak / frac: the LARGEST of the three datasheet ceilings this kernel can run against (no MFMA
            # by design), picked from the data -- VALU issue, the vector-memory request path, HBM; all three stay in `fractions`
            "bound": None, "achieved": None, "peak": None, "unit": None, "frac": None,
            "fractions": {"valu_issue": ext["issue_frac"], "valu_issue_weighted": ext["issue_frac_weighted"],
                          "valu_issue_weighted_static": ext["issue_frac_weighted_static"],
                          "valu_issue_weighted_at_clock": ext["issue_frac_weighted_at_clock"],
                          "l1_request": ext["l1_request_frac"], "hbm": ext["hbm_frac"]},
            "largest_datasheet_fraction": ext["bound"],
            "shader_clock_ghz": ext["shader_clock_ghz"],
            "valu_ginstr_s": ext["valu_ginstr_s"], "valu_peak_ginstr_s": VALU_PEAK_GINST,
            "valu_weighted_peak_ginstr_s": ext["valu_weighted_peak_ginstr_s"], "valu_mean_issue_cycles": ext["valu_mean_issue_cycles"],
            "valu_mean_issue_cycles_static": ext["valu_mean_issue_cycles_static"], "valu_class_shares": ext["valu_class_shares"],
            "lane_loads_per_ray": ext_lane_loads, "lds_nodes_per_ray": lds_nodes_per_ray, "lane_loads_g_s": ext["lane_loads_g_s"],
            "l1_request_peak_g_s": L1_REQUEST_PEAK_G,
            "lanes_per_instr": ext["lanes_per_instr"],
            # work-normalised: a build that issues MORE instructions per ray scores higher on `frac`; these do not
            "effective": ext["effective"],
            "valu_instr_per_ray": None, "valu_lane_instr_per_ray": None,
            "wait_share": ext["wait_share"],
            "traffic": ext["traffic"], "hbm_gbs": ext["hbm_gbs"], "hbm_frac": ext["hbm_frac"], "l2_hit_rate": ext["l2_hit_rate"],
            # SURVEY 8(d)'s byte model: what the traversal ASKS of the memory hierarchy if every record came from HBM.  Not a
            # bound (LDS serves half of the node visits, L2 most of the rest): above the HBM peak by construction
            "survey_byte_model": {"bytes_per_ray": b_ray, "bytes_per_launch": alg_bytes / launches, "gbs_if_all_from_hbm": survey_gbs,
                                  "note": "not a roofline: the top of the tree is read from LDS and L2"},
            "traffic_over_survey_model": (ext["traffic"] / (alg_bytes / launches)) if (ext["traffic"] and alg_bytes > 0) else None,
            "nodes_per_ray": nodes_per_ray, "tris_per_ray": tris_per_ray,
            "launches": int(launches), "avg_launch_ms": ext_ms / launches,
            "extend_ms": ext_ms, "shade_ms": st["shade_kernel_ms"], "connect_ms": st["connect_kernel_ms"],
            "other_kernels": {"k_trace<ConnectIO>": con, "k_shade": shd},
            # the counters were NOT measured in this run (they cannot be read in-process): they are the committed rocprofv3
            # passes over the same command line and the same device code; the times they are divided by ARE this run's
            "counters_from": ("committed PMC file " + pmc_src) if pmc else None,
            "pmc": pmc_src,
        }
        rows = {"valu_issue": (ext["valu_ginstr_s"], VALU_PEAK_GINST, "G wave64 VALU instr/s", ext["issue_frac"]),
                "l1_request": (ext["lane_loads_g_s"], L1_REQUEST_PEAK_G, "G 16-B lane loads/s", ext["l1_request_frac"]),
                "hbm": (ext["hbm_gbs"], HBM_PEAK_GBS, "GB/s", ext["hbm_frac"])}
        if ext["bound"] in rows:
            roof["bound"] = ext["bound"]
            roof["achieved"], roof["peak"], roof["unit"], roof["frac"] = rows[ext["bound"]]
        if pmc and pmc.get("k_trace_extend") and ext_traced0 > 0:
            c = pmc["k_trace_extend"]
            if "SQ_INSTS_VALU" in c:
                roof["valu_instr_per_ray"] = c["SQ_INSTS_VALU"] / ext_traced0
            if "SQ_THREAD_CYCLES_VALU" in c:
                roof["valu_lane_instr_per_ray"] = c["SQ_THREAD_CYCLES_VALU"] / ext_traced0
        out = {
            "metric": "Mrays/s (traced rays: extension + shadow; memoised camera segments excluded), ~1M-tri Mitsuba-style scene at 1080p",
            "value": rays / elapsed / 1e6,
            "unit": "Mrays/s",
            "msamples_per_s": samples / elapsed / 1e6,
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": scene_name,
                "triangles": int(st["num_triangles"]),
                "resolution": "%dx%d" % (W, H),
                "spp_per_step": S,
                "spp_timed": S * args.steps,
                "target_spp": 4096,
                "max_depth": 50,
                "parallelism": "tile%d" % world if world > 1 else "single",
                "collective": ("%s, %d rank(s), one gather of HDR tiles" % (dist.get_backend(), world)) if dist is not None else None,
                # the gather alone (device copy of the accumulate buffer + dist.gather + rank 0's scatter), max over ranks; inside `elapsed`
                "gather_ms": gather_s * 1e3 if dist is not None else None,
                # N = 1: gsp_download of the frame into host memory (drain + fold + one device-to-host copy); inside `elapsed`
                "download_ms": download_s * 1e3 if dist is None else None,
                "extension_rays": int(traced_ext),
                "shadow_rays": int(sh_rays),
                "primary_memo": bool(memoised > 0),
                "memoised_rays": int(memoised),  # camera-ray segments copied from the primary-hit memo: NOT in `value`
                "path_segments": int(ext_rays + sh_rays),  # what the reference traces for the same samples
                "bvh_build_ms": st["bvh_build_ms"],
                "scene_upload_ms": upload_s * 1e3,
                "library_digest": digest,
                "kernel_digest": kernel_digest,  # of the three render kernels' instruction streams: what the counter files belong to
                "per_rank": [{"rank": r, "elapsed_s": float(v[0]), "traced_rays": int(v[1]), "samples": int(v[2]), "pixels": int(v[3]),
                              "kernel_ms": float(v[4]), "launches": int(v[5]), "device_bytes": int(v[6]),
                              "render_call_s": float(v[7]), "scene_upload_s": float(v[8]), "gather_s": float(v[9])} for r, v in enumerate(per_rank)],
            },
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(sc, args, scene_name)
        ctx.close()  # (the other workloads size their own path pools from free memory)
        if world == 1 and not args.no_other_workloads:
            # after the timed region, outside `value`: BASELINE's configs 2, 3 and 5 (rank-0 share), ~1 s of rendering each
            out["config"]["other_workloads"] = other_workloads(g)
        print(json.dumps(out), flush=True)

    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

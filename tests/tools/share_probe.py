"""What one GPU of an 8-GPU job sees: Mrays/s on the 1/8 tile share of the 1080p bench frame against the full frame
(same scene, same total samples), plus the GPU-busy share of the wall time.  SURVEY 8(e) readiness without 8 GPUs."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g
from gpuspectral_amd import scenes, multigpu



def equal_spp():
    """--equal-spp: the strong-scaling curve that cannot be measured without an 8-GPU node, predicted on one GPU.
    For world = 2, 4, 8 EVERY share of the frame (gsp_tile_partition: interleaved 32 x 32 tiles) is rendered alone, at
    the SAME two sample counts; a share's time is fitted as  t(spp) = a + b * spp  (a = filling and draining the path pool,
    b = the steady state) and a world's step takes as long as its slowest share plus the gather:
        balance(world)              = mean(b) / max(b)                                  load balance of the partition alone
        predicted efficiency(world) = t_1(target) / (world * (max_r t_r(target) + gather))   at the job's target spp
    gather = the largest share's RGBA32F bytes over ONE xGMI link at 153 GB/s (the 7 senders use 7 distinct links)."""
    import numpy as np

    def one(ctx, W, H, ids, spp, params):
        ctx.frame_begin(W, H, ids)
        kw = dict(params=params) if params is not None else {}
        ctx.reset_stats(); t = time.time(); ctx.render(spp=spp, collect_kernel_times=1, **kw); ctx.sync(); dt = time.time() - t
        st = ctx.stats()
        return dt, st["traced_rays"], st["extend_kernel_ms"] + st["shade_kernel_ms"] + st["connect_kernel_ms"], st["extend_launches"]

    def run(ctx, W, H, spps, target, worlds, params=None, tile=None):
        rows = []
        for world in worlds:
            per = []
            for rank in range(world):
                ids = None if world == 1 else (multigpu.partition(W, H, rank, world) if tile is None else scenes.tile_pixel_ids(W, H, rank, world, tile))
                one(ctx, W, H, ids, spps[0], params)  # warm: pool sized for this share, queues touched
                lo, hi = one(ctx, W, H, ids, spps[0], params), one(ctx, W, H, ids, spps[1], params)
                b_ = (hi[0] - lo[0]) / (spps[1] - spps[0])
                a_ = lo[0] - b_ * spps[0]
                per.append(dict(rank=rank, pixels=ctx.num_pixels, t_lo=lo[0], t_hi=hi[0], a=a_, b=b_, t_target=a_ + b_ * target, rays_hi=hi[1], kernels_ms_hi=hi[2], launches_hi=hi[3]))
            bb = np.array([p["b"] for p in per]); tt = np.array([p["t_target"] for p in per]); rr = np.array([p["rays_hi"] for p in per], np.float64)
            gather_ms = 0.0 if world == 1 else max(p["pixels"] for p in per) * 16 / 153e9 * 1e3
            rows.append(dict(world=world, tile=tile or multigpu.TILE, spp_measured=list(spps), target_spp=target, shares=per,
                             fill_drain_ms_mean=round(float(np.mean([p["a"] for p in per])) * 1e3, 2),
                             s_per_spp_mean=float(bb.mean()), s_per_spp_max=float(bb.max()), balance=round(float(bb.mean() / bb.max()), 4),
                             rays_max_over_mean=round(float(rr.max() / rr.mean()), 4),
                             t_target_max_s=round(float(tt.max()), 4), gather_ms_one_link=round(gather_ms, 3)))
        base = rows[0]["t_target_max_s"] if rows[0]["world"] == 1 else None
        for r in rows:
            if base:
                r["predicted_speedup"] = round(base / (r["t_target_max_s"] + r["gather_ms_one_link"] / 1e3), 3)
                r["predicted_efficiency"] = round(r["predicted_speedup"] / r["world"], 4)
            print(json.dumps({k: v for k, v in r.items() if k != "shares"}), flush=True)
            for p in r["shares"]:
                print("    rank %d: %d pixels  %.4f s @ %d spp  %.4f s @ %d spp  (a %.1f ms, b %.4f ms/spp)  %.1f Mrays  kernels %.1f ms  %d launches" % (
                    p["rank"], p["pixels"], p["t_lo"], spps[0], p["t_hi"], spps[1], p["a"] * 1e3, p["b"] * 1e3, p["rays_hi"] / 1e6, p["kernels_ms_hi"], p["launches_hi"]), flush=True)
        return rows

    from gpuspectral_amd import abi
    print("# tests/tools/share_probe.py --equal-spp: every share of the frame rendered alone on ONE MI355X at equal spp", flush=True)
    print("== headline: interior(1_000_000), 1920 x 1080, shares at 128 and 512 spp, target 4096 spp", flush=True)
    with g.Context(0) as ctx:
        ctx.upload_scene(scenes.interior(1_000_000, seed=7))
        rows = run(ctx, 1920, 1080, (128, 512), 4096, (1, 2, 4, 8))
        worst = min(r["balance"] for r in rows)
        if worst < 1 / 1.03:
            print("== a share is > 3 %% above the mean (balance %.4f): 16 x 16 tiles" % worst, flush=True)
            run(ctx, 1920, 1080, (128, 512), 4096, (1, 8), tile=16)
    print("== config 5: caustics(1_000_000), 4096 x 4096, max depth 32, shares at 16 and 64 spp, target 8192 spp", flush=True)
    with g.Context(0) as ctx:
        ctx.upload_scene(scenes.caustics(1_000_000, seed=11))
        p = abi.default_render_params(16, 0)
        p.max_depth = 32
        run(ctx, 4096, 4096, (16, 64), 8192, (1, 8), params=p)


if "--equal-spp" in sys.argv:
    equal_spp()
    sys.exit(0)

W, H = 1920, 1080
sc = scenes.interior(1_000_000, seed=7)
with g.Context(0) as ctx:
    ctx.upload_scene(sc)
    for world in (1, 2, 4, 8):
        ids = multigpu.partition(W, H, 0, world)
        ctx.frame_begin(W, H, ids)
        spp = 64 * world
        ctx.render(spp=spp); ctx.sync()
        best = None
        for rep in range(2):
            ctx.reset_stats(); t = time.time(); ctx.render(spp=4 * spp, first_timestamp=spp * (1 + 4 * rep), collect_kernel_times=1); ctx.sync(); dt = time.time() - t
            st = ctx.stats()
            r = dict(share="1/%d" % world, pixels=ctx.num_pixels, spp=4 * spp, seconds=round(dt, 3),
                     mrays_per_s=round(st["traced_rays"] / dt / 1e6, 1),
                     kernels_ms=round(st["extend_kernel_ms"] + st["shade_kernel_ms"] + st["connect_kernel_ms"], 1),
                     busy=round((st["extend_kernel_ms"] + st["shade_kernel_ms"] + st["connect_kernel_ms"]) / (dt * 1e3), 4),
                     launches=st["extend_launches"], rays_per_launch=round(st["extension_rays"] / max(1, st["extend_launches"]) / 1e6, 2))
            best = r if best is None or r["mrays_per_s"] > best["mrays_per_s"] else best
        print(json.dumps(best), flush=True)

"""Measurement build only (-DGSP_SHADE_PROFILE, scripts/build_variant.sh shadeprof "-DGSP_SHADE_PROFILE"): where do the
lanes and the cycles of k_shade go?  Per region of the kernel: how often a wave entered it, the lanes enabled, the wave's
wall cycles inside it and the share of the tile loop that is, plus how many sort keys (BSDF types / miss / beyond the
queue) share a wave after the tile's counting sort.

    GSP_LIB_PATH=$PWD/gpuspectral_amd/lib/variants/shadeprof.so python scripts/shade_lane_profile.py [interior coffee materials]
"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import gpuspectral_amd as g  # noqa: E402
from gpuspectral_amd import abi, scenes  # noqa: E402

TYPES = ["diffuse", "smooth_dielectric", "smooth_conductor", "smooth_plastic", "rough_conductor", "smooth_floor", "rough_floor", "rough_plastic"]
REGIONS = [(0, "tile loop (all of it)"), (1, "  tile load + counting sort (4 barriers)"), (2, "  record fetch from LDS"),
           (3, "  vertex: shade_vertex + add_emitted"), (4, "    packet gather, normal, frame, wo"), (5, "    sampleBSDF (switch)"),
           (6, "    light sample"), (7, "    evalBSDF (switch)"), (8, "    NEE / emission / roulette / next state"),
           (9, "  deaths + compaction (2 barriers, 1 atomic pair)"), (10, "  connect_vertex x 2 + queue writes")]
PR_SAMPLE_T0, PR_EVAL_T0, PR_TYPES, PR_COUNT = 16, 24, 32, 48


def load(name):
    if name == "interior":
        return scenes.interior(1_000_000, seed=7), (1920, 1080)
    if name == "materials":
        return scenes.cornell_materials(96), (1024, 1024)
    if name == "caustics":
        return scenes.caustics(1_000_000, seed=11), (1024, 1024)
    return abi.SceneArrays.load(os.path.join(ROOT, "tests", "golden", "ref_scenes", name + ".npz")), (1280, 720)


def main():
    L = ctypes.CDLL(os.environ["GSP_LIB_PATH"])
    out = (ctypes.c_ulonglong * (PR_COUNT * 4))()
    with g.Context(0) as ctx:
        for name in sys.argv[1:] or ["interior", "coffee", "materials"]:
            sc, (W, H) = load(name)
            ctx.upload_scene(sc)
            ctx.frame_begin(W, H)
            ctx.render(spp=4)
            ctx.sync()
            L.gsp_debug_shade_profile(out)  # (clears)
            ctx.reset_stats()
            ctx.render(spp=32, first_timestamp=4, collect_kernel_times=1)
            ctx.sync()
            st = ctx.stats()
            L.gsp_debug_shade_profile(out)
            r = [[out[4 * k + j] for j in range(4)] for k in range(PR_COUNT)]
            tile_cyc = max(1, r[0][2])
            print("## %s  (%d triangles, %dx%d, 32 spp; %d shaded vertices, k_shade %.1f ms with the profile's own overhead)"
                  % (name, st["num_triangles"], W, H, st["shaded_vertices"], st["shade_kernel_ms"]))
            print("%-52s %12s %7s %9s %9s %9s" % ("region", "wave entries", "lanes", "cyc/entry", "% cycles", "% idle-lane cyc"))

            def row(label, e):
                n, lanes, cyc, cl = e
                if n == 0:
                    return
                print("%-52s %12d %7.1f %9.0f %9.1f %9.1f" % (label, n, lanes / n, cyc / n, 100.0 * cyc / tile_cyc, 100.0 * (64 * cyc - cl) / 64 / tile_cyc))

            for k, label in REGIONS:
                row(label, r[k])
            for t, tn in enumerate(TYPES):
                row("      sample " + tn, r[PR_SAMPLE_T0 + t])
            for t, tn in enumerate(TYPES):
                row("      eval   " + tn, r[PR_EVAL_T0 + t])
            waves = sum(r[PR_TYPES + k][0] for k in range(11))
            print("sort keys per wave after the tile sort: " + "  ".join("%d: %.1f %%" % (k, 100.0 * r[PR_TYPES + k][0] / max(1, waves)) for k in range(1, 11) if r[PR_TYPES + k][0]))
            # what straddling costs in the two switches: issued (every wave pays the whole case for each type it holds) against
            # the same vertices packed into type-pure waves
            for what, base in (("sampleBSDF", PR_SAMPLE_T0), ("evalBSDF", PR_EVAL_T0)):
                issued = sum(r[base + t][2] for t in range(8))
                packed = sum(r[base + t][2] * (r[base + t][1] / (64.0 * r[base + t][0])) for t in range(8) if r[base + t][0])
                if issued:
                    print("%s: wave-cycles issued %.3g, with type-pure full waves %.3g (%.0f %%)" % (what, issued, packed, 100.0 * packed / issued))
            print(flush=True)


if __name__ == "__main__":
    main()

"""Randomised GPU-vs-oracle parity: small random scenes with every BSDF type at ordinary and extreme parameters
(alpha -> 0, ior 1, zero / >1 reflectance, huge k), random transforms (mirrored, sheared scale), several lights,
random cameras.  Usage: python tests/tools/fuzz_parity.py [n_scenes] [first_seed] [dormant|nee0|updates|stream]
`nee0`: gsp_render_params.nee = 0 (RenderParams.nee; the other side of rayhit.rchit's `if (NEE)` branches).
`updates`: scene seed+1 is reached from scene seed's geometry by gsp_update_* calls where the object lists agree (else a
fresh upload): exercises the per-frame edit path with random transforms / materials / tables / cameras.
`stream` (+dormant, +lanes2: e.g. `stream+dormant+lanes2`): the same edits, but one sample per frame and NO sync between frames (r05: the table and geometry version rings --
samples of up to a dozen scenes in one launch); the accumulate buffer after 10 frames against the oracle's running mean.
`dormant`: every scene also gets the dormant-feature extension (tests/textured.py: random uv, random textures on the
texturable records, a random environment map; every third scene with an sRGB table, one wall removed so paths escape)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from gpuspectral_amd import scenes, abi


def random_scene(seed):
    rng = np.random.RandomState(seed)
    b = scenes.SceneBuilder()
    meshes = [b.add_mesh(*scenes.rect_mesh()), b.add_mesh(*scenes.box_mesh()), b.add_mesh(*scenes.sphere_mesh(12, 8)),
              b.add_mesh(*scenes.torus_mesh(12, 8))]
    u = lambda lo, hi: float(rng.uniform(lo, hi))
    ext = lambda choices: float(choices[rng.randint(len(choices))])

    def material():
        t = rng.randint(8)
        rgb = tuple(rng.choice([0.0, 0.2, 0.8, 1.0, 1.5], 3))
        if t == 0: return b.diffuse(rgb)
        if t == 1: return b.dielectric(ext([1.0, 1.0001, 1.3, 1.5, 2.4]), ext([1.0, 1.33]))
        if t == 2: return b.mirror(ext([0.0, 0.2, 1.5]))
        if t == 3: return b.plastic(rgb, ext([1.0, 1.3, 1.9]))
        if t == 4:
            return b.rough_conductor(tuple(rng.choice([0.0, 0.14, 1.0, 3.0], 3)), tuple(rng.choice([0.0, 0.5, 4.0, 50.0], 3)),
                                     ext([1e-4, 0.01, 0.1, 0.7, 3.0]), rgb)
        if t == 5: return b.smooth_floor(rgb, ext([0.0, 0.04, 1.0]))
        if t == 6: return b.rough_floor(rgb, ext([0.0, 0.04, 1.0]), ext([1e-4, 0.1, 1.0]))
        return b.rough_plastic(rgb, ext([1e-4, 0.05, 0.5, 2.0]), ext([1.0, 1.3, 1.9]))

    # a closed-ish room so paths bounce, then random clutter
    room = b.add_mesh(*scenes.box_mesh())
    b.add_object(room, scenes.trs((0, 1, 0), (2.5, 1.5, 2.5)), material(), twofaced=True)
    for _ in range(rng.randint(3, 9)):
        s = (u(0.1, 0.8) * ext([1, 1, -1]), u(0.1, 0.8), u(0.1, 0.8) * ext([1, 1, -1]))
        b.add_object(meshes[rng.randint(len(meshes))], scenes.trs((u(-1.5, 1.5), u(0.1, 1.9), u(-1.5, 1.5)), s, u(0, 360)),
                     material(), twofaced=bool(rng.randint(2)))
    for _ in range(rng.randint(1, 4)):
        b.add_object(meshes[0], scenes.trs((u(-1, 1), u(1.5, 2.4), u(-1, 1)), u(0.05, 0.6), u(0, 360)), b.diffuse((0, 0, 0)),
                     twofaced=bool(rng.randint(2)), emission=tuple(rng.choice([0.5, 5.0, 40.0], 3)))
    b.camera_lookat((u(-1.8, 1.8), u(0.3, 2.0), u(-1.8, 1.8)), (u(-0.5, 0.5), u(0.5, 1.5), u(-0.5, 0.5)), fov_deg=u(25, 90))
    return b.build()


def check(ctx, oracle_mod, seed, W=40, H=28, spp=3, dormant=False, nee=1):
    sc = random_scene(seed)
    if dormant:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import textured

        sc = textured.decorate(sc, seed=seed, textures=seed % 5 != 0, envmap=seed % 7 != 0,
                               decode=textured.srgb_table() if seed % 3 == 0 else None)
        if seed % 2 == 0:  # open the room: drop the last two triangles of the room box (instance 0) so that paths reach the sky
            sc.instances["vertex_count"][0] -= 6
    ctx.upload_scene(sc)
    ctx.frame_begin(W, H)
    ctx.reset_stats()
    p = abi.default_render_params()
    p.disable_nee = 0 if nee else 1
    ctx.render(spp=spp, params=p)
    st = ctx.stats()
    img = ctx.download().reshape(-1, 4)
    ref, so = oracle_mod.Oracle(sc).render(W, H, spp=spp, params=p)
    same = np.array_equal(img, ref, equal_nan=True)
    rays = st["extension_rays"] == so["extension_rays"] and st["shadow_rays"] == so["shadow_rays"]
    return same and rays, int((img != ref).any(1).sum()), sc.num_triangles


def mutate(sc, seed):
    """A random per-frame edit of `sc` that keeps its meshes: transforms (translation, mirrored / sheared scale), material
    handles swapped among the objects, twofaced flips, emission changes, every BSDF table and the lights rescaled, a moved
    camera.  Returns (edited copy, what changed)."""
    import copy

    rng = np.random.RandomState((seed * 7919 + 13) % (1 << 32))
    sc2 = copy.deepcopy(sc)
    inst = sc2.instances.copy()
    n = len(inst)
    changed = set()
    for i in range(n):
        if rng.rand() < 0.5:
            t = inst["transform"][i].copy()
            t[12:15] += rng.uniform(-0.3, 0.3, 3).astype(np.float32)
            if rng.rand() < 0.3:
                t[0:3] *= np.float32(rng.choice([-1.0, 0.5, 1.7]))  # first column: mirror / non-uniform scale
            inst["transform"][i] = t
            changed.add("instances")
        if rng.rand() < 0.25:
            inst["bsdf"][i] = inst["bsdf"][rng.randint(n)]
            changed.add("instances")
        if rng.rand() < 0.15:
            inst["twofaced"][i] ^= 1
            changed.add("instances")
        if rng.rand() < 0.1:
            inst["emission"][i] = rng.choice([0.0, 0.5, 8.0], 3).astype(np.float32)
            changed.add("instances")
    sc2.instances = inst
    if rng.rand() < 0.7:
        bs = [b.copy() for b in sc2.bsdfs]
        for b in bs:
            for name in b.dtype.names or ():
                if name != "has_texture" and len(b) and rng.rand() < 0.5:
                    b[name] = (b[name] * np.float32(rng.choice([0.5, 0.9, 1.2]))).astype(np.float32)
        sc2.bsdfs = bs
        lights = sc2.lights.copy()
        if len(lights):
            lights["radiance"] = (lights["radiance"] * np.float32(rng.choice([0.3, 1.0, 2.0]))).astype(np.float32)
        sc2.lights = lights
        changed.add("tables")
    if rng.rand() < 0.7:
        m = np.array(sc2.to_world, np.float32).copy()
        m[12:15] += rng.uniform(-0.2, 0.2, 3).astype(np.float32)
        sc2.to_world = m
        sc2.fov = np.float32(float(sc2.fov) * rng.choice([0.8, 1.0, 1.3]))
        changed.add("camera")
    return sc2, changed


INSTANCE_UPDATES = [0, 0]  # gsp_update_instances calls of check_updates that changed something, and how many of them refitted


def check_updates(ctx, oracle_mod, seed, W=40, H=28, spp=3):
    """random_scene(seed) is uploaded and rendered, then edited in place through gsp_update_camera / _tables / _instances
    (two rounds of edits) -- every frame against the oracle on the scene as it stands."""
    sc = random_scene(seed)
    ctx.upload_scene(sc)
    bad = 0
    for step in range(3):
        if step:
            sc, what = mutate(sc, seed * 4 + step)
            if "tables" in what:
                ctx.update_tables(sc)
            if "instances" in what:
                r0 = ctx.stats()["scene_refits"]
                ctx.update_instances(sc.instances)
                INSTANCE_UPDATES[0] += 1
                INSTANCE_UPDATES[1] += ctx.stats()["scene_refits"] - r0  # (the rest rebuilt: the edit outgrew the refit bound)
            if "camera" in what:
                ctx.update_camera(sc.to_world, sc.fov)
        ctx.frame_begin(W, H)
        ctx.reset_stats()
        ctx.render(spp=spp)
        st = ctx.stats()
        img = ctx.download().reshape(-1, 4)
        ref, so = oracle_mod.Oracle(sc).render(W, H, spp=spp)
        if not (np.array_equal(img, ref, equal_nan=True) and st["extension_rays"] == so["extension_rays"] and st["shadow_rays"] == so["shadow_rays"]):
            bad += 1
    return bad == 0, bad, sc.num_triangles


STREAM = [0, 0, 0]  # edits that changed something, refits, edits that first let the samples in flight finish


def check_stream(ctx, oracle_mod, seed, W=40, H=28, frames=10, dormant=False):
    """random_scene(seed), then `frames` one-sample frames, each behind a round of random edits (transforms incl. mirrored
    ones, material assignments, emission, table values, camera), with no sync in between: the samples of every frame finish on
    the versions of the tables and of the geometry they were generated under."""
    sc = random_scene(seed)
    if dormant:  # textures + environment map: the <TEX, VER> instantiations
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import textured

        sc = textured.decorate(sc, seed=seed, textures=seed % 5 != 0, envmap=seed % 7 != 0,
                               decode=textured.srgb_table() if seed % 3 == 0 else None)
    ctx.upload_scene(sc)
    ctx.frame_begin(W, H)
    acc = None
    for k in range(frames):
        if k:
            sc, what = mutate(sc, seed * 16 + k)
            if "instances" in what:
                ctx.update_instances(sc.instances)
            if "tables" in what:
                ctx.update_tables(sc)
            if "camera" in what:
                ctx.update_camera(sc.to_world, sc.fov)
        ctx.render(spp=1, first_timestamp=k)
        acc, _ = oracle_mod.Oracle(sc).render(W, H, spp=1, first_timestamp=k, accum=acc)
    img = ctx.download().reshape(-1, 4)
    st = ctx.stats()
    STREAM[0] += st["scene_updates"]
    STREAM[1] += st["scene_refits"]
    STREAM[2] += st["scene_drains"]
    ndiff = int((~((img == acc) | (np.isnan(img) & np.isnan(acc)))).any(1).sum())
    return ndiff == 0, ndiff, sc.num_triangles


if __name__ == "__main__":
    import gpuspectral_amd as g
    import oracle as O
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    s0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    dormant = len(sys.argv) > 3 and sys.argv[3] == "dormant"
    nee = 0 if len(sys.argv) > 3 and sys.argv[3] == "nee0" else 1
    bad = []
    opts = abi.CtxOptions(lanes=2) if len(sys.argv) > 3 and "lanes2" in sys.argv[3] else None
    with g.Context(0, options=opts) as ctx:
        for seed in range(s0, s0 + n):
            if len(sys.argv) > 3 and sys.argv[3] == "updates":
                ok, ndiff, tris = check_updates(ctx, O, seed)
            elif len(sys.argv) > 3 and sys.argv[3].startswith("stream"):  # stream | stream+dormant | stream+lanes2 | ...
                ok, ndiff, tris = check_stream(ctx, O, seed, dormant="dormant" in sys.argv[3])
            else:
                ok, ndiff, tris = check(ctx, O, seed, dormant=dormant, nee=nee)
            if not ok:
                bad.append((seed, ndiff))
                print("seed %d: MISMATCH (%d pixels, %d tris)" % (seed, ndiff, tris), flush=True)
    if STREAM[0]:
        print("stream: %d edits changed something, %d refits, %d edits first let the samples in flight finish" % tuple(STREAM))
    if INSTANCE_UPDATES[0]:
        print("gsp_update_instances: %d edits, %d refitted the tree, %d rebuilt it" % (INSTANCE_UPDATES[0], INSTANCE_UPDATES[1], INSTANCE_UPDATES[0] - INSTANCE_UPDATES[1]))
    print("%d scenes%s, %d mismatching: %s" % (n, " with the dormant-feature extension" if dormant else (" with nee = 0" if nee == 0 else ""), len(bad), bad))

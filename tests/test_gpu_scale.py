"""Scale edge of the boundary: a scene whose wide BVH has more than 2^23 nodes.  r03 packed the node index of a traversal
stack entry into 23 bits and refused such scenes AFTER the whole build; r04 packs 25 bits (the pushed
group needs 7, pt_trace.h pack_group) and checks the triangle limit -- fewer than 2^25 -- before any device work."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def grid_scene(G):
    """G x G unit cells of a flat floor at y = 0, two triangles each (A: x-fraction >= z-fraction, B: the other half), one
    small emissive quad above it.  Triangle id of cell (i, j) = 2 * (i * G + j) + {0: A, 1: B}."""
    from gpuspectral_amd import abi

    i, j = np.meshgrid(np.arange(G, dtype=np.float32), np.arange(G, dtype=np.float32), indexing="ij")
    i, j = i.reshape(-1), j.reshape(-1)
    z0 = np.zeros_like(i)
    def v(a, b):
        return np.stack([a, z0, b], 1)
    p00, p10, p11, p01 = v(i, j), v(i + 1, j), v(i + 1, j + 1), v(i, j + 1)
    tris = np.stack([p00, p11, p10, p00, p01, p11], 1).reshape(-1, 3)  # A = (00, 11, 10), B = (00, 01, 11): normals +y
    n = len(tris)
    light = np.array([[1, 9, 1], [2, 9, 1], [2, 9, 2], [1, 9, 1], [2, 9, 2], [1, 9, 2]], np.float32) * np.float32(G / 4)  # normals -y
    light[:, 1] = 9.0
    sc = abi.SceneArrays()
    sc.positions = np.concatenate([tris, light]).astype(np.float32)
    nrm = np.zeros_like(sc.positions)
    nrm[:n, 1] = 1.0
    nrm[n:, 1] = -1.0
    sc.normals = nrm
    inst = np.zeros(2, abi.INSTANCE_DT)
    eye = np.eye(4, dtype=np.float32).reshape(16)
    inst["transform"][:] = eye
    inst["bsdf"] = abi.bsdf_handle(abi.BSDF_DIFFUSE, 0)
    inst["twofaced"] = 1
    inst["first_vertex"] = (0, n)
    inst["vertex_count"] = (n, 6)
    inst["emission"][1] = (30, 30, 30)
    sc.instances = inst
    d = np.zeros(1, abi.DIFFUSE_DT)
    d["reflectance"] = (0.6, 0.6, 0.6)
    sc.bsdfs[0] = d
    lt = np.zeros(2, abi.LIGHT_DT)
    for k in range(2):
        lt["positions"][k][:, :3] = light[3 * k:3 * k + 3]
        lt["radiance"][k][:3] = (30, 30, 30)
    sc.lights = lt
    # camera: above the floor's centre, looking down
    m = np.eye(4, dtype=np.float32)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = (1, 0, 0), (0, 0, 1), (0, 1, 0), (G / 2, 6, G / 2)  # (raygen flips d.y: this looks DOWN)
    sc.to_world = m.T.reshape(16).copy()  # glm memory order (column-major)
    sc.fov = np.float32(0.9)
    return sc


def test_twenty_million_triangles_use_the_25_bit_node_index():
    import gpuspectral_amd as g

    G = 3200
    sc = grid_scene(G)
    ntri = 2 * G * G
    assert sc.num_triangles == ntri + 2 and ntri > (1 << 24)
    with g.Context(0) as ctx:
        ctx.upload_scene(sc)
        st = ctx.stats()
        assert st["num_triangles"] == ntri + 2
        assert st["num_bvh_nodes"] > (1 << 23), st["num_bvh_nodes"]  # beyond the old index width: the point of the test
        print("scale test: %d triangles, %d wide nodes (2^23 = %d), depth %d, upload + build %.0f ms, %.1f GB on the device"
              % (st["num_triangles"], st["num_bvh_nodes"], 1 << 23, st["bvh_depth"], st["bvh_build_ms"], st["device_bytes"] / 1e9))
        rng = np.random.RandomState(3)
        k = 20000
        ci, cj = rng.randint(0, G, k), rng.randint(0, G, k)
        half = rng.randint(0, 2, k)  # 0: a point of triangle A (x-fraction 0.7, z-fraction 0.2), 1: of B (0.3, 0.6)
        fx = np.where(half == 0, 0.7, 0.3).astype(np.float32)
        fz = np.where(half == 0, 0.2, 0.6).astype(np.float32)
        rays = np.zeros((k, 8), np.float32)
        rays[:, 0], rays[:, 1], rays[:, 2] = ci + fx, 5.0, cj + fz
        rays[:, 5] = -1.0
        rays[:, 7] = 1e10
        hit = ctx.trace(rays)
        want = 2 * (ci.astype(np.int64) * G + cj) + half
        assert (hit["prim"] == want).all()
        assert np.abs(hit["t"] - 5.0).max() < 1e-5
        sh = rays.copy()
        sh[:, 3], sh[:, 7] = 0.01, 4.9
        assert (ctx.trace(sh, any_hit=True)["prim"] < 0).all()  # stops short of the floor: unoccluded
        sh[:, 7] = 5.1
        assert (ctx.trace(sh, any_hit=True)["prim"] >= 0).all()
        # a slanted bundle from one point: long traversals through the deep end of the tree
        o = np.array([G / 2 + 0.31, 3.0, G / 2 + 0.17], np.float32)
        tgt = np.stack([ci + fx, np.zeros(k, np.float32), cj + fz], 1).astype(np.float32)
        dvec = tgt - o
        dist = np.linalg.norm(dvec.astype(np.float64), axis=1)
        rays2 = np.zeros((k, 8), np.float32)
        rays2[:, 0:3] = o
        rays2[:, 4:7] = (dvec / dist[:, None]).astype(np.float32)
        rays2[:, 7] = 1e10
        h2 = ctx.trace(rays2)
        ok = h2["prim"] == want  # (a ray that grazes the shared edge of a cell may report the neighbour triangle)
        assert ok.mean() > 0.999 and (h2["prim"] >= 0).all()
        assert np.abs(h2["t"][ok] - dist[ok]).max() < 1e-3 * dist.max()
        # and the integrator runs on it: deterministic, finite, lit
        ctx.frame_begin(96, 64)
        ctx.render(spp=4)
        a = ctx.download().copy()
        ctx.frame_begin(96, 64)
        ctx.render(spp=4)
        b = ctx.download()
        assert np.array_equal(a, b) and np.isfinite(a).all() and a[..., :3].max() > 0


def test_triangle_limit_is_checked_before_any_device_work():
    """gsp_upload_scene refuses 2^25 triangles and more at once (no multi-gigabyte upload, no build), with the documented
    message.  The instance below CLAIMS 2^25 triangles over a vertex array that is only declared, never read."""
    import ctypes as C

    import gpuspectral_amd as g
    from gpuspectral_amd import abi

    sc = abi.SceneArrays()
    inst = np.zeros(1, abi.INSTANCE_DT)
    inst["transform"][0] = np.eye(4, dtype=np.float32).reshape(16)
    inst["bsdf"] = abi.bsdf_handle(abi.BSDF_DIFFUSE, 0)
    inst["vertex_count"] = 3 * (1 << 25)
    sc.instances = inst
    d = np.zeros(1, abi.DIFFUSE_DT)
    sc.bsdfs[0] = d
    desc = sc.desc()
    one = np.zeros(3, np.float32)
    desc.positions = one.ctypes.data  # never dereferenced: the count check comes first
    desc.normals = one.ctypes.data
    desc.num_vertices = 3 * (1 << 25)
    with g.Context(0) as ctx:
        rc = ctx._L.gsp_upload_scene(ctx._h, C.byref(desc))
        assert rc == 4  # GSP_ERR_SCENE
        assert "too many triangles" in ctx._L.gsp_last_error(ctx._h).decode()

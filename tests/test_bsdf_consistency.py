"""Mathematical pins of the oracle's BSDF functions that do NOT come from restating the reference's text a second time
("a 3 % BSDF mistake shared by oracle and kernels would pass everything"): every sampling routine against the pdf
it reports (normalisation, sampled-direction histograms), sample() against eval(), closed forms where textbooks have them (Lambert,
Snell, Fresnel at normal incidence and at the critical angle, the mirror direction), energy bounds.  The reference's known
quirks are asserted as what they are (rough plastic's eval clamps D |wh.z| at 0.01 and its sample does not;
rough conductor is Beckmann-sampled and GGX-valued).  CPU only: the GPU is bit-equal to the oracle elsewhere."""
import numpy as np
import pytest

NON_DELTA = (0, 3, 4, 5, 6, 7)  # types with a non-delta lobe (1 = smooth dielectric, 2 = mirror: delta only)


def unit(v):
    v = np.asarray(v, np.float64)
    return (v / np.linalg.norm(v)).astype(np.float32)


WOS = [unit([0, 0, 1]), unit([0.6, 0, 0.8]), unit([0.3, -0.9, 0.316])]


@pytest.fixture(scope="module")
def orc(oracle_mod):
    from gpuspectral_amd import scenes

    sc = scenes.cornell_materials(8)
    return oracle_mod.Oracle(sc), sc


def hemisphere_grid(nt, nphi):
    """Midpoint grid uniform in (theta, phi) -- fine where a narrow lobe around the normal needs it: directions [nt, nphi, 3] +
    the solid angle of every cell [nt, nphi]."""
    th = (np.arange(nt) + 0.5) / nt * (np.pi / 2)
    ph = (np.arange(nphi) + 0.5) / nphi * 2 * np.pi
    t, p = np.meshgrid(th, ph, indexing="ij")
    w = np.stack([np.sin(t) * np.cos(p), np.sin(t) * np.sin(p), np.cos(t)], -1).astype(np.float32)
    return w, np.sin(t) * (np.pi / 2 / nt) * (2 * np.pi / nphi)


def samples(o, h, wo, n, first=7000):
    return np.array([o.bsdf_sample(h, wo, first + s) for s in range(n)])


@pytest.mark.parametrize("t", NON_DELTA)
def test_pdf_integrates_to_the_probability_of_the_lobe(orc, t):
    """Integral of eval().pdf over the upper hemisphere == share of sample() calls that take the non-delta lobe (1 for the
    pure lobes; at grazing wo part of a microfacet lobe points below the surface, so <= 1 there)."""
    o, _ = orc
    W, dw = hemisphere_grid(240, 360)
    flat, dwf = W.reshape(-1, 3), dw.reshape(-1)
    for wo in WOS:
        ev = np.array([o.bsdf_eval(t << 16, wo, wi) for wi in flat])
        total = float((ev[:, 3] * dwf).sum())
        S = samples(o, t << 16, wo, 3000)
        p_lobe = float((S[:, 7] == 0).mean())
        if t in (0, 3, 5):  # Lambertian lobes: exact up to the grid and the Monte-Carlo share
            assert abs(total - p_lobe) < 0.012, (t, wo, total, p_lobe)
        elif t in (4, 6):  # Beckmann half vectors: everything above the horizon at steep wo, a cut lobe at grazing wo
            assert p_lobe == 1.0 and (abs(total - 1.0) < 0.01 if wo[2] > 0.5 else 0.85 < total <= 1.005), (t, wo, total)
        else:  # rough plastic: eval's pdf clamps D |wh.z| at 0.01 from below (rayhit.rchit:577): slightly above 1
            assert p_lobe == 1.0 and 1.0 <= total < 1.03, (t, wo, total)
        # energy: what the lobe reflects of a unit irradiance stays bounded (reflectances of this scene go up to ~1.09 / pi)
        albedo = (ev[:, :3] * (flat[:, 2] * dwf)[:, None]).sum(0)
        assert np.all(albedo < 1.25), (t, wo, albedo)


@pytest.mark.parametrize("t", NON_DELTA)
def test_sample_and_eval_agree(orc, t):
    """f and pdf returned by sample() == eval() at the sampled direction (two separate pieces of code in the reference)."""
    o, _ = orc
    for wo in WOS:
        S = samples(o, t << 16, wo, 1500)
        clamp_raised = 0
        for s in S:
            if s[7] != 0 or s[6] == 0:
                continue
            e = o.bsdf_eval(t << 16, wo, s[:3])
            assert np.allclose(e[:3], s[3:6], rtol=3e-5, atol=1e-30), (t, wo, s, e)
            if t == 7:  # the clamp of eval only ever raises the pdf
                assert e[3] >= s[6] * (1 - 3e-5), (wo, s, e)
                clamp_raised += e[3] > s[6] * 1.001
            else:
                assert abs(e[3] - s[6]) <= 3e-5 * s[6], (t, wo, s, e)
        if t == 7:
            assert clamp_raised > 100  # (the quirk is there: alpha = 0.05 leaves D tiny over most of the diffuse lobe)


@pytest.mark.parametrize("t", (0, 4, 6))
def test_sampled_directions_follow_the_pdf(orc, t):
    """Histogram of sample() over 6 x 12 cells of (cos theta, phi) against the integral of eval().pdf over each cell."""
    o, _ = orc
    nt, nphi, sub = 6, 12, 20
    W, dw = hemisphere_grid(nt * sub, nphi * sub)
    for wo in WOS[:2]:
        pdf = np.array([o.bsdf_eval(t << 16, wo, wi)[3] for wi in W.reshape(-1, 3)]).reshape(nt * sub, nphi * sub)
        expect = (pdf * dw).reshape(nt, sub, nphi, sub).sum((1, 3))
        n = 20000
        S = samples(o, t << 16, wo, n)
        up = S[:, 2] > 0
        ti = np.minimum((np.arccos(np.clip(S[up, 2], -1, 1)) / (np.pi / 2) * nt).astype(int), nt - 1)
        pi = np.minimum((np.mod(np.arctan2(S[up, 1], S[up, 0]), 2 * np.pi) / (2 * np.pi) * nphi).astype(int), nphi - 1)
        got = np.zeros((nt, nphi))
        np.add.at(got, (ti, pi), 1.0 / n)
        sigma = np.sqrt(np.maximum(expect, 1e-9) / n)
        assert np.all(np.abs(got - expect) < 5 * sigma + 0.02 * expect + 2e-4), (t, wo, np.abs(got - expect).max())


def test_closed_forms(orc):
    o, sc = orc
    # Lambert: f = rho / pi, pdf = cos / pi, f cos / pdf = rho for every sample
    rho = np.asarray(sc.bsdfs[0]["reflectance"][0][:3], np.float64)
    for wo in WOS:
        S = samples(o, 0, wo, 200)
        assert np.allclose(S[:, 3:6], rho / np.pi, rtol=2e-6)
        assert np.allclose(S[:, 6], np.maximum(S[:, 2] / np.pi, 1e-6), rtol=2e-6)
    # smooth conductor: the mirror direction, and (eta = 0 in this scene: Fr = 1) f |cos| = 1
    S = o.bsdf_sample(2 << 16, WOS[1], 5)
    assert np.allclose(S[:3], [-WOS[1][0], -WOS[1][1], WOS[1][2]], atol=1e-7) and S[7] == 1
    assert np.allclose(S[3:6] * abs(S[2]), 1.0, rtol=1e-6)
    # smooth dielectric: reflect with probability Fr, else refract along Snell's direction with (no / nt)^2 (1 - Fr) / |cos|
    b = sc.bsdfs[1][0]
    n_in, n_out = float(b["ior_in"]), float(b["ior_out"])
    for wo in WOS:
        S = samples(o, 1 << 16, wo, 400)
        refl = S[:, 2] > 0
        assert refl.any() and (~refl).any() and np.all(S[:, 7] == 1)
        ci = float(wo[2])
        st = n_out / n_in * np.sqrt(1 - ci * ci)  # entering from outside (wo.z > 0)
        ct = np.sqrt(1 - st * st)
        rs = (n_out * ci - n_in * ct) / (n_out * ci + n_in * ct)
        rp = (n_in * ci - n_out * ct) / (n_in * ci + n_out * ct)
        fr = 0.5 * (rs * rs + rp * rp)  # unpolarised Fresnel reflectance (Born & Wolf)
        assert np.allclose(S[refl, 6], fr, rtol=2e-5) and np.allclose(S[~refl, 6], 1 - fr, rtol=2e-5)
        assert abs(refl.mean() - fr) < 4 * np.sqrt(fr * (1 - fr) / len(S)) + 1e-3
        assert np.allclose(S[refl, :3], [-wo[0], -wo[1], wo[2]], atol=1e-6)
        t = S[~refl][0, :3]
        assert abs(np.linalg.norm(t) - 1) < 1e-5 and abs(np.hypot(t[0], t[1]) - st) < 1e-5 and t[2] < 0  # Snell
        if st > 1e-6:
            assert np.allclose(t[:2] / np.hypot(t[0], t[1]), -wo[:2] / np.hypot(wo[0], wo[1]), atol=1e-5)  # plane of incidence
        assert np.allclose(S[~refl, 3] * np.abs(S[~refl, 2]), (n_out / n_in) ** 2 * (1 - fr), rtol=3e-5)
    # normal incidence: ((n - 1) / (n + 1))^2
    S = samples(o, 1 << 16, WOS[0], 50)
    assert np.allclose(S[S[:, 2] > 0, 6], ((n_in - n_out) / (n_in + n_out)) ** 2, rtol=1e-5)
    # from inside beyond the critical angle: total internal reflection, probability 1, no variate drawn
    crit = n_out / n_in
    wo_in = unit([np.sqrt(1 - 0.2 ** 2), 0, -0.2])
    assert np.sqrt(1 - 0.2 ** 2) > crit
    S = o.bsdf_sample(1 << 16, wo_in, 9)
    assert S[7] == 1 and S[6] == 1.0 and np.allclose(S[:3], [-wo_in[0], -wo_in[1], wo_in[2]], atol=1e-6)


def test_rough_conductor_value_is_the_ggx_microfacet_brdf(orc):
    """f(wo, wi) = F D_ggx(wh) G_smith(wo, wi) / (4 cos_o cos_i) with the exact Fresnel reflectance of a complex index (evaluated
    here in complex arithmetic, not in the reference's real-valued expansion) -- taken, as the reference does
    (rayhit.rchit:509,521), at the angle of wo to the SURFACE normal, not to the half vector: the value is therefore not
    reciprocal, f(wo, wi) / F(|wo.z|) is; pdf = D_beckmann(wh) cos_h / (4 wo.wh)."""
    o, sc = orc
    b = sc.bsdfs[4][0]
    eta, k, refl = (np.asarray(b[n][:3], np.float64) for n in ("eta", "k", "reflectance"))
    a = float(b["alpha"])

    def fresnel(c):
        n = eta + 1j * k
        root = np.sqrt(n * n - (1 - c * c))
        return 0.5 * (np.abs((c - root) / (c + root)) ** 2 + np.abs((n * n * c - root) / (n * n * c + root)) ** 2)

    rng = np.random.RandomState(3)
    checked = 0
    for _ in range(300):
        wo = unit([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.15, 1)])
        wh = unit([rng.normal(0, 0.12), rng.normal(0, 0.12), 1.0])
        wi = (2 * np.dot(wo, wh) * wh - wo).astype(np.float32)
        if wi[2] <= 0.05:
            continue
        wh = unit(wo.astype(np.float64) + wi.astype(np.float64)).astype(np.float64)
        e = o.bsdf_eval(4 << 16, wo, wi)
        c = float(np.dot(wo.astype(np.float64), wh))
        ch2 = float(wh[2]) ** 2
        th2 = (float(wh[0]) ** 2 + float(wh[1]) ** 2) / ch2
        D = 1.0 / (np.pi * a * a * ch2 * ch2 * (1 + th2 / (a * a)) ** 2)
        lam = lambda w: 0.5 * (-1 + np.sqrt(1 + a * a * (float(w[0]) ** 2 + float(w[1]) ** 2) / float(w[2]) ** 2))
        G = 1.0 / (1 + lam(wo) + lam(wi))
        f = refl * fresnel(float(wo[2])) * D * G / (4 * float(wo[2]) * float(wi[2]))
        assert np.allclose(e[:3], f, rtol=1e-4), (wo, wi, e[:3], f)
        Db = np.exp(-th2 / (a * a)) / (np.pi * a * a * ch2 * ch2)
        assert abs(e[3] - Db * float(wh[2]) / (4 * c)) <= 1e-4 * e[3], (wo, wi, e[3])
        back = o.bsdf_eval(4 << 16, wi, wo)
        assert np.allclose(back[:3] * fresnel(float(wo[2])), e[:3] * fresnel(float(wi[2])), rtol=1e-4)  # reciprocal up to F
        checked += 1
    assert checked > 150


def test_light_sampling_is_uniform_over_the_emitters_with_the_area_to_solid_angle_pdf(orc):
    """sampleLight: a light picked uniformly, a point uniform on its triangle (u = 1 - sqrt(e1), v = e2 sqrt(e1): the centroid of the
    samples is the triangle's), pdf = d^2 / (|cos| A) / numLights, radiance only towards the side the emitter faces -- recomputed here from the
    geometry in float64."""
    o, sc = orc
    tri = np.asarray(sc.lights["positions"], np.float64)[:, :, :3]
    rad = np.asarray(sc.lights["radiance"], np.float64)[:, :3]
    nl = len(tri)
    e1, e2 = tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]
    nrm = np.cross(e1, e2)
    area = 0.5 * np.linalg.norm(nrm, axis=1)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    for pos in (np.array([0.1, 0.6, -0.2]), np.array([-0.7, 1.2, 0.5]), np.array([0.0, 2.5, 0.0])):  # the last one: above the lamp
        picked = np.zeros(nl, int)
        centroid = np.zeros((nl, 3))
        for s in range(6000):
            r = o.sample_light(pos.astype(np.float32), 100 + s)
            p = r[:3].astype(np.float64)
            # which triangle? barycentrics inside [0, 1] and on its plane
            hit = None
            for k in range(nl):
                d = p - tri[k, 0]
                m = np.array([[e1[k] @ e1[k], e1[k] @ e2[k]], [e1[k] @ e2[k], e2[k] @ e2[k]]])
                u, v = np.linalg.solve(m, [d @ e1[k], d @ e2[k]])
                if abs(d @ nrm[k]) < 1e-5 and u > -1e-5 and v > -1e-5 and u + v < 1 + 1e-5:
                    hit = k
            assert hit is not None, (pos, p)
            picked[hit] += 1
            centroid[hit] += p
            toward = pos - p
            dist2 = toward @ toward
            cos_l = (toward / np.sqrt(dist2)) @ nrm[hit]
            # one-sided emission (rayhit.rchit:123-153): no radiance when the emitter faces away from the shaded point; the pdf
            # is the area-to-solid-angle one either way
            assert np.allclose(r[3:6], rad[hit] if cos_l > 0 else 0.0, rtol=1e-6)
            want = dist2 / (abs(cos_l) * area[hit]) / nl
            assert abs(r[6] - want) <= 2e-5 * max(want, 1e-12), (pos, p, r[6], want)
        assert np.all(np.abs(picked / picked.sum() - 1.0 / nl) < 4 * np.sqrt(0.25 / picked.sum()))
        for k in range(nl):
            assert np.allclose(centroid[k] / picked[k], tri[k].mean(0), atol=4 * np.sqrt(area[k] / picked[k]))


def _fresnel_dielectric(c, no, nt):
    """Unpolarised Fresnel reflectance of a dielectric interface, cos of the angle on the `no` side (1 beyond the critical angle)."""
    s2 = (no / nt) ** 2 * (1 - c * c)
    if s2 >= 1:
        return 1.0
    ct = np.sqrt(1 - s2)
    rs = (no * c - nt * ct) / (no * c + nt * ct)
    rp = (nt * c - no * ct) / (nt * c + no * ct)
    return 0.5 * (rs * rs + rp * rp)


def test_plastic_values_follow_their_formulas(orc):
    """Smooth plastic's diffuse lobe and rough plastic's diffuse + GGX-Smith lobes in float64 from the record values: substrate
    term rho (1 - Fi)(1 - Fo) eta^2 / (pi (1 - rho Ri)), Ri = 1 - eta^2 (1 - (20 pi R0 + 1) / 21) (rayhit.rchit:320-324), Fresnel at
    the surface normal (smooth) / at the half vector (rough); pdfs as sampled (the rough one with eval's clamp)."""
    o, sc = orc
    rng = np.random.RandomState(11)
    for t in (3, 7):
        b = sc.bsdfs[t][0]
        rho = np.asarray(b["diffuse"][:3], np.float64)
        no, nt, r0 = float(b["ior_out"]), float(b["ior_in"]), float(b["r0"])
        eta = no / nt
        ri = 1 - eta * eta * (1 - (np.pi * 20 * r0 + 1) / 21)
        for _ in range(300):
            wo = unit([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.1, 1)])
            wi = unit([rng.uniform(-1, 1), rng.uniform(-1, 1), rng.uniform(0.1, 1)])
            if t == 7 and rng.rand() < 0.5:  # near the specular peak as well
                wh = unit([rng.normal(0, 0.05), rng.normal(0, 0.05), 1.0])
                wi = (2 * np.dot(wo, wh) * wh - wo).astype(np.float32)
                if wi[2] < 0.1:
                    continue
            e = o.bsdf_eval(t << 16, wo, wi)
            wo64, wi64 = wo.astype(np.float64), wi.astype(np.float64)
            cos_pdf = max(wi64[2] / np.pi, 1e-6)
            if t == 3:
                fi, fo = _fresnel_dielectric(wo64[2], no, nt), _fresnel_dielectric(wi64[2], no, nt)
                f = rho * (1 - fi) * (1 - fo) * eta * eta / (np.pi * (1 - rho * ri))
                pdf = (1 - fi) * cos_pdf
            else:
                a = float(b["alpha"])
                wh = (wo64 + wi64) / np.linalg.norm(wo64 + wi64)
                fi, fo = _fresnel_dielectric(abs(wh @ wo64), no, nt), _fresnel_dielectric(abs(wh @ wi64), no, nt)
                ch2 = wh[2] ** 2
                th2 = (wh[0] ** 2 + wh[1] ** 2) / ch2
                d_ggx = 1.0 / (np.pi * a * a * ch2 * ch2 * (1 + th2 / (a * a)) ** 2)
                lam = lambda w: 0.5 * (-1 + np.sqrt(1 + a * a * (w[0] ** 2 + w[1] ** 2) / w[2] ** 2))
                g = 1.0 / (1 + lam(wo64) + lam(wi64))
                f = rho * (1 - fi) * (1 - fo) * eta * eta / (np.pi * (1 - rho * ri)) + fi * d_ggx * g / (4 * wo64[2] * wi64[2])
                d_b = np.exp(-th2 / (a * a)) / (np.pi * a * a * ch2 * ch2)
                pdf = 0.5 * max(d_b * wh[2], 0.01) / (4 * abs(wo64 @ wh)) + 0.5 * cos_pdf
            assert np.allclose(e[:3], f, rtol=2e-4, atol=1e-7), (t, wo, wi, e[:3], f)
            assert abs(e[3] - pdf) <= 2e-4 * pdf, (t, wo, wi, e[3], pdf)

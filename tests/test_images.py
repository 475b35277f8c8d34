"""Image readers of the loader's dormant features (gpuspectral_amd/host/Image.cpp) against independent decoders.

PNG is lossless: the own inflate + unfilter must reproduce PIL's pixels exactly.  JPEG decoders may differ in the IDCT's
rounding: within 4 levels of libjpeg (PIL) everywhere, 0.1 on average, including 4:2:2 / 4:2:0 files (triangle-filter
chroma up-sampling) and restart intervals.  PFM and Radiance .hdr against files written here.
"""
import os
import struct

import numpy as np
import pytest

PIL = pytest.importorskip("PIL.Image")


def pattern(h, w, seed=0):
    rng = np.random.RandomState(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([127 + 120 * np.sin(x / 9.0 + y / 17.0), 127 + 120 * np.cos(x / 5.0), (x * 3 + y * 5) % 256], -1)
    return np.clip(img + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)


def pil_rgb_bottom_up(path):
    return np.asarray(PIL.open(path).convert("RGB"))[::-1]


@pytest.mark.parametrize("case", ["rgb", "rgba", "grey", "grey-alpha", "palette", "noise-level9", "stored", "rgb16", "1x1"])
def test_png_matches_pil_exactly(tmp_path, case):
    from gpuspectral_amd import host

    p = str(tmp_path / (case + ".png"))
    rng = np.random.RandomState(3)
    if case == "rgb":
        PIL.fromarray(pattern(37, 53)).save(p)
    elif case == "rgba":
        PIL.fromarray(np.dstack([pattern(20, 31), rng.randint(0, 256, (20, 31), dtype=np.uint8)]), "RGBA").save(p)
    elif case == "grey":
        PIL.fromarray(pattern(33, 17)[..., 0], "L").save(p)
    elif case == "grey-alpha":
        PIL.fromarray(np.dstack([pattern(9, 14)[..., 0], rng.randint(0, 256, (9, 14), dtype=np.uint8)]), "LA").save(p)
    elif case == "palette":
        PIL.fromarray(pattern(40, 40)).quantize(16).save(p)
    elif case == "noise-level9":
        PIL.fromarray(rng.randint(0, 256, (300, 400, 3)).astype(np.uint8)).save(p, compress_level=9)
    elif case == "stored":
        PIL.fromarray(pattern(64, 64)).save(p, compress_level=0)
    elif case == "rgb16":
        # 16-bit RGB written by hand: PNG keeps the high byte first, the reader keeps the high byte
        import zlib

        hi = pattern(5, 7)
        lo = pattern(5, 7, seed=9)
        raw = b"".join(b"\x00" + np.stack([hi, lo], -1)[y].tobytes() for y in range(5))

        def chunk(t, d):
            return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

        open(p, "wb").write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 7, 5, 16, 2, 0, 0, 0)) +
                            chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
        got = host.load_bitmap(p)
        assert np.array_equal(got[..., :3], hi[::-1])
        return
    else:
        PIL.fromarray(pattern(1, 1)).save(p)
    got = host.load_bitmap(p)
    assert np.array_equal(got[..., :3], pil_rgb_bottom_up(p))
    assert (got[..., 3] == 255).all()  # Loader.cpp:80: A = 0xFF whatever the file holds


@pytest.mark.parametrize("size", [(123, 211), (16, 16), (17, 33), (8, 8), (1, 1)])
@pytest.mark.parametrize("kw", [dict(subsampling=0, quality=90), dict(subsampling=1, quality=85), dict(subsampling=2, quality=75),
                                dict(subsampling=2, quality=95, optimize=True), dict(subsampling=2, quality=80, restart_marker_blocks=3),
                                dict(subsampling=0, quality=30)], ids=["444", "422", "420", "420-optimised-tables", "420-restart", "444-q30"])
def test_jpeg_close_to_libjpeg(tmp_path, kw, size):
    from gpuspectral_amd import host

    p = str(tmp_path / "t.jpg")
    PIL.fromarray(pattern(*size)).save(p, **kw)
    got = host.load_bitmap(p)[..., :3].astype(int)
    err = np.abs(got - pil_rgb_bottom_up(p).astype(int))
    assert err.max() <= 4 and (err.size < 3 * 256 or err.mean() < 0.1), (err.max(), err.mean())


def test_jpeg_greyscale_and_unsupported(tmp_path):
    from gpuspectral_amd import host

    p = str(tmp_path / "g.jpg")
    PIL.fromarray(pattern(50, 70)[..., 0], "L").save(p, quality=90)
    got = host.load_bitmap(p)
    assert np.abs(got[..., 0].astype(int) - pil_rgb_bottom_up(p)[..., 0].astype(int)).max() <= 2
    assert np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])
    q = str(tmp_path / "prog.jpg")
    PIL.fromarray(pattern(32, 32)).save(q, progressive=True)
    with pytest.raises(host.GspError, match="progressive"):
        host.load_bitmap(q)


def test_bad_files_raise_like_the_reference(tmp_path):
    """loadTexture throws on an unreadable file (Loader.cpp:70-72); so do truncated / foreign files here."""
    from gpuspectral_amd import host

    with pytest.raises(host.GspError, match="cannot open"):
        host.load_bitmap(str(tmp_path / "missing.png"))
    p = str(tmp_path / "ok.png")
    PIL.fromarray(pattern(30, 30)).save(p)
    data = open(p, "rb").read()
    for cut in (10, 40, len(data) // 2, len(data) - 13):
        q = str(tmp_path / ("cut%d.png" % cut))
        open(q, "wb").write(data[:cut])
        with pytest.raises(host.GspError):
            host.load_bitmap(q)
    j = str(tmp_path / "ok.jpg")
    PIL.fromarray(pattern(30, 30)).save(j)
    jd = open(j, "rb").read()
    q = str(tmp_path / "cut.jpg")
    open(q, "wb").write(jd[:200])
    with pytest.raises(host.GspError):
        host.load_bitmap(q)
    t = str(tmp_path / "text.png")
    open(t, "w").write("not an image at all")
    with pytest.raises(host.GspError, match="not a PNG or JPEG"):
        host.load_bitmap(t)
    rng = np.random.RandomState(1)
    for k in range(40):  # flipped bytes anywhere: an error or an image, never a crash
        b = bytearray(data if k % 2 else jd)
        for _ in range(3):
            b[rng.randint(8, len(b))] ^= 1 << rng.randint(8)
        q = str(tmp_path / "fuzz.bin")
        open(q, "wb").write(bytes(b))
        try:
            host.load_bitmap(q)
        except host.GspError:
            pass


def write_rgbe(path, rgbe, rle):
    h, w = rgbe.shape[:2]
    with open(path, "wb") as f:
        f.write(b"#?RADIANCE\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y %d +X %d\n" % (h, w))
        for y in range(h):
            if not rle:
                f.write(rgbe[y].tobytes())
                continue
            f.write(bytes([2, 2, w >> 8, w & 255]))
            for c in range(4):
                row, x = rgbe[y, :, c], 0
                while x < w:
                    run = 1
                    while x + run < w and run < 127 and row[x + run] == row[x]:
                        run += 1
                    if run >= 3:
                        f.write(bytes([128 + run, row[x]]))
                        x += run
                    else:
                        n = min(w - x, 100)
                        f.write(bytes([n]) + row[x:x + n].tobytes())
                        x += n


@pytest.mark.parametrize("rle", [False, True])
def test_radiance_hdr(tmp_path, rle):
    from gpuspectral_amd import host

    rng = np.random.RandomState(5)
    h, w = 9, 40
    rgbe = rng.randint(0, 256, (h, w, 4)).astype(np.uint8)
    rgbe[:, 5:25, :] = rgbe[:, 5:6, :]  # runs
    rgbe[2, 30] = (9, 9, 9, 0)        # exponent 0 = black
    rgbe[..., 3] = np.clip(rgbe[..., 3], 100, 150)
    rgbe[2, 30, 3] = 0
    p = str(tmp_path / "e.hdr")
    write_rgbe(p, rgbe, rle)
    got = host.load_hdr_bitmap(p)
    scale = np.where(rgbe[..., 3:4] > 0, np.ldexp(1.0, rgbe[..., 3:4].astype(int) - 136), 0.0)
    want = (rgbe[..., :3] * scale).astype(np.float32)[::-1]
    assert got.shape == (h, w, 4) and np.array_equal(got[..., :3], want) and (got[..., 3] == 1).all()


def test_pfm_round_trip_and_byte_orders(tmp_path):
    from gpuspectral_amd import host

    rng = np.random.RandomState(2)
    img = rng.uniform(0, 5, (7, 11, 4)).astype(np.float32)  # top-down, as the frame buffer
    p = str(tmp_path / "a.pfm")
    host.write_pfm(p, img)
    got = host.load_hdr_bitmap(p)
    assert np.array_equal(got[..., :3], img[::-1, :, :3]) and (got[..., 3] == 1).all()
    q = str(tmp_path / "be.pfm")
    with open(q, "wb") as f:  # big-endian, single channel
        f.write(b"Pf\n11 7\n1.0\n")
        f.write(img[::-1, :, 0].astype(">f4").tobytes())
    got = host.load_hdr_bitmap(q)
    assert np.array_equal(got[..., 0], img[::-1, :, 0]) and np.array_equal(got[..., 0], got[..., 2])


def test_reference_textures_if_present():
    """The reference's own texture files (Modern Hall, CC-BY 3.0; tests/golden/ref_scenes/staircase2_textures.tar)."""
    import io
    import tarfile

    from gpuspectral_amd import host

    tar = os.path.join(os.path.dirname(__file__), "golden", "ref_scenes", "staircase2_textures.tar")
    if not os.path.exists(tar):
        pytest.skip("texture fixture absent")
    import tempfile

    with tempfile.TemporaryDirectory() as d, tarfile.open(tar) as t:
        t.extractall(d)
        for name, size in (("Tiles.jpg", (894, 894)), ("wood5.jpg", (1200, 1600)), ("Wallpaper.jpg", (512, 512))):
            p = os.path.join(d, "staircase2", "textures", name)
            got = host.load_bitmap(p)
            assert got.shape[:2] == size
            err = np.abs(got[..., :3].astype(int) - pil_rgb_bottom_up(p).astype(int))
            assert err.max() <= 4 and err.mean() < 0.1

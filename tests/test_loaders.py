"""Host loaders / BVH builder / PNG output (reference: scene.cpp:3-213, camera.cpp:3-17,
bvh.cpp:16-144, main.cpp:19-42).  Expected counts and areas are the ones SURVEY.md §8c(7)
and §8a Q3 record for the shipped scenes."""
import ctypes as C
import os

import numpy as np
import pytest
from PIL import Image

import oracle_lib as O
import scene_util as SU
import tinyraytracing_amd as T
from conftest import get_scene


def test_back_counts_and_light():
    s = get_scene("back", 64, 64)
    i = s.info
    assert (i["n_vertices"], i["n_vt"], i["n_vn"], i["n_triangles"]) == (36, 42, 66, 26)
    assert i["n_lights"] == 1 and i["n_materials"] == 4
    assert abs(s.light_area(0) - 27300.0) < 1e-6
    f = s.flat.contents
    assert f.n_light_tris == 4 and f.lights[0].tri_count == 4
    # cumulative CDF (Triangle::area, scene.cpp:203): 4 equal triangles of 6825
    cum = [f.light_tris[k].cum_area for k in range(4)]
    assert np.allclose(cum, [6825, 13650, 20475, 27300], rtol=1e-6)
    assert tuple(f.lights[0].radiance) == (34.0, 24.0, 8.0)
    m = f.materials[f.lights[0].mat]
    assert m.is_emissive == 1 and tuple(m.radiance) == (34.0, 24.0, 8.0)
    # back.mtl writes Kt, which the loader ignores: Tr stays 0
    names = [s.material_name(k) for k in range(4)]
    w = f.materials[names.index("back:DiffuseWhite")]
    assert np.allclose(tuple(w.Kd), (0.79, 0.76, 0.73)) and tuple(w.Tr) == (0, 0, 0) and w.Ns == 1 and w.Ni == 1


def test_veach_counts_and_light_areas():
    s = get_scene("veach-mis", 64, 36)
    i = s.info
    assert (i["n_vertices"], i["n_vt"], i["n_vn"], i["n_triangles"]) == (1421, 1421, 1421, 2332)
    assert i["n_lights"] == 3
    areas = [s.light_area(k) for k in range(3)]
    assert np.allclose(areas, [0.0311, 3.106, 12.43], rtol=2e-2)
    assert s.flat.contents.n_light_tris == 760 * 3


def test_staircase_counts_lights_textures():
    s = get_scene("staircase", 64, 36)
    i = s.info
    assert (i["n_vertices"], i["n_vt"], i["n_vn"], i["n_triangles"]) == (19350, 19350, 19350, 31407)
    assert i["n_lights"] == 6
    areas = [s.light_area(k) for k in range(6)]
    assert np.allclose(areas, [0.182, 0.062, 517, 36.4, 15.6, 124], rtol=2e-2)
    f = s.flat.contents
    assert f.n_textures == 3
    dims = sorted((f.textures[k].width, f.textures[k].height) for k in range(3))
    assert dims == [(512, 512), (894, 894), (1600, 1200)]
    # multi-line radiance attribute (staircase.xml:10-12)
    assert np.allclose(tuple(f.lights[3].radiance), (2.742004577636719, 2.1547576084136963, 0.9237708320617676))
    glass = [k for k in range(i["n_materials"]) if s.material_name(k) == "Glass"][0]
    assert f.materials[glass].Ni == 1.5 and np.allclose(tuple(f.materials[glass].Tr), (0.8, 1.0, 0.95))


def _decode(path):
    lib = T._abi.load_host()
    w, h = C.c_int(), C.c_int()
    assert lib.trth_decode_jpeg(os.fsencode(path), C.byref(w), C.byref(h), None, 0) == 0, lib.trth_last_error()
    buf = np.empty((h.value, w.value, 3), np.uint8)
    assert lib.trth_decode_jpeg(os.fsencode(path), C.byref(w), C.byref(h), buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size) == 0
    return buf


def test_jpeg_decoder_is_bit_identical_to_libjpeg_on_the_shipped_textures():
    """Material::readinMap (material.cpp:3-11: cv::imread): host/jpeg.cpp must yield libjpeg's texels."""
    for name in ("Tiles.jpg", "Wallpaper.jpg", "wood5.jpg"):      # 4:4:4, 4:2:0, 4:2:0
        path = os.path.join(T.SCENES_DIR, "staircase", "textures", name)
        assert np.array_equal(_decode(path), np.asarray(Image.open(path).convert("RGB"))), name
    # and that is what ends up in the flat scene
    f = get_scene("staircase", 64, 36).flat.contents
    tx = [f.textures[k] for k in range(3) if (f.textures[k].width, f.textures[k].height) == (512, 512)][0]
    ref = np.asarray(Image.open(os.path.join(T.SCENES_DIR, "staircase", "textures", "Wallpaper.jpg")).convert("RGB"))
    assert np.array_equal(np.ctypeslib.as_array(tx.rgb, shape=(512, 512, 3)), ref)


@pytest.mark.parametrize("size", [(64, 48), (37, 53), (1, 1), (17, 8), (250, 3)])
@pytest.mark.parametrize("subsampling", [0, 1, 2])                 # 4:4:4, 4:2:2 (h2v1), 4:2:0 (h2v2)
def test_jpeg_decoder_on_generated_images(tmp_path, size, subsampling):
    rng = np.random.default_rng(size[0] * 31 + subsampling)
    w, h = size
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 5 + yy * 3) % 256, (xx * yy) % 256, rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
    path = str(tmp_path / "t.jpg")
    Image.fromarray(img).save(path, quality=85, subsampling=subsampling)
    assert np.array_equal(_decode(path), np.asarray(Image.open(path).convert("RGB")))
    gray = str(tmp_path / "g.jpg")
    Image.fromarray(img[:, :, 2]).save(gray, quality=70)
    assert np.array_equal(_decode(gray), np.asarray(Image.open(gray).convert("RGB")))


@pytest.mark.parametrize("size", [(64, 48), (37, 53), (1, 1), (17, 8), (250, 3), (300, 200)])
def test_progressive_jpeg_decoder_is_bit_identical_to_libjpeg(tmp_path, size):
    """SOF2 files (round 4; cv::imread reads them like any JPEG): DC / AC first and refinement scans with end-of-band runs into coefficient arrays, then the
    baseline pipeline — against PIL (libjpeg-turbo) at three qualities and samplings, with restart markers and with optimised Huffman tables, and in grey."""
    from PIL import ImageFile
    old_block = ImageFile.MAXBLOCK
    ImageFile.MAXBLOCK = 1 << 22  # (PIL's progressive writer needs the whole file in one buffer)
    try:
        rng = np.random.default_rng(size[0] * 7 + size[1])
        w, h = size
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 5 + yy * 3) % 256, (xx * yy) % 256, rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
        path = str(tmp_path / "p.jpg")
        for subsampling in (0, 1, 2):
            for quality in (30, 85, 97):
                for kw in ({}, {"restart_marker_blocks": 3}, {"optimize": True}):
                    Image.fromarray(img).save(path, quality=quality, subsampling=subsampling, progressive=True, **kw)
                    assert open(path, "rb").read().find(b"\xff\xc2") > 0  # really SOF2
                    assert np.array_equal(_decode(path), np.asarray(Image.open(path).convert("RGB"))), (subsampling, quality, kw)
        Image.fromarray(img[:, :, 2]).save(path, quality=70, progressive=True)
        assert np.array_equal(_decode(path), np.asarray(Image.open(path).convert("RGB")))
    finally:
        ImageFile.MAXBLOCK = old_block


def test_jpeg_decoder_rejects_what_it_does_not_handle(tmp_path):
    lib = T._abi.load_host()
    w, h = C.c_int(), C.c_int()
    cmyk = str(tmp_path / "c.jpg")
    Image.fromarray(np.zeros((16, 16, 4), np.uint8), "CMYK").save(cmyk)
    assert lib.trth_decode_jpeg(os.fsencode(cmyk), C.byref(w), C.byref(h), None, 0) != 0      # four components: the .ppm sidecar route
    bad = str(tmp_path / "x.jpg")
    open(bad, "wb").write(b"\xff\xd8\xff\xe0garbage")
    assert lib.trth_decode_jpeg(os.fsencode(bad), C.byref(w), C.byref(h), None, 0) != 0
    assert lib.trth_decode_jpeg(os.fsencode(str(tmp_path / "missing.jpg")), C.byref(w), C.byref(h), None, 0) != 0


def test_decoders_on_hostile_headers_and_coefficients(tmp_path):
    """The two findings of tools/fuzz_decoders.py (mutation fuzzing under ASan + UBSan, ~1 M files): a header announcing an image the file cannot hold is
    rejected BEFORE anything is allocated for it, and coefficients far outside what an encoder writes go through the inverse DCT in wrapping arithmetic
    (tools/sanitize_cpu.sh runs this test under UBSan: no signed overflow)."""
    import struct
    import zlib
    import time
    lib = T._abi.load_host()
    w, h = C.c_int(), C.c_int()
    good = str(tmp_path / "g.jpg")
    Image.fromarray(np.random.default_rng(3).integers(0, 256, (32, 32, 3), dtype=np.uint8)).save(good, quality=95, subsampling=0)
    raw = bytearray(open(good, "rb").read())
    sof = raw.index(b"\xff\xc0")
    huge = bytearray(raw)
    huge[sof + 5:sof + 9] = struct.pack(">HH", 65535, 65535)  # 4.3 G pixels in a 3 KB file
    p = str(tmp_path / "huge.jpg")
    open(p, "wb").write(bytes(huge))
    t = time.time()
    assert lib.trth_decode_jpeg(os.fsencode(p), C.byref(w), C.byref(h), None, 0) != 0
    assert time.time() - t < 1.0
    # every quantiser 255: the coefficients of a quality-95 file, times ~100
    loud = bytearray(raw)
    at = 0
    while True:
        at = loud.find(b"\xff\xdb", at)
        if at < 0:
            break
        n = struct.unpack(">H", loud[at + 2:at + 4])[0]
        for i in range(at + 5, at + 2 + n):
            if (i - (at + 4)) % 65 != 0:
                loud[i] = 255
        at += 2 + n
    p = str(tmp_path / "loud.jpg")
    open(p, "wb").write(bytes(loud))
    assert lib.trth_decode_jpeg(os.fsencode(p), C.byref(w), C.byref(h), None, 0) == 0 and (w.value, h.value) == (32, 32)

    def chunk(kind, data):
        return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 32768, 32768, 16, 6, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(bytes(64))) + chunk(b"IEND", b"")
    p = str(tmp_path / "huge.png")
    open(p, "wb").write(png)
    t = time.time()
    assert lib.trth_decode_png(os.fsencode(p), C.byref(w), C.byref(h), None, 0) != 0
    assert time.time() - t < 1.0


def _decode_png(path):
    lib = T._abi.load_host()
    w, h = C.c_int(), C.c_int()
    assert lib.trth_decode_png(os.fsencode(path), C.byref(w), C.byref(h), None, 0) == 0, lib.trth_last_error()
    buf = np.empty((h.value, w.value, 3), np.uint8)
    assert lib.trth_decode_png(os.fsencode(path), C.byref(w), C.byref(h), buf.ctypes.data_as(C.POINTER(C.c_uint8)), buf.size) == 0
    return buf


def _write_png(path, width, height, depth, ctype, samples, interlace=False, palette=None, level=6, filters=None):
    """A PNG written by hand (PIL writes neither interlaced files nor 16-bit RGB): `samples` [h, w, channels] of `depth`-bit values."""
    import struct
    import zlib
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    samples = np.asarray(samples).reshape(height, width, channels)

    def pack(img):  # rows of one (sub-)image -> filtered scanlines
        hh, ww = img.shape[:2]
        out = bytearray()
        prev = None
        for y in range(hh):
            row = img[y].reshape(-1)
            if depth == 16:
                raw = bytearray(); [raw.extend(struct.pack(">H", int(v))) for v in row]
            elif depth == 8:
                raw = bytearray(int(v) for v in row)
            else:
                bits = "".join(format(int(v), f"0{depth}b") for v in row)
                bits += "0" * (-len(bits) % 8)
                raw = bytearray(int(bits[i:i + 8], 2) for i in range(0, len(bits), 8))
            ft = 0 if filters is None else filters[y % len(filters)]
            bpp = max(1, channels * depth // 8)
            line = bytearray(len(raw))
            for x in range(len(raw)):
                a = raw[x - bpp] if x >= bpp else 0
                b = prev[x] if prev is not None else 0
                c = prev[x - bpp] if (prev is not None and x >= bpp) else 0
                if ft == 0: pred = 0
                elif ft == 1: pred = a
                elif ft == 2: pred = b
                elif ft == 3: pred = (a + b) >> 1
                else:
                    pp = a + b - c; pa, pb, pc = abs(pp - a), abs(pp - b), abs(pp - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                line[x] = (raw[x] - pred) & 255
            out.append(ft); out.extend(line)
            prev = raw
        return bytes(out)
    if interlace:
        body = b""
        for x0, y0, dx, dy in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
            sub = samples[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                body += pack(sub)
    else:
        body = pack(samples)

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)
    z = zlib.compress(body, level)
    data = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 1 if interlace else 0))
    if palette is not None:
        data += chunk(b"PLTE", bytes(int(v) for v in np.asarray(palette).reshape(-1)))
    half = len(z) // 2
    data += chunk(b"IDAT", z[:half]) + chunk(b"tEXt", b"k\x00v") + chunk(b"IDAT", z[half:]) + chunk(b"IEND", b"")  # two IDAT chunks, an ancillary one between
    open(path, "wb").write(data)


def test_png_decoder_matches_pil_on_every_mode_pil_writes(tmp_path):
    """Material::readinMap (material.cpp:3-11: cv::imread, default flag): a PNG texture comes out as 8-bit RGB — palette expanded, grey
    replicated, alpha dropped (host/png.cpp; own inflate).  Files written by PIL in all its PNG modes, odd sizes, every compression level."""
    rng = np.random.default_rng(5)
    for k, (w, h) in enumerate([(64, 48), (37, 53), (1, 1), (17, 8), (250, 3), (300, 200)]):
        yy, xx = np.mgrid[0:h, 0:w]
        rgb = np.stack([(xx * 5 + yy * 3) % 256, (xx * yy) % 256, rng.integers(0, 256, (h, w))], -1).astype(np.uint8)
        alpha = rng.integers(0, 256, (h, w), dtype=np.uint8)
        cases = {"RGB": Image.fromarray(rgb), "RGBA": Image.fromarray(np.dstack([rgb, alpha])), "L": Image.fromarray(rgb[..., 0]),
                 "LA": Image.fromarray(np.dstack([rgb[..., 0], alpha]), "LA"), "P": Image.fromarray(rgb).quantize(37), "1": Image.fromarray(rgb[..., 2] > 127)}
        for mode, im in cases.items():
            path = str(tmp_path / f"t{k}_{mode}.png")
            im.save(path, compress_level=[0, 1, 6, 9][k % 4], optimize=(k % 2 == 1))
            ref = np.asarray(Image.open(path).convert("RGBA" if mode in ("RGBA", "LA") else "RGB"))[..., :3]  # alpha dropped, not blended
            assert np.array_equal(_decode_png(path), ref), (mode, w, h)


def test_png_decoder_on_hand_written_files(tmp_path):
    """What PIL does not write: 16-bit samples (the high byte is kept, libpng's strip_16), 1 / 2 / 4-bit grey (x * 255 / (2^d - 1)) and palettes,
    Adam7 interlace at sizes with empty passes, every scanline filter, IDAT split in two with an ancillary chunk between."""
    rng = np.random.default_rng(9)
    for (w, h) in ((1, 1), (2, 3), (5, 5), (9, 17), (33, 8), (40, 31)):
        for interlace in (False, True):
            for depth, ctype in ((16, 2), (16, 6), (16, 0), (16, 4), (8, 2), (8, 6), (8, 4), (8, 0), (4, 0), (2, 0), (1, 0), (8, 3), (4, 3), (2, 3), (1, 3)):
                channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
                smp = rng.integers(0, 1 << depth, (h, w, channels))
                pal = rng.integers(0, 256, (1 << min(depth, 8), 3)) if ctype == 3 else None
                path = str(tmp_path / "h.png")
                _write_png(path, w, h, depth, ctype, smp, interlace=interlace, palette=pal, level=int(rng.integers(0, 10)), filters=[0, 1, 2, 3, 4, 4, 3, 2, 1])
                s8 = (smp >> 8) if depth == 16 else smp
                if ctype == 3:
                    ref = pal[smp[..., 0]]
                elif ctype in (0, 4):
                    g = s8[..., 0] * 255 // ((1 << depth) - 1) if depth < 8 else s8[..., 0]
                    ref = np.stack([g, g, g], -1)
                else:
                    ref = s8[..., :3]
                assert np.array_equal(_decode_png(path), ref.astype(np.uint8)), (w, h, interlace, depth, ctype)


def test_png_decoder_rejects_corrupt_input(tmp_path):
    lib = T._abi.load_host()
    w, h = C.c_int(), C.c_int()
    good = str(tmp_path / "g.png")
    Image.fromarray(np.arange(48, dtype=np.uint8).reshape(4, 4, 3)).save(good)
    raw = open(good, "rb").read()
    for name, data in (("truncated", raw[: len(raw) // 2]), ("signature", b"\x89PNX" + raw[4:]), ("garbage", raw[:40] + bytes(200)), ("empty", b"")):
        p = str(tmp_path / (name + ".png"))
        open(p, "wb").write(data)
        assert lib.trth_decode_png(os.fsencode(p), C.byref(w), C.byref(h), None, 0) != 0, name
    assert lib.trth_decode_png(os.fsencode(str(tmp_path / "missing.png")), C.byref(w), C.byref(h), None, 0) != 0
    # a flipped bit in the compressed stream must not crash the decoder (whether it is noticed depends on where it lands)
    rng = np.random.default_rng(1)
    for _ in range(200):
        b = bytearray(raw)
        i = int(rng.integers(33, len(b) - 12))
        b[i] ^= 1 << int(rng.integers(0, 8))
        p = str(tmp_path / "flip.png")
        open(p, "wb").write(bytes(b))
        lib.trth_decode_png(os.fsencode(p), C.byref(w), C.byref(h), None, 0)


def test_scene_with_a_png_texture_loads_and_renders_like_the_same_texels_as_ppm(tmp_path):
    """map_Kd pointing at a PNG goes through Material::readinMap -> host/png.cpp; the flat scene holds PIL's texels and the oracle renders the same
    image as with the texels handed over as a PPM."""
    import scene_util as SU
    rng = np.random.default_rng(3)
    tex = rng.integers(0, 256, (32, 48, 3), dtype=np.uint8)
    Image.fromarray(np.dstack([tex, rng.integers(0, 256, (32, 48), dtype=np.uint8)])).save(str(tmp_path / "t.png"))  # RGBA: alpha is dropped
    with open(str(tmp_path / "t.ppm"), "wb") as f:
        f.write(b"P6\n48 32\n255\n" + tex.tobytes())
    lines = ["vt 0 0", "vt 1 0", "vt 1 1", "vt 0 1", "vn 0 0 1", "v -1 -1 0", "v 1 -1 0", "v 1 1 0", "v -1 1 0", "v -1 -1 2", "v 1 -1 2", "v 1 1 2", "v -1 1 2"]
    body = "usemtl tx\nf 1/1/1 2/2/1 3/3/1\nf 1/1/1 3/3/1 4/4/1\nusemtl lamp\nf 5/1/1 7/3/1 6/2/1\nf 5/1/1 8/4/1 7/3/1\n"
    imgs = {}
    for ext in ("png", "ppm"):
        mtl = SU.MTL_BASIC + f"newmtl tx\nKd 0.5 0.5 0.5\nKs 0 0 0\nNs 1\nNi 1\nmap_Kd t.{ext}\n"
        SU.write_scene(tmp_path, "s" + ext, "\n".join(lines) + "\n" + body, mtl, lights=[("lamp", (5, 5, 5))], eye=(0, 0, 1), lookat=(0, 0, 0))
        s = SU.load(tmp_path, "s" + ext)
        f = s.flat.contents
        assert f.n_textures == 1 and (f.textures[0].width, f.textures[0].height) == (48, 32)
        assert np.array_equal(np.ctypeslib.as_array(f.textures[0].rgb, shape=(32, 48, 3)), tex)
        imgs[ext] = O.render(s.flat, T.make_params(32, 32, 4, 7))[0]
        s.close()
    assert np.array_equal(imgs["png"], imgs["ppm"]) and imgs["png"].max() > 0


def test_camera_setup_and_resolution_override():
    s = get_scene("back", 1024, 1024)
    c = s.flat.contents.camera
    # camera.cpp:3-17 for the Cornell camera: w = (0,0,-1), u = (-1,0,0), v = (0,1,0)
    h = np.tan(np.radians(np.float64(np.float32(39.3077))) / 2)
    vh = np.float32(2 * h)
    assert np.allclose(tuple(c.eye), (278, 273, -800))
    assert np.allclose(tuple(c.vertical), (0, vh, 0), atol=1e-6)
    assert np.allclose(tuple(c.horizontal), (-vh, 0, 0), atol=1e-6)
    assert np.allclose(tuple(c.lower_left_corner), (278 + vh / 2, 273 - vh / 2, -799), atol=1e-4)
    wide = get_scene("back", 1920, 1080).flat.contents.camera
    assert np.isclose(abs(wide.horizontal[0]) / wide.vertical[1], 1920 / 1080, rtol=1e-6)  # aspect recomputed (scene.cpp:15)


def test_bvh_is_well_formed_for_both_builders():
    for builder in ("sweep", "binned"):
        s = T.Scene.named("staircase", 64, 36, builder=builder)
        f = s.flat.contents
        seen = np.zeros(f.n_tris, bool)
        stack, depth, visited = [(0, 1)], 0, 0
        while stack:
            n, d = stack.pop()
            visited += 1
            depth = max(depth, d)
            nd = f.nodes[n]
            for ref, lo, hi in ((nd.child0, nd.lo0, nd.hi0), (nd.child1, nd.lo1, nd.hi1)):
                if ref & 0x80000000:
                    first, cnt = ref & 0x07FFFFFF, (ref >> 27) & 15
                    assert 1 <= cnt <= 8
                    assert not seen[first:first + cnt].any()
                    seen[first:first + cnt] = True
                    v = np.ctypeslib.as_array(f.tri_v, shape=(f.n_tris, 3, 3))[first:first + cnt].reshape(-1, 3)
                    # padded by 0.001 (bvh.cpp:31-40)
                    assert np.all(v.min(0) - 0.001 >= np.array(lo) - 1e-6) and np.all(v.max(0) + 0.001 <= np.array(hi) + 1e-6)
                else:
                    assert 0 < ref < f.n_nodes
                    stack.append((ref, d + 1))
        assert seen.all() and visited == f.n_nodes and depth == f.bvh_depth


def test_obj_slot_order_quirk_and_extensions(tmp_path):
    """scene.cpp:149-152: `vt` seen before any `vn` -> tokens are v/vt/vn, otherwise v/vn/vt."""
    verts = "v 0 0 0\nv 1 0 0\nv 0 1 0\n"
    std = verts + "vt 0.25 0.75\nvn 0 0 1\nusemtl white\nf 1/1/1 2/1/1 3/1/1\n"
    swapped = verts + "vn 0 0 1\nvt 0.25 0.75\nusemtl white\nf 1/1/1 2/1/1 3/1/1\n"
    for name, text in (("std", std), ("swapped", swapped)):
        SU.write_scene(tmp_path, name, text, SU.MTL_BASIC)
        s = SU.load(tmp_path, name)
        a = s.arrays()
        assert np.allclose(a["tri_vn"][0], [[0, 0, 1]] * 3) and np.allclose(a["tri_vt"][0], [[0.25, 0.75]] * 3)
    # accepted beyond the reference: v//vn, bare v (face normal), negative indices; only the first 3 tokens of a face are used
    ext = verts + "v 1 1 0\nvn 0 0 1\nusemtl white\nf 1//1 2//1 3//1\nf 1 2 3 4\nf -4//-1 -3//-1 -2//-1\n"
    SU.write_scene(tmp_path, "ext", ext, SU.MTL_BASIC)
    s = SU.load(tmp_path, "ext")
    assert s.info["n_triangles"] == 3
    assert np.allclose(s.arrays()["tri_vn"][:, :, 2], 1.0)


def test_loader_errors_are_reported_not_fatal(tmp_path):
    with pytest.raises(T.TrtError, match="failed"):
        T.Scene.load(str(tmp_path / "nope.xml"), "x.obj", "x.mtl", str(tmp_path))
    SU.write_scene(tmp_path, "bad", "v 0 0 0\nusemtl white\nf 1/1/1 2/1/1 3/1/1\n", SU.MTL_BASIC)
    with pytest.raises(T.TrtError, match="out of range"):
        SU.load(tmp_path, "bad")
    with pytest.raises(T.TrtError):
        T.Scene.named("back", 64, 64, leaf_num=16)


def test_tonemap_matches_reference_transfer():
    x = np.array([[[0.0, 1.0, 0.5], [2.0, 0.001, 0.2176]]], np.float32)
    got = T.tonemap(x)
    exp = np.clip(np.power(x.astype(np.float64), np.float64(np.float32(1.0) / np.float32(2.2))) * 255, 0, 255).astype(np.uint8)  # main.cpp:34
    assert np.array_equal(got, exp)
    assert got[0, 0, 1] == 255 and got[0, 1, 0] == 255


def test_png_writer_roundtrip(tmp_path):
    rng = np.random.default_rng(3)
    for (h, w) in ((1, 1), (7, 13), (200, 333)):   # the last one spans several 64 KiB stored blocks
        img = rng.random((h, w, 3)).astype(np.float32)
        path = str(tmp_path / f"t{h}x{w}.png")
        T.imshow(img, path)
        back = np.asarray(Image.open(path))
        assert back.shape == (h, w, 3) and np.array_equal(back, T.tonemap(img))


def test_synthetic_scenes_build():
    s = T.Scene.named("soup", 64, 36, n=5000)
    assert s.info["n_triangles"] == 26 - 12 + 5000
    v = s.arrays()["tri_v"]
    assert np.isfinite(v).all()
    b = T.Scene.named("blob", 64, 36, n=2000)
    n = b.info["n_triangles"] - 14
    assert n >= 2000 and n % 20 == 0
    vn = b.arrays()["tri_vn"]
    assert np.allclose(np.linalg.norm(vn.reshape(-1, 3), axis=1), 1.0, atol=1e-4)
    # deterministic
    s2 = T.Scene.named("soup", 64, 36, n=5000)
    assert np.array_equal(np.sort(s.arrays()["tri_v"].reshape(-1)), np.sort(s2.arrays()["tri_v"].reshape(-1)))


def test_polygons_are_fanned_only_on_request(tmp_path):
    """An `f` line with four or five vertices: the reference (scene.cpp:162) and the default loader keep the first three tokens;
    triangulate_polygons turns it into the fan (v0,v1,v2), (v0,v2,v3), ... — also for the light's area / CDF."""
    obj = ("v 0 0 0\nv 2 0 0\nv 2 2 0\nv 0 2 0\nv -1 1 0\nvt 0 0\nvn 0 0 1\n"
           "usemtl lamp\nf 1/1/1 2/1/1 3/1/1 4/1/1\n"
           "usemtl white\nf 1/1/1 2/1/1 3/1/1 4/1/1 5/1/1\n")
    SU.write_scene(tmp_path, "poly", obj, SU.MTL_BASIC, lights=[("lamp", (1, 1, 1))], w=16, h=16)
    s3 = SU.load(tmp_path, "poly")
    assert s3.info["n_triangles"] == 2 and abs(s3.light_area(0) - 2.0) < 1e-5
    sp = SU.load(tmp_path, "poly", triangulate_polygons=True)
    assert sp.info["n_triangles"] == 2 + 3 and abs(sp.light_area(0) - 4.0) < 1e-5
    tv = sp.arrays()["tri_v"]
    areas = 0.5 * np.linalg.norm(np.cross(tv[:, 1] - tv[:, 0], tv[:, 2] - tv[:, 0]), axis=1)
    assert abs(areas.sum() - (4.0 + 5.0)) < 1e-4   # the quad twice over + the pentagon's extra triangle (area 1)

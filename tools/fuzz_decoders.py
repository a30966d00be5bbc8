#!/usr/bin/env python3
"""Mutation fuzzing of the texture decoders (tinyraytracing_amd/host/jpeg.cpp, png.cpp) under AddressSanitizer + UBSan.

Textures are files a scene's .mtl names: the decoders read untrusted bytes.  This builds a sanitised copy of libtrt_host.so under /tmp, writes seed
images with PIL (baseline / progressive JPEG at several samplings and restart intervals; PNG in every mode PIL writes, interlaced or not, 16 bit, palette),
and feeds mutated copies to trth_decode_jpeg / trth_decode_png in a child process: byte flips, runs of random bytes, truncations, duplicated and dropped
spans, and — for PNG — mutations INSIDE the chunks with the CRC recomputed and inside the zlib stream with the Adler-32 left wrong or the stream
re-deflated, so that the mutation reaches the filter / inflate / Adam7 code instead of dying at the first checksum.  A decoder may reject a file (that is the
expected outcome) but must not crash, read or write out of bounds, overflow a signed integer, or allocate without bound; every case is written to disk before
it is decoded, so a crash names its input.

--target scene does the same to the text loaders (host/scene.cpp: the .xml / .obj / .mtl of scenes/back, mutated byte-wise and token-wise — indices out of
range, zero, negative, huge; NaN / inf / 1e39 coordinates; missing and repeated lines) followed by both BVH builders and the flattening: a scene is loaded or
refused, never a crash, and never a hang (10 s alarm per case).

usage: tools/fuzz_decoders.py [--target images|scene] [--seconds 120] [--seed 1]      (exit code 0 = no finding)"""
import argparse
import ctypes as C
import io
import os
import random
import struct
import subprocess
import sys
import time
import zlib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORK = "/tmp/trt_fuzz_dec"
SAN = ("-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1 -std=c++17 -fPIC -fopenmp -ffp-contract=off "
       "-march=x86-64-v3").split()


def build():
    os.makedirs(WORK, exist_ok=True)
    lib = os.path.join(WORK, "libtrt_host_san.so")
    src = [os.path.join(ROOT, "tinyraytracing_amd/host", f + ".cpp") for f in ("scene", "bvh", "synth", "image_out", "jpeg", "png", "capi")]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in src):
        subprocess.check_call(["g++", *SAN, "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tinyraytracing_amd/host"), "-shared", "-o", lib, *src])
    return lib


def seeds():
    import numpy as np
    from PIL import Image, ImageFile
    ImageFile.MAXBLOCK = 1 << 22
    rng = np.random.default_rng(7)
    out = []

    def img(w, h, mode="RGB"):
        a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        a[:, :, 0] = (np.arange(w)[None, :] * 255 // max(w - 1, 1)).astype(np.uint8)  # some structure: long Huffman runs and real matches
        return Image.fromarray(a, "RGB").convert(mode)
    for (w, h) in ((17, 13), (64, 48), (33, 70)):
        for sub in (0, 1, 2):
            for prog in (False, True):
                b = io.BytesIO()
                img(w, h).save(b, "JPEG", quality=80, subsampling=sub, progressive=prog)
                out.append(("jpeg", b.getvalue()))
        b = io.BytesIO()
        img(w, h, "L").save(b, "JPEG", quality=70)
        out.append(("jpeg", b.getvalue()))
    for (w, h) in ((9, 7), (40, 33)):
        for mode in ("RGB", "RGBA", "L", "LA", "P", "1", "I;16"):
            b = io.BytesIO()
            im = img(w, h, "L").convert("I;16") if mode == "I;16" else img(w, h, mode)
            im.save(b, "PNG", compress_level=6)
            out.append(("png", b.getvalue()))
        out.append(("png", interlaced_png(w, h, rng)))
    return out


def chunk(kind, data):
    return struct.pack(">I", len(data)) + kind + data + struct.pack(">I", zlib.crc32(kind + data) & 0xFFFFFFFF)


def interlaced_png(w, h, rng):
    """an Adam7 file written by hand (PIL does not write them): colour type 2, 8 bit, filter 0 on every row"""
    import numpy as np
    a = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    raw = b""
    for (x0, y0, dx, dy) in ((0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)):
        sub = a[y0::dy, x0::dx]
        if sub.shape[0] == 0 or sub.shape[1] == 0:
            continue
        for row in sub:
            raw += b"\0" + np.ascontiguousarray(row).tobytes()
    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b"")


def png_chunks(data):
    pos, out = 8, []
    while pos + 12 <= len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        out.append((data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]))
        pos += 12 + n
    return out


def mutate_bytes(r, b):
    b = bytearray(b)
    if not b:
        return bytes(b)
    k = r.random()
    if k < 0.35:
        for _ in range(r.randint(1, 6)):
            b[r.randrange(len(b))] ^= 1 << r.randrange(8)
    elif k < 0.55:
        p = r.randrange(len(b))
        for i in range(p, min(len(b), p + r.randint(1, 24))):
            b[i] = r.randrange(256)
    elif k < 0.70:
        del b[r.randrange(len(b)):]
    elif k < 0.80:
        p, n = r.randrange(len(b)), r.randint(1, 64)
        b[p:p] = b[p:p + n]
    elif k < 0.90:
        p = r.randrange(len(b))
        del b[p:p + r.randint(1, 64)]
    else:
        p = r.randrange(len(b))
        b[p:p + 1] = bytes([r.choice((0, 0xFF, 0x7F, 0x80, b[p]))]) * r.randint(1, 4)
    return bytes(b)


def mutate_png(r, data):
    k = r.random()
    if k < 0.25:
        return mutate_bytes(r, data)  # dies at a CRC or the signature, mostly
    ch = png_chunks(data)
    idat = b"".join(d for t, d in ch if t == b"IDAT")
    others = [(t, d) for t, d in ch if t != b"IDAT" and t != b"IEND"]
    if k < 0.50:  # header / palette / transparency fields, CRC made right
        i = r.randrange(len(others))
        others[i] = (others[i][0], mutate_bytes(r, others[i][1]))
        if others[i][0] == b"IHDR" and len(others[i][1]) >= 13 and r.random() < 0.5:  # keep the size sane half of the time so that the body is reached
            others[i] = (b"IHDR", struct.pack(">II", r.randint(1, 70), r.randint(1, 70)) + others[i][1][8:])
    elif k < 0.75:  # inside the deflate stream
        idat = mutate_bytes(r, idat)
    else:  # inside the filtered scanlines, re-deflated: valid stream, arbitrary filter bytes / pixel data / length
        try:
            raw = zlib.decompress(idat)
        except zlib.error:
            raw = b""
        idat = zlib.compress(mutate_bytes(r, raw), r.choice((0, 1, 6, 9)))
    parts = [idat] if r.random() < 0.7 else [idat[:len(idat) // 3], idat[len(idat) // 3:len(idat) // 2], idat[len(idat) // 2:]]
    return b"\x89PNG\r\n\x1a\n" + b"".join(chunk(t, d) for t, d in others) + b"".join(chunk(b"IDAT", p) for p in parts) + chunk(b"IEND", b"")


def child(lib_path, seed, seconds):
    lib = C.CDLL(lib_path)
    for f in (lib.trth_decode_jpeg, lib.trth_decode_png):
        f.argtypes = [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_void_p, C.c_uint64]
        f.restype = C.c_int
    r = random.Random(seed)
    S = seeds()
    path = os.path.join(WORK, f"case_{seed}.bin")
    t_end = time.time() + seconds
    n = ok = 0
    buf = (C.c_uint8 * (1 << 24))()
    while time.time() < t_end:
        kind, data = S[r.randrange(len(S))]
        m = mutate_png(r, data) if kind == "png" else mutate_bytes(r, data)
        for _ in range(r.randint(0, 2)):
            m = mutate_bytes(r, m)
        with open(path, "wb") as f:
            f.write(m)
        w, h = C.c_int(0), C.c_int(0)
        fn = lib.trth_decode_png if kind == "png" else lib.trth_decode_jpeg
        rc = fn(path.encode(), C.byref(w), C.byref(h), buf, len(buf))
        n += 1
        ok += rc == 0
    print(f"seed {seed}: {n} mutated files, {ok} decoded, {n - ok} rejected, no finding", flush=True)


def mutate_text(r, text):
    lines = text.split("\n")
    k = r.random()
    if k < 0.25:
        return mutate_bytes(r, text.encode("latin-1")).decode("latin-1")
    if k < 0.45 and lines:  # drop / repeat / swap lines
        i = r.randrange(len(lines))
        op = r.randrange(3)
        if op == 0:
            del lines[i:i + r.randint(1, 5)]
        elif op == 1:
            lines[i:i] = lines[i:i + r.randint(1, 5)] * r.randint(1, 3)
        else:
            j = r.randrange(len(lines))
            lines[i], lines[j] = lines[j], lines[i]
        return "\n".join(lines)
    # replace tokens of some lines
    specials = ["0", "-1", "-999999999", "4294967296", "2147483647", "99999999999999999999", "nan", "inf", "-inf", "1e39", "-1e39", "1e-46", "", "/", "//", "1/", "/1/", "1//1",
                "a", "0x10", "1.5.5", "-", "+", "\"", "<", ">", "=", "\0", "\t", " " * 50, "9" * 400]
    for _ in range(r.randint(1, 4)):
        i = r.randrange(len(lines))
        tok = lines[i].split(" ")
        if not tok:
            continue
        j = r.randrange(len(tok))
        if "/" in tok[j] and r.random() < 0.6:
            parts = tok[j].split("/")
            parts[r.randrange(len(parts))] = r.choice(specials)
            tok[j] = "/".join(parts)
        else:
            tok[j] = r.choice(specials)
        lines[i] = " ".join(tok)
    return "\n".join(lines)


def child_scene(lib_path, seed, seconds):
    import signal
    lib = C.CDLL(lib_path)
    lib.trth_scene_load_opts.restype = C.c_void_p
    lib.trth_scene_load_opts.argtypes = [C.c_char_p] * 4 + [C.c_int] * 3
    lib.trth_scene_build.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.trth_scene_flat.restype = C.c_void_p
    lib.trth_scene_flat.argtypes = [C.c_void_p]
    lib.trth_scene_free.argtypes = [C.c_void_p]
    r = random.Random(seed)
    base = os.path.join(ROOT, "scenes", "back")
    src = {e: open(os.path.join(base, "back." + e), encoding="latin-1").read() for e in ("xml", "obj", "mtl")}
    d = os.path.join(WORK, f"scene_{seed}")
    os.makedirs(d, exist_ok=True)
    t_end = time.time() + seconds
    n = ok = 0
    while time.time() < t_end:
        which = r.choice(("xml", "obj", "obj", "mtl"))
        cur = dict(src)
        cur[which] = mutate_text(r, cur[which])
        if r.random() < 0.2:
            cur[which] = mutate_text(r, cur[which])
        for e in cur:
            with open(os.path.join(d, "s." + e), "w", encoding="latin-1", newline="") as f:
                f.write(cur[e])
        signal.alarm(10)
        s = lib.trth_scene_load_opts(*(os.path.join(d, "s." + e).encode() for e in ("xml", "obj", "mtl")), d.encode(), r.choice((0, 16)), r.choice((0, 9)), r.randrange(2))
        if s:
            if lib.trth_scene_build(s, r.choice((1, 2, 8, 0, -3, 1000)), r.randrange(2)) == 0 and lib.trth_scene_flat(s):
                ok += 1
            lib.trth_scene_free(s)
        signal.alarm(0)
        n += 1
    print(f"seed {seed}: {n} mutated scenes, {ok} loaded and built, {n - ok} refused, no finding", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--target", default="images", choices=("images", "scene"))
    ap.add_argument("--seconds", type=int, default=120)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--child", default=None)
    a = ap.parse_args()
    if a.child:
        (child if a.target == "images" else child_scene)(a.child, a.seed, a.seconds)
        return 0
    lib = build()
    gcc = lambda n: subprocess.check_output(["gcc", "-print-file-name=" + n], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=gcc("libasan.so") + " " + gcc("libubsan.so"), ASAN_OPTIONS="detect_leaks=0:halt_on_error=1:allocator_may_return_null=1:max_allocation_size_mb=4096",
               UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1", OMP_NUM_THREADS="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", lib, "--target", a.target, "--seed", str(a.seed), "--seconds", str(a.seconds)], env=env)
    if p.returncode != 0:
        print(f"FINDING: the child ended with {p.returncode}; its last input is {WORK}/case_{a.seed}.bin (images) or {WORK}/scene_{a.seed}/ (scene)")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Pre-decodes the JPEG textures of scenes/ into binary PPM sidecars (<name>.jpg.ppm).

The reference decodes map_Kd with cv::imread (material.cpp:6).  Material::readinMap() decodes
baseline JPEG itself (host/jpeg.cpp, bit-identical to libjpeg); for anything else (progressive or
CMYK JPEG, PNG, ...) it falls back to a "<file>.ppm" sidecar, which this script writes with PIL.
The shipped staircase textures are baseline JPEGs and need no sidecar.
"""
import glob
import os
import sys

from PIL import Image

root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scenes")
for path in sorted(glob.glob(os.path.join(root, "**", "*.jpg"), recursive=True)):
    im = Image.open(path)
    info = (im.format, im.mode, im.size, "progressive" if im.info.get("progressive") else "baseline")
    rgb = im.convert("RGB")
    out = path + ".ppm"
    with open(out, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % rgb.size)
        f.write(rgb.tobytes())
    print(path, info, "->", out, os.path.getsize(out), file=sys.stderr)

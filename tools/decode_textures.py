#!/usr/bin/env python3
"""Pre-decodes the JPEG textures of scenes/ into binary PPM sidecars (<name>.jpg.ppm).

The reference decodes map_Kd with cv::imread (material.cpp:6), i.e. libjpeg's default
decoder (integer IDCT, fancy chroma upsampling).  OpenCV and the libjpeg headers are not
available to the C++ host code, so Material::readinMap() reads these sidecars instead.
PIL decodes with the same libjpeg(-turbo) defaults.  Run once; the outputs are committed.
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

"""Helpers to write tiny OBJ/MTL/XML scenes for edge-case tests."""
import os

import tinyraytracing_amd as T

XML = """<?xml version="1.0" encoding="utf-8"?>
<camera type="perspective" width="{w}" height="{h}" fovy="{fovy}">
	<eye x="{eye[0]}" y="{eye[1]}" z="{eye[2]}"/>
	<lookat x="{lookat[0]}" y="{lookat[1]}" z="{lookat[2]}"/>
	<up x="0.0" y="1.0" z="0.0"/>
</camera>
{lights}
"""


def write_scene(tmpdir, name, obj_text, mtl_text, lights=(), w=32, h=32, fovy=40.0, eye=(0, 0, 5), lookat=(0, 0, 0)):
    d = str(tmpdir)
    lights_xml = "\n".join(f'<light mtlname="{n}" radiance="{r[0]},{r[1]},{r[2]}"/>' for n, r in lights)
    with open(os.path.join(d, name + ".xml"), "w") as f:
        f.write(XML.format(w=w, h=h, fovy=fovy, eye=eye, lookat=lookat, lights=lights_xml))
    with open(os.path.join(d, name + ".obj"), "w") as f:
        f.write(obj_text)
    with open(os.path.join(d, name + ".mtl"), "w") as f:
        f.write(mtl_text)
    return d


def load(tmpdir, name, leaf_num=2, builder="auto", width=0, height=0):
    d = str(tmpdir)
    s = T.Scene.load(os.path.join(d, name + ".xml"), os.path.join(d, name + ".obj"), os.path.join(d, name + ".mtl"), d, width, height)
    s.build_bvh(leaf_num, builder)
    return s


MTL_BASIC = """newmtl white
Kd 0.7 0.7 0.7
Ks 0 0 0
Ns 1
Ni 1
newmtl lamp
Kd 0 0 0
Ks 0 0 0
Ns 1
Ni 1
newmtl shiny
Kd 0.3 0.2 0.1
Ks 0.5 0.5 0.5
Ns 50
Ni 1
newmtl glass
Kd 0.5 0.5 0.5
Ks 0 0 0
Tr 0.8 1 0.95
Ns 1
Ni 1.5
"""


def quad(x0, x1, y0, y1, z, vbase, nz=1.0):
    """Two triangles of an axis-aligned quad in the plane z; returns (lines, next vertex base)."""
    lines = [f"v {x0} {y0} {z}", f"v {x1} {y0} {z}", f"v {x1} {y1} {z}", f"v {x0} {y1} {z}"]
    b = vbase
    faces = [f"f {b}/1/{{n}} {b+1}/1/{{n}} {b+2}/1/{{n}}", f"f {b}/1/{{n}} {b+2}/1/{{n}} {b+3}/1/{{n}}"]
    return lines, faces, vbase + 4

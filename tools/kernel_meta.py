#!/usr/bin/env python3
"""Registers / LDS / scratch of every kernel of libtrt_hip.so, read from the gfx950 assembly hipcc leaves behind with --save-temps
(no GPU needed).  usage: tools/kernel_meta.py [extra hipcc flags...]   -> one line per kernel + the instantiation count."""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    d = tempfile.mkdtemp(prefix="kmeta")
    flags = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-mllvm",
             "-amdgpu-atomic-optimizer-strategy=None", f"-I{ROOT}/include", f"-I{ROOT}/tinyraytracing_amd/csrc"] + sys.argv[1:]
    subprocess.run(["/opt/rocm/bin/hipcc"] + flags + ["--save-temps", "-c", "-o", "x.o", f"{ROOT}/tinyraytracing_amd/csrc/trt_api.hip"], cwd=d, check=True,
                   stderr=subprocess.DEVNULL)
    s = open([x for x in glob.glob(d + "/*.s") if "gfx950" in x][0]).read()
    ks = re.findall(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S)
    names = subprocess.run(["c++filt"], input="\n".join(k for k, _ in ks), capture_output=True, text=True).stdout.splitlines()
    for (name, body), dn in zip(ks, names):
        g = lambda key: re.search(r"\.amdhsa_" + key + r" (\d+)", body)
        dn = re.sub(r"\(.*", "", dn).replace("void trtd::", "")
        vg, acc = int(g("next_free_vgpr").group(1)), g("accum_offset")
        arch = int(acc.group(1)) if acc else vg
        print(f"{dn:70s} vgpr {arch:>4} (+{vg - arch} acc) sgpr {g('next_free_sgpr').group(1):>3} lds {g('group_segment_fixed_size').group(1):>6} scratch {g('private_segment_fixed_size').group(1)}")
    print(len(ks), "kernel instantiations")


if __name__ == "__main__":
    main()

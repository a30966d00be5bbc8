#!/usr/bin/env python3
"""CPU baseline leg of bench.py: the oracle — the reference's algorithm restated (oracle/oracle.cpp,
kind "port") — on ONE socket's physical cores, one pinned OpenMP thread per core (OMP_PLACES=cores OMP_PROC_BIND=close),
on a bounded sample of the workload: all pixels at as many samples per pixel as fit the time budget (Mrays/s does not
depend on spp).  Prints one JSON object.  Never touches the GPU.
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def socket0_physical_cores():
    """One logical CPU per physical core of the first package this process may run on."""
    allowed = sorted(os.sched_getaffinity(0))
    cores = {}
    pkg0 = None
    for cpu in allowed:
        base = f"/sys/devices/system/cpu/cpu{cpu}/topology"
        try:
            pkg = int(open(base + "/physical_package_id").read())
            core = int(open(base + "/core_id").read())
        except OSError:
            pkg, core = 0, cpu
        if pkg0 is None:
            pkg0 = pkg
        if pkg == pkg0 and core not in cores:
            cores[core] = cpu
    return sorted(cores.values()), pkg0


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scene", default="back")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--tris", type=int, default=None)
    ap.add_argument("--leaf", type=int, default=None)
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=0x5EED0001)
    ap.add_argument("--seconds", type=float, default=15.0)
    ap.add_argument("--fixed-nee", action="store_true")
    a = ap.parse_args()

    cpus, pkg = socket0_physical_cores()
    # the OpenMP binding goes into the environment BEFORE libgomp exists in this process (it is loaded with liboracle.so below)
    os.sched_setaffinity(0, set(cpus))
    os.environ.update(OMP_NUM_THREADS=str(len(cpus)), OMP_PLACES="cores", OMP_PROC_BIND="close")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import tinyraytracing_amd as T

    threads = len(os.sched_getaffinity(0))
    scene = T.Scene.named(a.scene, a.width, a.height, leaf_num=a.leaf, n=a.tris)
    fx = T.TRT_FLAG_FIXED_NEE if a.fixed_nee else 0
    _, s1 = O.render(scene.flat, T.make_params(a.width, a.height, 1, a.seed, flags=fx), threads=threads)
    spp = int(max(1, min(a.spp, a.seconds / max(s1.seconds, 1e-3))))
    s = s1
    if spp > 1:
        _, s = O.render(scene.flat, T.make_params(a.width, a.height, spp, a.seed, flags=fx), threads=threads)
    print(json.dumps({"value": round(s.rays / s.seconds / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
                      "cpu_model": cpu_model(), "socket": pkg, "binding": "one thread per physical core of one socket, OMP_PLACES=cores OMP_PROC_BIND=close",
                      "sample": f"{a.scene} {a.width}x{a.height}, {spp} spp of every pixel ({s.rays} rays, {s.seconds:.2f} s), OpenMP over rows"}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot path on N MI355X (one process per GPU).

A step = one full render of the workload (default: the Cornell-style box `back` of
example-scenes-cg22 at 1920x1080, 256 spp — the configuration BASELINE.json's metric is quoted on).
With N > 1 the image rows are dealt to the ranks in interleaved 8-row stripes and gathered on rank 0
with a single RCCL gather inside the timed region (total work fixed: strong scaling).

Prints ONE JSON line on rank 0 (see README / DESIGN.md §Measurement for every field).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); 6290 GB/s is the measured copy rate

WORKLOADS = {
    "back": "example-scenes-cg22 test/back Cornell-style box (26 triangles)",
    "veach-mis": "example-scenes-cg22 veach-mis (2332 triangles, 3 lights)",
    "staircase": "example-scenes-cg22 staircase (31407 triangles, 6 lights, 3 textures)",
    "soup": "synthetic 1M random triangles in the Cornell box (deep BVH stress)",
    "blob": "synthetic displaced geodesic sphere (Stanford-style mesh)",
}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scene", default="back", choices=sorted(WORKLOADS))
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--tris", type=int, default=None, help="triangle count of the synthetic scenes")
    ap.add_argument("--leaf", type=int, default=None, help="triangles per BVH leaf (the reference uses 8); default: 8 for scenes of <= 64 triangles, else 2")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=None)
    ap.add_argument("--mem-gb", type=float, default=0.0, help="HBM budget for path state (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--overlap", action="store_true", help="TRT_FLAG_OVERLAP: two passes in flight (+4-5 %% Mrays/s; per-kernel timings then overlap)")
    ap.add_argument("--fixed-nee", action="store_true", help="TRT_FLAG_FIXED_NEE: unbiased light sampling + occlusion-test shadow rays (not the parity mode; not the headline)")
    ap.add_argument("--also-overlap", action="store_true", help="after the timed steps, time the same steps again with TRT_FLAG_OVERLAP and report it as with_pass_overlap")
    ap.add_argument("--save-png", default=None)
    return ap.parse_args()


def algorithmic_bytes(st):
    """SURVEY.md §8d canonical sizes: 32 B per child box + reference of a visited inner node (64 B for a
    two-child node, 128 B for a four-child one: trt_stats.inner_node_bytes), triangle test 48 B, ray record
    32 B read, hit record 16 B written, shaded hit 64 B, generated ray 32 B written, framebuffer 12 B / pixel."""
    closest_rays = st.rays_camera + st.rays_indirect
    nb = st.inner_node_bytes or 64
    b_closest = nb * st.inner_visits[0] + 48 * st.tri_tests[0] + (32 + 16) * closest_rays
    b_shadow = nb * st.inner_visits[1] + 48 * st.tri_tests[1] + (32 + 16) * st.rays_shadow
    b_shade = 64 * st.shaded_hits + 32 * (st.rays_shadow + st.rays_indirect)
    b_gen = 32 * st.rays_camera
    return {"trace_closest": b_closest, "trace_shadow": b_shadow, "shade": b_shade, "gen_primary": b_gen}


def main():
    a = parse()
    import torch
    import tinyraytracing_amd as T
    from tinyraytracing_amd import dist as D

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 bench.py --gpus N ...")
        raise SystemExit(f"--gpus {a.gpus} does not match WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    # TRT_BENCH_BACKEND=gloo: rehearsal of the multi-rank logic on a box with fewer GPUs than ranks (ranks share
    # device local_rank % device_count, the gather goes through host memory); the real run is nccl = RCCL over xGMI.
    backend = os.environ.get("TRT_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    seed = a.seed if a.seed is not None else {"back": T.SEED_BACK, "veach-mis": 0x5EED0002, "staircase": T.SEED_STAIRCASE,
                                               "soup": T.SEED_SOUP, "blob": T.SEED_BLOB}[a.scene]
    t0 = time.time()
    scene = T.Scene.named(a.scene, a.width, a.height, leaf_num=a.leaf, n=a.tris)
    t_load = time.time() - t0
    renderer = T.Renderer(scene, local_rank)
    budget = int(a.mem_gb * (1 << 30))

    ov = (T.TRT_FLAG_OVERLAP if a.overlap else 0) | (T.TRT_FLAG_FIXED_NEE if a.fixed_nee else 0)
    p_time = D.shard_params(a.width, a.height, a.spp, seed, rank, world, flags=T.TRT_FLAG_TIMING | ov, mem_budget=budget)
    p_count = D.shard_params(a.width, a.height, a.spp, seed, rank, world, flags=T.TRT_FLAG_TIMING | T.TRT_FLAG_COUNT | (T.TRT_FLAG_FIXED_NEE if a.fixed_nee else 0), mem_budget=budget)
    nrows = len(T.rows_selected(p_time))
    out = torch.empty((nrows, a.width, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
    stream = torch.cuda.current_stream().cuda_stream

    def step(p):
        def fn(pp):
            st = renderer.render_into(pp, out, stream)
            return out, st
        # every rank renders its stripes, then ONE gather of the packed stripes to rank 0
        return D.render_distributed(fn, a.width, a.height, a.spp, seed, dist=dist, device=f"cuda:{local_rank}",
                                    flags=p.flags, mem_budget=budget)

    # counting pass (untimed): inner-node visits / triangle tests for the algorithmic bytes
    img, st_count = step(p_count)
    for _ in range(max(a.warmup - 1, 0)):
        step(p_time)

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    sync()
    t_begin = time.perf_counter()
    kernel_ms = [0.0] * 8
    launches = [0] * 8
    rays_rank = 0
    render_ms = 0.0
    for _ in range(a.steps):
        img, st = step(p_time)
        rays_rank += st.rays
        render_ms += st.render_ms
        for k in range(8):
            kernel_ms[k] += st.kernel_ms[k]
            launches[k] += st.launches[k]
    sync()
    elapsed = time.perf_counter() - t_begin
    # --also-overlap: beside the contract's number, the same steps with two sample passes in flight
    # (TRT_FLAG_OVERLAP, what render()/tinyrt use).  Not `value`: per-kernel hipEvent times of overlapping passes
    # contain each other.  Off by default so that a rocprofv3 run of the default command sees the timed launches only.
    overlap_extra = None
    if world == 1 and not a.overlap and a.also_overlap:
        p_ov = D.shard_params(a.width, a.height, a.spp, seed, rank, world, flags=T.TRT_FLAG_OVERLAP | (T.TRT_FLAG_FIXED_NEE if a.fixed_nee else 0), mem_budget=budget)
        step(p_ov)
        sync()
        t_ov = time.perf_counter()
        rays_ov = 0
        for _ in range(a.steps):
            rays_ov += step(p_ov)[1].rays
        sync()
        t_ov = time.perf_counter() - t_ov
        overlap_extra = {"value": round(rays_ov / t_ov / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(t_ov / a.steps * 1e3, 3)}
    if dist is not None:
        red_dev = f"cuda:{local_rank}" if backend == "nccl" else "cpu"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
        rr = torch.tensor([rays_rank], dtype=torch.int64, device=red_dev)
        dist.all_reduce(rr)
        rays_total = int(rr.item())
    else:
        rays_total = rays_rank

    if rank == 0:
        mrays = rays_total / elapsed / 1e6
        # roofline of the dominant kernel of THIS rank: algorithmic bytes (counting pass, same seed -> same
        # counts every step) / its summed launch time in the timed region (hipEvents on the launch stream)
        by = algorithmic_bytes(st_count)
        names = T.KERNEL_NAMES
        dom = max(range(len(names)), key=lambda k: kernel_ms[k])
        dom_name = names[dom]
        dom_bytes_step = by.get(dom_name, 0)
        dom_ms = kernel_ms[dom]
        achieved = dom_bytes_step * a.steps / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        traffic = None  # measured HBM bytes per launch of that kernel (PMC passes, committed under profiles/), when this is that workload
        tpath = os.path.join(ROOT, "profiles", f"hbm_traffic_{a.scene}_{a.height}p_{a.spp}spp.json")
        if world == 1 and a.width * 9 == a.height * 16 and os.path.exists(tpath):
            traffic = json.load(open(tpath))["bytes_per_launch"].get(dom_name)
        roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "launches_per_step": launches[dom] // max(a.steps, 1),
                    "avg_launch_ms": round(dom_ms / max(launches[dom], 1), 5),
                    "algorithmic_bytes_per_launch": int(dom_bytes_step * a.steps / max(launches[dom], 1)),
                    "frac_of_measured_copy_peak": round(achieved / 6290.0, 4)}
        if traffic and launches[dom] and dom_ms > 0:
            # the same launches' measured HBM-side bytes (PMC passes, profiles/) over the live launch duration
            meas = traffic / (dom_ms / launches[dom] * 1e-3) / 1e9
            roofline["traffic_GBps"] = round(meas, 1)
            roofline["traffic_frac"] = round(meas / HBM_PEAK_GBS, 4)
        if achieved > HBM_PEAK_GBS:
            roofline["note"] = ("algorithmic bytes (every node / triangle record a ray touches) exceed the HBM peak because the scene is "
                                "served from L1/L2/Infinity Cache; the kernel is bound by VALU issue and the L1 tag rate, not by HBM (DESIGN.md 5)")
        total_bytes_step = sum(by.values()) + 12 * a.width * nrows
        kernels = {names[k]: {"ms_per_step": round(kernel_ms[k] / a.steps, 3), "launches_per_step": launches[k] // max(a.steps, 1),
                              "algorithmic_GBps": round(by.get(names[k], 0) * a.steps / (kernel_ms[k] * 1e-3) / 1e9, 1) if kernel_ms[k] > 0 and names[k] in by else None}
                   for k in range(len(names))}
        result = {
            "metric": "Mrays/s", "value": round(mrays, 2), "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "f32", "data": f"scene '{a.scene}' ({WORKLOADS[a.scene]}), counter RNG seed {seed:#x}",
            "config": {"workload": f"{WORKLOADS[a.scene]}, {a.width}x{a.height}, {a.spp} spp", "scene": a.scene, "width": a.width,
                       "height": a.height, "spp": a.spp, "triangles": scene.info["n_triangles"], "bvh_nodes": scene.arrays()["n_nodes"],
                       "leaf_num": a.leaf if a.leaf is not None else T.default_leaf(a.scene, scene.info["n_triangles"]), "overlap_passes": bool(a.overlap), "fixed_nee": bool(a.fixed_nee), "tiling": "single GPU" if world == 1 else f"rows interleaved in 8-row stripes over {world} GPUs + one RCCL gather"},
            "rays_per_step": rays_total // a.steps,
            "rays_rank0": {"camera": st.rays_camera, "shadow": st.rays_shadow, "indirect": st.rays_indirect},
            "device_render_ms_per_step_rank0": round(render_ms / a.steps, 3),
            "hbm_algorithmic_GBps_rank0": round(total_bytes_step * a.steps / (render_ms * 1e-3) / 1e9, 1) if render_ms > 0 else None,
            "simd_utilisation_traversal": {
                "inner_steps": round((st_count.inner_visits[0] + st_count.inner_visits[1]) / (64.0 * st_count.wave_steps[0]), 3) if st_count.wave_steps[0] else None,
                "leaf_steps": round((st_count.tri_tests[0] + st_count.tri_tests[1]) / (64.0 * st_count.wave_steps[1]), 3) if st_count.wave_steps[1] else None,
                "visits_per_ray": round((st_count.inner_visits[0] + st_count.inner_visits[1]) / max(st_count.rays, 1), 2),
                "tri_tests_per_ray": round((st_count.tri_tests[0] + st_count.tri_tests[1]) / max(st_count.rays, 1), 2)},
            "roofline": roofline, "kernels_rank0": kernels, "passes": st.passes, "max_path_vertices": st.max_bounces + 1,
            "scene_load_build_s": round(t_load, 2), "with_pass_overlap": overlap_extra,
        }
        if a.save_png and img is not None:
            T.imshow(img.cpu().numpy(), a.save_png)
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(a, scene, seed)
            result["gpu_over_cpu"] = round(mrays / result["cpu_baseline"]["value"], 1)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(a, scene, seed):
    """The oracle (oracle/liboracle.so: the reference algorithm — unordered, unculled traversal — restated
    in C++ with OpenMP over rows) timed on this box's host cores on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    import tinyraytracing_amd as T
    threads = len(os.sched_getaffinity(0))
    fx = T.TRT_FLAG_FIXED_NEE if a.fixed_nee else 0
    p1 = T.make_params(a.width, a.height, 1, seed, flags=fx)
    _, s1 = O.render(scene.flat, p1, threads=threads)
    spp = int(max(1, min(a.spp, a.cpu_seconds / max(s1.seconds, 1e-3))))
    if spp > 1:
        _, s = O.render(scene.flat, T.make_params(a.width, a.height, spp, seed, flags=fx), threads=threads)
    else:
        s = s1
    return {"value": round(s.rays / s.seconds / 1e6, 3), "unit": "Mrays/s", "cores": threads, "kind": "port",
            "sample": f"{a.scene} {a.width}x{a.height}, {spp} spp of every pixel ({s.rays} rays, {s.seconds:.2f} s), OpenMP over rows"}


if __name__ == "__main__":
    main()

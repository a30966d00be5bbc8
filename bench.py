#!/usr/bin/env python3
"""bench.py — Mrays/s of the path-tracing hot path on N MI355X (one process per GPU).

A step = one full render of the workload (default: the Cornell-style box `back` of
example-scenes-cg22 at 1920x1080, 256 spp — the configuration BASELINE.json's metric is quoted on).
With N > 1 the image rows are dealt to the ranks in interleaved 8-row stripes and gathered on rank 0
with a single RCCL gather inside the timed region (total work fixed: strong scaling).

Prints ONE JSON line on rank 0 (DESIGN.md §5 explains every field):
  value / ms_per_step     wall clock of the K timed steps (barrier + synchronize on both sides, max over ranks)
  roofline                dominant kernel of the headline: algorithmic bytes (SURVEY.md §8d canonical sizes x counts
                          from an untimed counting render) / that kernel's hipEvent time in one profiling step after the timed
                          region; `traffic` = its HBM-side bytes per launch from two rocprofv3 --pmc child passes of the same
                          command (FETCH_SIZE, WRITE_SIZE; run before this process touches the GPU — live_traffic()), null with
                          the reason in `traffic_note` when the profiler cannot run; `traffic_from_profiles` = the committed
                          counter files of profiles/ beside it
  extra_workloads         (N = 1) the other rows of BASELINE.md §3 — veach-mis, staircase at 1080p/256 spp, the 1 M-triangle
                          soup of config 3 and config 5's 10 M-triangle scene at 3840x2160 / 64 spp per pass — timed the same
                          way with fewer steps, each with its own roofline
  cpu_baseline            the oracle ("port") on one socket's physical cores, pinned, in a child process
"""
import argparse
import json
import os
import subprocess
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
SEEDS = {"back": 0x5EED0001, "veach-mis": 0x5EED0002, "staircase": 0x5EED0004, "soup": 0x5EED0003, "blob": 0x5EED0005}
# (scene, spp, steps, width, height, triangles): BASELINE.md §3's other rows, timed after the headline when N = 1 — the other two cg22
# scenes at the headline's size, config 3, and config 5's scene at the resolution and per-pass queue length of config 5
EXTRA = [("veach-mis", 256, 2, None, None, None), ("staircase", 256, 1, None, None, None), ("soup", 64, 2, None, None, None),
         ("blob", 64, 1, 3840, 2160, 10_000_000)]


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
    ap.add_argument("--builder", default="auto", choices=["auto", "sweep", "binned", "lbvh"], help="BVH builder: the host SAH builders, or the GPU LBVH builder of include/trt_build.h")
    ap.add_argument("--seed", type=lambda s: int(s, 0), default=None)
    ap.add_argument("--mem-gb", type=float, default=0.0, help="HBM budget for path state (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip extra_workloads (veach-mis, staircase, soup after the headline)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample")
    ap.add_argument("--overlap", action="store_true", help="TRT_FLAG_OVERLAP: two passes in flight (+4-5 %% Mrays/s; per-kernel timings then overlap)")
    ap.add_argument("--fixed-nee", action="store_true", help="TRT_FLAG_FIXED_NEE: unbiased light sampling + occlusion-test shadow rays (not the parity mode; not the headline)")
    ap.add_argument("--no-overlap-extra", action="store_true", help="skip the second timing of the same steps with TRT_FLAG_OVERLAP (reported as with_pass_overlap; what render()/tinyrt "
                    "ship with).  rocprofv3 runs pass it so that every launch the profiler sees is a non-overlapped one")
    ap.add_argument("--also-overlap", action="store_true", help="(default now; kept for old command lines)")
    ap.add_argument("--group", type=int, default=0, metavar="N", help="(N=1 process) time the C boundary of the multi-GPU path instead: trt_group_render_device over a "
                    "group of N entries, all naming device 0 on a one-GPU box (its overhead against trt_render_device is then on record)")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 --pmc child passes (FETCH_SIZE, WRITE_SIZE) that measure roofline.traffic live "
                    "(they run by default for the contract's own command: N = 1, the headline workload, extras on; about 30 s)")
    ap.add_argument("--traffic", action="store_true", help="run those passes for any other N = 1 command too")
    ap.add_argument("--save-png", default=None)
    return ap.parse_args()


def algorithmic_bytes(st):
    """SURVEY.md §8d canonical sizes: 32 B per child box + reference of a visited inner node (64 B for a two-child node
    or a compressed four-child one, 128 B for an exact four-child one: trt_stats.inner_node_bytes), triangle test 48 B, ray
    record 32 B read, hit record 16 B written, shaded hit 64 B, generated ray 32 B written, framebuffer 12 B / pixel."""
    closest_rays = st.rays_camera + st.rays_indirect
    nb = st.inner_node_bytes or 64
    b_closest = nb * st.inner_visits[0] + 48 * st.tri_tests[0] + (32 + 16) * closest_rays
    b_shadow = nb * st.inner_visits[1] + 48 * st.tri_tests[1] + (32 + 16) * st.rays_shadow
    b_shade = 64 * st.shaded_hits + 32 * (st.rays_shadow + st.rays_indirect)
    b_gen = 32 * st.rays_camera
    return {"trace_closest": b_closest, "trace_shadow": b_shadow, "shade": b_shade, "gen_primary": b_gen}


def traffic_from_profiles(scene, height, spp):
    """HBM-side bytes per launch measured by rocprofv3 PMC passes of this command in an EARLIER run and committed under
    profiles/ (tools/prof.sh + tools/pmc_summary.py) — reported beside the live numbers, never as `roofline.traffic`."""
    tpath = os.path.join(ROOT, "profiles", f"hbm_traffic_{scene}_{height}p_{spp}spp.json")
    if not os.path.exists(tpath):
        return None
    d = json.load(open(tpath))
    return {"file": os.path.relpath(tpath, ROOT), "source": d.get("source"), "bytes_per_launch": d["bytes_per_launch"]}


def live_traffic(a):
    """roofline.traffic measured LIVE: before this process touches the GPU, the same command (one warm-up + one timed step, no extras) is run
    twice as a child under `rocprofv3 --pmc FETCH_SIZE --kernel-trace` / `--pmc WRITE_SIZE --kernel-trace` — the counters in separate passes, with
    --kernel-trace only, as /opt/skills/guides/MI355X_MICROARCH.md prescribes — and the per-kernel sums are turned into HBM-side bytes per
    launch exactly as tools/pmc_summary.py does for the committed files (reads x 2: gfx950's FETCH_SIZE counts each 128-B line as 64 B; the
    counting kernels of the untimed counting render are left out).  Returns ({kernel: bytes per launch}, note); ({}, reason) when it cannot run —
    never a guess.  N = 1 only."""
    import shutil
    import tempfile
    if any(k.startswith("ROCPROF") or k.startswith("ROCP_") for k in os.environ):
        return {}, "not attempted: this process already runs under a profiler"
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return {}, "not attempted: rocprofv3 not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import pmc_summary as P
    tmp = tempfile.mkdtemp(prefix="trt_pmc_", dir="/tmp")
    child = [sys.executable, os.path.abspath(__file__), "--scene", a.scene, "--width", str(a.width), "--height", str(a.height), "--spp", str(a.spp),
             "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-extra", "--no-overlap-extra", "--no-traffic", "--builder", a.builder]
    if a.tris:
        child += ["--tris", str(a.tris)]
    if a.leaf:
        child += ["--leaf", str(a.leaf)]
    if a.seed is not None:
        child += ["--seed", hex(a.seed)]
    if a.fixed_nee:
        child += ["--fixed-nee"]
    env = dict(os.environ)
    env["TMPDIR"] = "/tmp"
    sums = {}
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter)
            r = subprocess.run([exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "--"] + child, cwd="/tmp", env=env,
                               capture_output=True, text=True, timeout=240)
            if r.returncode != 0:
                return {}, f"rocprofv3 --pmc {counter} failed (rc {r.returncode}): {(r.stderr or '').strip().splitlines()[-1:]}"
            sums[counter] = P.collect(d, counter)
        per = {}
        for k, (n, kib) in sums["FETCH_SIZE"].items():
            nm, counting = P.short(k)
            if nm and not counting:
                e = per.setdefault(nm, [0, 0.0, 0.0])
                e[0] += n
                e[1] += kib
        for k, (n, kib) in sums["WRITE_SIZE"].items():
            nm, counting = P.short(k)
            if nm and not counting:
                per.setdefault(nm, [0, 0.0, 0.0])[2] += kib
        out = {nm: int((2.0 * v[1] + v[2]) * 1024 / max(v[0], 1)) for nm, v in per.items() if v[0]}
        return out, ("live: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace only) over this command with --steps 1 --warmup 1, "
                     "run as child processes before the timed run; reads x 2 (gfx950 FETCH_SIZE), per launch of the non-counting kernels")
    except Exception as e:  # a profiler that cannot run leaves traffic null, with the reason
        return {}, f"failed: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


class Bench:
    def __init__(self, a, world, rank, local_rank, dist, backend):
        import torch
        import tinyraytracing_amd as T
        from tinyraytracing_amd import dist as D
        self.a, self.world, self.rank, self.local_rank, self.dist, self.backend = a, world, rank, local_rank, dist, backend
        self.torch, self.T, self.D = torch, T, D

    def sync(self):
        self.torch.cuda.synchronize()
        if self.dist is not None:
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def measure(self, scene_name, width, height, spp, steps, warmup, seed, leaf=None, tris=None, base_flags=0, also_overlap=False, save_png=None, builder=None):
        """Counting render (untimed), warm-up, then `steps` timed renders of one workload.  Returns the result fields."""
        T, D, torch, a = self.T, self.D, self.torch, self.a
        world, rank, local_rank, dist = self.world, self.rank, self.local_rank, self.dist
        t0 = time.time()
        scene = T.Scene.named(scene_name, width, height, leaf_num=leaf, n=tris, builder=builder or a.builder, device=local_rank)
        t_load = time.time() - t0
        renderer = T.Renderer(scene, local_rank)
        budget = int(a.mem_gb * (1 << 30))
        group = None
        if a.group > 0 and world == 1:
            # --group N: the C boundary of the multi-GPU path (trt_group_render_device: resident host threads, interleaved stripes, gather,
            # un-interleave; image left on the first device) with N entries naming this device — all its overhead, none of its speed-up
            renderer.close()
            group = T.GroupRenderer(scene, [local_rank] * a.group)
            if budget == 0:  # N handles share one device's memory: give each its share instead of three quarters of what is free
                budget = int(torch.cuda.mem_get_info(local_rank)[0] * 0.6 / a.group)
        fx = T.TRT_FLAG_FIXED_NEE if a.fixed_nee else 0
        # The timed steps run WITHOUT TRT_FLAG_TIMING (two hipEventRecords per launch, ~60 launches per step: at one rank of eight a
        # step is ~12 ms and they show); per-kernel times come from ONE extra step with the events on, after the timed region.
        flags_time = base_flags | fx
        flags_prof = T.TRT_FLAG_TIMING | base_flags | fx
        flags_count = T.TRT_FLAG_COUNT | fx
        p_rank = D.shard_params(width, height, spp, seed, rank, world, flags=flags_time, mem_budget=budget)
        nrows = len(T.rows_selected(p_rank))
        out = torch.empty((nrows, width, 3), dtype=torch.float32, device=f"cuda:{local_rank}")
        stream = torch.cuda.current_stream().cuda_stream

        step_times = D.StepTimes()  # this rank's marks of the TIMED steps only (render | gather + un-interleave)

        def step(flags, times=None):
            if group is not None:
                pg = T.make_params(width, height, spp, seed, flags=flags, mem_budget=budget)
                st_g, _ = group.render_into(pg, out)
                return out, st_g

            def fn(pp):
                return out, renderer.render_into(pp, out, stream)
            # every rank renders its stripes, then ONE gather of the packed stripes to rank 0
            return D.render_distributed(fn, width, height, spp, seed, dist=dist, device=f"cuda:{local_rank}", flags=flags, mem_budget=budget, times=times)

        img, st_count = step(flags_count)  # inner-node visits / triangle tests for the algorithmic bytes
        for _ in range(max(warmup - 1, 0)):
            step(flags_time)
        self.sync()
        t_begin = time.perf_counter()
        rays_rank = 0
        render_ms = 0.0
        st = st_count
        for _ in range(steps):
            img, st = step(flags_time, step_times)
            rays_rank += st.rays
            render_ms += st.render_ms
        self.sync()
        elapsed = time.perf_counter() - t_begin
        rank_render_ms, rank_gather_ms = step_times.totals_ms() if group is None else (0.0, 0.0)
        # the profiling step: the same render once more with hipEvents around every launch (on the launch stream, inside the
        # library); its per-kernel sums are scaled to `steps` so that every per-step figure below keeps its meaning
        img, st_prof = step(flags_prof)
        self.sync()
        kernel_ms = [st_prof.kernel_ms[k] * steps for k in range(8)]
        launches = [st_prof.launches[k] * steps for k in range(8)]
        # beside the contract's number, the same steps with two sample passes in flight (TRT_FLAG_OVERLAP, what render()/tinyrt
        # use).  Not `value`: per-kernel hipEvent times of overlapping passes contain each other.  On by default at N = 1
        # (--no-overlap-extra skips it, e.g. so that a rocprofv3 run of the command sees the timed launches and the profiling step only).
        overlap_extra = None
        if world == 1 and also_overlap and not (base_flags & T.TRT_FLAG_OVERLAP) and spp >= 2:
            step(T.TRT_FLAG_OVERLAP | fx)
            self.sync()
            t_ov = time.perf_counter()
            rays_ov = 0
            for _ in range(steps):
                rays_ov += step(T.TRT_FLAG_OVERLAP | fx)[1].rays
            self.sync()
            t_ov = time.perf_counter() - t_ov
            overlap_extra = {"value": round(rays_ov / t_ov / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(t_ov / steps * 1e3, 3)}
        if dist is not None:
            red_dev = f"cuda:{local_rank}" if self.backend == "nccl" else "cpu"
            tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
            rr = torch.tensor([rays_rank], dtype=torch.int64, device=red_dev)
            dist.all_reduce(rr)
            rays_total = int(rr.item())
            # every rank's own step, for the first run on a real node: [device render ms (trt_stats), render ms and gather ms between this
            # rank's marks (a rank that finishes early waits in the gather), k_tail ms, rays] per step
            mine = torch.tensor([render_ms / steps, rank_render_ms / steps, rank_gather_ms / steps, st_prof.kernel_ms[T.KERNEL_NAMES.index("tail")] if "tail" in T.KERNEL_NAMES else 0.0,
                                 rays_rank / steps], dtype=torch.float64, device=red_dev)
            every = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(every, mine)
            per_rank = [{"rank": r, "device_render_ms": round(float(v[0]), 3), "render_ms": round(float(v[1]), 3), "gather_wait_ms": round(float(v[2]), 3),
                         "tail_ms": round(float(v[3]), 3), "rays_per_step": int(v[4])} for r, v in enumerate(every)]
        else:
            rays_total = rays_rank
            per_rank = None
        res = None
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
            achieved = dom_bytes_step * steps / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
            roofline = {"bound": "hbm", "kernel": dom_name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": round(achieved / HBM_PEAK_GBS, 4),
                        "traffic": None,  # filled below from the live PMC passes (live_traffic) when they ran; null with the reason otherwise
                        "launches_per_step": launches[dom] // max(steps, 1),
                        "avg_launch_ms": round(dom_ms / max(launches[dom], 1), 5),
                        "algorithmic_bytes_per_launch": int(dom_bytes_step * steps / max(launches[dom], 1)),
                        "frac_of_measured_copy_peak": round(achieved / 6290.0, 4)}
            lt = getattr(self, "live", None)
            if lt and (scene_name, width, height, spp) == lt["key"]:
                roofline["traffic_note"] = lt["note"]
                if dom_name in lt["bytes"] and launches[dom] and dom_ms > 0:
                    tb = lt["bytes"][dom_name]
                    roofline["traffic"] = tb  # HBM-side bytes per launch of the dominant kernel, measured by the PMC passes of this very invocation
                    roofline["traffic_GBps"] = round(tb / (dom_ms / launches[dom] * 1e-3) / 1e9, 1)
                    roofline["traffic_frac_of_peak"] = round(roofline["traffic_GBps"] / HBM_PEAK_GBS, 4)
                    roofline["traffic_over_algorithmic"] = round(tb / max(roofline["algorithmic_bytes_per_launch"], 1), 3)
                    roofline["traffic_all_kernels"] = lt["bytes"]
            tp = traffic_from_profiles(scene_name, height, spp) if (world == 1 and width * 9 == height * 16) else None
            if tp and dom_name in tp["bytes_per_launch"] and launches[dom] and dom_ms > 0:
                tb = tp["bytes_per_launch"][dom_name]
                meas = tb / (dom_ms / launches[dom] * 1e-3) / 1e9
                roofline["traffic_from_profiles"] = {"file": tp["file"], "bytes_per_launch": tb, "GBps_at_live_launch_time": round(meas, 1),
                                                     "frac": round(meas / HBM_PEAK_GBS, 4),
                                                     "over_algorithmic": round(tb / max(roofline["algorithmic_bytes_per_launch"], 1), 3)}
            # the same account for every path kernel (the dominant one changes with the workload: since round 4 `back` spends 28 / 26 / 23 ms in closest hits /
            # shade / shadow rays): algorithmic bytes per launch, live launch time, and the live counter bytes where the PMC passes ran
            per_kernel = {}
            for k in range(len(names)):
                if names[k] in by and kernel_ms[k] > 0 and launches[k]:
                    alg_pl = by[names[k]] * steps / launches[k]
                    e = {"avg_launch_ms": round(kernel_ms[k] / launches[k], 5), "algorithmic_bytes_per_launch": int(alg_pl),
                         "achieved_GBps": round(alg_pl / (kernel_ms[k] / launches[k] * 1e-3) / 1e9, 1)}
                    e["frac"] = round(e["achieved_GBps"] / HBM_PEAK_GBS, 4)
                    if lt and (scene_name, width, height, spp) == lt["key"] and names[k] in lt["bytes"]:
                        e["traffic"] = lt["bytes"][names[k]]
                        e["traffic_GBps"] = round(e["traffic"] / (kernel_ms[k] / launches[k] * 1e-3) / 1e9, 1)
                        e["traffic_frac_of_peak"] = round(e["traffic_GBps"] / HBM_PEAK_GBS, 4)
                    per_kernel[names[k]] = e
            roofline["by_kernel"] = per_kernel
            if achieved > HBM_PEAK_GBS:
                roofline["note"] = ("algorithmic bytes (every node / triangle record a ray touches) exceed the HBM peak because the scene is "
                                    "served from L1/L2/Infinity Cache; the kernel is bound by VALU/SALU issue and the CU's texture-address rate, not by HBM (DESIGN.md 5)")
            total_bytes_step = sum(by.values()) + 12 * width * nrows
            kernels = {names[k]: {"ms_per_step": round(kernel_ms[k] / steps, 3), "launches_per_step": launches[k] // max(steps, 1),
                                  "algorithmic_GBps": round(by.get(names[k], 0) * steps / (kernel_ms[k] * 1e-3) / 1e9, 1) if kernel_ms[k] > 0 and names[k] in by else None}
                       for k in range(len(names))}
            res = {
                "value": round(mrays, 2), "unit": "Mrays/s", "steps": steps, "warmup": warmup, "ms_per_step": round(elapsed / steps * 1e3, 3),
                "data": f"scene '{scene_name}' ({WORKLOADS[scene_name]}), counter RNG seed {seed:#x}",
                "config": {"workload": f"{WORKLOADS[scene_name]}, {width}x{height}, {spp} spp", "scene": scene_name, "width": width,
                           "height": height, "spp": spp, "triangles": scene.info["n_triangles"], "bvh_nodes": scene.arrays()["n_nodes"], "bvh_builder": builder or a.builder,
                           "leaf_num": leaf if leaf is not None else T.default_leaf(scene_name, scene.info["n_triangles"]),
                           "inner_node_bytes": st_count.inner_node_bytes,
                           "overlap_passes": bool(base_flags & T.TRT_FLAG_OVERLAP), "fixed_nee": bool(a.fixed_nee),
                           "group_entries_on_this_device": a.group if group is not None else None,
                           "tiling": "single GPU" if world == 1 else f"rows interleaved in 8-row stripes over {world} GPUs + one RCCL gather"},
                "rays_per_step": rays_total // steps,
                "rays_rank0": {"camera": st.rays_camera, "shadow": st.rays_shadow, "indirect": st.rays_indirect},
                "device_render_ms_per_step_rank0": round(render_ms / steps, 3),
                "hbm_algorithmic_GBps_rank0": round(total_bytes_step * steps / (render_ms * 1e-3) / 1e9, 1) if render_ms > 0 else None,
                "simd_utilisation_traversal": {
                    "inner_steps": round((st_count.inner_visits[0] + st_count.inner_visits[1]) / (64.0 * st_count.wave_steps[0]), 3) if st_count.wave_steps[0] else None,
                    "leaf_steps": round((st_count.tri_tests[0] + st_count.tri_tests[1]) / (64.0 * st_count.wave_steps[1]), 3) if st_count.wave_steps[1] else None,
                    "visits_per_ray": round((st_count.inner_visits[0] + st_count.inner_visits[1]) / max(st_count.rays, 1), 2),
                    "tri_tests_per_ray": round((st_count.tri_tests[0] + st_count.tri_tests[1]) / max(st_count.rays, 1), 2)},
                "roofline": roofline, "kernels_rank0": kernels, "passes": st.passes, "max_path_vertices": st.max_bounces + 1,
                "scene_load_build_s": round(t_load, 2), "with_pass_overlap": overlap_extra,
                # how the numbers above were taken (rounds 1-2 had hipEvents around every launch of the timed steps; since round 3 they have none)
                "timing": {"events_in_timed_steps": False, "kernel_ms_from": f"1 profiling step (TRT_FLAG_TIMING) after the timed region x {steps}",
                           "value_from": "wall clock of the timed steps between barriers, max over ranks"},
                "per_rank": per_rank,
            }
            if save_png and img is not None:
                T.imshow(img.cpu().numpy(), save_png)
        renderer.close()
        if group is not None:
            group.close()
        del out
        torch.cuda.empty_cache()
        return res, scene


def self_launch(a):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks (one process per GPU) the way the contract's
    launcher does — python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 — as a CHILD process, before
    this process has imported torch or touched a GPU (an exec after GPU initialisation is not allowed on this pool), relay its output
    and exit with its return code."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    print(f"bench.py: --gpus {a.gpus} without WORLD_SIZE: launching {' '.join(cmd[1:9])} ...", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(a))
    live = None
    contract_cmd = (a.scene, a.width, a.height, a.spp) == ("back", 1920, 1080, 256) and not a.no_extra and not a.fixed_nee and not a.overlap and a.leaf is None and a.tris is None
    if a.gpus == 1 and "WORLD_SIZE" not in os.environ and not a.no_traffic and not a.group and (contract_cmd or a.traffic):
        t_pmc = time.time()
        by, note = live_traffic(a)  # child processes; this process has not touched the GPU yet
        live = {"key": (a.scene, a.width, a.height, a.spp), "bytes": by, "note": note + f" ({time.time() - t_pmc:.0f} s)"}
    import torch
    import tinyraytracing_amd as T

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
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

    B = Bench(a, world, rank, local_rank, dist, backend)
    B.live = live
    seed = a.seed if a.seed is not None else SEEDS[a.scene]
    res, scene = B.measure(a.scene, a.width, a.height, a.spp, a.steps, a.warmup, seed, leaf=a.leaf, tris=a.tris,
                           base_flags=T.TRT_FLAG_OVERLAP if a.overlap else 0, also_overlap=not a.no_overlap_extra, save_png=a.save_png)
    if rank == 0:
        result = {"metric": "Mrays/s", "value": res["value"], "unit": "Mrays/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
                  "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32"}
        result.update({k: v for k, v in res.items() if k not in result})
        headline = (a.scene, a.width, a.height, a.spp) == ("back", 1920, 1080, 256)
        if world == 1 and headline and not a.no_extra and not a.fixed_nee and not a.overlap:
            extra = []
            for name, spp, steps, ew, eh, etris in EXTRA:
                r, sc = B.measure(name, ew or a.width, eh or a.height, spp, steps, 1, SEEDS[name], tris=etris)
                extra.append({k: r[k] for k in ("value", "unit", "steps", "ms_per_step", "config", "rays_per_step", "roofline", "kernels_rank0", "simd_utilisation_traversal")})
                sc.close()
            result["extra_workloads"] = extra
        if world == 1 and not a.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(a, seed)
            if result["cpu_baseline"] and result["cpu_baseline"].get("value"):
                result["gpu_over_cpu"] = round(res["value"] / result["cpu_baseline"]["value"], 1)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(a, seed):
    """The oracle (oracle/liboracle.so: the reference algorithm — unordered, unculled traversal — restated in C++ with
    OpenMP over rows) timed on ONE socket's physical cores of this box, one pinned thread per core, on a bounded sample
    of the same workload.  Runs in a child process (tools/cpu_baseline.py) so that OMP_PLACES /
    OMP_PROC_BIND take effect before any OpenMP runtime is initialised, and so that it never shares this process' GPU."""
    cmd = [sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), "--scene", a.scene, "--width", str(a.width), "--height", str(a.height),
           "--spp", str(a.spp), "--seed", hex(seed), "--seconds", str(a.cpu_seconds)]
    if a.tris:
        cmd += ["--tris", str(a.tris)]
    if a.leaf:
        cmd += ["--leaf", str(a.leaf)]
    if a.fixed_nee:
        cmd += ["--fixed-nee"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, check=True).stdout.strip().splitlines()[-1]
        return json.loads(out)
    except Exception as e:  # a baseline that could not run is reported as such, never guessed
        return {"value": None, "unit": "Mrays/s", "cores": 0, "kind": "port", "sample": f"failed: {e}"}


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""What every rank of an N-GPU render does, measured on ONE GPU (VERDICT r03 task 3): for N in {2, 4, 8} each rank's interleaved row
stripes (trt_params.row_block / row_mod / row_rem, the partitioning of SURVEY.md §8e) are rendered one after the other on device 0 with
the shipped kernels, timed like bench.py's steps (wall clock, no event records), then once more with TRT_FLAG_TIMING for the per-kernel
split.  The ranks of a real node run side by side on their own GPUs, so the step of the job is the SLOWEST rank's time plus the gather:

    predicted speed-up(N) = T(1 GPU, whole image) / (max over ranks T(rank) + gather)

with the gather priced at the payload over one xGMI link (every sender has its own link to the root: W * H * 12 B / N per rank at
~50 GB/s achievable of the 153 GB/s peak, + 20 us launch) — an estimate, NOT a measurement: no gather between distinct devices has run.
The limiter is named per row: imbalance (max / mean rank time), the path-length tail (k_tail and the late small bounces do not shrink
with the image) or the host round trips (launches per step are the same for a rank as for the whole image).

usage: tools/stripe_balance.py SCENE [--spp S] [--width W --height H] [--tris N] [--blocks 4,8,16] [--ranks 2,4,8] [--reps 2] > table"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import tinyraytracing_amd as T  # noqa: E402

SEEDS = {"back": T.SEED_BACK, "veach-mis": 0x5EED0002, "staircase": T.SEED_STAIRCASE, "soup": T.SEED_SOUP, "blob": T.SEED_BLOB}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("--spp", type=int, default=256)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--tris", type=int, default=None)
    ap.add_argument("--blocks", default="8")
    ap.add_argument("--ranks", default="2,4,8")
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("--json", default=None)
    ap.add_argument("--overlap", action="store_true", help="TRT_FLAG_OVERLAP: two sample passes in flight (what render() / tinyrt ship with)")
    a = ap.parse_args()
    W, H, spp, seed = a.width, a.height, a.spp, SEEDS[a.scene]
    s = T.Scene.named(a.scene, W, H, n=a.tris)
    r = T.Renderer(s, 0)
    out = torch.empty((H, W, 3), dtype=torch.float32, device="cuda:0")

    def timed(rows):
        p = T.make_params(W, H, spp, seed, rows=rows, flags=T.TRT_FLAG_OVERLAP if a.overlap else 0)
        r.render_into(p, out)  # warm-up (allocations, caches)
        torch.cuda.synchronize()
        best = 1e30
        st = None
        for _ in range(a.reps):
            t = time.perf_counter()
            st = r.render_into(p, out)
            torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t)
        pt = T.make_params(W, H, spp, seed, rows=rows, flags=T.TRT_FLAG_TIMING)
        stt = r.render_into(pt, out)
        torch.cuda.synchronize()
        k = {T.KERNEL_NAMES[i]: stt.kernel_ms[i] for i in range(len(T.KERNEL_NAMES))}
        launches = int(sum(stt.launches))
        return best * 1e3, st.rays, k, launches, sum(k.values())

    full_ms, full_rays, full_k, full_launches, full_ksum = timed(None)
    print(f"# {a.scene} {W}x{H} {spp} spp, {s.info['n_triangles']} triangles: whole image on one GPU {full_ms:.2f} ms/step, {full_rays / full_ms / 1e3:.0f} Mrays/s, "
          f"{full_launches} launches, kernels {full_ksum:.2f} ms (tail {full_k['tail']:.2f}), outside kernels {full_ms - full_ksum:.2f} ms")
    print("| N | row_block | rank ms: min / mean / max | max / mean | tail ms (max rank) | outside kernels ms (max rank) | launches | gather est. ms | predicted speed-up | efficiency | limiter |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    res = {"scene": a.scene, "width": W, "height": H, "spp": spp, "one_gpu_ms": full_ms, "rows": []}
    for n in [int(x) for x in a.ranks.split(",")]:
        for rb in [int(x) for x in a.blocks.split(",")]:
            ms, tails, outside, launches = [], [], [], []
            for rank in range(n):
                m, _, k, ln, ksum = timed((rb, n, rank))
                ms.append(m)
                tails.append(k["tail"])
                outside.append(m - ksum)
                launches.append(ln)
            mx, mean = max(ms), sum(ms) / n
            gather = W * H * 12.0 / n / 50e9 * 1e3 + 0.02
            sp = full_ms / (mx + gather)
            i_max = ms.index(mx)
            # which of the three costs more of the gap to perfect scaling (full_ms / n)
            gap = mx - full_ms / n
            parts = {"imbalance": mx - mean, "tail + small launches": max(mean - full_ms / n - (sum(outside) / n - (full_ms - full_ksum) / n), 0.0),
                     "host round trips": max(sum(outside) / n - (full_ms - full_ksum) / n, 0.0)}
            lim = max(parts, key=parts.get) if gap > 0 else "-"
            print(f"| {n} | {rb} | {min(ms):.2f} / {mean:.2f} / {mx:.2f} | {mx / mean:.3f} | {tails[i_max]:.2f} | {outside[i_max]:.2f} | {launches[i_max]} | {gather:.2f} | "
                  f"{sp:.2f}x | {sp / n:.2f} | {lim} ({', '.join(f'{k} {v:.2f}' for k, v in parts.items())} ms of {gap:.2f}) |", flush=True)
            res["rows"].append({"n": n, "row_block": rb, "rank_ms": ms, "tail_ms": tails, "outside_kernels_ms": outside, "launches": launches,
                                "gather_est_ms": gather, "predicted_speedup": sp})
    if a.json:
        json.dump(res, open(a.json, "w"))
    r.close()


if __name__ == "__main__":
    main()

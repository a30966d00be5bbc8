"""World-size-2 test of the image-tiling path over gloo (CPU): row-interleaved sharding, the single
gather, un-interleaving.  The per-rank renderer here is the oracle (no GPU in this container); on the
GPU box the same dist.render_distributed drives Renderer.render_into and gathers over RCCL."""
import os
import subprocess
import sys

import numpy as np

import oracle_lib as O
import tinyraytracing_amd as T
from tinyraytracing_amd import dist as D
from conftest import get_scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np, torch, torch.distributed as dist
import oracle_lib as O, tinyraytracing_amd as T
from tinyraytracing_amd import dist as D
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:{port}", rank=int(sys.argv[1]), world_size={world})
scene = T.Scene.named("back", 40, 30)
def render_fn(p):
    out, st = O.render(scene.flat, p, threads=2)
    return out, st
img, st = D.render_distributed(render_fn, 40, 30, 3, 9, dist=dist, row_block=4)
rays = torch.tensor([st.rays_camera + st.rays_shadow + st.rays_indirect], dtype=torch.int64)
dist.all_reduce(rays)
if dist.get_rank() == 0:
    np.savez({out!r}, image=img.numpy().copy(), rays=rays.numpy())
# ADVICE r02: two image heights with the same padded stripe height (8 rows per rank at heights 16 and 12) in one process:
# rank 0's cached landing buffers must follow the height (a stale key returned an image of the old height)
for h2 in (16, 12, 16):
    sc2 = T.Scene.named("back", 24, h2)
    im2, _ = D.render_distributed(lambda p: O.render(sc2.flat, p, threads=2), 24, h2, 1, 4, dist=dist, row_block=8)
    if dist.get_rank() == 0:
        ref2, _ = O.render(sc2.flat, T.make_params(24, h2, 1, 4), threads=2)
        assert tuple(im2.shape) == (h2, 24, 3), im2.shape
        assert np.array_equal(im2.numpy(), ref2)
dist.destroy_process_group()
'''


def test_shard_helpers_cover_every_row_once():
    for world in (1, 2, 3, 4, 8):
        seen = np.zeros(1080, int)
        for r in range(world):
            p = D.shard_params(1920, 1080, 1, 0, r, world)
            ys = T.rows_selected(p)
            seen[ys] += 1
            assert len(ys) <= D.max_rows(1920, 1080, world)
        assert (seen == 1).all()
    # 8 GPUs, 8-row stripes over 1080 rows: 135 stripes -> 17 or 16 per rank
    counts = [len(T.rows_selected(D.shard_params(1920, 1080, 1, 0, r, 8))) for r in range(8)]
    assert max(counts) - min(counts) <= 8 and sum(counts) == 1080


def test_assemble_inverts_sharding():
    rng = np.random.default_rng(0)
    full = rng.random((30, 40, 3)).astype(np.float32)
    for world in (2, 3):
        stripes = []
        for r in range(world):
            ys = T.rows_selected(D.shard_params(40, 30, 1, 0, r, world, row_block=4))
            pad = np.zeros((D.max_rows(40, 30, world, 4), 40, 3), np.float32)
            pad[: len(ys)] = full[ys]
            stripes.append(pad)
        assert np.array_equal(D.assemble(stripes, 40, 30, world, 4), full)


def test_two_rank_gloo_render_equals_single_process(tmp_path):
    world, port = 2, 29500 + (os.getpid() % 2000)
    out = str(tmp_path / "dist.npz")
    script = str(tmp_path / "worker.py")
    with open(script, "w") as f:
        f.write(WORKER.format(root=ROOT, port=port, world=world, out=out))
    procs = [subprocess.Popen([sys.executable, script, str(r)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    logs = [p.communicate(timeout=600)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(logs)
    got = np.load(out)
    scene = get_scene("back", 40, 30)
    ref, st = O.render(scene.flat, T.make_params(40, 30, 3, 9))
    assert np.array_equal(got["image"], ref)            # independent of the number of ranks, bit for bit
    assert int(got["rays"][0]) == st.rays


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE (how the driver may start the scaling runs): bench.py starts
    torch.distributed.run as a child before touching torch or a GPU and exits with the child's return code.  On this CPU-only
    container the two ranks come up, find no GPU and say so — which proves the launch plumbing; the GPU box runs the real thing."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, env=env)
    assert "launching -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1" in r.stderr, r.stderr[-2000:]
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        assert r.stderr.count("bench.py needs an MI355X") >= 1, r.stderr[-2000:]

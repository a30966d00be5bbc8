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
    np.savez({out!r}, image=img.numpy(), rays=rays.numpy())
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

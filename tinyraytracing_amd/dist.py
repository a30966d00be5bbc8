"""Image tiling across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm).  Every
(pixel, sample) of the image is independent given the read-only scene
(main.cpp:84-108 has no cross-pixel dependency), so the scene is replicated and
the image rows are dealt to the ranks in interleaved stripes of `row_block`
rows: rank r renders the rows y with (y // row_block) % world == r.  Because
the RNG stream is keyed by the global (pixel, sample), the assembled image is
bit-identical for every world size.  The only communication is ONE gather of
the packed float32 stripes to rank 0 over xGMI.
"""
import numpy as np

from . import make_params, rows_selected

ROW_BLOCK = 8


def shard_params(width, height, spp, seed, rank, world, row_block=ROW_BLOCK, **kw):
    """Params of the rows this rank renders."""
    return make_params(width, height, spp, seed, rows=(row_block, world, rank) if world > 1 else None, **kw)


_max_rows_cache = {}


def max_rows(width, height, world, row_block=ROW_BLOCK):
    """Rows of the most loaded rank (the gather uses equal-sized, padded buffers).  Rank 0 owns the first stripe of every
    round, so it is never behind: full rounds of `world` stripes give every rank row_block rows each, and the remainder
    goes to the lowest ranks first."""
    if world <= 1:
        return height
    key = (height, world, row_block)
    if key not in _max_rows_cache:
        rounds, rest = divmod(height, row_block * world)
        _max_rows_cache[key] = rounds * row_block + min(rest, row_block)
    return _max_rows_cache[key]


def assemble(stripes, width, height, world, row_block=ROW_BLOCK):
    """Un-interleaves per-rank packed stripes [world][rows_r, width, 3] into the full image."""
    img = np.empty((height, width, 3), np.float32)
    for r in range(world):
        p = make_params(width, height, 1, 0, rows=(row_block, world, r) if world > 1 else None)
        ys = rows_selected(p)
        img[ys] = np.asarray(stripes[r])[: len(ys)]
    return img


_row_index_cache = {}
_gather_cache = {}


def _row_indices(width, height, world, row_block, device):
    """Per rank: LongTensor of the image rows it owns (cached; used to un-interleave on the device)."""
    import torch
    key = (height, world, row_block, str(device))
    if key not in _row_index_cache:
        idx = []
        for r in range(world):
            p = make_params(width, height, 1, 0, rows=(row_block, world, r) if world > 1 else None)
            idx.append(torch.tensor(rows_selected(p), dtype=torch.long, device=device))
        _row_index_cache[key] = idx
    return _row_index_cache[key]


class StepTimes:
    """Where one rank's step goes (bench.py --gpus N): marks before the render, after it, and after the gather + un-interleave.
    Device tensors with RCCL: torch.cuda events on the current stream, read after the timed region (no host wait inside it);
    otherwise (gloo rehearsals, CPU tests) the host clock."""

    def __init__(self):
        self.marks = []  # per step: three events or three floats

    def mark(self, use_events):
        import time
        import torch
        if use_events:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        return time.perf_counter()

    def totals_ms(self):
        """(render ms, gather ms) summed over the recorded steps; call after a synchronize."""
        r = g = 0.0
        for a, b, c in self.marks:
            if isinstance(a, float):
                r += (b - a) * 1e3
                g += (c - b) * 1e3
            else:
                r += a.elapsed_time(b)
                g += b.elapsed_time(c)
        return r, g


def render_distributed(render_fn, width, height, spp, seed, dist=None, device=None, row_block=ROW_BLOCK, times=None, **kw):
    """Renders this rank's stripes with `render_fn(params) -> (array-or-tensor [rows, width, 3], stats)`
    and gathers them on rank 0.  Returns (image tensor [height, width, 3] on rank 0 else None, stats of
    this rank).  The image stays where the gather ran: on the GPU with RCCL (`device` given, backend
    nccl), on the CPU with gloo.  With more than one rank the returned tensor is rank 0's landing buffer and is
    overwritten by the next call: clone it to keep it.

    `dist` is torch.distributed (already initialised) or None for a single process.  `times`: a StepTimes that gets this step's marks."""
    import torch

    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    p = shard_params(width, height, spp, seed, rank, world, row_block, **kw)
    use_events = times is not None and device is not None and str(device).startswith("cuda") and (dist is None or dist.get_backend() == "nccl")
    m0 = times.mark(use_events) if times is not None else None
    out, stats = render_fn(p)
    t = out if torch.is_tensor(out) else torch.from_numpy(np.ascontiguousarray(out))
    if device is not None:
        t = t.to(device)
    m1 = times.mark(use_events) if times is not None else None
    if world == 1:
        if times is not None:
            times.marks.append((m0, m1, m1))
        return t.reshape(height, width, 3), stats
    pad_rows = max_rows(width, height, world, row_block)
    nrows = t.numel() // (width * 3)
    if nrows == pad_rows:
        buf = t.reshape(pad_rows, width, 3)
    else:
        buf = torch.zeros((pad_rows, width, 3), dtype=torch.float32, device=t.device)
        buf[:nrows] = t.reshape(nrows, width, 3)
    if buf.is_cuda and dist.get_backend() != "nccl":
        buf = buf.cpu()  # gloo gathers host tensors (tests, rehearsals); RCCL gathers device to device over xGMI
    # rank 0's landing buffers are kept between calls (bench.py calls this once per step: at 8 ranks a step is ~12 ms)
    gather_list = None
    if rank == 0:
        key = (height, pad_rows, width, world, row_block, str(buf.device))  # the image buffer hangs on the height, the stripes on row_block
        if key not in _gather_cache:
            _gather_cache.clear()
            _gather_cache[key] = ([torch.empty_like(buf) for _ in range(world)], torch.empty((height, width, 3), dtype=torch.float32, device=buf.device))
        gather_list = _gather_cache[key][0]
    dist.gather(buf, gather_list, dst=0)  # the single collective of the data path
    if rank != 0:
        if times is not None:
            times.marks.append((m0, m1, times.mark(use_events)))
        return None, stats
    img = _gather_cache[key][1]
    for r, idx in enumerate(_row_indices(width, height, world, row_block, buf.device)):
        img.index_copy_(0, idx, gather_list[r][: idx.numel()])
    if times is not None:
        times.marks.append((m0, m1, times.mark(use_events)))
    return img, stats

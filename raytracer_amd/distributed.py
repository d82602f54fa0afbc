"""Multi-GPU partition of the hot path: one process per GPU, `torch.distributed`.

Single frames: the image is cut into 8x8 tiles (one wavefront each); rank r owns the tiles with
tile_id % world == r (cyclic, for load balance), renders them into a packed buffer, and ONE gather over
RCCL/xGMI brings the packed buffers to rank 0, where a small kernel un-interleaves them into the row-major
image.  There is no other data-path collective: the scene is immutable and replicated.
Animations: whole frames are dealt round-robin (frame i -> rank i % world), no collective at all.

The tile index math lives here in plain Python/numpy too (`tile_layout`, `assemble_tiles_host`) so that the
N>1 path is covered by world_size-2 gloo tests on CPU; the GPU path uses rt_assemble_tiles_device.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import numpy as np

TILE = 8


def tiles_xy(width: int, height: int):
    return (width + TILE - 1) // TILE, (height + TILE - 1) // TILE


def tile_count(width: int, height: int, part: int, n_parts: int) -> int:
    """Mirror of rt_tile_count (include/rtcore.h)."""
    tx, ty = tiles_xy(width, height)
    n = tx * ty
    return (n - part + n_parts - 1) // n_parts if n > part else 0


def owned_tiles(width: int, height: int, part: int, n_parts: int) -> np.ndarray:
    tx, ty = tiles_xy(width, height)
    return np.arange(part, tx * ty, n_parts, dtype=np.int64)


def pack_tiles_host(image: np.ndarray, part: int, n_parts: int) -> np.ndarray:
    """Row-major image [h, w, 3] -> this part's packed tiles [n_owned, 64, 3] (pixels outside the image are 0).
    The layout rt_render_tiles_device writes."""
    h, w, _ = image.shape
    tx, _ = tiles_xy(w, h)
    ids = owned_tiles(w, h, part, n_parts)
    out = np.zeros((len(ids), 64, 3), dtype=image.dtype)
    for k, t in enumerate(ids):
        x0, y0 = (t % tx) * TILE, (t // tx) * TILE
        blk = image[y0:y0 + TILE, x0:x0 + TILE]
        buf = np.zeros((TILE, TILE, 3), dtype=image.dtype)
        buf[:blk.shape[0], :blk.shape[1]] = blk
        out[k] = buf.reshape(64, 3)
    return out


def assemble_tiles_host(width: int, height: int, parts: List[np.ndarray]) -> np.ndarray:
    """Packed tiles of every part (list indexed by part; each [>= n_owned, 64, 3], padding ignored) ->
    row-major image.  Mirror of rt_assemble_tiles_device."""
    n_parts = len(parts)
    tx, _ = tiles_xy(width, height)
    out = np.zeros((height, width, 3), dtype=parts[0].dtype)
    for p, buf in enumerate(parts):
        for k, t in enumerate(owned_tiles(width, height, p, n_parts)):
            x0, y0 = (t % tx) * TILE, (t // tx) * TILE
            blk = buf[k].reshape(TILE, TILE, 3)
            hh, ww = min(TILE, height - y0), min(TILE, width - x0)
            out[y0:y0 + hh, x0:x0 + ww] = blk[:hh, :ww]
    return out


def frames_for_rank(n_frames: int, rank: int, world: int) -> List[int]:
    """`animate`: frame i goes to rank i % world (animation_renderer.rs:118-121 renders them sequentially)."""
    return list(range(rank, n_frames, world))


def gather_frame(local_tiles, width: int, height: int, rank: int, world: int, dist,
                 assemble: Optional[Callable] = None):
    """The one exchange step.  `local_tiles`: torch tensor [n_owned_max * 64 * 3] f64 (padded to the largest
    part so every rank sends the same size), on the device the backend works with.  Returns, on rank 0, the
    gathered buffer [world, n_owned_max*64*3] passed through `assemble` (device kernel on GPU, numpy on CPU);
    None on the other ranks."""
    import torch

    if world == 1:
        gathered = local_tiles.unsqueeze(0)
    elif rank == 0:
        # one buffer, pre-sliced: the gather delivers every part where the un-interleave kernel reads it (no second copy)
        gathered = torch.empty((world, local_tiles.numel()), dtype=local_tiles.dtype, device=local_tiles.device)
        dist.gather(local_tiles, gather_list=[gathered[p] for p in range(world)], dst=0)
    else:
        dist.gather(local_tiles, gather_list=None, dst=0)
        return None
    return assemble(gathered) if assemble is not None else gathered


def render_frame_gpu(dev_scene, camera, options, rank: int, world: int, dist=None, want_image: bool = True, as_u8: bool = False):
    """Rank-local GPU leg + gather + assemble.  Returns the [h, w, 3] CUDA tensor on rank 0 (None elsewhere): f64, or —
    as_u8 — the u8 image of `Image::as_u8_rgb` (image.rs:84-89): every rank quantises its packed tiles on its device
    (`Color::as_scaled_values`, color.rs:55-66) BEFORE the gather, so 3 bytes per pixel travel over xGMI instead of 24
    (3.1 MB instead of 24.9 MB per rank for a 4K frame over 8 GPUs)."""
    import torch

    from . import renderer as R

    w, h = camera.width, camera.height
    stream = torch.cuda.current_stream().cuda_stream
    if world == 1:
        out = torch.empty((h, w, 3), dtype=torch.float64, device="cuda")
        dev_scene.render_device(camera, options, out.data_ptr(), stream)
        if as_u8:
            out8 = torch.empty((h, w, 3), dtype=torch.uint8, device="cuda")
            R.quantize_device(out.data_ptr(), h * w * 3, out8.data_ptr(), stream)
            return out8
        return out
    n_max = tile_count(w, h, 0, world)
    local = torch.zeros(n_max * 64 * 3, dtype=torch.float64, device="cuda")
    dev_scene.render_tiles(camera, options, rank, world, local.data_ptr(), stream)
    if as_u8:
        local8 = torch.empty(n_max * 64 * 3, dtype=torch.uint8, device="cuda")
        R.quantize_device(local.data_ptr(), local.numel(), local8.data_ptr(), stream)
        local = local8
    if not want_image:
        return None

    def assemble(gathered):
        out = torch.empty((h, w, 3), dtype=gathered.dtype, device="cuda")
        (R.assemble_tiles_u8 if as_u8 else R.assemble_tiles)(w, h, world, n_max, gathered.data_ptr(), out.data_ptr(), stream)
        return out

    return gather_frame(local, w, h, rank, world, dist, assemble)

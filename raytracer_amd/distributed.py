"""Multi-GPU frame sharding: one process per GPU (torch.distributed; backend "nccl" is RCCL on ROCm, "gloo" on CPU).

The reference is single-process.  Here a frame is cut into 32x8-pixel tiles dealt round-robin to the ranks
(tile t belongs to rank t % ranks: neighbouring tiles cost alike, so the deal balances the load), the scene is replicated
on every GPU, each rank renders its tiles, packs them densely and rank 0 gathers the packed buffers -- the only
exchange step -- and unpacks them into the full frame.

`tile_pixels` restates the device's tile -> pixel map (rt_kernels.hip: tilePixel) in numpy so that the sharding logic can
be exercised without a GPU.
"""
import numpy as np

TILE_W, TILE_H = 32, 8


def tiles_count(width, height, rank, ranks):
    total = ((width + TILE_W - 1) // TILE_W) * ((height + TILE_H - 1) // TILE_H)
    return (total - rank + ranks - 1) // ranks if rank < total else 0


def tile_pixels(width, height, rank, ranks):
    """For local pixel slot lp = 0 .. tiles_count*256-1 of `rank`: the image pixel index y*width+x, or -1 for the padding
    slots of partial edge tiles.  Inside a tile, slots run in four 8x8 blocks (one wavefront each)."""
    n = tiles_count(width, height, rank, ranks) * 256
    lp = np.arange(n, dtype=np.int64)
    tiles_x = (width + TILE_W - 1) // TILE_W
    gt = (lp >> 8) * ranks + rank
    tx, ty = gt % tiles_x, gt // tiles_x
    within = lp & 255
    sub, lane = within >> 6, within & 63
    x = tx * TILE_W + sub * 8 + (lane & 7)
    y = ty * TILE_H + (lane >> 3)
    ok = (x < width) & (y < height)
    return np.where(ok, y * width + x, -1)


def pack_tiles(frame, rank, ranks):
    """frame: [H, W, 3] array/tensor on the host -> dense [tiles*256, 3] of this rank's pixels (padding slots zero)."""
    import torch
    h, w = frame.shape[:2]
    pix = torch.from_numpy(tile_pixels(w, h, rank, ranks))
    flat = torch.as_tensor(frame).reshape(-1, 3)
    out = torch.zeros(len(pix), 3, dtype=flat.dtype)
    ok = pix >= 0
    out[ok] = flat[pix[ok]]
    return out


def unpack_tiles(packed, frame, rank, ranks):
    import torch
    h, w = frame.shape[:2]
    pix = torch.from_numpy(tile_pixels(w, h, rank, ranks))
    ok = pix >= 0
    frame.reshape(-1, 3)[pix[ok]] = torch.as_tensor(packed).reshape(-1, 3)[ok]


def gather_packed(packed, sizes, rank, world, dst=0, bufs=None):
    """Gathers every rank's packed tile buffer (flat float tensor, sizes[r] elements for rank r) on `dst`.
    Equal sizes go through one gather collective; ragged sizes (tile count not divisible by the ranks) through
    point-to-point sends into the root.  Returns the list of buffers on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    if rank == dst and bufs is None:
        bufs = [torch.empty(sizes[r], dtype=packed.dtype, device=packed.device) for r in range(world)]
    if world == 1:
        bufs[0].copy_(packed)
        return bufs
    if len(set(sizes)) == 1:
        dist.gather(packed, bufs, dst=dst)
    elif rank == dst:
        reqs = [dist.irecv(bufs[r], src=r) for r in range(world) if r != dst]
        bufs[dst].copy_(packed)
        for q in reqs:
            q.wait()
    else:
        dist.send(packed, dst=dst)
    return bufs

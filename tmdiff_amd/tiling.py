"""Tile helpers for large scenes (SURVEY 5 "long context", 8f N3; reference data/LRHR_dataset.py:17-53).

``invPatch`` / ``patch_16`` / ``unpatch_16`` keep the reference's names and row-major tile order; ``split_tiles`` /
``merge_tiles`` generalise them to any grid.  ``sample_tiled`` is BASELINE config 5: a large scene is cut into
independent tiles that are sharded over the ranks of the process group (no collective during sampling, one
all-gather at the end).  WavBEST is fully convolutional with zero padding, so tiled output differs from
full-frame output within 3x3x3-receptive-field distance of the tile seams -- same as the reference's tiling.
"""
import torch

from . import dist as tdist


def split_tiles(img, th, tw):
    """[B, C, H, W] -> [B * (H/th) * (W/tw), C, th, tw], tiles in row-major order per image."""
    b, c, h, w = img.shape
    assert h % th == 0 and w % tw == 0, "scene must be a whole number of tiles"
    t = img.reshape(b, c, h // th, th, w // tw, tw).permute(0, 2, 4, 1, 3, 5)
    return t.reshape(-1, c, th, tw).contiguous()


def merge_tiles(tiles, rows, cols):
    """inverse of split_tiles: [B*rows*cols, C, th, tw] -> [B, C, rows*th, cols*tw]"""
    n, c, th, tw = tiles.shape
    b = n // (rows * cols)
    t = tiles.reshape(b, rows, cols, c, th, tw).permute(0, 3, 1, 4, 2, 5)
    return t.reshape(b, c, rows * th, cols * tw).contiguous()


def invPatch(img_MS):
    """(1, c, h, w) -> (4, c, h/2, w/2): the four quadrants (reference :17-25)."""
    return split_tiles(img_MS, img_MS.shape[2] // 2, img_MS.shape[3] // 2)


def patch_16(img_MSs):
    """(16, c, h, w) -> (c, 4h, 4w) (reference :28-37)."""
    return merge_tiles(torch.as_tensor(img_MSs), 4, 4)[0]


def unpatch_16(patch):
    """(c, 4h, 4w) -> (16, c, h, w) (reference :40-53)."""
    p = torch.as_tensor(patch)
    return split_tiles(p[None], p.shape[1] // 4, p.shape[2] // 4)


@torch.no_grad()
def sample_tiled(diffusion, scene, prompt, tile=64, method="dpmsolver", steps=20, max_batch=32):
    """Fuse a large scene tile by tile.  ``scene`` = {'MS': [B,C,H,W], 'PAN': [B,1,H,W]} (Res optional).
    Tiles are sharded over the ranks; each rank samples its share in batches of ``max_batch`` and the fused
    tiles are gathered and stitched on every rank."""
    ms, pan = scene["MS"], scene["PAN"]
    rows, cols = ms.shape[2] // tile, ms.shape[3] // tile
    tiles = {"MS": split_tiles(ms, tile, tile), "PAN": split_tiles(pan, tile, tile)}
    tiles["Res"] = split_tiles(scene["Res"], tile, tile) if "Res" in scene else torch.zeros_like(tiles["MS"])
    n = tiles["MS"].shape[0]
    world = tdist.dist.get_world_size() if tdist.dist.is_initialized() else 1
    rank = tdist.dist.get_rank() if tdist.dist.is_initialized() else 0
    pad = (-n) % world                                   # equal shares so that the final all-gather is regular
    if pad:
        tiles = {k: torch.cat([v, v[:pad]]) for k, v in tiles.items()}
    mine = tdist.shard_batch(tiles, rank, world) if world > 1 else tiles
    outs = []
    for lo in range(0, mine["MS"].shape[0], max_batch):
        part = {k: v[lo:lo + max_batch].contiguous() for k, v in mine.items()}
        outs.append(diffusion.sample(part, prompt, method=method, **({"steps": steps} if method == "dpmsolver" else {})))
    fused = tdist.gather_images(torch.cat(outs))[:n]
    return merge_tiles(fused, rows, cols)

"""Host-side tile bookkeeping for the multi-GPU image split.

The image is cut into 8x8-pixel tiles numbered along a blocked curve (8x8 blocks of tiles,
blocks row-major, tiles row-major inside a block with row iy rotated by iy: tile_order());
rank r of N renders positions r, r+N, ... of that order into a compact buffer [numLocalTiles][64] float4 (numLocalTiles = ceil(numTiles / N), the
same on every rank so one gather of equal-sized buffers suffices).  Rank 0 receives
[N][numLocalTiles][64][4] and un-permutes it (on the GPU: srtResolveTiles; the numpy
version here is the host mirror used by the CPU multi-process tests)."""
import numpy as np

TILE_W = TILE_H = 8
TILE_PIXELS = 64
TILE_BLOCK = 8  # SRT_TILE_BLOCK in include/srt_hip.h


def tiles_xy(width, height):
    return (width + TILE_W - 1) // TILE_W, (height + TILE_H - 1) // TILE_H


def num_tiles(width, height):
    tx, ty = tiles_xy(width, height)
    return tx * ty


def num_local_tiles(width, height, nranks):
    return (num_tiles(width, height) + nranks - 1) // nranks


def tile_order(width, height, block=TILE_BLOCK):
    """Position of every row-major tile id (ty*tilesX + tx) along the blocked curve
    (csrc/srt_device.h srtOrderFromTile)."""
    tx, ty = tiles_xy(width, height)
    y, x = np.divmod(np.arange(tx * ty), tx)
    by, iy = np.divmod(y, block)
    bx, ix = np.divmod(x, block)
    bh = np.minimum(block, ty - by * block)
    bw = np.minimum(block, tx - bx * block)
    return by * block * tx + bx * block * bh + iy * bw + (ix + iy) % bw


def owner(position, nranks):
    """(rank, local index) of a position of the tile order."""
    return position % nranks, position // nranks


def untile(gathered, width, height, nranks):
    """gathered: array (nranks, numLocalTiles, 64, C) -> image (height, width, C)."""
    tx, ty = tiles_xy(width, height)
    nloc = num_local_tiles(width, height, nranks)
    g = np.asarray(gathered).reshape(nranks, nloc, TILE_H, TILE_W, -1)
    pos = tile_order(width, height)
    # tile-major -> [ty][tx][8][8][C] -> image
    t = g[pos % nranks, pos // nranks].reshape(ty, tx, TILE_H, TILE_W, -1)
    img = t.transpose(0, 2, 1, 3, 4).reshape(ty * TILE_H, tx * TILE_W, -1)
    return img[:height, :width]


def tile_image(image, nranks):
    """Inverse of untile for tests: image (H, W, C) -> (nranks, numLocalTiles, 64, C), zero padded."""
    h, w, c = image.shape
    tx, ty = tiles_xy(w, h)
    nloc = num_local_tiles(w, h, nranks)
    pad = np.zeros((ty * TILE_H, tx * TILE_W, c), image.dtype)
    pad[:h, :w] = image
    t = pad.reshape(ty, TILE_H, tx, TILE_W, c).transpose(0, 2, 1, 3, 4).reshape(tx * ty, TILE_PIXELS, c)
    out = np.zeros((nranks, nloc, TILE_PIXELS, c), image.dtype)
    pos = tile_order(w, h)
    out[pos % nranks, pos // nranks] = t
    return out

"""The N>1 path on CPU: world_size-2 gloo processes shard the tiles of one image (interleaved,
SURVEY 8e), each fills its compact tile buffer, one gather of equal-sized buffers to rank 0,
rank 0 un-permutes.  The per-tile values come from the CPU oracle's render of the full image, so
the test checks the partition / gather / un-permute logic bench.py uses, not the kernel."""
import importlib
import os
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, w, h, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tiles = importlib.import_module("sexy-raytracer_amd.tiles")
    bench = importlib.import_module("bench")
    rng = np.random.default_rng(42)
    image = rng.random((h, w, 4)).astype(np.float32)  # every rank can regenerate the reference image
    full = tiles.tile_image(image, world)              # (world, nloc, 64, 4)
    local = torch.from_numpy(full[rank].copy())        # what this rank's kernel would have written
    gathered = bench.gather_tiles(local, rank, world)
    dist.barrier()
    if rank == 0:
        out = tiles.untile(gathered.numpy(), w, h, world)
        q.put(bool(np.array_equal(out, image)))
    dist.destroy_process_group()


def test_two_rank_tile_gather_gloo():
    ctx = mp.get_context("spawn")
    for port, (w, h) in ((29611, (426, 240)), (29612, (70, 45))):
        q = ctx.Queue()
        procs = [ctx.Process(target=_worker, args=(r, 2, port, w, h, q)) for r in range(2)]
        for p in procs:
            p.start()
        ok = q.get(timeout=120)
        for p in procs:
            p.join(60)
            assert p.exitcode == 0
        assert ok


def test_tile_ownership_is_balanced():
    sys.path.insert(0, ROOT)
    tiles = importlib.import_module("sexy-raytracer_amd.tiles")
    n = tiles.num_tiles(1920, 1080)
    for world in (1, 2, 4, 8):
        counts = np.bincount(np.arange(n) % world, minlength=world)
        assert counts.max() - counts.min() <= 1
        assert tiles.num_local_tiles(1920, 1080, world) == counts.max()

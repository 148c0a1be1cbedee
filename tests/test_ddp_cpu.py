"""CPU, world_size 2, gloo: the data-parallel helper's segment-wise all-reduce + parameter broadcast.
The engine itself needs a GPU; here a stand-in model object with the same flat-buffer interface checks the
collective logic (segment ranges cover the flat gradient exactly once; averaged gradients; broadcast)."""
import os

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pixelwiseregression_amd import PixelwiseRegression
        from pixelwiseregression_amd.ddp import DataParallel
        torch.manual_seed(100 + rank)                      # different initial weights per rank
        m = PixelwiseRegression(4, stage=2, label_size=16, features=32, level=1, norm_method="batch")
        ddp = DataParallel(m)
        gathered = [torch.empty_like(m.flat_parameters()) for _ in range(world)]
        dist.all_gather(gathered, m.flat_parameters())
        same_params = all(torch.equal(gathered[0], g) for g in gathered)
        rngs = m.segment_ranges()
        cover = torch.zeros(m.flat_parameters().numel())
        for b, e in rngs:
            cover[b:e] += 1
        g = m.flat_grad()
        g.copy_(torch.arange(g.numel(), dtype=torch.float32) * (rank + 1))
        for seg in range(len(rngs)):
            ddp.segment_done(m, seg, len(rngs))
        ddp.finish(m)
        expect = torch.arange(g.numel(), dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        ret[rank] = (same_params, bool((cover == 1).all()), torch.allclose(g, expect, rtol=1e-6), len(rngs))
    finally:
        dist.destroy_process_group()


def test_ddp_segments_world2():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + os.getpid() % 1000
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    for r in range(world):
        same, cover, ok, nseg = ret[r]
        assert same, "parameters not broadcast"
        assert cover, "segment ranges must tile the flat gradient exactly once"
        assert ok, "averaged gradient wrong"
        assert nseg == 3

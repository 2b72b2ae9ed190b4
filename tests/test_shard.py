"""Multi-GPU plumbing on CPU: world_size-2 gloo process group, scatter -> (stand-in work) -> gather."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from audio_denoising_amd.shard import gather_rows, scatter_rows, shard_range


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 256, 2048, 8192):
        for world in (1, 2, 3, 8):
            spans = [shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
    assert shard_range(2048, 8, 3) == (768, 1024)          # BASELINE config 4: 256 streams per GPU
    with pytest.raises(ValueError):
        shard_range(8, 2, 2)


def _worker(rank, world, port, total, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        full = torch.arange(total * 6, dtype=torch.float32).reshape(total, 6) if rank == 0 else None
        mine = scatter_rows(full, total, (6,), torch.float32, "cpu")
        lo, hi = shard_range(total, world, rank)
        assert mine.shape == (hi - lo, 6)
        assert torch.equal(mine, torch.arange(total * 6, dtype=torch.float32).reshape(total, 6)[lo:hi])
        back = gather_rows(mine * 2.0, total)
        if rank == 0:
            q.put(bool(torch.equal(back, full * 2.0)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total", [5, 8])
def test_scatter_gather_world2_gloo(total):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True

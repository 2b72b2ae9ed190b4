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


class _OneHopPipe:
    """Stand-in for the software-pipelined hop: launch i reads in_buf NOW (its front half) and completes hop i-1 into the `out` of hop i-1
    (its Griffin-Lim half) -- the data hazards the double-buffered ingress loop of bench.py has to respect."""

    def __init__(self):
        self.pending = None

    def submit(self, frames, out):
        if self.pending is not None:
            f, o = self.pending
            o.copy_(f * 2.0)
        self.pending = (frames.clone(), out)          # (the front half has consumed `frames` when submit returns: launch order)

    submit_group = submit                             # (a hop group: the same hazards with G hops per launch -- frames / out are (G, B, n))

    def flush(self):
        if self.pending is not None:
            f, o = self.pending
            o.copy_(f * 2.0)
            self.pending = None


def _ingress_worker(rank, world, port, total, steps, q, group=1):
    """bench.ingress_variant's order of operations (there: scatter/gather on a second HIP stream with events between the same steps; here the
    calls themselves, in that order): scatter(0); per step i: hop(i) -> scatter(i+1) -> gather(i-1); flush; gather(n-1).  Every step carries
    different audio; in_buf / out_buf are double-buffered, `out=` forms, uneven shards."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        lo, hi = shard_range(total, world, rank)
        B = hi - lo
        G = group           # hops per launch (the headline's schedule: 4); a unit of audio is (G, total, 4)
        audio = [torch.arange(G * total * 4, dtype=torch.float32).reshape(G, total, 4) + 1000.0 * i for i in range(steps)] if rank == 0 else [None] * steps
        big_out = torch.zeros(G, total, 4) if rank == 0 else None
        in_buf = [torch.empty(G, B, 4) for _ in range(2)]
        out_buf = [torch.zeros(G, B, 4) for _ in range(2)]
        pipe = _OneHopPipe()
        ok = True

        def scatter(i, s):
            for h in range(G):
                scatter_rows(None if audio[i] is None else audio[i][h], total, (4,), torch.float32, "cpu", out=in_buf[s][h])

        def gather(s):
            for h in range(G):
                gather_rows(out_buf[s][h], total, out=None if big_out is None else big_out[h])
        scatter(0, 0)
        for i in range(steps):
            s = i & 1
            pipe.submit_group(in_buf[s], out_buf[s])              # launch i: front halves of unit i, completes unit i-1 into out_buf[(i-1)&1]
            if i + 1 < steps:
                scatter(i + 1, s ^ 1)
            if i >= 1:
                gather(s ^ 1)
                if rank == 0:
                    ok = ok and bool(torch.equal(big_out, audio[i - 1] * 2.0))
        pipe.flush()
        gather((steps - 1) & 1)
        if rank == 0:
            ok = ok and bool(torch.equal(big_out, audio[steps - 1] * 2.0))
            q.put(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("total,group", [(5, 1), (7, 1), (5, 4), (7, 4)])
def test_double_buffered_ingress_order_round_trips_on_uneven_shards(total, group):
    """the scatter -> hop -> gather loop of bench.py's `ingress_variant`, with its double buffering and the one-hop delay of the
    pipelined hop, on two gloo ranks with uneven shards (3 + 2, 4 + 3 rows): every step's audio must come back doubled, step by step -- one hop
    per launch, and the headline's schedule since round 4: units of four hops per launch (bench.py: ingress_variant(group=4))."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ingress_worker, args=(r, 2, port, total, 6, q, group)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True

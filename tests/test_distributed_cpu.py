"""world_size-2 gloo test (CPU) of the N>1 path: disjoint scene shards per rank and the
all-gather of rendered batches (one collective per dtype buffer)."""
import os

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from stillleben_amd import parallel


def _fake_batch(seeds, H=6, W=8):
    """Stand-in for a rendered batch: content is a pure function of the scene seeds."""
    s = torch.tensor(seeds, dtype=torch.int64)
    rgb = (s[:, None, None, None] % 251).to(torch.uint8).expand(-1, H, W, 4).contiguous()
    coord = (s[:, None, None, None].float() * 0.5).expand(-1, H, W, 4).contiguous()
    inst = (s[:, None, None, None] % 7).to(torch.int16).expand(-1, H, W, 1).contiguous()
    return [rgb, coord, inst]


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_items, batch = 3, 4
    seeds = parallel.shard_seeds(rank, world, n_items, batch)
    gather = parallel.BatchGatherer(dist, world)
    ok = True
    for k in range(n_items):
        mine = _fake_batch(seeds[k])
        allb = gather(mine)
        for r in range(world):
            exp = _fake_batch(parallel.shard_seeds(r, world, n_items, batch)[k])
            for a, b in zip(allb, exp):
                ok = ok and torch.equal(a[r], b)
    # overlapped form used by bench.py: asynchronous collectives into a 2-deep staging ring; the
    # third call reuses the first set, the views of the last `depth` calls stay valid
    agather = parallel.BatchGatherer(dist, world, depth=2)
    keep = []
    for k in range(n_items):
        views, works = agather(_fake_batch(seeds[k]), async_op=True)
        for w in works:
            w.wait()
        keep.append(views)
    ok = ok and keep[0][0].data_ptr() == keep[2][0].data_ptr() and keep[0][0].data_ptr() != keep[1][0].data_ptr()
    for k in (1, 2):
        for r in range(world):
            exp = _fake_batch(parallel.shard_seeds(r, world, n_items, batch)[k])
            for a, b in zip(keep[k], exp):
                ok = ok and torch.equal(a[r], b)
    # streamed form (bench.py --gather compact | full): a rank's whole chunk goes through the exchange in pieces of 3 scenes (the last
    # one ragged), double-buffered staging; every piece arrives complete and in rank order, the byte count is what was handed over
    chunk = _fake_batch([1000 * rank + i for i in range(8)])
    stream = parallel.ChunkedGatherer(parallel.BatchGatherer(dist, world, depth=2), piece=3)
    got = {}

    def on_piece(first, views, works):
        for w in works:
            w.wait()
        got[first] = [v.clone() for v in views]        # (valid until two further pieces of this shape have been issued)

    works = stream(chunk, on_piece=on_piece)
    ok = ok and sorted(got) == [0, 3, 6] and stream.pieces == 3 and len(works) == 3 * len(chunk)
    ok = ok and stream.bytes_sent == sum(t.numel() * t.element_size() for t in chunk)
    for first, views in got.items():
        for r in range(world):
            exp = _fake_batch([1000 * r + i for i in range(first, min(first + 3, 8))])
            for a, b in zip(views, exp):
                ok = ok and torch.equal(a[r], b)
    # max-over-ranks timing reduction used by bench.py
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    ok = ok and float(t) == float(world)
    dist.barrier()
    q.put((rank, ok, [s for item in seeds for s in item]))
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29531 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok, _ in res)
    seeds = [set(s) for _, _, s in sorted(res)]
    assert not (seeds[0] & seeds[1])            # disjoint shards
    assert len(seeds[0]) == len(seeds[1]) == 12


def test_static_partition():
    parts = [parallel.shard_scenes(512, r, 8) for r in range(8)]
    assert all(len(p) == 64 for p in parts)     # BASELINE config 3: 512 scenes, 64 per GPU
    assert sorted(sum(parts, [])) == list(range(512))

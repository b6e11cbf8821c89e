"""world_size-2 `gloo` test of the multi-process plumbing used by bench.py --gpus N (CPU only):
image sharding is a partition, the timed region returns MAX over ranks, gathered per-image results
come back in image order."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import time
    from yolo_for_turbines_amd import dist as ydist
    d = ydist.init(backend="gloo")
    assert d is not None and d.get_world_size() == world
    lo, hi = ydist.shard_range(128, rank, world)                    # Config 5: 128 images over the ranks
    sleep = 0.02 * (rank + 1)                                       # rank 1 is the slow one
    elapsed = ydist.timed_steps(lambda: time.sleep(sleep), steps=5, warmup=1, dist=d)
    counts = torch.arange(lo, hi, dtype=torch.int32)                # stand-in for per-image kept counts
    allc = ydist.gather_counts(counts, d)
    q.put((rank, lo, hi, elapsed, allc.tolist()))
    d.barrier()
    d.destroy_process_group()


def test_two_rank_gloo_sharding_and_timing():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, t0, c0), (r1, lo1, hi1, t1, c1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 64, 64, 128)
    assert t0 == pytest.approx(t1) and t0 >= 5 * 0.04 * 0.9          # MAX over ranks = the slow rank's time
    assert c0 == c1 == list(range(128))


def test_shard_range_is_a_partition():
    from yolo_for_turbines_amd.dist import shard_range
    for n in (0, 1, 7, 32, 128, 129):
        for world in (1, 2, 3, 8):
            cover = []
            for r in range(world):
                lo, hi = shard_range(n, r, world)
                assert 0 <= lo <= hi <= n
                cover += list(range(lo, hi))
            assert cover == list(range(n))


def _bucket_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from yolo_for_turbines_amd import dist as ydist
    d = ydist.init(backend="gloo")
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in [(8, 4, 3, 3), (8,), (8,), (16, 8, 1, 1), (16,), (5,)]]
    params[5].requires_grad_(False)                                   # frozen parameters get no slot
    gb = ydist.GradBuckets(params, d, bucket_mb=4 * 300 / 1024 / 1024)   # ~300 floats per bucket -> several buckets
    assert len(gb.buckets) >= 2 and id(params[5]) not in gb.slot
    for step in range(2):                                             # buckets are reusable across steps
        gb.begin()
        for i, p in enumerate(params[:5]):
            g = gb.view(p)
            assert g.shape == p.shape
            g.copy_(torch.full(p.shape, float((rank + 1) * (i + 1) * (step + 1))))
            gb.ready(p)
        gb.finish()
        vals = [float(gb.view(p).mean()) for p in params[:5]]
        q.put((rank, step, vals))
    d.barrier()
    d.destroy_process_group()


def test_gradient_buckets_average_over_ranks():
    """DP parity definition of SURVEY §8e: the exchanged gradient is the arithmetic mean of the per-rank
    gradients (rank r contributes (r+1)*k -> mean 1.5*k for two ranks), for every bucket, every step."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world * 2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, step, vals in res:
        assert vals == pytest.approx([1.5 * (i + 1) * (step + 1) for i in range(5)])

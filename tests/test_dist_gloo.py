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


# ------------------------------------------------------------------ bench.py --gpus N starts its own ranks
def test_bench_self_launch_command_line():
    """`python bench.py --gpus N` outside torchrun must start `python -m torch.distributed.run --nproc-per-node N ...
    bench.py <same flags>` as a CHILD process (bench contract); --dry-launch prints that command instead of running it."""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "7", "--warmup", "2", "--dry-launch"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    argv = json.loads(out.stdout.strip().splitlines()[-1])["argv"]
    assert argv[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert "--nproc-per-node=4" in argv and argv[argv.index("--master-addr") + 1] == "127.0.0.1"
    k = argv.index(os.path.join(ROOT, "bench.py"))
    assert argv[k + 1:] == ["--gpus", "4", "--steps", "7", "--warmup", "2"]          # the caller's flags, minus --dry-launch
    # under torchrun (WORLD_SIZE set) the script is a rank, not a launcher: on this CPU box it stops at the GPU check
    env2 = dict(env, WORLD_SIZE="4", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dry-launch"], env=env2,
                          capture_output=True, text=True, timeout=300)
    if not torch.cuda.is_available():
        assert out2.returncode != 0 and "needs an MI355X" in out2.stderr


def test_bench_self_launch_relays_one_json_line(monkeypatch, capsys):
    """The parent relays exactly rank 0's JSON line (library banners on the child's stdout are dropped) and the exit code."""
    import argparse
    import json
    sys.path.insert(0, ROOT)
    import bench
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.torch.cuda, "device_count", lambda: 8)
    line = json.dumps({"metric": "images/sec at 416x416 (fwd)", "n_gpus": 2, "value": 1.0})
    monkeypatch.setattr(bench, "launch_argv", lambda n, port, rest: [sys.executable, "-c", f"print('RCCL version banner'); print({line!r})"])
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    assert bench.self_launch(argparse.Namespace(gpus=2, dry_launch=False)) == 0
    assert capsys.readouterr().out.strip() == line
    monkeypatch.setattr(bench, "launch_argv", lambda n, port, rest: [sys.executable, "-c", "import sys; print('{}'); sys.exit(3)"])
    with pytest.raises(SystemExit) as e:
        bench.self_launch(argparse.Namespace(gpus=2, dry_launch=False))
    assert e.value.code == 3


def test_gradient_buckets_do_not_alias_a_live_grad():
    """AccumulateGrad keeps the bucket VIEW as p.grad. If that tensor is still alive at the next backward
    (zero_grad(set_to_none=False), micro-batch accumulation) the kernels would overwrite it and autograd would add the
    bucket to itself: begin() must give such a parameter its own copy first."""
    from yolo_for_turbines_amd.dist import GradBuckets
    params = [torch.nn.Parameter(torch.zeros(s)) for s in [(4, 3), (7,), (2, 2, 2)]]
    gb = GradBuckets(params, None, bucket_mb=1.0)
    gb.begin(params)
    for i, p in enumerate(params):
        gb.view(p).fill_(float(i + 1))
        gb.ready(p)
    gb.finish()
    for p in params:
        p.grad = gb.view(p)                       # what AccumulateGrad does with the tensor the backward returned
    params[1].grad = torch.full((7,), 9.0)        # a gradient that does NOT live in the bucket stays untouched
    keep = params[1].grad
    gb.begin(params)                              # second backward starts
    for i, p in enumerate(params):
        gb.view(p).fill_(100.0)                   # the kernels overwrite the bucket
    assert params[1].grad is keep
    for i in (0, 2):
        assert params[i].grad.data_ptr() != gb.view(params[i]).data_ptr()
        assert float(params[i].grad.mean()) == float(i + 1)          # old value preserved -> `p.grad += new` is a real sum

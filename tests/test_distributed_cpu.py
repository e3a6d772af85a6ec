"""CPU tests of the data-parallel path: world_size-2 gloo runs of the bucketed gradient all-reduce
over the flat gradient arena (speech_amd.train.BucketedAllReduce + FlatAdam's arena layout).
The optimizer *step* is a HIP kernel and is covered by the GPU tests; here the exchange logic is
checked: buckets tile the arena from its end, every rank ends with the same summed gradients,
replicas start identical, hooks fire once per parameter."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _tiny_model():
    torch.manual_seed(0)
    return torch.nn.Sequential(torch.nn.Linear(64, 300), torch.nn.Tanh(), torch.nn.Linear(300, 300),
                               torch.nn.Tanh(), torch.nn.Linear(300, 7))


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from speech_amd.train import BucketedAllReduce, FlatAdam
    model = _tiny_model()
    if rank == 1:  # replicas must be made identical by the initial broadcast
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    opt = FlatAdam(model.parameters(), lr=1e-3)
    dp = BucketedAllReduce(opt, bucket_mb=0.05)  # ~13k floats per bucket -> several buckets
    res = {"buckets": [tuple(b) for b in dp.buckets], "total": opt.total,
           "start": opt.flat.clone()}
    torch.manual_seed(100 + rank)
    x = torch.randn(16, 64)
    opt.zero_grad()
    model(x).pow(2).mean().backward()
    local = None
    # what this rank alone would have produced
    ref_model = _tiny_model()
    ref_model(x).pow(2).mean().backward()
    local = torch.cat([torch.nn.functional.pad(p.grad.reshape(-1), (0, (-p.numel()) % 4))
                       for p in ref_model.parameters()])
    dp.finish()
    res["grad"] = opt.flat_grad.clone()
    res["local"] = local
    res["views_ok"] = all(p.grad.data_ptr() == opt.flat_grad.data_ptr() + 4 * off
                          for p, off in zip(opt.plist, opt.offsets))
    # second step: hooks re-arm
    opt.zero_grad()
    model(x).pow(2).mean().backward()
    dp.finish()
    res["grad2"] = opt.flat_grad.clone()
    torch.save(res, os.path.join(out, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_bucketed_allreduce_world2(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / "rank0.pt", weights_only=False)
    r1 = torch.load(tmp_path / "rank1.pt", weights_only=False)
    # replicas identical after the initial broadcast from rank 0
    assert torch.equal(r0["start"], r1["start"])
    # buckets tile [0, total) contiguously, last parameters first, every slice 16-byte aligned
    b = r0["buckets"]
    assert len(b) >= 2
    assert b[0][1] == r0["total"] and b[-1][0] == 0
    for (lo, hi, n), (lo2, hi2, _) in zip(b[:-1], b[1:]):
        assert hi2 == lo and lo < hi and n >= 1 and lo % 4 == 0
    assert sum(x[2] for x in b) == 6
    # all-reduce = sum of the per-rank gradients, identical on both ranks
    assert torch.equal(r0["grad"], r1["grad"])
    assert torch.allclose(r0["grad"], r0["local"] + r1["local"], rtol=1e-5, atol=1e-7)
    assert r0["views_ok"] and r1["views_ok"]
    assert torch.allclose(r0["grad2"], r0["grad"], rtol=1e-6, atol=1e-8)


def test_lr_schedule_matches_reference_formula():
    """train.py:17-24."""
    from speech_amd.train import lr_lambda_linear_warmup as f
    assert f(0, 3500, 100000) == 0.0
    assert f(1750, 3500, 100000) == pytest.approx(0.5)
    assert f(3500, 3500, 100000) == pytest.approx(1.0)
    assert f(100000, 3500, 100000) == 0.0
    assert f(51750, 3500, 100000) == pytest.approx(0.5)
    assert f(200000, 3500, 100000) == 0.0


_DYING_RANK = """
import os, sys, time
import torch, torch.distributed as dist
dist.init_process_group("gloo")
if dist.get_rank() == 1:
    os._exit(3)                      # dies before the collective: rank 0 would wait in it for the backend's timeout
t = torch.ones(4)
dist.all_reduce(t)
time.sleep(120)
"""


def test_launcher_ends_the_survivors_when_a_rank_dies(tmp_path):
    """bench.spawn_ranks: one of two gloo ranks exits non-zero before the first collective; the launcher must notice,
    terminate the rank that is blocked in the all-reduce and return non-zero within seconds (it used to wait for the
    ranks one after another with no deadline)."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    script = tmp_path / "rank.py"
    script.write_text(_DYING_RANK)
    t0 = time.time()
    rc = bench.spawn_ranks(2, deadline_s=60, log_dir=str(tmp_path), cmd=[sys.executable, str(script)])
    dt = time.time() - t0
    assert rc == 3 and dt < 30, (rc, dt)
    assert (tmp_path / "bench_rank0.stderr").exists() and (tmp_path / "bench_rank1.stderr").exists()


def test_launcher_deadline(tmp_path):
    """... and a set of ranks that simply never finishes is ended at the deadline with exit code 124."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    script = tmp_path / "sleep.py"
    script.write_text("import time\ntime.sleep(120)\n")
    t0 = time.time()
    rc = bench.spawn_ranks(2, deadline_s=2, log_dir=str(tmp_path), cmd=[sys.executable, str(script)])
    assert rc == 124 and time.time() - t0 < 20


def test_shard_sampler_covers_every_item_once():
    from speech_amd.train import ShardSampler
    for n in (0, 1, 7, 16):
        for world in (1, 2, 3):
            for shuffle in (False, True):
                got = sorted(i for r in range(world) for i in ShardSampler(n, r, world, shuffle))
                assert got == list(range(n))
                assert sum(len(ShardSampler(n, r, world, shuffle)) for r in range(world)) == n

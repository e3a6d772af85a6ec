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

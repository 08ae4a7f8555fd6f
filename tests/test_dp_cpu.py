"""Data-parallel gradient averaging (maskplanner_amd/dp.py) with world_size 2 on the gloo backend (CPU).

The N>1 path of bench.py is: shard the batch by sample, backward locally, bucketed all-reduce of the flat gradient
buffers overlapped with backward, average, optimizer step.  Here the same reducer runs on two CPU processes and
must reproduce the single-process gradient of the mean loss over the global batch, and keep replicas identical
after optimizer steps.
"""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _model():
    torch.manual_seed(7)
    return torch.nn.Sequential(torch.nn.Linear(6, 32), torch.nn.ReLU(), torch.nn.Linear(32, 32), torch.nn.ReLU(),
                               torch.nn.Linear(32, 5))


def _data():
    g = torch.Generator().manual_seed(11)
    return torch.randn(8, 6, generator=g), torch.randn(8, 5, generator=g)


def _worker(rank, world, port, bucket_bytes, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from maskplanner_amd import dp
    r, _, w = dp.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    model = _model()
    red = dp.BucketedGradAllReduce(model.parameters(), bucket_bytes=bucket_bytes)
    opt = torch.optim.Adam(model.parameters(), lr=1e-2)
    x, y = _data()
    shard = slice(rank * 4, rank * 4 + 4)
    grads = None
    for step in range(3):
        red.zero_grad()
        loss = ((model(x[shard]) - y[shard]) ** 2).mean()
        loss.backward()
        red.finish()
        if step == 0:
            grads = [p.grad.clone() for p in model.parameters()]
        opt.step()
    out[rank] = dict(n_buckets=len(red.buckets), grads=grads, params=[p.detach().clone() for p in model.parameters()])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes", [256, 1 << 20])
def test_bucketed_allreduce_matches_global_batch_gradient(bucket_bytes):
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), bucket_bytes, out), nprocs=world, join=True)
    # single-process reference: mean over the global batch == average of the per-shard means (equal shard sizes)
    model = _model()
    x, y = _data()
    (0.5 * (((model(x[:4]) - y[:4]) ** 2).mean() + ((model(x[4:]) - y[4:]) ** 2).mean())).backward()
    want = [p.grad for p in model.parameters()]
    assert out[0]["n_buckets"] == out[1]["n_buckets"]
    assert out[0]["n_buckets"] > 1 if bucket_bytes == 256 else out[0]["n_buckets"] == 1
    for r in range(world):
        for g, w in zip(out[r]["grads"], want):
            torch.testing.assert_close(g, w, rtol=1e-6, atol=1e-7)
    for a, b in zip(out[0]["params"], out[1]["params"]):
        assert torch.equal(a, b), "replicas diverged after optimizer steps"


def test_single_process_reducer_is_a_noop():
    from maskplanner_amd import dp
    model = _model()
    red = dp.BucketedGradAllReduce(model.parameters(), bucket_bytes=512)
    assert red.buckets == []  # world size 1: autograd owns .grad, no flat copies
    x, y = _data()
    red.zero_grad()
    ((model(x) - y) ** 2).mean().backward()
    red.finish()
    ref = _model()
    ((ref(x) - y) ** 2).mean().backward()
    for p, q in zip(model.parameters(), ref.parameters()):
        torch.testing.assert_close(p.grad, q.grad)
    red.zero_grad()
    assert all(p.grad is None for p in model.parameters())
    assert red.grad_bytes() == sum(p.numel() * 4 for p in model.parameters())


def _gather_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from maskplanner_amd import dp
    from maskplanner_amd.factor_heads import FactorAdam
    dp.init_from_env(backend="gloo")
    fa = FactorAdam({}, {}, lr=1e-3)
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.randn(3, 8, generator=g)
    ga = torch.randn(3, 5, generator=g)
    gb = torch.randn(3, 7, generator=g)
    got = fa._gather([x, ga, gb])
    out[rank] = [t.clone() for t in got] + [x, ga, gb]
    dist.barrier()
    dist.destroy_process_group()


def test_factor_gather_concatenates_rank_rows():
    """Under DP the head gradient dW = sum_r g_r^T x_r is rebuilt from ALL ranks' factors: one all-gather of the
    packed (x, g) rows; every rank must see rank 0's rows first, then rank 1's, split back into the original widths."""
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gather_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for k in range(3):
        want = torch.cat([out[0][3 + k], out[1][3 + k]], dim=0)
        for r in range(world):
            assert torch.equal(out[r][k], want)
    # the product of the gathered factors is the sum of the per-rank gradients
    x, g = out[0][0], out[0][1]
    want = out[0][4].t() @ out[0][3] + out[1][4].t() @ out[1][3]
    torch.testing.assert_close(g.t() @ x, want)


def _worker_deferred(rank, world, port, out):
    """The replayed-graph protocol of harness.TrainStep (MASKPLANNER_DP_GRAPH=1) on CPU: gradients appear in STATIC tensors
    without any hook firing; rearm() points .grad at them and finish() exchanges every bucket."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from maskplanner_amd import dp
    dp.init_from_env(backend="gloo")
    model = _model()
    red = dp.BucketedGradAllReduce(model.parameters(), bucket_bytes=256)
    red.deferred = True
    x, y = _data()
    shard = slice(rank * 4, rank * 4 + 4)
    # "recording": one backward whose gradient tensors become the static ones (hooks are no-ops)
    red.zero_grad()
    ((model(x[shard]) - y[shard]) ** 2).mean().backward()
    static = [(p, p.grad) for p in model.parameters()]
    first = None
    for step in range(2):
        # "replay": the static tensors are overwritten in place, Python sees nothing; .grad may point anywhere (the flat views)
        fresh = torch.autograd.grad(((model(x[shard]) - y[shard]) ** 2).mean(), list(model.parameters()))
        with torch.no_grad():
            for (_, g), f in zip(static, fresh):
                g.copy_(f)
        red.rearm(static)
        red.finish()
        if step == 0:
            first = [p.grad.clone() for p in model.parameters()]
    out[rank] = dict(grads=first, again=[p.grad.clone() for p in model.parameters()])
    dist.barrier()
    dist.destroy_process_group()


def test_deferred_exchange_of_static_gradients_matches_global_batch_gradient():
    world = 2
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker_deferred, args=(world, _free_port(), out), nprocs=world, join=True)
    model = _model()
    x, y = _data()
    (0.5 * (((model(x[:4]) - y[:4]) ** 2).mean() + ((model(x[4:]) - y[4:]) ** 2).mean())).backward()
    want = [p.grad for p in model.parameters()]
    for r in range(world):
        for key in ("grads", "again"):       # the second round starts from .grad pointing at the flat views: rearm() resets that
            for g, w in zip(out[r][key], want):
                torch.testing.assert_close(g, w, rtol=1e-6, atol=1e-7)

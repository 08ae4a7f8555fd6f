"""Execute the data-parallel exchange code on the real RCCL backend with ONE rank and check it changes nothing.

A gpurun box has one GPU and RCCL refuses two ranks on one device, so this is how the bucket / hook / stream code of
maskplanner_amd/dp.py and the factor all-gather of factor_heads.py run on real hardware before the driver's multi-GPU
bench.  With one rank, AVG all-reduce and all-gather are identities:
  1. a torch-only MLP (deterministic kernels) trained 5 Adam steps with and without the forced collectives must end with
     bit-identical weights;
  2. FactorAdam fed the same factors with and without the gather must produce bit-identical weights;
  3. the full MaskPlanner step with forced collectives must track the bypassed one (its dW GEMMs use fp32 atomics, so
     two runs of the SAME configuration already differ in the last bits; the report carries that spread too).
Prints one JSON line.
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def mlp_run(force):
    import torch
    import maskplanner_amd.dp as dp
    dp.FORCE_COLLECTIVES = force
    torch.manual_seed(7)
    net = torch.nn.Sequential(torch.nn.Linear(64, 256), torch.nn.ReLU(), torch.nn.Linear(256, 256), torch.nn.ReLU(),
                              torch.nn.Linear(256, 8)).cuda()
    frozen = torch.nn.Linear(8, 8).cuda()          # a parameter that never receives a gradient (its bucket path)
    params = list(net.parameters()) + list(frozen.parameters())
    red = dp.BucketedGradAllReduce(params, bucket_bytes=128 << 10)   # several buckets
    assert red.active == force
    opt = torch.optim.Adam(params, lr=1e-2)
    x = torch.randn(512, 64, device="cuda")
    y = torch.randn(512, 8, device="cuda")
    for _ in range(5):
        red.zero_grad()
        loss = (net(x) - y).square().mean()
        loss.backward()
        red.finish()
        opt.step()
    torch.cuda.synchronize()
    return torch.cat([p.detach().flatten() for p in params]).cpu(), len(red.buckets)


def factor_run(force):
    import torch
    import maskplanner_amd.dp as dp
    from maskplanner_amd.factor_heads import FactorAdam
    dp.FORCE_COLLECTIVES = force
    torch.manual_seed(11)
    w1 = torch.nn.Parameter(torch.randn(5994, 1024, device="cuda") * 0.02)
    w2 = torch.nn.Parameter(torch.randn(1200, 1024, device="cuda") * 0.02)
    store = {}
    opt = FactorAdam({"a": w1, "b": w2}, store, lr=1e-3)
    for i in range(3):
        gen = torch.Generator(device="cuda").manual_seed(100 + i)
        x = torch.randn(32, 1024, device="cuda", generator=gen)
        store["a"] = (x, torch.randn(32, 5994, device="cuda", generator=gen))
        store["b"] = (x, torch.randn(32, 1200, device="cuda", generator=gen))   # shared input, as fc3 / fc_normals
        opt.step()
    torch.cuda.synchronize()
    return torch.cat([w1.detach().flatten(), w2.detach().flatten()]).cpu()


def full_run(force, steps=3, dp_graph=False, record_exchange=True):
    import torch
    import maskplanner_amd.dp as dp
    from maskplanner_amd.harness import TrainStep
    dp.FORCE_COLLECTIVES = force
    os.environ["MASKPLANNER_DP_GRAPH"] = "1" if dp_graph else "0"
    os.environ["MASKPLANNER_DP_COLLECTIVES_GRAPH"] = "1" if record_exchange else "0"
    ts = TrainStep("cuboids", B=4, N=1024, seed=4321)
    assert ts.reducer.active == force and ts.dp_graph == (force and dp_graph)
    losses = [float(ts.step()) for _ in range(steps)]
    torch.cuda.synchronize()
    if dp_graph:
        assert ts._graph is not None and ts._graph_b is not None and ts._static_grads, "the data-parallel step was not recorded"
        # [r6] the exchange recorded into the backward graph: two graphs; launched eagerly between replays (r5): three
        assert ts._dp_recorded == record_exchange and (ts._graph_b2 is None) == record_exchange, (ts._dp_recorded, ts._graph_b2 is None)
    os.environ["MASKPLANNER_DP_GRAPH"] = "0"
    os.environ.pop("MASKPLANNER_DP_COLLECTIVES_GRAPH", None)
    return losses


def main():
    import torch
    import torch.distributed as dist
    port = int(sys.argv[1]) if len(sys.argv) > 1 else 29611
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)
    w0, _ = mlp_run(False)
    w1, nb = mlp_run(True)
    f0, f1 = factor_run(False), factor_run(True)
    plain, plain2, forced = full_run(False), full_run(False), full_run(True)
    # the recorded data-parallel step (8 steps: 4 replays): [r6] exchange + dense Adam inside the backward graph; r5's form: launched eagerly between replays
    plain8, forced_graph8 = full_run(False, steps=8), full_run(True, steps=8, dp_graph=True)
    forced_graph8_eager_exchange = full_run(True, steps=8, dp_graph=True, record_exchange=False)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps({"mlp_identical": bool(torch.equal(w0, w1)), "mlp_buckets": nb, "factor_identical": bool(torch.equal(f0, f1)),
                      "plain": plain, "plain2": plain2, "forced": forced, "plain8": plain8, "forced_graph8": forced_graph8,
                      "forced_graph8_eager_exchange": forced_graph8_eager_exchange}))


if __name__ == "__main__":
    main()

"""Worker of tests/test_gpu_dp.py: one data-parallel rank.  Several ranks share the box's single GPU over the gloo backend
(RCCL wants one GPU per rank), which exercises the same control flow as the RCCL run: rank-sharded batch, SyncBN exchanges,
bucketed gradient all-reduce, factor all-gather, graph replay with the exchange outside the graphs.
    python tests/dp_worker.py MODE RANK WORLD PORT OUT
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
mode, rank, world, port, out = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0",
                  MASKPLANNER_DIST_BACKEND="gloo")
import faulthandler  # noqa: E402
faulthandler.dump_traceback_later(240, exit=True)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
from maskplanner_amd import dp, pointnet2_utils as pu, synthetic, sync_bn  # noqa: E402
from maskplanner_amd.pointnet2_cls_ssg import maskplanner_model  # noqa: E402

if world > 1:
    dp.init_from_env()
torch.cuda.set_device(0)
result = {}

if mode == "syncbn":
    # the global batch of 16 clouds; rank r of `world` takes its contiguous shard.  Same weights everywhere (same seed).
    B, N = 16, 1024
    cat = synthetic.CATEGORIES["cuboids"]
    batch = synthetic.make_batch(77, B, N, "cuboids", "cuboid")
    sl = slice(rank * B // world, (rank + 1) * B // world)
    torch.manual_seed(5)
    model = maskplanner_model(cat, hidden_size=(128, 128)).cuda().train()
    model.dropout.p = 0.0
    if world > 1:
        sync_bn.enable(model)
    w = [torch.randn(B, cat.out_vectors, 24, generator=torch.Generator().manual_seed(1)),
         torch.randn(B, cat.max_n_strokes, cat.out_vectors, generator=torch.Generator().manual_seed(2))]
    pc = batch["point_cloud"][sl].cuda()
    starts = [s[sl].cuda() for s in batch["fps_start"]]
    def averaged_grads():
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        if world > 1:     # what the gradient exchange does: average over ranks
            for g in grads.values():
                dist.all_reduce(g)
                g /= world
        model.zero_grad()
        return {n: g.cpu() for n, g in grads.items()}

    # (1) the encoder alone under a linear functional of its output: isolates the SyncBN backward of the fused kernels
    wf = torch.randn(B, 1024, generator=torch.Generator().manual_seed(3))
    with pu.fps_start_override(starts):
        feat = model.encode(pc.permute(0, 2, 1))
    running = {n: b.detach().cpu().clone() for n, b in model.named_buffers() if "running" in n and n.startswith("sa")}
    ((feat * wf[sl].cuda()).sum() / (B // world)).backward()
    enc_grads = averaged_grads()
    # (2) through the heads (a second forward pass: the fused backward frees its saved activations)
    with pu.fps_start_override(starts):
        feat = model.encode(pc.permute(0, 2, 1))
        outp, sm, conf, _ = model.heads(feat)
    running.update({n: b.detach().cpu().clone() for n, b in model.named_buffers() if "running" in n and not n.startswith("sa")})
    loss = ((outp * w[0][sl].cuda()).sum() + (sm * w[1][sl].cuda()).sum() + conf.sum()) / (B // world)
    loss.backward()
    grads = averaged_grads()
    result = dict(feat=feat.detach().cpu(), out=outp.detach().cpu(), sm=sm.detach().cpu(), conf=conf.detach().cpu(), sl=(sl.start, sl.stop),
                  grads=grads, enc_grads=enc_grads,
                  running=running)
elif mode in ("dp_graph", "dp_eager", "dp_graph_trip"):
    from maskplanner_amd.harness import TrainStep
    os.environ["MASKPLANNER_DP_GRAPH"] = "0" if mode == "dp_eager" else "1"
    if mode == "dp_graph_trip":      # the replica guard's failure branch: the second guarded step pretends the replicas differ
        os.environ["MASKPLANNER_DP_GUARD_TRIP"] = "0"
    from maskplanner_amd import ops
    ops.DETERMINISTIC = True       # ordered scatter kernels: what is left to differ between two runs is the dW atomics of the MLP kernels
    ts = TrainStep("cuboids", B=4, N=1024, hidden_size=(128, 128), rank=rank, seed=9)
    ts.model.dropout.p = 0.0
    losses, early = [], None
    for i in range(8):
        losses.append(float(ts.step()))
        if i == 1:
            torch.cuda.synchronize()
            early = torch.cat([p.detach().reshape(-1) for p in ts.model.parameters()]).cpu()
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in ts.model.parameters()])
    ref = flat.clone()
    dist.broadcast(ref, src=0)
    worst = {}
    for n, p_ in ts.model.named_parameters():
        r0 = p_.detach().clone()
        dist.broadcast(r0, src=0)
        d = float((p_.detach() - r0).abs().max())
        if d > 0:
            worst[n] = d
    result = dict(losses=losses, replica_diff=float((flat - ref).abs().max()), graph=ts._graph is not None and ts._graph_b is not None,
                  params=flat.cpu(), params_after_2=early, diverged=worst, fell_back=ts.dp_fell_back)
else:
    raise SystemExit(f"unknown mode {mode}")

torch.save(result, out)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()

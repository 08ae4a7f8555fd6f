#!/usr/bin/env python3
"""Headline benchmark: point-clouds/s, forward + loss + backward + Adam step, MaskPlanner cuboids_v2
(N=5120 points, B=32 per GPU, PointNet++ SSG encoder + asymm_chamfer_v9 loss) -- BASELINE.json configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU; the batch is sharded by sample (weak scaling: 32 clouds per GPU), gradients are averaged with
a bucketed RCCL all-reduce overlapped with backward (maskplanner_amd/dp.py).  Inputs are synthetic (the reference's
dataset is not public) and resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     -- the entry point with the largest device time among the library's kernels, timed with HIP events
                  on the launch stream inside the timed region; algorithmic bytes / flops per launch from DESIGN.md.
  cpu_baseline -- the CPU restatement of the same step (oracle/: C for FPS / ball query / kNN / LAP + torch fp32
                  for the MLP algebra) timed on this box's host cores on a bounded sample (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3     # fp32 vector == fp32-input MFMA peak


def measured_traffic(kernel):
    """HBM bytes per launch of `kernel` from the most recent committed PMC passes (profiles/rNN_traffic.json, made
    by tools/make_profile_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this bench)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    k = json.load(open(files[-1]))["kernels"].get(kernel)
    return None if k is None else k["hbm_bytes"]


def collect_kernel_profile(lib):
    """Per-kernel device time from the library's own launch hooks (HIP events on the launch stream, recorded inside
    the timed region).  -> {kernel name: dict(calls, ms, flops, bytes)}"""
    import ctypes
    buf = ctypes.create_string_buffer(1 << 16)
    n = lib.mp_profiler_collect(buf, len(buf))
    out = {}
    if n > 0:
        for line in buf.value.decode().strip().split("\n"):
            name, calls, ms, flops, nbytes = line.split("\t")
            out[name] = dict(calls=int(calls), ms=float(ms), flops=float(flops), bytes=float(nbytes))
    return out


def cpu_baseline(cat, N, seed):
    """The oracle's restatement of ONE training step (forward + loss + backward) on the host cores."""
    from maskplanner_amd import synthetic as syn
    from maskplanner_amd.loss_handler import maskplanner_loss_config
    from maskplanner_amd.pointnet2_cls_ssg import maskplanner_model
    from oracle import oracle as O
    from oracle import torch_ref as T
    O.build()
    Bc = 16  # ~10 s of host work
    batch = syn.make_batch(seed, Bc, N, cat.name, "cuboid")
    torch.manual_seed(seed)
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k)
          for k, v in maskplanner_model(cat).state_dict().items()}
    cfg = maskplanner_loss_config()
    t0 = time.perf_counter()
    out, sm, conf = T.strokemasks_forward(sd, batch["point_cloud"], [s.numpy() for s in batch["fps_start"]], train=True,
                                          out_vectors=cat.out_vectors, n_masks=cat.max_n_strokes)
    loss = T.asymm_v6_loss(out, batch["traj"], sm, conf, batch["stroke_ids"], batch["traj_as_pc"], cfg)
    loss.backward()
    dt = time.perf_counter() - t0
    return {"value": Bc / dt, "unit": "point-clouds/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 step of forward+loss+backward on {Bc} clouds of N={N} (no optimizer step), {dt:.1f} s"}


def main():
    if os.environ.get("MASKPLANNER_FAULT_DUMP"):   # debugging aid: dump all stacks and exit if the run hangs
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MASKPLANNER_FAULT_DUMP"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=5120)
    ap.add_argument("--category", default="cuboids")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every step kernel by kernel (no hipGraph replay)")
    ap.add_argument("--overlap-sampling", type=int, default=None, help="1/0: next batch's FPS + ball query on a second stream")
    args = ap.parse_args()

    from maskplanner_amd import dp
    rank, local, world = dp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the MaskPlanner hot path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    local = local % torch.cuda.device_count()   # (several ranks per GPU only happen in the gloo dry run)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from maskplanner_amd import _lib, ops, synthetic
    from maskplanner_amd.harness import TrainStep
    _lib.load()  # fail loudly if the HIP library is missing
    cat = synthetic.CATEGORIES[args.category]
    ts = TrainStep(cat, B=args.batch, N=args.points, device=dev, rank=rank, graph=False if args.no_graph else None,
                   overlap_sampling=None if args.overlap_sampling is None else bool(args.overlap_sampling))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    lib = _lib.load()
    # set-up, before the contract's warmup: the steps TrainStep needs to create optimizer state and record its hipGraph (three
    # eager steps + the recording one), so that even --warmup 0 times replays and not the recording
    while ts.use_graph and ts._graph is None:
        ts.step()
    if ts._graph is not None:
        ts.eager_step()   # the profiled steps of the timed region launch eagerly on this stream: warm its allocator blocks too
    for _ in range(args.warmup):
        ts.step()
    barrier()
    # one event per step on the compute stream: the median step time is reported beside the contract's mean (a shared
    # host makes the mean jittery; nothing is synchronised inside the timed region)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    profiled_steps = 0
    for i in range(args.steps):
        marks[i].record()
        # per-kernel HIP events (two records per library launch) on every 20th timed step, rank 0 only: the hooks
        # cost ~0.5 ms per profiled step, so sampling keeps the headline number honest
        prof = rank == 0 and i % 20 == 0
        if prof:
            lib.mp_profiler_enable(1)
            profiled_steps += 1
        loss = ts.eager_step() if prof else ts.step()   # the timing hooks sit in the launch path, which a graph replay skips
        if prof:
            lib.mp_profiler_enable(0)
    if marks:
        marks[-1].record()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    final_loss = float(loss.detach())

    if rank == 0:
        ms = dt / args.steps * 1e3
        kernels = collect_kernel_profile(lib)
        # dominant kernel = largest total device time on the step's own stream; its binding roof from the algorithmic work
        # model.  (With pipelined sampling the first-level FPS -- a latency-bound chain of dependent arg-max steps, one
        # workgroup per cloud -- runs on the second stream underneath the step and is not on the critical path.)
        on_path = [k for k in kernels if not (ts.overlap and k.startswith("fps_kernel"))] or list(kernels)
        dom = max(on_path, key=lambda k: kernels[k]["ms"])
        d = kernels[dom]
        avg_s = d["ms"] / d["calls"] * 1e-3
        flops, nbytes = d["flops"] / d["calls"], d["bytes"] / d["calls"]
        if flops / (FP32_PEAK_TFLOPS * 1e12) >= nbytes / (HBM_PEAK_GBS * 1e9):
            bound, ach, peak, unit = "mfma", flops / avg_s / 1e12, FP32_PEAK_TFLOPS, "TFLOP/s"
        else:
            bound, ach, peak, unit = "hbm", nbytes / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
        line = {
            "metric": "point-clouds/sec fwd+bwd (N=5120, B=32)", "value": args.batch * world * args.steps / dt,
            "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cat.name}_v2 N={args.points} B={args.batch}/GPU SSG encoder + asymm_chamfer_v9 loss "
                                   f"(forward+loss+backward+Adam), S={cat.out_vectors} M={cat.max_n_strokes}",
                       "parallelism": f"dp{world}", "global_batch": args.batch * world, "grad_allreduce_MB":
                           round(ts.reducer.grad_bytes() / 1e6, 1),
                       "launch": ("hipGraph replay of the recorded step" + (" (two graphs; head optimizer on its own stream under the next encoder forward)"
                                                                             if ts._graph_b is not None else ""))
                       if ts._graph is not None else "eager (kernel by kernel)",
                       "sampling": "next batch's first-level FPS + ball query on a second stream, under the step" if ts.overlap
                       else "in line"},
            "roofline": {"kernel": dom, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                         "traffic": measured_traffic(dom), "avg_us": avg_s * 1e6, "launches_per_step": d["calls"] / profiled_steps,
                         "flops_per_launch": flops, "bytes_per_launch": nbytes},
            # the kernels BASELINE.json names, each against BOTH roofs (algorithmic flops vs the fp32 peak, algorithmic
            # bytes vs HBM): FPS is latency-bound, ball query / kNN are fp32-VALU bound, grouping is the HBM-bound one
            "named_kernels": {k: {"us_per_launch": round(v["ms"] * 1e3 / v["calls"], 1),
                                  "tflops": round(v["flops"] / v["calls"] / (v["ms"] / v["calls"] * 1e-3) / 1e12, 2),
                                  "frac_fp32_peak": round(v["flops"] / v["calls"] / (v["ms"] / v["calls"] * 1e-3) / 1e12 / FP32_PEAK_TFLOPS, 4),
                                  "alg_GBps": round(v["bytes"] / v["calls"] / (v["ms"] / v["calls"] * 1e-3) / 1e9, 1),
                                  "frac_hbm": round(v["bytes"] / v["calls"] / (v["ms"] / v["calls"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}
                              for k, v in kernels.items() if k.split("<")[0] in
                              ("fps_kernel", "ball_query_kernel", "knn1_kernel", "knn_kernel", "group_kernel", "group_bwd_atomic_kernel")},
            "kernels_us_per_step": {k: round(v["ms"] * 1e3 / profiled_steps, 1) for k, v in
                                    sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])},
            "final_loss": final_loss,
        }
        if marks:
            per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps) if i % 10 != 0)  # unprofiled steps
            if per_step:
                line["step_ms_median"] = per_step[len(per_step) // 2]
                line["step_ms_min"] = per_step[0]
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cat, args.points, 1235)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

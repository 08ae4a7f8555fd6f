#!/usr/bin/env python3
"""Headline benchmark: point-clouds/s, forward + loss + backward + Adam step, MaskPlanner cuboids_v2
(N=5120 points, B=32 per GPU, PointNet++ SSG encoder + asymm_chamfer_v9 loss) -- BASELINE.json configs[1].

    python bench.py [--gpus N] [--steps K] [--warmup W]

With N > 1 and no torchrun environment the script launches its own workers (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 ...` of itself, before anything touches the GPU); under torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  One process per GPU; the batch is sharded by sample (weak
scaling: 32 clouds per GPU), gradients are exchanged over RCCL (maskplanner_amd/dp.py).  Inputs are synthetic (the
reference's dataset is not public) and resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

`value` is the training step of `harness.TrainStep` (the path's ceiling: resident batch, hipGraph replay, factor heads).
Extra objects in the line (rank 0, N=1):
  roofline     -- the library kernel with the largest device time on the step's own stream; its duration is the mean over the K
                  timed replays (wall-clock stamps recorded around it inside the graph: `timed_by`; the per-kernel table comes from one
                  eagerly launched step at the end of the warm-up); algorithmic bytes / flops per launch from DESIGN.md; `traffic` from the committed PMC passes of the
                  run's own config.  For the split-plane kernels (fp32 products as six bf16 MFMA products) `achieved` / `peak` / `frac`
                  are the EXECUTED matrix-core figures against the bf16 dense peak; `algorithmic` keeps the fp32-equivalent rate,
                  `executed` both executed fractions and `binding` the larger of them.
  named_kernels-- FPS / ball query / kNN / grouping against both roofs, `effective_scan_GBps`, and for FPS
                  the measured latency floor (the same kernel without distance arithmetic) and the fraction of it.
  dropin_path  -- the reference's own loop body (train_maskplanner.py:182-227) on the drop-in modules: torch.optim.Adam over
                  all parameters, a fresh host batch per step, compute() -> numpy, loss.item(); the model call and the loss call
                  replay the graphs they record themselves (maskplanner_amd/graphed.py).  `--path dropin` makes this the headline
                  `value` instead.
  streamed_inputs -- the harness step fed a fresh HOST batch every step (PCIe-inclusive): batch k+1 is collated onto the device
                  and sampled on the second stream during step k.  `--stream-batches K` makes this the headline run.
  ucube        -- the same step on U[-1,1]^3 clouds (sparse balls: full-scan ball query, heavy padding).
  cpu_baseline -- the CPU restatement of the same step (oracle/) on this box's host cores, bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
FP32_PEAK_TFLOPS = 157.3     # fp32 vector == fp32-input MFMA peak
BF16_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA


def _split_template(name):
    base, _, rest = name.partition("<")
    args = [a.strip() for a in rest.rstrip(">").split(",")] if rest else []
    return base.strip(), args


def _variant_ok(tag_base, name_base, name_args):
    """The library's launch tags name the operand variant in the kernel name (`pos_gemm_bf16_kernel`, `dw_gemm_split_kernel`,
    `bwd_fused_bf16_kernel`), rocprofv3 prints it as a template argument: PREC (last) of the tiled GEMMs -- 0 fp32 MFMA, 1 bf16,
    3 split planes --, ONE (last) of the position-stream kernels."""
    bf16, split, split2 = "_bf16_" in tag_base, "_split_" in tag_base, "_split2_" in tag_base
    if name_base in ("pos_gemm_kernel", "dw_gemm_kernel"):
        return name_args[-1:] == [("1" if bf16 else "3" if split else "2" if split2 else "0")]
    if name_base in ("fwd_chunk_kernel", "bwd_fused_kernel") and name_args and name_args[-1] in ("true", "false"):
        return (name_args[-1] == "true") == bf16 if len(name_args) >= (8 if name_base == "fwd_chunk_kernel" else 6) else not bf16
    return True


def measured_traffic(kernel, config_key=None):
    """HBM bytes per launch of `kernel` from the most recent committed PMC passes (profiles/rNN_traffic.json, made by
    tools/make_profile_summary.py from separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs of this bench; `config_key`: the table of
    one of the other BASELINE configs -- tools/profile_configs.sh -- before the default one).  The library tags a launch with the template
    arguments it chose; rocprofv3 prints every argument including defaulted ones, so a tag matches the profile name whose argument
    list it is a prefix of."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
    if not files:
        return None
    doc = json.load(open(files[-1]))
    tables = ([doc.get("configs", {}).get(config_key)] if config_key else []) + [doc["kernels"]]
    kernel = kernel.split("[")[0]            # ([r6] a launch-shape suffix "[N x S]" of the tag is not part of the profiled kernel's name)
    tbase, args = _split_template(kernel)
    base = tbase.replace("_bf16_kernel", "_kernel").replace("_split_kernel", "_kernel").replace("_split2_kernel", "_kernel")
    for table in tables:
        if not table:
            continue
        if kernel in table:
            return table[kernel]["hbm_bytes"]
        hits = []
        for name, v in table.items():
            b, a = _split_template(name)
            if b == base and a[:len(args)] == args and _variant_ok(tbase, b, a):
                hits.append(v["hbm_bytes"])
        if hits:
            return max(hits)      # (several instantiations behind one tag, e.g. fps_kernel<256, 20, false|true>: the larger figure)
    return None


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


def read_kernel_marks(lib, last_n):
    """[r5] The marked kernels' durations over their last `last_n` executions inside the replayed graphs (mp_profiler_mark: wall-clock stamps
    around them, recorded with the graph).  -> {kernel name: dict(calls = samples, ms = their sum, flops, bytes)}"""
    import ctypes
    buf = ctypes.create_string_buffer(1 << 14)
    n = lib.mp_profiler_read_marks(buf, len(buf), int(last_n))
    out = {}
    if n > 0:
        for line in buf.value.decode().strip().split("\n"):
            name, calls, ms, flops, nbytes = line.split("\t")
            if float(ms) > 0:
                out[name] = dict(calls=int(calls), ms=float(ms), flops=float(flops), bytes=float(nbytes))
    return out


def _host_cpus():
    """CPUs this process may actually use: the affinity mask, capped by the cgroup's CPU quota when there is one (a 128-thread box that grants
    a container 16 CPUs' worth of time runs 128 threads slower than 16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = max(1, min(n, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cat, N, seed):
    """The oracle's restatement of ONE training step (forward + loss + backward) on the host cores.  [r6] The thread count is CHOSEN: r5 ran
    torch's default (one thread per hardware thread of the box, 128) and reported 2.6 point-clouds/s where the build container's 8 cores give
    6.8 -- these boxes grant a container far less CPU time than they show threads.  One untimed pass, one timed pass at 8 / 16 / 32 threads
    (those the host allows), then two more at the best: median of its three.  Beside it, read from the committed timing of the IMPORTED reference
    in the build container (oracle/time_reference.py; /root/reference does not travel), the reference's own forward + backward."""
    import torch
    from maskplanner_amd import synthetic as syn
    from maskplanner_amd.loss_handler import maskplanner_loss_config
    from maskplanner_amd.pointnet2_cls_ssg import maskplanner_model
    from oracle import oracle as O
    from oracle import torch_ref as T
    O.build()
    Bc = 32  # the bench batch
    batch = syn.make_batch(seed, Bc, N, cat.name, "cuboid")
    torch.manual_seed(seed)
    state = maskplanner_model(cat).state_dict()
    cfg = maskplanner_loss_config()

    def one_pass():
        sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point and "running" not in k) for k, v in state.items()}
        t0 = time.perf_counter()
        out, sm, conf = T.strokemasks_forward(sd, batch["point_cloud"], [s.numpy() for s in batch["fps_start"]], train=True,
                                              out_vectors=cat.out_vectors, n_masks=cat.max_n_strokes)
        loss = T.asymm_v6_loss(out, batch["traj"], sm, conf, batch["stroke_ids"], batch["traj_as_pc"], cfg)
        loss.backward()
        return time.perf_counter() - t0
    avail, before = _host_cpus(), torch.get_num_threads()
    spent = [0.0]

    def timed(n):
        torch.set_num_threads(n)        # (torch's pool and, through the shared OpenMP runtime, the C oracle's loops)
        dt = one_pass()
        spent[0] += dt
        return dt
    cands = sorted({min(c, avail) for c in (8, 16, 32)})
    try:
        timed(cands[0])                 # untimed: allocator, lazy initialisation
        first = {}
        for n in cands:
            if spent[0] > 45.0 and first:
                break
            first[n] = timed(n)
        best = min(first, key=first.get)
        times = [first[best]]
        while len(times) < 3 and spent[0] < 60.0:
            times.append(timed(best))
    finally:
        torch.set_num_threads(before)
    dt = sorted(times)[len(times) // 2]
    out = {"value": Bc / dt, "unit": "point-clouds/s", "cores": best, "kind": "port",
           "sample": f"forward+loss+backward of one B={Bc} batch of N={N} clouds (no optimizer step) on {best} threads (host grants {avail}; one pass each at "
                     + ", ".join(f"{n}: {t:.1f} s" for n, t in first.items()) + f"), median of {len(times)} passes at {best}: "
                     + ", ".join(f"{t:.1f}" for t in times) + " s"}
    try:        # the imported reference itself, timed where it can be imported (the build container): carried, not re-measured here
        import glob
        ref = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_reference_cpu_timing.json")))[-1]))
        out["reference_in_build_container"] = {
            "value": ref["point_clouds_per_s_forward_backward_without_loss"], "unit": "point-clouds/s", "cores": ref["cores"], "kind": "reference",
            "sample": f"the imported reference's forward + backward at B={ref['B']}, N={ref['N']} (loss excluded: pytorch3d is not installable), "
                      f"median of {ref['repeats']} on the build container's {ref['cores']} cores (oracle/time_reference.py, committed under profiles/); "
                      "not measured on this box: /root/reference does not travel"}
    except Exception:
        pass
    return out


def inference_b1(cat, N, dev, reps=50):
    """[r5] Batch-of-one inference latency, the one latency figure the reference itself prints (test_maskplanner.py:253-257, 299:
    `model(point_cloud[:1, ...])` in eval mode): device-synchronised wall-clock per forward, median of `reps`, launched kernel by kernel
    (what the unchanged test script does) and replayed from a recorded hipGraph."""
    import torch
    from maskplanner_amd import synthetic as syn
    from maskplanner_amd.pointnet2_cls_ssg import maskplanner_model
    torch.manual_seed(7)
    model = maskplanner_model(cat).to(dev).eval()
    pc = syn.make_batch(99, 1, N, cat.name, "cuboid")["point_cloud"].to(dev).permute(0, 2, 1)

    def fwd():
        with torch.no_grad():
            return model(pc)

    def timed(fn):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return ts[len(ts) // 2], ts[0]
    from maskplanner_amd import graphed
    was_enabled = graphed.ENABLED       # (a run started with MASKPLANNER_DROPIN_GRAPH=0 keeps its later legs eager: restored below)
    graphed.ENABLED = False
    e_med, e_min = timed(fwd)
    graphed.ENABLED = True
    a_med, a_min = timed(fwd)         # [r5] what the unchanged test script gets: the forward replays the graph it recorded itself after three calls
    graphed.ENABLED = False
    out = {"eager_ms_median": e_med, "eager_ms_min": e_min, "unchanged_script_ms_median": a_med, "unchanged_script_ms_min": a_min,
           "unit": "ms per forward (B=1)", "reps": reps,
           "what": f"eval-mode forward of one N={N} cloud (FPS 512 + 128, ball queries, three set abstractions, heads), host wall-clock "
                   "including the synchronisation; FPS start indices drawn per call like the reference (pointnet2_utils.py:77).  eager: launched "
                   "kernel by kernel (MASKPLANNER_DROPIN_GRAPH=0); unchanged_script: model(x) as test_maskplanner.py:253-257 calls it (the forward "
                   "replays the graph it recorded itself, maskplanner_amd/graphed.py); graph: a bare replay with fixed FPS starts (lower bound)"}
    try:
        from maskplanner_amd import pointnet2_utils as pu
        starts = [torch.zeros(1, dtype=torch.long, device=dev), torch.zeros(1, dtype=torch.long, device=dev)]

        def fwd_fixed():
            with torch.no_grad(), pu.fps_start_override(list(starts)):
                return model(pc)
        for _ in range(3):
            fwd_fixed()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        from maskplanner_amd.harness import recording
        with recording(g):
            fwd_fixed()
        g_med, g_min = timed(g.replay)
        out.update({"graph_ms_median": g_med, "graph_ms_min": g_min})
    except Exception as exc:       # the eager figure stands on its own
        out["graph_error"] = f"{type(exc).__name__}: {exc}"[:200]
    finally:
        graphed.ENABLED = was_enabled
    return out


def streamed_leg(args, k):
    """The harness step fed a fresh HOST batch every step, as a child process of its own (`bench.py --stream-batches 4`)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--stream-batches", "4", "--steps", str(k), "--warmup", str(max(args.warmup, 8)),
           "--batch", str(args.batch), "--points", str(args.points), "--category", args.category, "--no-side-legs", "--no-cpu-baseline"]
    try:
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    except Exception as exc:
        return {"error": f"{type(exc).__name__}: {exc}"[:300]}
    return {"value": d["value"], "unit": d["unit"], "ms_per_step": d["ms_per_step"], "steps": d["steps"], "step_ms_median": d.get("step_ms_median"),
            "final_loss": d.get("final_loss"),
            "what": "4 host batches of ragged dataset items in rotation; batch k+1 goes through the device collate (pinned flat copy per key + "
                    "pad kernel) and its FPS / ball query on the second stream during step k (a child process: bench.py --stream-batches 4)"}


def dp_overhead_leg():
    """[r5] The N > 1 launch path against the N = 1 path on this one GPU (tools/dp_overhead.py, a child process with one forced RCCL rank):
    what a future 1 -> 8 comparison must subtract before it reads the rest as communication."""
    import socket
    for _attempt in range(2):      # (RCCL's start-up next to a process that holds the GPU fails now and then on shared boxes: one retry)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dp_overhead.py"), str(port), "30"], capture_output=True, text=True,
                                 timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
        except subprocess.TimeoutExpired:
            continue
        lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
        if out.returncode == 0 and lines:
            return json.loads(lines[-1])
    return {"error": "tools/dp_overhead.py did not finish", "stderr": (out.stderr[-300:] if "out" in dir() else "")}


def self_launch(args):
    """`python bench.py --gpus N` typed as is: start one worker per GPU with torch.distributed.run.  Nothing in this process
    has touched the GPU yet (device_count() does not initialise it); the workers inherit HSA_ENABLE_IPC_MODE_LEGACY=0, which
    RCCL's dmabuf IPC needs on this driver."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def kernel_tables(kernels, profiled_steps, floors):
    """named_kernels (the kernels BASELINE.json names, each against BOTH roofs + SURVEY 8d's effective scan rate) and the
    per-step time of every tagged kernel."""
    scan_bytes_per_flop = {"fps_kernel": 20.0 / 8.0, "ball_query_kernel": 12.0 / 8.0, "knn1_kernel": 4.0 / 3.0, "knn_kernel": 4.0 / 3.0,
                           "knn1_screen_kernel": 4.0 / 3.0}
    named = {}
    for k, v in kernels.items():
        base = k.split("<")[0].split("[")[0]        # ([r6] "[N x S]": one row per launch shape -- ball query of each level, each search of the loss)
        # FPS, ball query, the nearest-neighbour searches of the chamfer terms (every kernel of knn.hip: direct, screened, the
        # plane pre-pass, the backward scatter) and the grouping gathers
        if not (base in ("fps_kernel", "ball_query_kernel") or base.startswith("knn") or base.startswith("group_")
                or base.startswith("chamfer_")):
            continue
        t = v["ms"] / v["calls"] * 1e-3
        fl, by = v["flops"] / v["calls"], v["bytes"] / v["calls"]
        e = {"us_per_launch": round(t * 1e6, 1), "tflops": round(fl / t / 1e12, 2), "frac_fp32_peak": round(fl / t / 1e12 / FP32_PEAK_TFLOPS, 4),
             "alg_GBps": round(by / t / 1e9, 1), "frac_hbm": round(by / t / 1e9 / HBM_PEAK_GBS, 4)}
        if base in scan_bytes_per_flop:
            # bytes_scan = pairs x operand bytes (each pair touches its operand once): > 100 % of HBM is expected (LDS reuse)
            e["effective_scan_GBps"] = round(fl * scan_bytes_per_flop[base] / t / 1e9, 1)
        if k in floors:
            e["latency_floor_us"] = round(floors[k], 1)
            e["frac_latency_floor"] = round(floors[k] / (t * 1e6), 4)
        named[k] = e
    per_step = {k: round(v["ms"] * 1e3 / profiled_steps, 1) for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}
    if os.environ.get("MASKPLANNER_BENCH_RATES"):     # diagnostic: [us per step, launches per step, algorithmic GB/s, TFLOP/s] of every tagged kernel
        per_step = {k: [per_step[k], round(v["calls"] / profiled_steps, 1), round(v["bytes"] / max(v["ms"], 1e-9) / 1e6, 0),
                        round(v["flops"] / max(v["ms"], 1e-9) / 1e9, 1)] for k, v in sorted(kernels.items(), key=lambda kv: -kv[1]["ms"])}
    return named, per_step


def fps_floors(ts, lib):
    """S * t_iter for the two FPS launches of the encoder: mp_fps_floor_f32 (the same kernel, same launch shape and per-step
    reduce + barrier chain, no distance arithmetic), median of 9 launches each, HIP events on the launch stream."""
    import torch
    from maskplanner_amd import ops
    out = {}
    xyz = ts.batch["point_cloud"]
    B = xyz.shape[0]
    for m in ts._plan_levels():
        N = xyz.shape[1]
        S = m.npoint
        start = torch.zeros(B, dtype=torch.long, device=xyz.device)
        idx = torch.empty(B, S, dtype=torch.long, device=xyz.device)
        new_xyz = torch.empty(B, S, 3, device=xyz.device)
        times = []
        for _ in range(12):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            ops._run("fps_floor", xyz, lib.mp_fps_floor_f32, xyz.data_ptr(), B, N, S, start.data_ptr(), idx.data_ptr(), new_xyz.data_ptr())
            b.record()
            times.append((a, b))
        torch.cuda.synchronize()
        us = sorted(a.elapsed_time(b) * 1e3 for a, b in times[3:])
        tag = {5120: "fps_kernel<512, 10>", 512: "fps_kernel<64, 8>", 10240: "fps_kernel<512, 20>"}.get(N)
        if tag:
            out[tag] = us[len(us) // 2]
        xyz = new_xyz
    return out


def time_steps(step, steps, warmup, barrier, profile=None):
    """The contract's timing: W untimed steps, then exactly K steps bracketed by barrier + synchronize.  Returns (seconds,
    per-step device milliseconds of the unprofiled steps, last loss)."""
    import torch
    for _ in range(warmup):
        step(False)
    barrier()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    t0 = time.perf_counter()
    loss = None
    for i in range(steps):
        marks[i].record()
        loss = step(profile is not None and profile(i))
    marks[-1].record()
    barrier()
    dt = time.perf_counter() - t0
    per_step = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps) if not (profile is not None and profile(i)))
    return dt, per_step, loss


def main():
    if os.environ.get("MASKPLANNER_FAULT_DUMP"):   # debugging aid: dump all stacks and exit if the run hangs
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["MASKPLANNER_FAULT_DUMP"]), exit=True)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32, help="clouds per GPU")
    ap.add_argument("--points", type=int, default=5120)
    ap.add_argument("--category", default="cuboids")
    ap.add_argument("--dist", default="cuboid", choices=["cuboid", "ucube"], help="point distribution of the synthetic clouds")
    ap.add_argument("--path", default="harness", choices=["harness", "dropin"],
                    help="harness: TrainStep (resident batch, graph replay, factor heads); dropin: the reference's own loop body")
    ap.add_argument("--encoder", default="ssg", choices=["ssg", "msg"], help="msg: two multi-scale set abstractions (BASELINE configs[4])")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="operand type of the grouped-MLP MFMAs (accumulation is fp32)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-side-legs", action="store_true", help="skip the dropin_path / ucube legs of the default run")
    ap.add_argument("--no-graph", action="store_true", help="launch every step kernel by kernel (no hipGraph replay)")
    ap.add_argument("--overlap-sampling", type=int, default=None, help="1/0: next batch's FPS + ball query on a second stream")
    ap.add_argument("--sync-bn", action="store_true", help="data-parallel runs: BatchNorm statistics over the global batch")
    ap.add_argument("--stream-batches", type=int, default=0,
                    help="K > 0: rotate K host batches through the step, each collated onto the device during the previous step")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args))

    import torch
    import torch.distributed as dist
    from maskplanner_amd import dp
    rank, local, world = dp.init_from_env()
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the MaskPlanner hot path has no CPU fallback")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    local = local % torch.cuda.device_count()   # (several ranks per GPU only happen in the gloo dry run)
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    from maskplanner_amd import _lib, synthetic
    from maskplanner_amd.harness import DropInLoop, TrainStep
    lib = _lib.load()  # fail loudly if the HIP library is missing
    cat = synthetic.CATEGORIES[args.category]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def make_harness(dist_points, stream=None):
        kw = {}
        if (args.stream_batches if stream is None else stream):
            kw.update(stream_batches=args.stream_batches if stream is None else stream)
        if args.encoder != "ssg" or args.dtype != "f32":
            kw.update(encoder=args.encoder, mlp_dtype=args.dtype)
        if args.sync_bn:
            kw.update(sync_bn=True)
        ts = TrainStep(cat, B=args.batch, N=args.points, device=dev, rank=rank, graph=False if args.no_graph else None,
                       overlap_sampling=None if args.overlap_sampling is None else bool(args.overlap_sampling),
                       dist_points=dist_points, **kw)
        # set-up, before the contract's warmup: the steps TrainStep needs to create optimizer state and record its hipGraph
        # (three eager steps + the recording one), so that even --warmup 0 times replays and not the recording
        while ts.use_graph and ts._graph is None:
            ts.step()
        if ts._graph is not None:
            ts.eager_step()   # the profiled step launches eagerly on this stream: warm its allocator blocks too
        return ts

    def run_harness(ts, steps, warmup, profile_every=50):
        def step(prof):
            if prof and rank == 0:
                lib.mp_profiler_enable(1)
            loss = ts.eager_step() if prof else ts.step()   # the timing hooks sit in the launch path, which a graph replay skips
            if prof and rank == 0:
                lib.mp_profiler_enable(0)
            return loss
        # per-kernel HIP events (two records per library launch) on ONE timed step per `profile_every` x 4 steps (the middle one of the
        # default 100), rank 0 only: a profiled step is launched kernel by kernel from Python (the hooks live in the launch path, which
        # a graph replay skips) and costs ~2 ms more than a replayed one; it IS part of the timed region
        # EVERY rank launches that step eagerly (rank 0 alone records events): a replayed data-parallel step issues its collectives in
        # another order than an eager one (the factor all-gather sits behind the heads' backward, in front of the bucket all-reduces),
        # so a rank that left the replay alone would pair its all-reduce with the others' all-gather -- a deadlock ([r4] found with two
        # gloo ranks on one GPU; the profiled step used to be rank 0's alone)
        # [r5] That step now runs as the LAST step of the warm-up (the per-kernel table is a breakdown, not part of the rate): the K timed steps
        # are K replays.  The roofline kernel's own duration still comes from the timed region -- the candidates for the largest kernel were
        # marked while the graphs were recorded (mp_profiler_mark: a wall-clock stamp kernel in front of and behind each), so every replay
        # timestamps them and read_kernel_marks() returns their mean over the K timed steps.  MASKPLANNER_BENCH_PROFILE_IN_TIMED=1: r4's way.
        n_prof = 0
        if profile_every and os.environ.get("MASKPLANNER_BENCH_PROFILE_IN_TIMED", "0") == "1":
            every = 4 * profile_every
            prof = lambda i: i % every == min(every, steps) // 2
            dt, per_step, loss = time_steps(step, steps, warmup, barrier, prof)
            n_prof = len([i for i in range(steps) if prof(i)])
        else:
            if profile_every:
                tail = min(2, max(warmup - 1, 0))        # two replays between that step and the timed region (the pipelined sampling plan
                for _ in range(max(warmup - 1 - tail, 0)):   # and the optimizer stream find their rhythm again)
                    step(False)
                step(True)
                for _ in range(tail):
                    step(False)
                warmup, n_prof = 0, 1
            dt, per_step, loss = time_steps(step, steps, warmup, barrier, None)
        return dt, per_step, float(loss.detach()), n_prof

    def run_dropin(steps, warmup, dist_points="cuboid", adam_kwargs=None, info=None):
        """[r6] The loop draws from 32 host batches, each padded to its own maximum like the reference's collate does (paintnet_ODv1.py:738-747):
        ~30 distinct (n_segments, n_points) widths, visited in shuffled order.  `info` receives what the recorded calls did DURING THE TIMED
        STEPS (replayed / eager / recorded; recordings' capacities) and the pool's widths."""
        loop = DropInLoop(cat, B=args.batch, N=args.points, device=dev, rank=rank, dist_points=dist_points, adam_kwargs=adam_kwargs)
        last = [0.0]

        def step(_prof):
            last[0] = loop.step()
            return torch.tensor(last[0])
        for _ in range(20):
            step(False)     # optimizer state, allocator, the libraries' lazy kernel selection (a stall of tens of ms in the first steps), and
                            # the recordings of maskplanner_amd/graphed.py (three eager calls, then one recording per capacity bucket that turns up)
        for _ in range(warmup):
            step(False)
        before = loop.graph_stats()
        dt, per_step, _ = time_steps(step, steps, 0, barrier)
        if info is not None:
            after = loop.graph_stats()
            timed = {who: {k: after[who][k] - before[who][k] for k in ("replayed", "eager", "recorded")} for who in ("model", "loss")}
            for who in timed:
                n = sum(timed[who].values())
                timed[who]["hit_rate"] = timed[who]["replayed"] / n if n else 0.0
            w = loop.widths()
            info.update({"timed_steps": timed, "whole_run": after, "host_batches": len(loop.host_batches), "distinct_gt_widths": len(w),
                         "gt_width_range": {"n_segments": [w[0][0], w[-1][0]], "n_points": [min(x[1] for x in w), max(x[1] for x in w)]}})
        return dt, per_step, last[0]

    line = None
    marks = {}
    if args.path == "harness":
        # (before the recording) the kernels that are the step's largest in one BASELINE config or another: at most one launch each per step
        lib.mp_profiler_mark(os.environ.get("MASKPLANNER_BENCH_MARKS", "bwd_roles_kernel<3|bwd_stream16_kernel<3, 128, 128>|bwd_fused_kernel<3, 128, 128>"
                                                                       "|bwd_fused_bf16_kernel<3, 128, 128>").encode())
        ts = make_harness(args.dist)
        dt, per_step, final_loss, profiled_steps = run_harness(ts, args.steps, args.warmup)
        if rank == 0:
            marks = read_kernel_marks(lib, args.steps)          # (their last K executions: the K timed replays)
        lib.mp_profiler_mark(None)                  # (the side legs' recordings carry no marks)
        ts.check()      # (outside the timed region) a failed stroke-mask matching or a non-finite loss raises here
    else:
        dropin_info = {}
        dt, per_step, final_loss = run_dropin(args.steps, args.warmup, args.dist, info=dropin_info,
                                              adam_kwargs={"fused": True} if os.environ.get("MASKPLANNER_BENCH_FUSED_ADAM") == "1" else None)
        ts, profiled_steps = None, 0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    if rank == 0:
        ms = dt / args.steps * 1e3
        enc = "SSG encoder" if args.encoder == "ssg" else "MSG encoder (two multi-radius set abstractions)"
        line = {
            "metric": f"point-clouds/sec fwd+bwd (N={args.points}, B={args.batch})", "value": args.batch * world * args.steps / dt,
            "unit": "point-clouds/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"{cat.name}_v2 N={args.points} B={args.batch}/GPU {enc} + asymm_chamfer_v9 loss "
                                   f"(forward+loss+backward+Adam), S={cat.out_vectors} M={cat.max_n_strokes}, {args.dist} clouds",
                       "parallelism": f"dp{world}", "global_batch": args.batch * world, "path": args.path},
            "final_loss": final_loss,
        }
        if args.path == "dropin":
            line["recorded_calls"] = dropin_info
        if per_step:
            line["step_ms_median"] = per_step[len(per_step) // 2]
            line["step_ms_min"] = per_step[0]
        if ts is not None:
            line["config"].update({
                "grad_allreduce_MB": round(ts.reducer.grad_bytes() / 1e6, 1),
                "launch": ("hipGraph replay of the recorded step" + (" (two graphs; head optimizer on its own stream under the next encoder forward)"
                                                                      if ts._graph_b is not None else ""))
                if ts._graph is not None else "eager (kernel by kernel)",
                "sampling": "next batch's FPS + ball queries (every sampling level) on a second stream, under the step" if ts.overlap else "in line",
                "batchnorm": "global-batch statistics (SyncBN)" if getattr(ts, "sync_bn", False) else "per-replica statistics",
                "inputs": ("a fresh host batch per step, collated onto the device during the previous step" if getattr(ts, "_stream", None)
                           else "one batch resident in HBM")})
            kernels = collect_kernel_profile(lib)
            # dominant kernel = largest total device time on the step's own stream; its binding roof from the algorithmic work
            # model.  (With pipelined sampling the first-level FPS -- a latency-bound chain of dependent arg-max steps, one
            # workgroup per cloud -- runs on the second stream underneath the step and is not on the critical path.)
            # (likewise the factor Adam of the head matrices: seven launches on its own stream underneath the encoder backward)
            off_path = lambda k: (ts.overlap and k.startswith("fps_kernel")) or (ts._graph_b is not None and k.startswith("adam_lowrank"))
            on_path = [k for k in kernels if not off_path(k)] or list(kernels)
            dom = max(on_path, key=lambda k: kernels[k]["ms"])
            split = args.dtype == "f32" and os.environ.get("MP_SA_SPLIT", "1") != "0"
            # which table of profiles/rNN_traffic.json this run's kernels are looked up in (tools/profile_configs.sh)
            cfg_key = (f"containers_msg_{args.dtype}" if (args.encoder == "msg" and args.category == "containers")
                       else args.category if (args.category != "cuboids" and args.encoder == "ssg" and args.dtype == "f32") else None)

            def roof(k):
                """One kernel against its roofs.  `achieved` / `peak` / `frac` price the ALGORITHMIC work per launch
                against the peak of the dtype the path computes in (fp32-input MFMA 157.3 TF, or bf16 with --dtype bf16) or
                against HBM, whichever the work model says binds.  `executed` is what the hardware actually does for the same launch
                -- the split-plane kernels run each fp32 product as six bf16 MFMA products (sa_mlp.hip: split3), so the matrix cores see
                6x the flops at the bf16 rate -- and `binding` names the larger of the executed matrix-core and HBM fractions: the roof
                the kernel is really under."""
                d = kernels[k]
                avg_s = d["ms"] / d["calls"] * 1e-3
                eager_avg_s, timed_by = avg_s, "HIP events around the launch in one eagerly launched step"
                if k in marks:              # [r5] the kernel inside the replayed graphs, averaged over the K timed steps
                    avg_s = marks[k]["ms"] / marks[k]["calls"] * 1e-3
                    timed_by = (f"wall-clock stamps around the kernel inside the replayed graph, mean of the last {marks[k]['calls']} timed steps "
                                "(the two stamp kernels' boundaries, ~3 us, are inside the interval)")
                flops, nbytes = d["flops"] / d["calls"], d["bytes"] / d["calls"]
                # [r5] the BACKWARD position-stream kernels run two planes / three products per fp32 product (sa_mlp.hip: split2) unless MP_BWD_PLANES=3
                bwd2 = (os.environ.get("MP_BWD_PLANES", "2") != "3" and any(t in k for t in ("bwd_fused", "bwd_roles"))) or "_split2_" in k
                planes = (3.0 if bwd2 else 6.0) if (split and "bf16" not in k and "stream16" not in k and any(t in k for t in ("fused", "roles", "chunk", "bwd_first", "gemm", "lean"))) else 1.0
                # the matrix-core roof is the one the kernel's INSTRUCTIONS run under: bf16 dense peak for the plane products of a split
                # kernel (`planes` executed bf16 products per algorithmic fp32 product) and for --dtype bf16, the fp32-input MFMA peak otherwise
                ex_peak = BF16_PEAK_TFLOPS if (planes > 1 or "bf16" in k or "stream16" in k) else FP32_PEAK_TFLOPS      # (stream16: the one-plane bf16 kernels)
                ex_flops = planes * flops
                if ex_flops / (ex_peak * 1e12) >= nbytes / (HBM_PEAK_GBS * 1e9):
                    bound, ach, peak, unit = "mfma", ex_flops / avg_s / 1e12, ex_peak, "TFLOP/s"
                else:
                    bound, ach, peak, unit = "hbm", nbytes / avg_s / 1e9, HBM_PEAK_GBS, "GB/s"
                r = {"kernel": k, "bound": bound, "achieved": ach, "peak": peak, "unit": unit, "frac": ach / peak,
                     "traffic": measured_traffic(k, cfg_key), "avg_us": avg_s * 1e6, "launches_per_step": d["calls"] / max(profiled_steps, 1),
                     "flops_per_launch": flops, "bytes_per_launch": nbytes, "timed_by": timed_by, "eager_step_avg_us": eager_avg_s * 1e6}
                ex = {"mfma_TFLOPs": ex_flops / avg_s / 1e12, "mfma_peak": ex_peak, "mfma_frac": ex_flops / avg_s / 1e12 / ex_peak,
                      "mfma_unit": "executed bf16 TFLOP/s (dense bf16 peak)" if ex_peak == BF16_PEAK_TFLOPS else "fp32 TFLOP/s (fp32-input MFMA peak)",
                      "hbm_GBps": nbytes / avg_s / 1e9, "hbm_peak": HBM_PEAK_GBS, "hbm_frac": nbytes / avg_s / 1e9 / HBM_PEAK_GBS}
                if planes > 1:
                    ex["what"] = f"fp32 contraction as {int(planes)} bf16 plane products per fp32 product (v_mfma_f32_32x32x16_bf16, fp32 accumulate)"
                    r["algorithmic"] = {"TFLOPs": flops / avg_s / 1e12, "fp32_mfma_peak": FP32_PEAK_TFLOPS, "frac_fp32_mfma": flops / avg_s / 1e12 / FP32_PEAK_TFLOPS}
                    if bound == "mfma":
                        r["achieved_is"] = f"executed bf16 plane products ({int(planes)} per fp32 product), not fp32-equivalent flops"
                r["executed"] = ex
                r["binding"] = {"roof": "mfma" if ex["mfma_frac"] >= ex["hbm_frac"] else "hbm", "frac": max(ex["mfma_frac"], ex["hbm_frac"])}
                return r
            line["roofline"] = roof(dom)
            if split:
                line["config"]["contraction"] = ("fp32 operands as bf16 planes on the bf16 matrix cores: forward three planes / six plane products per fp32 product, "
                                                 + ("backward three planes / six products (MP_BWD_PLANES=3)" if os.environ.get("MP_BWD_PLANES", "2") == "3"
                                                    else "backward (gradients) two planes / three products"))
            # kernels that run on the other streams underneath the step (not candidates for `roofline.kernel`, which is the largest
            # kernel of the step's own chain): reported here with their own fractions instead of being dropped
            side_k = [k for k in kernels if off_path(k)]
            if side_k:
                line["side_stream"] = {k: {kk: vv for kk, vv in roof(k).items() if kk != "kernel"} for k in side_k}
            floors = fps_floors(ts, lib) if world == 1 else {}
            line["named_kernels"], line["kernels_us_per_step"] = kernel_tables(kernels, max(profiled_steps, 1), floors)
        side = world == 1 and not args.no_side_legs and args.path == "harness" and args.dist == "cuboid" and args.encoder == "ssg"
        if side:
            # the drop-in figure next to the harness figure
            k = max(10, min(args.steps, 40))
            dinfo = {}
            ddt, dper, dloss = run_dropin(k, 3, info=dinfo)
            dmed = dper[len(dper) // 2]
            line["dropin_path"] = {"value": args.batch / dmed * 1e3, "unit": "point-clouds/s", "ms_per_step": dmed, "steps": k,
                                   "step_ms_median": dmed, "ms_per_step_mean": ddt / k * 1e3, "final_loss": dloss, "recorded_calls": dinfo,
                                   "what": "train_maskplanner.py:182-227 loop body on the drop-in modules: torch.optim.Adam on all parameters, "
                                           "fresh host batch per step (H2D inside the step), FPS starts drawn per call, compute() -> numpy, loss.item(); "
                                           "[r6] the host batches are padded per batch like the reference's collate pads them (paintnet_ODv1.py:738-747: "
                                           "32 batches, ~30 distinct ground-truth widths, shuffled); model(...) and loss_handler.compute(...) replay graphs "
                                           "they recorded from their own eager code (maskplanner_amd/graphed.py: the loss's ground-truth buffers at "
                                           "capacity buckets of 128 rows; MASKPLANNER_DROPIN_GRAPH=0: launched op by op); `recorded_calls.timed_steps` = "
                                           "how many of the timed calls replayed; value / ms_per_step are the MEDIAN step (the loop follows the host, "
                                           "and on these shared hosts single steps stall for tens of ms: the mean is reported beside it)"}
            try:        # the same loop with the one-word change `torch.optim.Adam(..., fused=True)` of train_maskplanner.py:159
                fdt, fper, floss = run_dropin(k, 3, adam_kwargs={"fused": True})
                fmed = fper[len(fper) // 2]
                line["dropin_path"]["fused_adam"] = {"value": args.batch / fmed * 1e3, "ms_per_step": fmed, "ms_per_step_mean": fdt / k * 1e3,
                                                     "final_loss": floss, "what": "torch.optim.Adam(model.parameters(), lr=..., fused=True)"}
            except Exception as exc:
                line["dropin_path"]["fused_adam"] = {"error": f"{type(exc).__name__}: {exc}"[:200]}
            # PCIe-inclusive: a fresh host batch every step, collated + sampled on the second stream under the previous step.  In a child
            # process: this runtime spreads the streams a process creates over four hardware queues in creation order, and the streamed
            # step keeps three of them busy -- after the legs above (graph recordings create streams of their own) its streams land
            # wherever the count has got to, two of them on one queue (2.44 ms instead of 2.03; NOTEBOOK.md, "hardware-queue count")
            del ts
            torch.cuda.empty_cache()
            line["streamed_inputs"] = streamed_leg(args, k)
            # U-cube clouds: sparse balls => full-scan ball query and heavy padding
            torch.cuda.empty_cache()
            tu = make_harness("ucube", stream=0)
            lib.mp_profiler_collect(None, 0)
            udt, uper, uloss, uprof = run_harness(tu, k, max(args.warmup, 8), profile_every=20)
            uk = collect_kernel_profile(lib)
            unamed, _ = kernel_tables(uk, max(uprof, 1), {})
            line["ucube"] = {"value": args.batch * k / udt, "unit": "point-clouds/s", "ms_per_step": udt / k * 1e3, "steps": k,
                             "step_ms_median": uper[len(uper) // 2] if uper else None, "final_loss": uloss, "named_kernels": unamed}
        if side and args.batch == 32:
            # [r5] the reference's OWN batch size: config=[maskplanner,cuboids_v2,longx_v2] merges to batch_size 64 (configs/maskplanner/cuboids_v2.yaml:12
            # over asymm_chamfer_v9.yaml:3; utils/args.py:77-94; README.md:115) -- harness and drop-in loop
            torch.cuda.empty_cache()
            k = max(10, min(args.steps, 30))
            args.batch = 64
            try:
                tb = make_harness("cuboid", stream=0)
                bdt, bper, bloss, _ = run_harness(tb, k, max(args.warmup, 8), profile_every=None)
                del tb
                torch.cuda.empty_cache()
                ddt, dper, dloss = run_dropin(k, 3)
                dmed = dper[len(dper) // 2]
                line["b64"] = {"harness": {"value": 64 * k / bdt, "unit": "point-clouds/s", "ms_per_step": bdt / k * 1e3, "steps": k,
                                           "step_ms_median": bper[len(bper) // 2] if bper else None, "final_loss": bloss},
                               "dropin_path": {"value": 64 / dmed * 1e3, "unit": "point-clouds/s", "ms_per_step": dmed, "steps": k,
                                               "ms_per_step_mean": ddt / k * 1e3, "final_loss": dloss},
                               "what": "the same step at the reference's own batch size (configs/maskplanner/cuboids_v2.yaml:12: batch_size 64); "
                                       "every head kernel takes 64 rows (no fallback route)"}
            finally:
                args.batch = 32
            torch.cuda.empty_cache()
            line["inference_b1"] = inference_b1(cat, args.points, dev)
            line["dp_overhead"] = dp_overhead_leg()
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(cat, args.points, 1235)
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

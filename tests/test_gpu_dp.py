"""The data-parallel path with TWO ranks on the GPU box's single MI355X (gloo backend between them; tests/dp_worker.py).

RCCL needs one GPU per rank, so the multi-GPU bench cannot be run here; what can be run is everything around the collective:
the rank-sharded batch, SyncBN's per-layer exchanges inside the fused kernels' calls, the bucketed gradient all-reduce, the
factor all-gather and the two-graph step with the exchange outside the graphs.
"""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _run(mode, world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    tmp = tempfile.mkdtemp(prefix="mp_dp_")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    # ([r3] SyncBN levels keep the factorised first layer -- mp_sa_mlp_{fwd,bwd}_gather_ex carry the exchange hook --, so both sides of
    # the comparison run the default route)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), mode, str(r), str(world), port,
                               os.path.join(tmp, f"r{r}.pt")], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(os.path.join(tmp, f"r{r}.pt")) for r in range(world)]


def _close(a, b, what, rtol, atol):
    err = float((a - b).abs().max())
    scale = max(float(b.abs().max()), 1.0)
    assert err <= atol + rtol * scale, f"{what}: {err:.3e} (scale {scale:.3e})"


def test_syncbn_two_ranks_reproduce_the_single_process_global_batch():
    """north_star: loss parity of the data-parallel run with the single-device run.  16 clouds in one process (plain train-mode
    BatchNorm over the whole batch) vs 2 ranks x 8 clouds with SyncBN: predictions of every sample within 1e-5, running
    statistics equal, rank-averaged parameter gradients equal up to the max-pool routing noise."""
    (one,) = _run("syncbn", 1)
    two = _run("syncbn", 2)
    for r in two:
        a, b = r["sl"]
        _close(r["feat"], one["feat"][a:b], "encoder feature", 1e-5, 1e-5)
        # behind the heads' BatchNorm1d layers (statistics over 16 rows of near-identical cuboid features: channels with a tiny
        # variance amplify the 1e-7 differences of the encoder feature) a little more is allowed
        _close(r["out"], one["out"][a:b], "out", 2e-4, 1e-5)
        _close(r["sm"], one["sm"][a:b], "sm_out", 2e-4, 1e-5)
        _close(r["conf"], one["conf"][a:b], "mask_conf", 2e-4, 1e-5)
        for n, v in one["running"].items():
            _close(r["running"][n], v, n, 1e-5, 1e-6)
    def noise_only(n):      # a bias in front of a train-mode BatchNorm: its gradient is rounding noise
        return n.endswith("bias") and ("mlp_convs" in n or n in ("fc1.bias", "fc2.bias", "sm_fc1.bias", "sm_fc2.bias"))
    # the encoder's gradients under a linear functional of its output: the SyncBN backward (global sums in the dZ constants,
    # local sums in dgamma / dbeta, then the average over ranks) reproduces the single-process gradient
    errs = {n: float((two[0]["enc_grads"][n] - g).norm() / g.norm().clamp_min(1e-12)) for n, g in one["enc_grads"].items()
            if not noise_only(n)}
    # (gradients through max-pools are discontinuous in the forward values: a 1e-6 forward difference re-routes the gradient of
    # the few groups whose two largest members are that close, which shows as ~2e-3 here -- the same floor as between the HIP
    # path and the CPU oracle, tests/test_gpu_bf16.py; a wrong SyncBN constant would show as O(1))
    bad = {n: e for n, e in errs.items() if e > 2e-2}
    assert not bad and sorted(errs.values())[len(errs) // 2] < 5e-3, (bad, sorted(errs.values())[len(errs) // 2])
    # through the heads (their BatchNorm1d amplification included): direction and size
    for n, g in one["grads"].items():
        if noise_only(n):
            continue
        _close(two[0]["grads"][n], g, "grad " + n, 2e-2, 2e-5)
        assert torch.equal(two[0]["grads"][n], two[1]["grads"][n])


def test_two_graph_step_replays_under_data_parallelism():
    """The default N > 1 launch path: graphs A (encoder forward) and B (heads, loss, backward) replayed, gradient all-reduce,
    dense Adam and the factor all-gather + Adam launched eagerly after B.  Replicas stay identical and the trajectory tracks
    the kernel-by-kernel data-parallel run."""
    g = _run("dp_graph", 2)
    e = _run("dp_eager", 2)
    assert all(r["graph"] for r in g) and not any(r["graph"] for r in e)
    assert max(r["replica_diff"] for r in g) == 0.0 and max(r["replica_diff"] for r in e) == 0.0, (g[1]["diverged"], e[1]["diverged"])
    lg, le = np.array(g[0]["losses"]), np.array(e[0]["losses"])
    assert np.isfinite(lg).all() and lg[-1] < lg[0]
    # same first loss; then two optimisation walks that differ by the summation order of the dW atomics only (the scatter kernels run
    # in their ordered form: ops.DETERMINISTIC), on 4-cloud batches with train-mode BatchNorm
    assert abs(lg[0] - le[0]) <= 1e-4 * abs(le[0]) and np.allclose(lg[:4], le[:4], rtol=5e-2) and np.allclose(lg, le, rtol=0.1), (lg, le)
    # the weights themselves after two steps: Adam moves every parameter by ~lr = 1e-3 per step; a missed bucket or a stale factor in the
    # deferred exchange would move whole tensors differently, atomics noise flips the update of a few near-zero gradients
    pg, pe = g[0]["params_after_2"], e[0]["params_after_2"]
    off = float(((pg - pe).abs() > 2e-4).float().mean())
    assert off < 0.02, off


def test_four_ranks_replay_the_two_graph_step():
    """The same launch path with FOUR ranks on the one GPU (gloo): the first 8-GPU run must not be the first time more than two
    ranks execute the graph replay + deferred exchange + all-gathered factor Adam."""
    g = _run("dp_graph", 4)
    assert all(r["graph"] for r in g) and not any(r["fell_back"] for r in g)
    assert max(r["replica_diff"] for r in g) == 0.0, g[1]["diverged"]
    lg = np.array(g[0]["losses"])
    assert np.isfinite(lg).all() and lg[-1] < lg[0]


def test_replica_guard_falls_back_to_eager_launches():
    """harness.TrainStep._dp_guard: when the replicas of a graph-replayed data-parallel step differ (here: the test hook says so on
    the second guarded step), every rank drops the graphs, takes rank 0's weights and optimizer state and continues kernel by
    kernel -- replicas identical afterwards, the loss keeps falling."""
    g = _run("dp_graph_trip", 2)
    assert all(r["fell_back"] for r in g) and not any(r["graph"] for r in g)
    assert max(r["replica_diff"] for r in g) == 0.0, g[1]["diverged"]
    lg = np.array(g[0]["losses"])
    assert np.isfinite(lg).all() and lg[-1] < lg[0]


def test_syncbn_step_replays_from_graphs_on_rccl():
    """[r4] SyncBN no longer forces eager launches when the backend is RCCL: the per-layer all-reduces of the library's exchange hook are
    recorded into the step's graphs (tools/syncbn_graph_probe.py: one forced RCCL rank -- the box has one GPU).  The replayed trainer
    records, stays finite and tracks the kernel-by-kernel one (same seeds; fp32 atomics make two runs differ in the last bits, which
    Adam amplifies over the steps)."""
    import json
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for _attempt in range(2):       # (RCCL's start-up in a child of a process that holds the GPU fails now and then on the shared boxes: one retry)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "syncbn_graph_probe.py"), port, "8", "2048"], capture_output=True,
                             text=True, timeout=600, env=env)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    res = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    e, g = res["eager"], res["graph"]
    assert e["sync_bn"] and g["sync_bn"] and not e["recorded"] and g["recorded"], res
    assert e["finite"] and g["finite"]
    le, lg = np.array(e["losses"]), np.array(g["losses"])
    assert np.allclose(lg[:3], le[:3], rtol=2e-3) and np.allclose(lg, le, rtol=8e-2), res       # (the first three steps are eager in both)
    assert lg[-1] < 0.9 * lg[0]


def test_bench_runs_with_two_ranks():
    """bench.py under torch.distributed.run with two ranks (gloo between them, both on the box's one GPU): the contract's JSON line comes
    back with n_gpus = 2.  [r4] Regression: the step bench.py profiles is launched eagerly on EVERY rank -- when rank 0 alone left the
    replay, its bucket all-reduce met the other ranks' factor all-gather (a replayed data-parallel step orders them the other way) and
    the job hung."""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASKPLANNER_DIST_BACKEND="gloo", MASKPLANNER_FAULT_DUMP="240")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--batch", "8", "--points", "2048", "--no-cpu-baseline",
           "--no-side-legs"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=400, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 6 and line["value"] > 0 and np.isfinite(line["final_loss"])
    assert line["config"]["global_batch"] == 16


def test_bench_control_flow_with_eight_ranks():
    """[r5] `bench.py --gpus 8` exactly as the driver launches it (torch.distributed.run, eight ranks) at a reduced batch, the ranks sharing the
    box's one GPU over gloo: rendezvous, rank-sharded batches, the recorded data-parallel step with its replica guard, the eagerly
    profiled step on every rank, MAX-over-ranks timing, one JSON line from rank 0 with n_gpus = 8.  (No 8-GPU box is available to this
    build: the first real run must not be the first time eight ranks execute this control flow.)"""
    import json
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = str(s.getsockname()[1])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASKPLANNER_DIST_BACKEND="gloo", MASKPLANNER_FAULT_DUMP="500")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1", "--master-port", port,
           os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "6", "--warmup", "2", "--batch", "4", "--points", "1024", "--no-cpu-baseline",
           "--no-side-legs"]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-2500:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["steps"] == 6 and line["value"] > 0 and np.isfinite(line["final_loss"])
    assert line["config"]["global_batch"] == 32 and line["config"]["parallelism"] == "dp8"

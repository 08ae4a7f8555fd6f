"""[r5] What the N > 1 LAUNCH PATH costs on one GPU, so that a 1 -> 8 comparison is like for like: the data-parallel step ([r6] graphs A and B with
the bucket all-reduces and the dense Adam recorded into B, then factor all-gather + factor Adam on the optimizer stream; r5 /
MASKPLANNER_DP_COLLECTIVES_GRAPH=0: graphs A, B1, B2 recorded without collectives, everything else launched eagerly -- harness.TrainStep with
dp.exchanging()) driven by ONE forced RCCL rank (every collective runs, each is an identity), next to the two-graph N = 1 step, alternating,
same process.  Prints one JSON line.   python tools/dp_overhead.py [port] [steps] [B] [N]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def main():
    import torch
    import torch.distributed as dist
    import maskplanner_amd.dp as dp
    from maskplanner_amd.harness import TrainStep
    port = int(sys.argv[1]) if len(sys.argv) > 1 else 29633
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    B = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    N = int(sys.argv[4]) if len(sys.argv) > 4 else 5120
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1)

    def build(force):
        dp.FORCE_COLLECTIVES = force
        ts = TrainStep("cuboids", B=B, N=N)
        while ts.use_graph and ts._graph is None:
            ts.step()
        return ts

    def run(ts, k):
        for _ in range(5):
            ts.step()
        torch.cuda.synchronize()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        t0 = time.perf_counter()
        for i in range(k):
            marks[i].record()
            ts.step()
        marks[-1].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / k * 1e3
        per = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(k))
        return per[len(per) // 2], wall

    plain, forced = build(False), build(True)
    assert forced.dp_graph and not plain.dp_graph, "launch paths"
    recorded = bool(forced._dp_recorded)     # [r6] the exchange and the dense Adam are nodes of the backward graph (MASKPLANNER_DP_COLLECTIVES_GRAPH=0: r5's structure)
    assert recorded == (forced._graph_b2 is None), "two graphs with the recorded exchange, three with the eager one"
    res = {"n1_path_ms": [], "dp_path_one_rank_ms": []}
    for _ in range(3):
        dp.FORCE_COLLECTIVES = False
        res["n1_path_ms"].append(run(plain, steps)[0])
        dp.FORCE_COLLECTIVES = True
        res["dp_path_one_rank_ms"].append(run(forced, steps)[0])
    dist.barrier()
    dist.destroy_process_group()
    a, b = sorted(res["n1_path_ms"])[1], sorted(res["dp_path_one_rank_ms"])[1]
    print(json.dumps({"n1_path_ms": a, "dp_path_one_rank_ms": b, "overhead_ms": b - a, "rounds": res, "steps": steps, "B": B, "N": N,
                      "exchange_recorded": recorded,
                      "what": "median step, three alternations: two-graph N = 1 step vs the N > 1 launch path with one forced RCCL rank -- every collective "
                              "issued, each an identity.  [r6] exchange_recorded: the bucket all-reduces and the dense Adam are nodes of the backward "
                              "graph (two graphs, then factor all-gather + factor Adam on the optimizer stream); false: r5's structure (three graphs + "
                              "eager bucket all-reduce, dense Adam, factor all-gather, factor Adam)"}))


if __name__ == "__main__":
    main()

"""Re-time the IMPORTED reference per phase on this container's host cores -- TEST INFRASTRUCTURE, build
container only (/root/reference does not travel): FPS, square_distance, ball query, the model's forward (train mode) and backward, at the
bench configuration (cuboids_v2: B = 32, N = 5120, S = 999, M = 6), three repeats each, median; torch.set_num_threads(<cores>).  The loss
phase is omitted (pytorch3d is not installable here: any figure would time the stand-in, not the reference).

    python oracle/time_reference.py [--batch 32] [--points 5120] [--repeats 3] > profiles/r05_reference_cpu_timing.json
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def med(fn, n):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2], ts


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--points", type=int, default=5120)
    ap.add_argument("--repeats", type=int, default=3)
    a = ap.parse_args()
    import torch
    from oracle import ref_import as R
    from maskplanner_amd import synthetic as syn
    if not R.available():
        raise SystemExit("the reference checkout is only present in the build container")
    cores = os.cpu_count()
    torch.set_num_threads(cores)
    pu = R.pointnet2_utils()
    pc = R.pointnet2_cls_ssg()
    cat = syn.CATEGORIES["cuboids"]
    B, N = a.batch, a.points
    batch = syn.make_batch(1235, B, N, "cuboids", "cuboid")
    xyz = batch["point_cloud"]
    out = {"what": "imported reference (/root/reference), CPU-only torch " + torch.__version__, "cores": cores, "B": B, "N": N, "repeats": a.repeats,
           "phases_s": {}, "runs_s": {}}

    def rec(name, fn):
        m, ts = med(fn, a.repeats)
        out["phases_s"][name] = round(m, 3)
        out["runs_s"][name] = [round(t, 3) for t in ts]
        print(f"{name}: {m:.2f} s", file=sys.stderr)

    with torch.no_grad():
        fps1 = pu.farthest_point_sample(xyz, 512)
        new1 = pu.index_points(xyz, fps1)
        rec("farthest_point_sample 5120->512 + 512->128 (pointnet2_utils.py:65-86)",
            lambda: pu.farthest_point_sample(pu.index_points(xyz, pu.farthest_point_sample(xyz, 512)), 128))
        rec("square_distance [B,512,5120] (:21-42)", lambda: pu.square_distance(new1, xyz))
        fps2 = pu.farthest_point_sample(new1, 128)
        new2 = pu.index_points(new1, fps2)
        rec("query_ball_point SA1 + SA2 (:89-109)", lambda: (pu.query_ball_point(0.2, 32, xyz, new1), pu.query_ball_point(0.4, 64, new1, new2)))
    torch.manual_seed(0)
    model = pc.PointNet2Regressor_StrokeMasks(outdim=12, outdim_orient=12, weight_orient=0.25, out_vectors=cat.out_vectors, hidden_size=(1024, 1024),
                                              pred_stroke_masks=True, n_stroke_masks=cat.max_n_strokes, mask_confidence_scores=True).train()
    x = xyz.permute(0, 2, 1).contiguous()
    state = {}

    def fwd():
        state["outs"] = model(x)
    rec("model forward, train mode (pointnet2_cls_ssg.py:297-344)", fwd)

    def fwd_bwd():
        model.zero_grad()
        o, sm, conf, _ = model(x)
        (o.square().mean() + sm.square().mean() + conf.square().mean()).backward()
    m, ts = med(fwd_bwd, a.repeats)
    out["phases_s"]["forward + backward of a quadratic functional of the outputs (no loss_handler)"] = round(m, 3)
    out["runs_s"]["forward + backward"] = [round(t, 3) for t in ts]
    out["point_clouds_per_s_forward_backward_without_loss"] = round(B / m, 3)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

"""fp64 arbiter for the TRAIN-mode step (north_star: predictions and loss within 1e-5 fp32).

Two fp32 implementations of the same train-mode step -- the HIP path and the CPU oracle -- differ by more than 1e-5 behind the
heads' BatchNorm1d layers and in the gradients behind the max-pools, and each round explained that as amplification of fp32
rounding (tiny batch variances; arg-max re-routing).  This file turns the explanation into a measurement: `oracle/torch_ref.py`
is evaluated a third time in float64 (same discrete decisions: FPS, ball query, nearest neighbours, assignment are taken on the
float32 values; everything else in double), and every output, the loss and EVERY parameter gradient of the HIP path must be at
as close to that arbiter as the fp32 oracle is:

    || HIP - f64 ||  <=  factor * || oracle_fp32 - f64 ||  + floor         factor 2 (forward values, loss), 3 (gradients)

so whatever distance remains between HIP and the fp32 oracle is the fp32 arithmetic's own distance from the exact function, not a
kernel's.  Where the arbiter proves a tight bound directly (encoder feature, predictions, loss) that bound is asserted as well.

[r3] measured (relative L2 distance from the fp64 evaluation; HIP | fp32 oracle):
    B = 32, N = 5120:  feature 1.5e-6 | 5.4e-6   out 9.0e-6 | 3.0e-5   sm_out 8.2e-6 | 2.9e-5   loss 1.5e-7 | 2.3e-7
                       parameter gradients 1.3e-2 .. 1.6e-2 | 2.2e-2 .. 2.9e-2
    B = 8,  N = 1024:  feature 1.4e-6 | 3.5e-6   out 1.0e-5 | 2.7e-5   loss 3.6e-7 | 2.8e-8   gradients 7e-3 .. 9e-3 | 3.5e-3 .. 5e-3
The forward values of the HIP path are CLOSER to the exact function than the fp32 oracle's (fp64 BatchNorm sums, split-plane
contractions); the gradients of both sit ~1e-2 away because a gradient through a max-pool is discontinuous in the forward values
(a 1e-7 forward difference re-routes the few groups whose two largest members are that close): which of the two fp32 paths lands
nearer is a matter of which ties it happens to flip (HIP at B = 32, the oracle at B = 8), hence the factor 3 there.
[r4] tests/test_gpu_routing.py settles the gradients: with the discrete decisions of the HIP path (pool arg-max, nearest neighbours, ReLU
masks) imposed on the oracle, every HIP gradient is within 0.5e-4 .. 4.5e-4 of the fp64 evaluation and at most twice the fp32 oracle's
own distance (factor 2, no luck involved); the factor-3 check below stays as the unconditioned sanity check it always was.
"""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def _oracle_step(model_sd, batch, cat, cfg, dtype):
    from oracle import torch_ref as T
    sd = {}
    for k, v in model_sd.items():
        t = v.detach().clone()
        if t.dtype.is_floating_point:
            t = t.to(dtype)
            if "running" not in k:
                t.requires_grad_(True)
        sd[k] = t
    starts = [s.numpy() for s in batch["fps_start"]]
    pc = batch["point_cloud"].to(dtype)
    out, sm, conf, feat = T.strokemasks_forward(sd, pc, starts, train=True, out_vectors=cat.out_vectors, n_masks=cat.max_n_strokes,
                                                return_feat=True)
    loss = T.asymm_v6_loss(out, batch["traj"].to(dtype), sm, conf, batch["stroke_ids"], batch["traj_as_pc"].to(dtype), cfg)
    loss.backward()
    grads = {k: v.grad.detach() for k, v in sd.items() if v.requires_grad and v.grad is not None}
    return dict(feat=feat.detach(), out=out.detach(), sm=sm.detach(), conf=conf.detach(), loss=loss.detach(), grads=grads)


def _hip_step(model, batch, cfg):
    from maskplanner_amd import ops
    from maskplanner_amd import pointnet2_utils as pu
    from maskplanner_amd.loss_handler import LossHandler
    lh = LossHandler(["asymm_v6_chamfer_with_stroke_masks"], cfg)
    det, ops.DETERMINISTIC = ops.DETERMINISTIC, True
    try:
        with pu.fps_start_override(batch["fps_start"]):
            feat = model.encode(batch["point_cloud"].cuda().permute(0, 2, 1))
            out, sm, conf, _ = model.heads(feat)
        loss = lh.compute(return_list=False, y_pred=out, y=batch["traj"].cuda(), pred_stroke_masks=sm, mask_scores=conf,
                          seg_logits=None, stroke_ids=batch["stroke_ids"], traj_as_pc=batch["traj_as_pc"])
        loss.backward()
    finally:
        ops.DETERMINISTIC = det
    grads = {n: p.grad.detach().cpu() for n, p in model.named_parameters() if p.grad is not None}
    return dict(feat=feat.detach().cpu(), out=out.detach().cpu(), sm=sm.detach().cpu(), conf=conf.detach().cpu(),
                loss=loss.detach().cpu(), grads=grads)


def _noise_only(name):
    """Parameters whose exact gradient is 0 -- what an fp32 path returns there is rounding noise: a bias in front of a train-mode
    BatchNorm (the batch mean removes it), and the last BatchNorm shift of the encoder (it moves the global feature of every
    sample by the same vector, which fc1 / sm_fc1 -> train-mode BatchNorm1d remove; the fp64 evaluation returns ~1e-16)."""
    return name.endswith("bias") and ("mlp_convs" in name or name in ("fc1.bias", "fc2.bias", "sm_fc1.bias", "sm_fc2.bias", "sa3.mlp_bns.2.bias"))


def _arbitrate(hip, o32, o64, report):
    """-> list of violations.  Relative L2 distances from the arbiter, per tensor."""
    bad = []

    def one(name, h, a, b, factor=2.0, ulps=3e-7):
        scale = float(b.norm())
        eh, eo = float((h.double() - b).norm()), float((a.double() - b).norm())
        floor = ulps * scale + 1e-12          # a few ulp of fp32 on the tensor's own scale: below it "twice as close" means nothing
        report[name] = dict(hip=eh / max(scale, 1e-30), oracle_fp32=eo / max(scale, 1e-30), scale=scale)
        if eh > factor * eo + floor:
            bad.append((name, eh / max(scale, 1e-30), eo / max(scale, 1e-30)))
    for k in ("feat", "out", "sm", "conf"):
        one(k, hip[k], o32[k], o64[k])
    one("loss", hip["loss"], o32["loss"], o64["loss"], ulps=1e-6)     # a scalar: one path's rounding can cancel to ~0 by luck
    for n, g64 in o64["grads"].items():
        if _noise_only(n):
            # exact value 0: compare the noise itself against the size of the sibling weight gradient's entries
            ref = float(o64["grads"][n.replace("bias", "weight")].abs().mean()) if n.replace("bias", "weight") in o64["grads"] else 1.0
            report["grad " + n] = dict(hip=float(hip["grads"][n].abs().max()), oracle_fp32=float(o32["grads"][n].abs().max()), exact=float(g64.abs().max()))
            if float(hip["grads"][n].abs().max()) > 1e-2 * max(ref, 1e-12) + 10 * float(o32["grads"][n].abs().max()):
                bad.append(("grad " + n, float(hip["grads"][n].abs().max()), float(o32["grads"][n].abs().max())))
            continue
        one("grad " + n, hip["grads"][n], o32["grads"][n], g64, factor=3.0)
    return bad


@pytest.mark.parametrize("B,N,hidden,seed", [(32, 5120, (256, 256), 17), (8, 1024, (64, 64), 1515)])
def test_train_step_is_as_close_to_fp64_as_the_fp32_oracle(oracle, B, N, hidden, seed):
    """BASELINE configs[1] exactly (cuboids, N = 5120, B = 32; and the 8-cloud shape of fixture g15): train-mode BatchNorm in all
    13 layers, dropout off, forward + asymm_v6 loss + backward.  Every output, the loss and every parameter gradient of the HIP
    path: at most twice the fp32 oracle's distance from the fp64 evaluation."""
    from maskplanner_amd import pointnet2_cls_ssg as pc
    from maskplanner_amd import synthetic as syn
    from maskplanner_amd.loss_handler import maskplanner_loss_config
    cat = syn.CATEGORIES["cuboids"]
    batch = syn.make_batch(seed, B, N, "cuboids", "cuboid")
    torch.manual_seed(4)
    model = pc.maskplanner_model(cat, hidden_size=hidden)
    model.dropout.p = 0.0
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = maskplanner_loss_config()
    hip = _hip_step(model.cuda().train(), batch, cfg)
    o32 = _oracle_step(sd0, batch, cat, cfg, torch.float32)
    o64 = _oracle_step(sd0, batch, cat, cfg, torch.float64)
    report = {}
    bad = _arbitrate(hip, o32, o64, report)
    worst = sorted(((v["hip"], v.get("oracle_fp32", 0.0), k) for k, v in report.items() if "exact" not in v), reverse=True)[:8]
    out_dir = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out_dir):      # kept next to the profiles when run through tools/ (scratch on the GPU box)
        with open(os.path.join(out_dir, f"arbiter_B{B}_N{N}.json"), "w") as f:
            json.dump(report, f, indent=1)
    assert not bad, f"further from fp64 than the fp32 oracle allows: {bad[:10]}; worst: {worst}"
    # what the arbiter proves directly, on the scale of each quantity: the contract's 1e-5 for the encoder feature and the loss,
    # 2e-5 (relative L2) for the predictions behind the heads' train-mode BatchNorm1d -- where the fp32 oracle is at 3e-5
    assert report["feat"]["hip"] <= 1e-5 and report["loss"]["hip"] <= 1e-5, (report["feat"], report["loss"])
    assert max(report[k]["hip"] for k in ("out", "sm", "conf")) <= 2e-5, {k: report[k] for k in ("out", "sm", "conf")}
    # the fp32 oracle itself sits this far from the exact function behind the heads' BatchNorm1d: the looser direct tolerances
    # of test_gpu_modules.py::test_full_size_train_step_b32_vs_oracle are this number, not a kernel property
    print("arbiter: out", report["out"], "sm", report["sm"], "loss", report["loss"], "worst", worst[:4])

"""Max error of the full-size (B=4, N=5120, train mode) model output against the fp32 CPU oracle -- the quantity
tests/test_gpu_modules.py::test_full_size_forward_loss_backward_vs_oracle bounds."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from maskplanner_amd import pointnet2_cls_ssg as pc, pointnet2_utils as pu, synthetic as syn
from oracle import torch_ref as T
B, N = 4, 5120
cat = syn.CATEGORIES["cuboids"]
for seed in (7, 8, 9):
    batch = syn.make_batch(seed, B, N, "cuboids", "cuboid")
    torch.manual_seed(3)
    model = pc.maskplanner_model(cat, hidden_size=(256, 256))
    model.dropout.p = 0.0
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    model = model.cuda().train()
    with pu.fps_start_override(batch["fps_start"]), torch.no_grad():
        out, sm, conf, _ = model(batch["point_cloud"].cuda().permute(0, 2, 1))
    with torch.no_grad():
        o_out, o_sm, o_conf = T.strokemasks_forward(sd, batch["point_cloud"], [s.numpy() for s in batch["fps_start"]], train=True,
                                                    out_vectors=cat.out_vectors, n_masks=cat.max_n_strokes)
    e = (out.cpu() - o_out).abs().max().item()
    print(f"seed {seed}: max |out - oracle| = {e:.3e}  (scale {o_out.abs().max().item():.3f})")

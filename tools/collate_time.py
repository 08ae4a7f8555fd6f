"""Host-to-device cost of one batch through maskplanner_amd.collate (B=32 cuboids items: cloud 5120x3, ~999 segments x 24,
~2900 poses x 6): wall-clock per batch including the pad kernels, and the PCIe-inclusive step rate it implies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maskplanner_amd.collate import Paintnet_ODv1_CollateBatch
rng = np.random.default_rng(0)
items = []
for b in range(32):
    n_seg, n_pts = int(rng.integers(900, 1000)), int(rng.integers(2700, 2950))
    items.append({"point_cloud": rng.normal(size=(5120, 3)).astype(np.float32), "traj": rng.normal(size=(n_seg, 24)).astype(np.float32),
                  "traj_as_pc": rng.normal(size=(n_pts, 6)).astype(np.float32), "stroke_ids": np.sort(rng.integers(0, 6, size=n_seg)),
                  "stroke_ids_as_pc": np.sort(rng.integers(0, 6, size=n_pts)), "dirname": str(b), "n_strokes": 6})
collate = Paintnet_ODv1_CollateBatch({"load_extra_data": []})
nbytes = sum(v.nbytes for it in items for v in it.values() if isinstance(v, np.ndarray))
for _ in range(5): collate(items)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): batch = collate(items)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 50 * 1e3
print(f"collate on device: {ms:.3f} ms per batch of 32 ({nbytes / 1e6:.1f} MB of host arrays, {nbytes / ms / 1e6:.1f} GB/s end to end)")
step = float(sys.argv[1]) if len(sys.argv) > 1 else 3.6
print(f"PCIe-inclusive, not overlapped: {32 / (step + ms) * 1e3:.0f} point-clouds/s at a {step} ms step (resident inputs: {32 / step * 1e3:.0f})")

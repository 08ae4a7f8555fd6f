"""Time mp_mask_match_f32 at the stroke-mask shapes of the bench configurations (HIP events, 200 launches)."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from maskplanner_amd import ops  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    for name, B, M, S, n_ids in (("cuboids", 32, 6, 1280, 6), ("windows", 32, 6, 1280, 6), ("shelves", 32, 41, 1266, 41),
                                 ("containers", 32, 33, 1333, 33)):
        pred = torch.from_numpy((rng.normal(size=(B, M, S)) * 2).astype(np.float32)).cuda()
        ids = torch.from_numpy(rng.integers(0, n_ids, size=(B, S)).astype(np.float32)).cuda()
        for _ in range(5):
            ops.mask_match(pred, ids)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(200):
            ops.mask_match(pred, ids)
        e1.record()
        torch.cuda.synchronize()
        if "--phases" in sys.argv:   # library built with -DMP_MM_TIMING (tools/mask_match_phases.sh)
            u = ops.mask_match(pred, ids)[1].cpu().numpy()[:, 48:53]
            print("   phases us (stage+bitmap, unique+rank, sums, cost, lap):", np.round(u.mean(0), 2), "max", np.round(u.max(0), 2))
        print(f"{name:11s} B={B} M={M} S={S} ids={n_ids}: {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us per call (launch included)")


if __name__ == "__main__":
    main()

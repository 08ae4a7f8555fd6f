"""Forward + backward time of the multi-scale first level (N = 10240, B = 32, upstream widths) with the 96-wide interior layer carried
as 128 (MASKPLANNER_WIDEN_INTERIOR, default) and on the tiled kernels."""
import sys
import numpy as np
import torch

sys.path.insert(0, ".")
from maskplanner_amd import pointnet2_utils as pu, sa_mlp, synthetic as syn  # noqa: E402


def main():
    rng = np.random.default_rng(0)
    B, N = 32, 10240
    xyz = torch.from_numpy(syn.point_cloud(rng, B, N, "cuboid")).cuda().permute(0, 2, 1).contiguous()
    torch.manual_seed(0)
    msg = pu.PointNetSetAbstractionMsg(512, [0.1, 0.2, 0.4], [16, 32, 128], 0, [[32, 32, 64], [64, 64, 128], [64, 96, 128]]).cuda().train()
    for widen in (False, True, False, True):
        sa_mlp.WIDEN_INTERIOR = widen
        ts = []
        for it in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            _, out = msg(xyz, None)
            out.square().sum().backward()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        print(f"widen={widen}: {sorted(ts)[len(ts) // 2]:.3f} ms (fwd + bwd of the level, median of 6)")
        if "--kernels" in sys.argv:      # the library's per-launch timing hooks, one pass
            import ctypes
            from maskplanner_amd import _lib
            lib = _lib.load()
            lib.mp_profiler_collect(None, 0)
            lib.mp_profiler_enable(1)
            _, out = msg(xyz, None)
            out.square().sum().backward()
            torch.cuda.synchronize()
            lib.mp_profiler_enable(0)
            buf = ctypes.create_string_buffer(1 << 16)
            if lib.mp_profiler_collect(buf, len(buf)) > 0:
                rows = [l.split("\t") for l in buf.value.decode().strip().split("\n")]
                for name, calls, ms, _, _ in sorted(rows, key=lambda r: -float(r[2])):
                    print(f"      {float(ms) * 1e3:8.1f} us  x{calls}  {name}")


if __name__ == "__main__":
    main()

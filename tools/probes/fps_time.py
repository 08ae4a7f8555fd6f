"""[r6] fps_kernel timing at the bench shapes (B = 32: N = 5120 -> 512, N = 512 -> 128, N = 10240 -> 512; B = 1: N = 5120 -> 512), HIP events, median of 30."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from maskplanner_amd import ops, synthetic
import numpy as np

def timed(fn, n=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2], t[0]

rng = np.random.default_rng(0)
for B, N, S in ((32, 5120, 512), (32, 512, 128), (32, 10240, 512), (1, 5120, 512)):
    xyz = torch.from_numpy(synthetic.point_cloud(rng, B, N, "cuboid")).cuda()
    start = torch.zeros(B, dtype=torch.long, device="cuda")
    med, mn = timed(lambda: ops.fps(xyz, S, start))
    print(f"MP_FPS_SHAPE={os.environ.get('MP_FPS_SHAPE', '-')}  B={B} N={N} S={S}: median {med:.1f} us, min {mn:.1f} us", flush=True)

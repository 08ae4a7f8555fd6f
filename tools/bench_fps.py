"""Time farthest point sampling (B=32, N=5120 -> 512 and N=512 -> 128) on the GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maskplanner_amd import ops, synthetic as syn
rng = np.random.default_rng(0)
for (B, N, S) in [(32, 5120, 512), (32, 512, 128), (32, 10240, 512)]:
    xyz = torch.from_numpy(syn.point_cloud(rng, B, N, "cuboid")).cuda()
    start = torch.zeros(B, dtype=torch.long).cuda()
    for _ in range(3): ops.fps(xyz, S, start)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(20): ops.fps(xyz, S, start)
    b.record(); torch.cuda.synchronize()
    us = a.elapsed_time(b) / 20 * 1e3
    print(f"MP_FPS_THREADS={os.environ.get('MP_FPS_THREADS','-')}  B={B} N={N} S={S}: {us:8.1f} us  = {us / S:6.3f} us/step")

"""Stand-alone time of the ball queries of the encoders (HIP events, median of 30): SA1 / SA2 of the single-scale encoder on cuboid and ucube
clouds, and the multi-scale levels of config 5 as one scan vs one call per radius; MP_BQ_LEGACY=1 times the one-query-per-wave kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from maskplanner_amd import ops, synthetic as syn

def med(fn, n=30):
    for _ in range(5):
        fn()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
    return t[len(t) // 2]

B = 32
for dist in ("cuboid", "ucube"):
    for N, S, radii, Ks in ((5120, 512, [0.2], [32]), (512, 128, [0.4], [64]), (10240, 512, [0.1, 0.2, 0.4], [16, 32, 128]), (512, 128, [0.2, 0.4, 0.8], [32, 64, 128])):
        rng = np.random.default_rng(1)
        xyz = torch.from_numpy(syn.point_cloud(rng, B, N, dist)).cuda()
        start = torch.zeros(B, dtype=torch.long, device="cuda")
        _, new_xyz = ops.fps(xyz, S, start, return_xyz=True)
        one = med(lambda: ops.ball_query_multi(radii, Ks, xyz, new_xyz))
        per = med(lambda: [ops.ball_query(r, K, xyz, new_xyz) for r, K in zip(radii, Ks)])
        os.environ["MP_BQ_LEGACY"] = "1"
        old = med(lambda: [ops.ball_query(r, K, xyz, new_xyz) for r, K in zip(radii, Ks)])
        del os.environ["MP_BQ_LEGACY"]
        print(f"{dist:7s} N={N:5d} S={S} radii={radii}: one scan {one:7.1f} us | one call per radius {per:7.1f} us | r4 kernel {old:7.1f} us")

"""Time the batched device LAP against scipy on the host (999 x ~900 Euclidean costs, B = 32)."""
import os
import sys
import time

import numpy as np
import torch
from scipy.optimize import linear_sum_assignment

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maskplanner_amd import ops  # noqa: E402

rng = np.random.default_rng(0)
B, S, D = 32, 999, 24
out = torch.from_numpy(rng.uniform(-1, 1, (B, S, D)).astype(np.float32)).cuda()
tg = [torch.from_numpy(rng.uniform(-1, 1, (int(n), D)).astype(np.float32)).cuda() for n in rng.integers(850, 986, B)]
costs = [torch.cdist(out[b], t) for b, t in enumerate(tg)]
ops.lsap(costs[:2])
torch.cuda.synchronize()
t0 = time.perf_counter()
pairs, status = ops.lsap(costs)
torch.cuda.synchronize()
t1 = time.perf_counter()
host = [c.cpu().numpy() for c in costs]
t2 = time.perf_counter()
ref = [linear_sum_assignment(c) for c in host[:4]]
t3 = time.perf_counter()
ok = all(np.array_equal(pairs[b][0].cpu().numpy(), ref[b][0]) and np.array_equal(pairs[b][1].cpu().numpy(), ref[b][1]) for b in range(4))
print(f"device LAP, B={B}: {1e3 * (t1 - t0):.1f} ms for the batch; scipy: {1e3 * (t3 - t2) / 4:.1f} ms per sample "
      f"({1e3 * (t3 - t2) / 4 * B:.0f} ms for the batch, serial); identical on the 4 checked samples: {ok}")

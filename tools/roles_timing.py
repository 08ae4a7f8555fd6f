"""Per-phase cycle sums of bwd_roles_kernel from a -DMP_ROLES_TIMING build (tools/roles_timing.sh): SA2-shaped level, forward + backward a few times."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import sa_mlp, _lib

lib = _lib.load()
raw = ctypes.CDLL(os.environ["MASKPLANNER_HIP_LIB"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
torch.manual_seed(0)
S, K, C0, mlp = 128, 64, 131, [128, 128, 256]
convs, bns = torch.nn.ModuleList(), torch.nn.ModuleList()
last = C0
for c in mlp:
    convs.append(torch.nn.Conv2d(last, c, 1)); bns.append(torch.nn.BatchNorm2d(c)); last = c
convs, bns = convs.cuda(), bns.cuda()
x = torch.randn(B, S, K, C0).cuda().requires_grad_(True)
g = torch.randn(B, S, mlp[-1]).cuda()
buf = (ctypes.c_ulonglong * 16)()
n = 0
for it in range(8):
    if it == 3:
        torch.cuda.synchronize(); raw.mp_debug_roles_times(buf)
    y = sa_mlp.shared_mlp_max(x, convs, bns, layout="feats_first")
    (y * g).sum().backward()
    n += it >= 3
torch.cuda.synchronize()
raw.mp_debug_roles_times(buf)
waves = n * (B * S * K // 1024) * 4          # waves of each kind over the timed launches
chunks = 64
names = {0: ["products", "epilogue", "barrier wait", "loop total", "staging", "loop head"], 1: ["products", "staging", "barrier wait", "loop total", "load issue", "loop head"]}
for kind in (0, 1):
    v = [buf[kind * 8 + i] / waves / chunks for i in range(8)]
    print(("dX" if kind == 0 else "dW") + f" wave, cycles per chunk (B={B}): " + "  ".join(f"{names[kind][i]} {v[i]:.0f}" for i in range(6)))

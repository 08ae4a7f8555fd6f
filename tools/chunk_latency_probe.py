"""Is an iteration of the position-stream kernels bound by the memory round trip of its one-chunk-ahead prefetch?
The same per-workgroup work (1024 positions of SA2 / SA1 shapes) at B = 32 (one workgroup per CU, HBM loaded), 8, 4, 2 (a few CUs busy,
everything L2 / MALL resident): if the kernel time falls with B although every workgroup does the same 64 iterations, the iteration is
waiting for memory, not for the matrix pipe or the VALU."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd import sa_mlp, _lib

lib = _lib.load()


def collect():
    buf = ctypes.create_string_buffer(1 << 16)
    n = lib.mp_profiler_collect(buf, len(buf))
    out = {}
    if n > 0:
        for line in buf.value.decode().strip().split("\n"):
            name, calls, ms, flops, nbytes = line.split("\t")
            out[name] = float(ms) / int(calls) * 1e3
    return out


torch.manual_seed(0)
for S, K, C0, mlp in ((128, 64, 131, [128, 128, 256]), (512, 32, 3, [64, 64, 128])):
    for B in ([int(a) for a in sys.argv[1:]] or (32, 8, 4, 2)):
        convs, bns = torch.nn.ModuleList(), torch.nn.ModuleList()
        last = C0
        for c in mlp:
            convs.append(torch.nn.Conv2d(last, c, 1)); bns.append(torch.nn.BatchNorm2d(c)); last = c
        convs, bns = convs.cuda(), bns.cuda()
        x = torch.randn(B, S, K, C0).cuda().requires_grad_(C0 != 3)
        g = torch.randn(B, S, mlp[-1]).cuda()
        for it in range(6):
            if it == 2:
                torch.cuda.synchronize(); lib.mp_profiler_enable(1)
            y = sa_mlp.shared_mlp_max(x, convs, bns, layout="xyz_first" if C0 == 3 else "feats_first")
            (y * g).sum().backward()
        torch.cuda.synchronize(); lib.mp_profiler_enable(0)
        t = collect()
        print(f"S={S} K={K} B={B}: " + "  ".join(f"{k.split('(')[0][:34]} {v:.1f}" for k, v in sorted(t.items(), key=lambda kv: -kv[1])[:7]))

"""Practical ceiling of a 1:1 read/write stream on this device: torch's D2D copy (and an elementwise a*b+c -> out: 3 reads, 1 write) of
Adam-sized buffers, next to mp_adam_lowrank_f32 alone on the largest head matrix.  usage: python tools/copy_rate.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
def timeit(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
O, I = 11988, 1024
a = torch.randn(3, O, I, device="cuda"); b = torch.empty_like(a)
us = timeit(lambda: b.copy_(a)); print(f"copy 3 x [{O},{I}] fp32: {us:.1f} us = {2 * a.numel() * 4 / us / 1e6:.2f} TB/s (read + write)")
x, y, z = torch.randn(O, I, device="cuda"), torch.randn(O, I, device="cuda"), torch.randn(O, I, device="cuda"); o = torch.empty_like(x)
us = timeit(lambda: torch.addcmul(z, x, y, out=o)); print(f"addcmul (3 reads, 1 write): {us:.1f} us = {4 * x.numel() * 4 / us / 1e6:.2f} TB/s")
from maskplanner_amd import _lib
lib = _lib.load()
p, m, v = torch.randn(O, I, device="cuda"), torch.zeros(O, I, device="cuda"), torch.zeros(O, I, device="cuda")
xf, gf = torch.randn(32, I, device="cuda"), torch.randn(32, O, device="cuda")
st = torch.cuda.current_stream().cuda_stream
def adam():
    rc = lib.mp_adam_lowrank_f32(p.data_ptr(), m.data_ptr(), v.data_ptr(), xf.data_ptr(), gf.data_ptr(), 32, O, I, 1.0, 1e-4, 0.9, 0.999, 1e-8, 3, None, st)
    assert rc == 0, rc
us = timeit(adam); print(f"mp_adam_lowrank_f32 [{O},{I}], 32 factor rows, alone: {us:.1f} us = {6 * O * I * 4 / us / 1e6:.2f} TB/s (p, m, v read + written)")

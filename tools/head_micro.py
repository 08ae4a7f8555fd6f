"""Micro-benchmark of the head-block kernels (csrc/head_linear.hip) against the launches they replace: back-to-back launches between two
events, per-launch time.  usage: head_micro.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from maskplanner_amd import _lib, ops, factor_heads as fh
lib = _lib.load()
dev = "cuda"
def timeit(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n
st = torch.cuda.current_stream().cuda_stream
for B, I, O in [(32, 1024, 1024), (32, 1024, 16384), (32, 128, 1024), (32, 128, 16384), (32, 1024, 11988), (32, 512, 1024), (32, 256, 1024)]:
    lin = torch.nn.Linear(I, O).to(dev); bn = torch.nn.BatchNorm1d(O).to(dev)
    x = torch.randn(B, I, device=dev); y = torch.empty(B, O, device=dev); z = torch.empty(B, O, device=dev); stats = torch.empty(2, O, device=dev)
    g = torch.randn(B, O, device=dev); dz = torch.empty(B, O, device=dev); gx = torch.zeros(B, I, device=dev); gg = torch.empty(O, device=dev); gb = torch.empty(O, device=dev)
    def fwd_bn():
        lib.mp_head_block_fwd_f32(x.data_ptr(), lin.weight.data_ptr(), lin.bias.data_ptr(), B, I, O, 1, 1, 0.1, 1e-5, bn.weight.data_ptr(), bn.bias.data_ptr(),
                                  bn.running_mean.data_ptr(), bn.running_var.data_ptr(), z.data_ptr(), y.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), 0.0, None, 0, st)
    def fwd_plain():
        lib.mp_head_block_fwd_f32(x.data_ptr(), lin.weight.data_ptr(), lin.bias.data_ptr(), B, I, O, 0, 0, 0.0, 0.0, None, None, None, None, None, y.data_ptr(), None, None, 0.0, None, 0, st)
    def bwd():
        lib.mp_head_block_bwd_f32(g.data_ptr(), y.data_ptr(), z.data_ptr(), lin.weight.data_ptr(), B, I, O, 1, bn.weight.data_ptr(), stats[0].data_ptr(), stats[1].data_ptr(), 0.0,
                                  dz.data_ptr(), gg.data_ptr(), gb.data_ptr(), gx.data_ptr(), st)
    with torch.no_grad():
        t_lin = timeit(lambda: F.linear(x, lin.weight, lin.bias))
        t_rows = timeit(lambda: ops.bn_relu_rows(z, bn))
        t_dx = timeit(lambda: torch.mm(dz, lin.weight))
        fwd_bn()
        r = [timeit(fwd_bn), timeit(fwd_plain)]
        if O <= 4096: r.append(timeit(bwd))
    print(f"B{B} I{I} O{O}: F.linear {t_lin:.1f} bn_relu_rows {t_rows:.1f} mm(dz,W) {t_dx:.1f} | head_fwd bn {r[0]:.1f} plain {r[1]:.1f}" + (f" head_bwd {r[2]:.1f}" if len(r) > 2 else ""))


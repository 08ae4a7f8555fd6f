"""Streamed-input steps for a kernel trace: python tools/stream_trace_run.py [0|1 = collation off/on] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from maskplanner_amd.harness import TrainStep
on = (sys.argv[1] if len(sys.argv) > 1 else "1") == "1"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
ts = TrainStep("cuboids", B=32, N=5120, stream_batches=4)
while ts._graph is None:
    ts.step()
if not on:
    st = ts._stream
    st.collate_cloud = lambda: (st.stage["point_cloud"], st.stage_starts)
    st.collate_targets = lambda: None
    st.collated = lambda *e: None
    st.prefetch = lambda: None
for _ in range(n):
    ts.step()
torch.cuda.synchronize()

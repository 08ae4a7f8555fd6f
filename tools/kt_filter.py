import json,sys
d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][0])
k=d["kernels_us_per_step"]
print(round(d["ms_per_step"],3), round(sum(k.values())))
for n,v in k.items():
    if any(t in n for t in ("group", "gathered", "bwd_first", "0, 4", "fwd_chunk", "rc_stats", "dw_ci4")): print("   ", round(v,1), n[:80])

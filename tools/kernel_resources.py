"""VGPR / AGPR / scratch / occupancy / LDS of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage), names demangled.
usage: kernel_resources.py <file.hip> [name filter regex]"""
import re, subprocess, sys, os
src = os.path.abspath(sys.argv[1])
flt = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
cmd = ["hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-munsafe-fp-atomics", "--offload-arch=gfx950",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kres.o"]
txt = subprocess.run(cmd, capture_output=True, text=True, cwd=os.path.dirname(os.path.abspath(src))).stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]
names = [b.split("\n")[0].split(" [")[0].strip() for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
for b, d in zip(blocks, dem):
    d = d.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    if flt and not flt.search(d):
        continue
    g = lambda k: (re.search(re.escape(k) + r": (\d+)", b) or [0, "-1"])[1]
    vals = [g("VGPRs"), g("AGPRs"), g("ScratchSize [bytes/lane]"), g("Occupancy [waves/SIMD]"), g("LDS Size [bytes/block]")]
    print("%-78s vgpr %4s agpr %4s scratch %4s occ %2s lds %7s" % (d[:78], *vals))

import csv, collections, glob, sys
rows = list(csv.DictReader(open(glob.glob(sys.argv[1] + '/*/*_kernel_trace.csv')[0])))
agg = collections.OrderedDict()
for r in rows:
    k = (r['Kernel_Name'], r['Grid_Size_X'], r['Grid_Size_Y'], r['Grid_Size_Z'])
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    agg.setdefault(k, []).append(d)
tot = 0
filt = sys.argv[2].split(',') if len(sys.argv) > 2 else None
for k, v in agg.items():
    if filt and not any(f in k[0] for f in filt):
        continue
    tot += min(v)
    name = k[0].replace('(anonymous namespace)::', '')
    print(f"{min(v):9.1f} us  n={len(v):3d}  grid=({k[1]},{k[2]},{k[3]})  {name[:90]}")
print("total", round(tot, 1))

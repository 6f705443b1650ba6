#!/usr/bin/env python3
"""Per-(kernel, grid) average durations from a rocprofv3 --kernel-trace CSV (first quarter of launches skipped as warm-up)."""
import csv, collections, sys
rows = csv.DictReader(open(sys.argv[1]))
d = collections.defaultdict(list)
for r in rows:
    d[(r['Kernel_Name'][:44], r['Grid_Size_X'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
tot = sum(sum(v) for v in d.values())
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v2 = v[len(v) // 4:]
    if sum(v) / tot < 0.002:
        continue
    print(f"{k[0]:44s} grid {k[1]:>8s} n {len(v):4d} avg {sum(v2)/len(v2):8.1f} us min {min(v):8.1f}  share {100*sum(v)/tot:5.2f}%")

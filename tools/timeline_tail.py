#!/usr/bin/env python3
"""The last N microseconds of one replayed step from a rocprofv3 kernel_trace.csv, kernel by kernel (start, duration, #running beside it).
tools/timeline_tail.py <kernel_trace.csv> [us=1500]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Workgroup_Size_X", "?")) for r in rows)
adam = [i for i, k in enumerate(ks) if "adam_kernel" in k[2]]
ends = adam[1::2]
a, b = ends[-2], ends[-1]
step = ks[a + 1:b + 1]
t0, t1 = step[0][0], step[-1][1]
win = float(sys.argv[2]) * 1e3 if len(sys.argv) > 2 else 1.5e6
print(f"step span {(t1 - t0) / 1e3:.1f} us; kernels starting in its last {win / 1e3:.0f} us:")
for s, e, n, gx, bx in step:
    if s < t1 - win:
        continue
    beside = sum(1 for s2, e2, *_ in step if s2 < e and e2 > s) - 1
    name = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").replace("void ", "")[:60]
    print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:7.1f} us  beside {beside}  grid {gx}/{bx}  {name}")

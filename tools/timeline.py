#!/usr/bin/env python3
"""Timeline statistics of one replayed step from a rocprofv3 kernel_trace.csv (step = between the last two G-group adam kernels)."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
ks.sort()
adam = [i for i, k in enumerate(ks) if "adam_kernel" in k[2]]
# G adam = the longer of each pair; steps delimited by every second adam
ends = adam[1::2]
a, b = ends[-2], ends[-1]
step = ks[a + 1:b + 1]
t0, t1 = step[0][0], step[-1][1]
span = (t1 - t0) / 1e3
busy_sum = sum(e - s for s, e, _ in step) / 1e3
# union of intervals
ev = sorted([(s, 1) for s, e, _ in step] + [(e, -1) for s, e, _ in step])
cur = 0; last = t0; union = 0; conc_time = collections.Counter()
for t, d in ev:
    if cur > 0: union += t - last
    conc_time[cur] += t - last
    cur += d; last = t
print(f"step span {span:.1f} us, kernels {len(step)}, sum of durations {busy_sum:.1f} us, GPU busy (union) {union/1e3:.1f} us, idle {span - union/1e3:.1f} us")
print("time by #concurrent kernels:", {k: round(v / 1e3, 1) for k, v in sorted(conc_time.items())})
# gaps
gaps = []
cur_end = step[0][1]
for s, e, n in step[1:]:
    if s > cur_end: gaps.append(s - cur_end)
    cur_end = max(cur_end, e)
import statistics
if gaps: print(f"gaps: n={len(gaps)} total={sum(gaps)/1e3:.1f}us median={statistics.median(gaps)/1e3:.2f}us max={max(gaps)/1e3:.1f}us")
# short kernels
short = [(e - s) for s, e, _ in step if e - s < 5000]
print(f"kernels < 5us: {len(short)} totalling {sum(short)/1e3:.1f} us")
agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in step:
    key = n.replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").replace("void ", "")[:44]
    agg[key][0] += 1; agg[key][1] += e - s
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 14]:
    print(f"  {k:44s} {c:4d} {t/1e3:8.1f}us")

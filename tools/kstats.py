#!/usr/bin/env python3
"""Summarise a rocprofv3 kernel_stats.csv per train step: tools/kstats.py <csv> <steps in trace>"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"kernel time per step {tot/steps/1e6:.2f} ms, launches per step {sum(int(r['Calls']) for r in rows)/steps:.0f}")
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("_ZN12_GLOBAL__N_1", "").replace("void ", "")[:72]
    print(f"{n:72s} {int(r['Calls'])/steps:7.1f} {float(r['TotalDurationNs'])/steps/1e3:9.1f}us {float(r['AverageNs'])/1e3:8.1f}us {float(r['Percentage']):5.2f}%")

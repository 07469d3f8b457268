#!/usr/bin/env python3
"""Per kernel of a hipcc -S listing: number of global loads and of full drains (s_waitcnt vmcnt(0)) that are followed by
further loads -- i.e. serial memory round trips the scheduler left in straight-line code.  isa_roundtrips.py file.s"""
import re, sys
lines = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for si, st in enumerate(starts):
    en = starts[si + 1] if si + 1 < len(starts) else len(lines)
    seq = []
    for l in lines[st:en]:
        l = l.strip()
        if not l or l[0] in ";.":
            continue
        op = l.split()[0]
        if op.startswith(("global_load", "buffer_load", "flat_load")): seq.append("L")
        elif op.startswith("s_waitcnt") and "vmcnt(0)" in l: seq.append("W")
    s = "".join(seq)
    trips = len(re.findall(r"W(?=L)", s))
    name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", lines[st].split(":")[0])[:60]
    if s.count("L"):
        print(f"{name:62s} loads {s.count('L'):3d}  drains followed by loads {trips:3d}")

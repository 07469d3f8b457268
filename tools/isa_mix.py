#!/usr/bin/env python3
"""Instruction mix of the main (MFMA) loop of a kernel in a hipcc -S listing: isa_mix.py file.s name-substring"""
import re, sys, collections
lines = open(sys.argv[1]).read().split("\n")
starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l)]
for si, st in enumerate(starts):
    if sys.argv[2] not in lines[st]:
        continue
    en = starts[si + 1] if si + 1 < len(starts) else len(lines)
    body = lines[st:en]
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    best = None
    for i, l in enumerate(body):
        m = re.search(r"s_cbranch\w+\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            seg = body[labels[m.group(1)]:i + 1]
            if any("v_mfma" in x for x in seg) and (best is None or len(seg) > len(best)):
                best = seg
    c = collections.Counter()
    vc = collections.Counter()
    for l in best:
        l = l.strip()
        if not l or l[0] in ".;" or l.endswith(":"):
            continue
        op = l.split()[0]
        if op.startswith("v_mfma"): k = "mfma"
        elif op.startswith("v_"): k = "valu"; vc[op] += 1
        elif op.startswith("s_waitcnt") or op.startswith("s_barrier") or op.startswith("s_nop"): k = op
        elif op.startswith("s_"): k = "salu"
        else: k = op
        c[k] += 1
    print(lines[st][:90], "loop lines", len(best))
    print(dict(c))
    print(vc.most_common(30))
    if len(sys.argv) > 3:
        print("\n".join(best))

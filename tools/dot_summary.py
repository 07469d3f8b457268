#!/usr/bin/env python3
"""Node / edge census of a hipGraphDebugDotPrint dump: tools/dot_summary.py graph.dot"""
import collections, re, sys
txt = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'^\s*"?(graph_\w+|\w+)"?\s*\[(.*?)\];', txt, re.M | re.S):
    nid, attr = m.group(1), m.group(2)
    lab = re.search(r'label="(.*?)"', attr, re.S)
    nodes[nid] = lab.group(1) if lab else ""
edges = re.findall(r'"?(\w+)"?\s*->\s*"?(\w+)"?', txt)
kinds = collections.Counter()
for nid, lab in nodes.items():
    l = lab.lower()
    k = "memset" if "memset" in l else "memcpy" if "memcpy" in l else "event" if "event" in l else "empty" if "empty" in l else "kernel" if ("kernel" in l or "(" in l or "_z" in l) else "other"
    kinds[k] += 1
indeg, outdeg = collections.Counter(), collections.Counter()
for a, b in edges:
    outdeg[a] += 1; indeg[b] += 1
print(f"nodes {len(nodes)} kinds {dict(kinds)} edges {len(edges)} forks(out>1) {sum(1 for v in outdeg.values() if v > 1)} joins(in>1) {sum(1 for v in indeg.values() if v > 1)} "
      f"roots {sum(1 for n in nodes if indeg[n] == 0)} leaves {sum(1 for n in nodes if outdeg[n] == 0)} max_out {max(outdeg.values() or [0])} max_in {max(indeg.values() or [0])}")
names = collections.Counter(re.sub(r"\\n.*", "", lab)[:60] for lab in nodes.values())
for n, c in names.most_common(int(sys.argv[2]) if len(sys.argv) > 2 else 0):
    print(f"  {c:4d} {n}")

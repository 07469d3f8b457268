#!/usr/bin/env python3
"""Longest dependency chain of the captured training step.
tools/graph_critical_path.py step.dot [kernel_trace.csv]
step.dot: tools/graph_dot.py (hipGraphDebugDotPrint of the captured hipGraph: kernel nodes with grid / block / LDS, edges).
kernel_trace.csv: `rocprofv3 --kernel-trace --mangled-kernels --output-format csv -- python3 bench.py ...` of the same build; a
node's duration is the mean duration of the trace rows with its (kernel, grid, block) signature.  Prints the chain's length
with and without a per-node dispatch cost, the time by kernel family ON the chain, and the same for all nodes."""
import re, sys, csv, collections
dot = open(sys.argv[1]).read()
nodes = {}
for m in re.finditer(r'"graph_0_node_(\d+)"\[[^\]]*?label="\{\s*(\w+)(.*?)\}"\];', dot, re.S):
    nid, kind, body = int(m.group(1)), m.group(2), m.group(3)
    k = re.search(r'\| \{ID \| \d+ \| (\S+?)\\<\\<\\<\((\d+),(\d+),(\d+)\),\((\d+),(\d+),(\d+)\)', body)
    if k:
        gx, gy, gz, bx, by, bz = map(int, k.groups()[1:])
        nodes[nid] = (k.group(1), (gx * bx, gy * by, gz * bz), (bx, by, bz))
    else:
        nodes[nid] = (kind, None, None)
edges = [(int(a), int(b)) for a, b in re.findall(r'"graph_0_node_(\d+)" -> "graph_0_node_(\d+)"', dot)]
succ, pred = collections.defaultdict(list), collections.defaultdict(list)
for a, b in edges:
    succ[a].append(b); pred[b].append(a)
print(f"{len(nodes)} nodes ({sum(1 for n in nodes.values() if n[1])} kernels), {len(edges)} edges, "
      f"{sum(1 for n in nodes if len(succ[n]) > 1)} forks, {sum(1 for n in nodes if len(pred[n]) > 1)} joins")

dur = {}
if len(sys.argv) > 2:
    sig = collections.defaultdict(list)
    for r in csv.DictReader(open(sys.argv[2])):
        key = (r["Kernel_Name"].removesuffix(".kd"), (int(r["Grid_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"])),
               (int(r["Workgroup_Size_X"]), int(r["Workgroup_Size_Y"]), int(r["Workgroup_Size_Z"])))
        sig[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    byname = collections.defaultdict(list)
    for (n, g, b), v in sig.items():
        byname[n] += v
    miss = 0
    for nid, (name, grid, blk) in nodes.items():
        if grid is None:
            dur[nid] = 2.0                                   # memset / memcpy nodes
            continue
        v = sig.get((name, grid, blk))
        if v is None:
            v = byname.get(name); miss += 1
        dur[nid] = sum(v) / len(v) if v else 3.0
    print(f"durations: {len(sig)} signatures in the trace, {miss} nodes matched by name only")
else:
    dur = {n: 1.0 for n in nodes}

def fam(name):
    n = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)
    n = re.sub(r"^_Z\w*?\d+(at6native\d+)?", "", n)
    return n[:40]

order, indeg = [], {n: len(pred[n]) for n in nodes}
q = [n for n in nodes if indeg[n] == 0]
while q:
    n = q.pop(); order.append(n)
    for s in succ[n]:
        indeg[s] -= 1
        if indeg[s] == 0: q.append(s)
for node_cost in (0.0, 1.6):
    best, via = {}, {}
    for n in order:
        p = max(pred[n], key=lambda x: best[x], default=None)
        best[n] = (best[p] if p is not None else 0.0) + dur[n] + node_cost
        via[n] = p
    end = max(best, key=best.get)
    chain = []
    n = end
    while n is not None:
        chain.append(n); n = via[n]
    chain.reverse()
    total = sum(dur.values())
    print(f"\nper-node dispatch cost {node_cost} us: longest chain {best[end]:.0f} us over {len(chain)} nodes; all nodes {total:.0f} us "
          f"({total / best[end]:.2f}x the chain)")
    if node_cost: break
    if "--nodes" in sys.argv:
        for n in chain:
            print(f"    {n:4d} {dur[n]:7.1f} us  {fam(nodes[n][0])} {nodes[n][1]}{'  FORK' if len(succ[n]) > 1 else ''}{'  JOIN' if len(pred[n]) > 1 else ''}")
    on = collections.defaultdict(lambda: [0, 0.0]); al = collections.defaultdict(lambda: [0, 0.0])
    for n in chain:
        f = fam(nodes[n][0]); on[f][0] += 1; on[f][1] += dur[n]
    for n in nodes:
        f = fam(nodes[n][0]); al[f][0] += 1; al[f][1] += dur[n]
    print(f"{'kernel family':42s} {'on chain':>16s} {'all nodes':>16s}")
    for f, (c, t) in sorted(on.items(), key=lambda kv: -kv[1][1])[:28]:
        print(f"{f:42s} {c:5d} {t:8.0f} us {al[f][0]:5d} {al[f][1]:8.0f} us")
    # segments of the chain between fork / join nodes
    print("chain segments (between forks / joins): first node id, nodes, us, first kernel")
    seg_start, seg_t, seg_n = chain[0], 0.0, 0
    for i, n in enumerate(chain):
        seg_t += dur[n]; seg_n += 1
        if len(succ[n]) > 1 or (i + 1 < len(chain) and len(pred[chain[i + 1]]) > 1) or i + 1 == len(chain):
            print(f"  node {seg_start:4d} +{seg_n:4d} nodes {seg_t:8.0f} us  {fam(nodes[seg_start][0])}")
            if i + 1 < len(chain): seg_start, seg_t, seg_n = chain[i + 1], 0.0, 0

#!/usr/bin/env python3
"""Cost of one dependent kernel node in a replayed hipGraph: a chain of N tiny kernels on one stream, and the same N spread
over 2 / 4 forked streams.  Prints us per node."""
import torch
x = torch.zeros(40, 256, device="cuda")
ys = [torch.zeros(40, 256, device="cuda") for _ in range(4)]
def chain(n, streams):
    g = torch.cuda.CUDAGraph()
    ss = [torch.cuda.Stream() for _ in range(streams)]
    with torch.cuda.graph(g):
        cur = torch.cuda.current_stream()
        for s in ss:
            s.wait_stream(cur)
        for k, s in enumerate(ss):
            with torch.cuda.stream(s):
                for _ in range(n // streams):
                    ys[k].add_(1.0)
        for s in ss:
            cur.wait_stream(s)
    return g
for streams in (1, 2, 4):
    n = 400
    g = chain(n, streams)
    for _ in range(3): g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{n} tiny kernels over {streams} stream(s): {e0.elapsed_time(e1) * 100 / n:.2f} us per node, {e0.elapsed_time(e1) / 10:.3f} ms per replay")

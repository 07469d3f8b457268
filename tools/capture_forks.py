#!/usr/bin/env python3
"""Model-free reproducer for the hipStreamEndCapture crashes of round 2 (gpurun_out/r2_t11.log, tr_t.txt: both die in
torch/cuda/graphs.py capture_end after extra fork / join pairs were added to the captured train step).

Every pattern captures N fork / join pairs around trivial ast_scale launches (the same C-ABI helper the step uses) on the
capture plumbing the Trainer uses (torch.cuda.graph + Stream.wait_stream), replays the graph and checks the arithmetic.
ONE pattern per process (a crash must not take the other patterns with it):

    python tools/capture_forks.py PATTERN N           -> prints "OK <pattern> <n> nodes=<k>" and exits 0
    python tools/capture_forks.py --all               -> runs every (pattern, N) in child processes and prints a table

Patterns
  seq        N sequential fork/join pairs on ONE side stream, events created and destroyed inside the capture (wait_stream)
  fan        N side streams forked, then all joined (events destroyed inside the capture)
  keep       as fan, but every Event object is kept alive until the capture has ended
  nested     side stream forks a second-level stream, joined back level by level
  nested_pre    the same, but BOTH streams first join the capture from the origin stream (level-1 forks at the start)
  nested_fresh  the same as nested with a fresh pair of streams per iteration
  nested_direct second-level stream joined straight into the origin stream
  nested_keep   nested with every Event object kept alive until the capture has ended
  nested_via_main  the second-level stream is joined into the ORIGIN stream and the first-level stream then waits for the origin
  nested_helper    the nested pattern written with ast_amd.streams.fork / join (the guard routes the child's join through the origin)
  sibling       two first-level streams; one waits for an event of the other, both join the origin
  sibling_mutual  two first-level streams wait for each other in turn: a waits for b, b then waits for a (b waits on a stream that
             already depends on b -- the nested shape without a fork: what a pooled hand-over between flush streams does)
  tail       the side stream gets MORE work after its join event was recorded (an unjoined tail: what an autograd node that
             returns no gradient leaves behind when its backward runs on a side stream)
  unjoined   a forked stream is never joined (CUDA semantics: cudaErrorStreamCaptureUnjoined)
  bwd        autograd: a chain of N custom Functions whose backward forks a side stream for a leaf kernel (the weight-gradient
             side streams of DESIGN 8.4), joined by an end-of-backward engine callback
  bwd_leaf   as bwd, but the LAST node's input needs no gradient: its side-stream work is nobody's input
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd"))

PATTERNS = ("seq", "fan", "keep", "nested", "nested_pre", "nested_fresh", "nested_direct", "nested_keep", "nested_via_main", "nested_helper", "sibling", "sibling_mutual", "tail", "unjoined", "bwd", "bwd_leaf")


def run_pattern(pattern, n):
    import faulthandler
    faulthandler.enable()
    import torch
    from ast_amd._lib import check, lib, ptr

    dev = torch.device("cuda:0")
    x = torch.ones(1 << 12, device=dev)
    bufs = [torch.zeros(1 << 12, device=dev) for _ in range(n + 2)]

    def scale(src, dst, s, st):
        check(lib().ast_scale(ptr(src), None, float(s), ptr(dst), src.numel(), 0, st.cuda_stream), "ast_scale")

    sides = [torch.cuda.Stream(device=dev) for _ in range(max(2, 2 * n if pattern == "nested_fresh" else n))]
    # warm-up outside the capture (library load, allocator)
    for s in sides[:2]:
        s.wait_stream(torch.cuda.current_stream())
        scale(x, bufs[0], 1.0, s)
        torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    keep = []
    expect = None
    g = torch.cuda.CUDAGraph()

    class Node(torch.autograd.Function):
        """y = 2 t; the parameter w gets its "gradient" from a leaf kernel on a side stream and None through autograd (as the
        convolution Functions of ast_amd.ops do with their packed weights)."""
        pending = False

        @staticmethod
        def forward(ctx, t, w, i):
            ctx.i = i
            y = torch.empty_like(t)
            scale(t, y, 2.0, torch.cuda.current_stream())
            return y

        @staticmethod
        def backward(ctx, dy):
            dy = dy.contiguous()                        # (the gradient of sum() is an expanded 1-element tensor)
            main = torch.cuda.current_stream()
            side = sides[0]
            side.wait_stream(main)                      # leaf work (a "weight gradient") beside the data gradient
            scale(dy, bufs[ctx.i], 3.0, side)
            dy.record_stream(side)
            if not Node.pending:
                Node.pending = True
                torch.autograd.Variable._execution_engine.queue_callback(Node.join)
            if not ctx.needs_input_grad[0]:
                return None, None, None
            dx = torch.empty_like(dy)
            scale(dy, dx, 2.0, main)
            return dx, None, None

        @staticmethod
        def join():
            Node.pending = False
            torch.cuda.current_stream().wait_stream(sides[0])

    w_param = torch.ones(8, device=dev, requires_grad=True)
    t_in = x.clone().requires_grad_(pattern == "bwd")

    with torch.cuda.graph(g):
        main = torch.cuda.current_stream()
        if pattern == "seq":
            s = sides[0]
            for i in range(n):
                s.wait_stream(main)
                scale(x, bufs[i], i + 1, s)
                main.wait_stream(s)
            expect = [(i, i + 1.0) for i in range(n)]
        elif pattern in ("fan", "keep"):
            for i in range(n):
                if pattern == "keep":
                    e = torch.cuda.Event(); e.record(main); sides[i].wait_event(e); keep.append(e)
                else:
                    sides[i].wait_stream(main)
                scale(x, bufs[i], i + 1, sides[i])
            for i in range(n):
                if pattern == "keep":
                    e = torch.cuda.Event(); e.record(sides[i]); main.wait_event(e); keep.append(e)
                else:
                    main.wait_stream(sides[i])
            expect = [(i, i + 1.0) for i in range(n)]
        elif pattern in ("nested", "nested_pre", "nested_fresh", "nested_direct", "nested_keep"):
            def wait(dst, srcs):
                if pattern == "nested_keep":
                    e = torch.cuda.Event(); e.record(srcs); dst.wait_event(e); keep.append(e)
                else:
                    dst.wait_stream(srcs)
            if pattern == "nested_pre":
                for s in sides[:2]:
                    s.wait_stream(main)            # both streams join the capture as children of the ORIGIN stream first
            for i in range(n):
                a, b = (sides[(2 * i) % len(sides)], sides[(2 * i + 1) % len(sides)]) if pattern == "nested_fresh" else (sides[0], sides[1])
                wait(a, main)
                scale(x, bufs[i], i + 1, a)
                wait(b, a)                         # second-level fork: b waits on an event recorded on a forked stream
                scale(bufs[i], bufs[n], 1.0, b)
                if pattern == "nested_direct":
                    wait(main, b)
                else:
                    wait(a, b)
                wait(main, a)
            expect = [(i, i + 1.0) for i in range(n)]
        elif pattern == "nested_via_main":
            a, b = sides[0], sides[1]
            for i in range(n):
                a.wait_stream(main)
                scale(x, bufs[i], i + 1, a)
                b.wait_stream(a)
                scale(bufs[i], bufs[n], 1.0, b)
                main.wait_stream(b)                # join into the origin ...
                a.wait_stream(main)                # ... and let the first-level stream wait for the origin
                scale(bufs[n], bufs[n + 1], 1.0, a)
                main.wait_stream(a)
            expect = [(i, i + 1.0) for i in range(n)] + [(n + 1, float(n))]
        elif pattern == "nested_helper":
            from ast_amd import streams as ST
            a, b = sides[0], sides[1]
            with ST.capture_origin(main):
                for i in range(n):
                    ST.fork(a, main)
                    scale(x, bufs[i], i + 1, a)
                    ST.fork(b, a)
                    scale(bufs[i], bufs[n], 1.0, b)
                    ST.join(a, b)                  # would be the crashing a.wait_stream(b): goes through the origin stream
                    scale(bufs[n], bufs[n + 1], 1.0, a)
                    ST.join(main, a)
            expect = [(i, i + 1.0) for i in range(n)] + [(n + 1, float(n))]
        elif pattern == "sibling":
            a, b = sides[0], sides[1]
            for i in range(n):
                a.wait_stream(main)
                b.wait_stream(main)
                scale(x, bufs[i], i + 1, b)
                a.wait_stream(b)                   # first-level stream waits on an event of its sibling
                scale(bufs[i], bufs[n], 1.0, a)
                main.wait_stream(a)
                main.wait_stream(b)
            expect = [(i, i + 1.0) for i in range(n)] + [(n, float(n))]
        elif pattern == "sibling_mutual":
            a, b = sides[0], sides[1]
            for i in range(n):
                a.wait_stream(main)
                b.wait_stream(main)
                scale(x, bufs[i], i + 1, b)
                a.wait_stream(b)                   # a now depends on b
                scale(bufs[i], bufs[n], 1.0, a)
                b.wait_stream(a)                   # ... and b waits for a
                scale(bufs[n], bufs[n + 1], 1.0, b)
                main.wait_stream(a)
                main.wait_stream(b)
            expect = [(i, i + 1.0) for i in range(n)] + [(n, float(n)), (n + 1, float(n))]
        elif pattern == "tail":
            s = sides[0]
            for i in range(n):
                s.wait_stream(main)
                scale(x, bufs[i], i + 1, s)
                main.wait_stream(s)
                scale(x, bufs[n], 7.0, s)              # after the join event: nothing waits for this launch
            expect = [(i, i + 1.0) for i in range(n)]
        elif pattern == "unjoined":
            for i in range(n):
                sides[i].wait_stream(main)
                scale(x, bufs[i], i + 1, sides[i])
            for i in range(n - 1):
                main.wait_stream(sides[i])
            expect = [(i, i + 1.0) for i in range(n - 1)]
        elif pattern in ("bwd", "bwd_leaf"):
            h = t_in
            for i in range(n):
                h = Node.apply(h, w_param, i)
            h.sum().backward()
            # dy of node i = 2^(n-1-i); its leaf kernel stores 3 dy
            expect = [(i, 3.0 * 2.0 ** (n - 1 - i)) for i in range(n)]
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    for i, v in expect or []:
        got = float(bufs[i][5])
        assert abs(got - v) < 1e-6, (pattern, i, got, v)
    del keep
    print(f"OK {pattern} {n}", flush=True)


def main():
    if len(sys.argv) >= 2 and sys.argv[1] == "--all":
        ns = [int(a) for a in sys.argv[2:]] or [4, 8, 16, 32]
        pats = [p for p in os.environ.get("PATTERNS", ",".join(PATTERNS)).split(",") if p in PATTERNS]
        print(f"{'pattern':10s} " + " ".join(f"{'N=' + str(n):>12s}" for n in ns), flush=True)
        for p in pats:
            row = []
            for n in ns:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), p, str(n)], capture_output=True, text=True, timeout=300)
                if r.returncode == 0 and f"OK {p} {n}" in r.stdout:
                    row.append("ok")
                else:
                    err = (r.stderr or "").strip().splitlines()
                    key = "segv" if ("Segmentation fault" in r.stderr or r.returncode in (-11, 139)) else next(
                        (ln.split(":")[0][-28:] for ln in reversed(err) if "Error" in ln or "error" in ln), f"rc={r.returncode}")
                    row.append(key)
                    with open(os.path.join(ROOT, "gpurun_out", f"capture_forks_{p}_{n}.log"), "w") as f:
                        f.write(r.stdout + "\n--- stderr ---\n" + r.stderr)
            print(f"{p:10s} " + " ".join(f"{c:>12s}" for c in row), flush=True)
        return
    run_pattern(sys.argv[1], int(sys.argv[2]))


if __name__ == "__main__":
    main()

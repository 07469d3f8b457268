"""Stream fork / join helpers that are safe inside a hipGraph capture.

Round 2 lost two experiments (weight gradients on side streams, the ResBlock shortcut convolution on a side stream) to
segmentation faults inside hipStreamEndCapture.  The model-free reproducer tools/capture_forks.py (tests/test_gpu_capture_forks.py)
pins the cause down to ONE pattern (profiles/r03/capture_forks_*.txt, ROCm 7.2 / PyTorch 2.10):

    a.wait_stream(origin)     # a joins the capture
    b.wait_stream(a)          # b forks from the forked stream a           -- fine
    a.wait_stream(b)          # a waits for an event of ITS OWN CHILD      -- hipStreamEndCapture dereferences freed state: SIGSEGV
    origin.wait_stream(a)

Everything else captures and replays correctly: any number of first-level forks (sequential or concurrent, Event objects kept or
destroyed inside the capture), forks from forked streams, waits between sibling streams, joins of a second-level stream straight
into the origin stream, and autograd backward passes that fork per node.  An unjoined stream is reported as
hipErrorStreamCaptureUnjoined (an exception, not a crash).

`join(dst, src)` therefore routes a join whose destination is not the capture's origin stream THROUGH the origin:
origin.wait_stream(src); dst.wait_stream(origin).  The origin stream idles at that point in the patterns this package uses (it is
waiting for its branches anyway), so nothing is serialised that was not already.
"""
from __future__ import annotations

import torch

_origin = {}          # device index -> the stream a capture was begun on (set by capture_origin)
_parent = {}          # stream id -> the stream it was last forked from (inside the current capture_origin scope)


class capture_origin:
    """with capture_origin(stream): ...   -- names the stream torch.cuda.graph captures on for join()."""

    def __init__(self, stream=None):
        self.stream = stream

    def __enter__(self):
        s = self.stream or torch.cuda.current_stream()
        self.key = s.device.index
        self.old = _origin.get(self.key)
        _origin[self.key] = s
        _parent.clear()
        return s

    def __exit__(self, *exc):
        if self.old is None:
            _origin.pop(self.key, None)
        else:
            _origin[self.key] = self.old
        return False


def fork(side: torch.cuda.Stream, src: torch.cuda.Stream = None):
    """`side` continues after everything enqueued on `src` (default: the current stream) so far."""
    src = src or torch.cuda.current_stream()
    if side == src:
        return
    _parent[side.cuda_stream] = src
    side.wait_stream(src)


def _descends_from(s, anc):
    seen = 0
    while s is not None and seen < 64:
        s = _parent.get(s.cuda_stream)
        if s is not None and s == anc:
            return True
        seen += 1
    return False


def join(dst: torch.cuda.Stream, src: torch.cuda.Stream):
    """`dst` continues after everything enqueued on `src` so far.  Inside a capture, a join into the stream that `src` was forked
    from (directly or through other forks) goes through the capture's origin stream when `dst` is not the origin itself (see the
    module docstring: a forked stream must never wait on an event of its own child)."""
    if dst == src:
        return
    origin = _origin.get(dst.device.index)
    if origin is not None and torch.cuda.is_current_stream_capturing() and dst != origin and _descends_from(src, dst):
        origin.wait_stream(src)
        dst.wait_stream(origin)
    else:
        dst.wait_stream(src)

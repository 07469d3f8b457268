"""Stream fork / join patterns inside a hipGraph capture, model-free (tools/capture_forks.py): the captured train step forks
and joins up to ~20 streams (encoder branches, discriminator phase, per-bank flushes, weight prepare), and round 2 lost two
experiments to segmentation faults inside hipStreamEndCapture.  Every pattern runs in its own process (a crash must be an
assertion here, not the end of the test session)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "capture_forks.py")


def _run(pattern, n):
    return subprocess.run([sys.executable, TOOL, pattern, str(n)], capture_output=True, text=True, timeout=300)


# NOT in this list, on purpose: "nested" (a forked stream waits for an event of its own child stream), which segfaults inside
# hipStreamEndCapture on ROCm 7.2 (tools/capture_forks.py nested 4 / 16; profiles/r03/capture_forks_*.txt).  "nested_helper" is the
# same dependency structure written with ast_amd.streams.join, which routes that join through the capture's origin stream.
# "sibling_mutual" (a waits for b, then b waits for a) is the same shape without a fork and crashes the same way
# (profiles/r03/capture_forks_mutual.txt): hand-overs between side streams go through the origin (layers._DeferPool).
@pytest.mark.parametrize("pattern,n", [("seq", 32), ("fan", 24), ("keep", 24), ("nested_direct", 16), ("nested_via_main", 16), ("nested_helper", 16),
                                       ("sibling", 16), ("bwd", 32), ("bwd_leaf", 32)])
def test_joined_fork_patterns_capture_and_replay(pattern, n):
    r = _run(pattern, n)
    assert r.returncode == 0 and f"OK {pattern} {n}" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-1500:])


@pytest.mark.parametrize("pattern", ["unjoined"])
def test_unjoined_stream_is_an_error_not_a_crash(pattern):
    r = _run(pattern, 4)
    assert r.returncode == 1 and "unjoined work" in r.stderr, (r.returncode, r.stderr[-800:])

#!/usr/bin/env python3
"""Dump the captured training step's hipGraph as DOT (AST_GRAPH_DOT) -- input of tools/graph_critical_path.py.
tools/graph_dot.py out.dot"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["AST_GRAPH_DOT"] = os.path.abspath(sys.argv[1])
sys.path.insert(0, os.path.join(ROOT, "audio-style-transfer_amd")); sys.path.insert(0, ROOT)
sys.argv = ["bench.py", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-roofline", "--no-extras"]
import runpy
runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")

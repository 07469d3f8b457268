#!/bin/bash
# Is the patch-staged (halo) weight-gradient kernel latency-bound per tile trip?  time = trips x t_trip + flush(slices): vary the slices.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3; mkdir -p $O
{
for pg in 1 2; do for tgt in 96 192 384 768; do
  AST_WGRAD_HALO_MAXCD=128 AST_WGRAD_PG=$pg AST_WGRAD_WG_TARGET=$tgt WGRAD_REP=8 timeout -k 10 120 python tools/conv_bench.py b0c2,b1c2,b2c2 30 wgrad 2>&1 | grep -v amdgpu.ids | sed "s/^/halo pg=$pg wgs=$tgt /"
done; done
for tgt in 96 192 384 768; do
  AST_WGRAD_HALO_MAXCD=32 AST_WGRAD_PG=1 AST_WGRAD_WG_TARGET=$tgt WGRAD_REP=8 timeout -k 10 120 python tools/conv_bench.py b1c2,b2c2 30 wgrad 2>&1 | grep -v amdgpu.ids | sed "s/^/gathered pg=1 wgs=$tgt /"
done
} | tee $O/halo_probe.txt

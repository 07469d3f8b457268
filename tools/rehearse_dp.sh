#!/bin/bash
# Two data-parallel ranks on ONE GPU over gloo (RCCL needs one device per rank): rehearses bench.py --gpus 2 in the default
# loss-matched mode (eager with gloo: its collectives are not capturable) and in the --per-rank-bn mode (three graphs), f32 and bf16 wire.
export AST_DIST_BACKEND=gloo AST_ONE_GPU=1
O=gpurun_out/r3; mkdir -p $O
run() { # name, port, extra args...
  local name=$1 port=$2; shift 2
  timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $port bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline "$@" 2>$O/reh_$name.err | tail -1 | cut -c1-1200 > $O/reh_$name.txt || { tail -5 $O/reh_$name.err; return 1; }
  python -c "import json,sys; d=json.loads(open('$O/reh_$name.txt').read()); print('$name', d['ms_per_step'], d['config']['dp_semantics'][:40], '|', d['config']['collectives'], '|', d['config']['grad_exchange'][:30])"
}
run matched 29577 && run perrank 29578 --per-rank-bn && AST_GRAD_WIRE=bf16 run perrank_bf16wire 29579 --per-rank-bn

#!/bin/bash
# round 5: bench.py's N = 2 code path on a one-GPU box (ORBFE_BENCH_SHARE_GPU=1: both ranks on cuda:0, gloo instead of RCCL): the
# C2 weak-scaling line with its gather, and C5's shard + all-reduce(MAX) line.  Writes gpurun_out/r5v/.
mkdir -p gpurun_out/r5v
run() { # name, port, extra args
  ORBFE_BENCH_SHARE_GPU=1 timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $2 \
    bench.py --gpus 2 --steps 5 --warmup 2 ${@:3} > gpurun_out/r5v/$1.json 2> gpurun_out/r5v/$1.err
  echo "$1 rc=$?"
  python -c "
import json; d=json.load(open('gpurun_out/r5v/$1.json')); print(d['n_gpus'], '%.4g' % d['value'], d['unit'], 'ms %.4f' % d['ms_per_step'], d['scaling'], '|', d['config'].get('collective'))"
}
run bench_2rank_share 29511 --batch 256
run bench_c5_2rank_share 29512 --mode c5
run bench_2rank_share_exact 29513 --batch 256 --gather exact --scene survey
for f in gpurun_out/r5v/*.err; do tail -n 2 $f; done

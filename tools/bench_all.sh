#!/bin/bash
# all single-GPU bench lines of the round (one JSON line each)
out=gpurun_out/bench_lines.jsonl
: > $out
run() { echo "# python bench.py $*" >> $out; timeout -k 10 400 python bench.py --no_cpu_baseline "$@" 2>/dev/null >> $out; echo "done: $*"; }
run
run --da_policy ""
run --g_regularization none
run --n_labels 1000
run --workload c1
run --workload c2bf16
run --workload c3
run --workload c3fp32
run --workload c3 --g_regularization none
run --workload c4 --steps 3 --warmup 1
run --workload c5 --batch 16 --steps 2 --warmup 1
run --workload c5 --batch 32 --steps 2 --warmup 1

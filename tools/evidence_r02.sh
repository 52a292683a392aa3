#!/bin/bash
# Round-2 evidence of the final build, one gpurun call per part (each part fits the 1200 s limit):
#   tools/evidence_r02.sh lines    default bench line (config-2 headline + config-3 target) and every workload line
#   tools/evidence_r02.sh stats    rocprofv3 --kernel-trace --stats of config 3 (32 and 256 images) and config 2,
#                                  per-layer kbench tables (config 3 at 32 / 256 images, bf16 and staged)
#   tools/evidence_r02.sh pmc      the three --pmc passes over config 3 (32 and 256 images) and config 2
# Everything lands under gpurun_out/ev/; copy what is to be judged into profiles/ (README there).
part=$1
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/ev
E=gpurun_out/ev
case "$part" in
lines)
  echo "== default bench"; timeout -k 10 500 python bench.py > $E/bench_default.json 2> $E/bench_default.err; tail -c 400 $E/bench_default.json
  out=$E/bench_lines.jsonl
  : > $out
  run() { echo "# python bench.py $*" >> $out; timeout -k 10 300 python bench.py --no_cpu_baseline "$@" 2>/dev/null >> $out; echo "done: $*"; }
  run --workload c3
  run --workload c3 --batch 256
  run --workload c3 --graph
  run --workload c3 --batch 256 --g_regularization none
  run --workload c3 --batch 256 --da_policy ""
  run --workload c3staged
  run --workload c3fp32
  run --workload c2bf16
  run --workload c1
  run --workload c1 --graph
  run --workload c4 --steps 3 --warmup 1
  run --workload c5 --batch 32 --steps 2 --warmup 1
  run --gan_type ra-dragan
  ;;
stats)
  prof() {  # tag, bench args...
    tag=$1; shift
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/prof_$tag -o r -- python $R/bench.py --no_cpu_baseline --no_roofline --no_target --steps 4 "$@" > $R/$E/prof_$tag.log 2>&1)
    find $E/prof_$tag -name "*kernel_stats.csv" -exec cp {} $E/${tag}_kernel_stats.csv \;
    grep -h '"metric"' $E/prof_$tag.log | cut -c1-240
    rm -rf $E/prof_$tag
  }
  prof c3 --workload c3
  prof c3_b256 --workload c3 --batch 256
  prof c2
  timeout -k 10 200 python tools/kbench.py "" bf16 c3 > $E/kbench_c3_bf16.txt 2>&1; tail -n 1 $E/kbench_c3_bf16.txt
  timeout -k 10 200 python tools/kbench.py "" bf16 c3 8 > $E/kbench_c3_bf16_b256.txt 2>&1; tail -n 1 $E/kbench_c3_bf16_b256.txt
  timeout -k 10 200 python tools/kbench.py "" bf16-staged c3 > $E/kbench_c3_staged.txt 2>&1; tail -n 1 $E/kbench_c3_staged.txt
  ;;
pmc)
  bash tools/pmc.sh c3 --workload c3 && cp gpurun_out/pmc_c3.json gpurun_out/pmc_c3.txt $E/
  bash tools/pmc.sh c3_b256 --workload c3 --batch 256 && cp gpurun_out/pmc_c3_b256.json gpurun_out/pmc_c3_b256.txt $E/
  bash tools/pmc.sh c2 && cp gpurun_out/pmc_c2.json gpurun_out/pmc_c2.txt $E/
  ;;
*) echo "usage: $0 lines|stats|pmc"; exit 2;;
esac
echo "== done $part"

#!/bin/bash
# Round-3 evidence, one gpurun call per part:
#   tools/evidence_r03.sh lines    the default bench line (config 3 at global batch 256 + companions + fp32 config 2) and
#                                  the other workloads
#   tools/evidence_r03.sh stats    rocprofv3 --kernel-trace --stats of config 3 at 256 and 32 images, the per-shape table
#   tools/evidence_r03.sh pmc      the three --pmc passes over config 3 at 256 and at 32 images
# Everything lands under gpurun_out/ev3/; copy what is to be judged into profiles/ (README there).
part=$1
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/ev3
E=gpurun_out/ev3
case "$part" in
lines)
  echo "== default bench"; timeout -k 10 500 python bench.py > $E/bench_default.json 2> $E/bench_default.err; tail -c 300 $E/bench_default.json
  out=$E/bench_lines.jsonl
  : > $out
  run() { echo "# python bench.py $*" >> $out; timeout -k 10 300 python bench.py --no_cpu_baseline --no_companions "$@" 2>/dev/null >> $out; echo "done: $*"; }
  run --batch 32
  run --batch 32 --graph
  run --workload c2
  run --workload c2bf16
  run --workload c1
  run --workload c1 --graph
  run --workload c4 --batch 32 --steps 3 --warmup 1
  run --workload c5 --batch 32 --steps 2 --warmup 1
  run --gan_type ra-dragan --steps 4 --warmup 1
  ;;
stats)
  prof() {  # tag, bench args...
    tag=$1; shift
    (cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$E/prof_$tag -o r -- python3 $R/bench.py --no_cpu_baseline --no_roofline --no_companions --steps 4 --warmup 2 "$@" > $R/$E/prof_$tag.log 2>&1)
    find $E/prof_$tag -name "*kernel_stats.csv" -exec cp {} $E/${tag}_kernel_stats.csv \;
    grep -h '"metric"' $E/prof_$tag.log | cut -c1-200
    rm -rf $E/prof_$tag
  }
  prof c3_b256
  prof c3_b32 --batch 32
  timeout -k 10 200 python tools/shapes.py 128 96 256 $E/shapes_c3_b256.tsv --precision bf16 > $E/shapes_c3_b256.txt 2>&1; head -14 $E/shapes_c3_b256.txt
  ;;
pmc)
  bash tools/pmc.sh c3_b256 && cp gpurun_out/pmc_c3_b256.json gpurun_out/pmc_c3_b256.txt $E/
  bash tools/pmc.sh c3_b32 --batch 32 && cp gpurun_out/pmc_c3_b32.json gpurun_out/pmc_c3_b32.txt $E/
  ;;
*) echo "usage: $0 lines|stats|pmc"; exit 2;;
esac
echo "== done $part"

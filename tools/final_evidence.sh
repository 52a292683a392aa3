#!/bin/bash
# final-build evidence of the round: default bench line, rocprofv3 kernel stats, per-shape GEMM times, PMC passes, all bench lines
R=$GRAFT_REPO_ROOT
cd $R
echo "== default bench"; timeout -k 10 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err && tail -c 600 gpurun_out/bench_default.json
echo "== ra-dragan"; timeout -k 10 200 python bench.py --gan_type ra-dragan --no_cpu_baseline --no_roofline > gpurun_out/bench_radragan.json 2>/dev/null
echo "== shapes"; timeout -k 10 200 python tools/shapes.py 128 64 64 gpurun_out/gemm_dump_c2.csv > gpurun_out/gemm_shapes_c2.txt 2>&1
echo "== rocprof stats"
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c2 -o r -- python $R/bench.py --no_cpu_baseline --no_roofline --steps 4 > $R/gpurun_out/prof_c2.log 2>&1)
find gpurun_out/prof_c2 -name "*kernel_stats.csv" -exec cp {} gpurun_out/c2_kernel_stats.csv \;
find gpurun_out/prof_c2 -name "*kernel_trace.csv" -delete
echo "== pmc"; bash tools/pmc.sh c2
echo "== all lines"; bash tools/bench_all.sh
echo "== done"

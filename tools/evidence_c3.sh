R=$GRAFT_REPO_ROOT
cd $R
(cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c3 -o r -- python $R/bench.py --no_cpu_baseline --no_roofline --steps 4 --workload c3 > $R/gpurun_out/prof_c3.log 2>&1)
find gpurun_out/prof_c3 -name "*kernel_trace.csv" -delete
echo stats done
bash tools/pmc.sh c3 --workload c3
echo pmc done

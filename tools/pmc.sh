#!/bin/bash
# usage: tools/pmc.sh <tag> [bench args...]      three separate rocprofv3 --pmc passes over bench.py (3 iterations each)
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16"; do
  n=$(echo $pass | cut -d' ' -f1)
  timeout -k 10 280 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -o r -- python3 $R/bench.py --no_cpu_baseline --no_roofline --no_companions --steps 2 --warmup 1 "$@" > $R/gpurun_out/pmc_${tag}_$n.log 2>&1
  echo "pass $n done"
done
cd $R
python tools/summarize_pmc.py gpurun_out/pmc_${tag}.json --iterations 3 gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/pmc_${tag}_SQ_VALU_MFMA_BUSY_CYCLES > gpurun_out/pmc_${tag}.txt
rm -rf gpurun_out/pmc_${tag}_FETCH_SIZE gpurun_out/pmc_${tag}_WRITE_SIZE gpurun_out/pmc_${tag}_SQ_VALU_MFMA_BUSY_CYCLES

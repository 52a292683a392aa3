#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z_0-9]*(LDS|WAIT|BARRIER|ACTIVE_INST|BUSY|WAVE_CYCLES|INSTS_VALU_MFMA|INST_CYCLES)[A-Z_0-9]*)\b" | sort -u > $R/gpurun_out/pmc_names.txt
echo "counters: $(wc -l < $R/gpurun_out/pmc_names.txt)"
i=0
for pass in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" "SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout -k 10 120 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmcb_$i -o r -- python $R/tools/kbench.py "${KB_FILTER:-deconv N64 H32}" ${KB_MODE:-bf16} > $R/gpurun_out/pmcb_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python - <<'PY'
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
for f in glob.glob("gpurun_out/pmcb_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        if "nn_kernel" not in k and "tn_kernel" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k, d in agg.items():
    print(k)
    for c, v in sorted(d.items()):
        print("   %-32s %16.0f per launch" % (c, v / max(cnt[(k, c)], 1)))
PY
rm -rf gpurun_out/pmcb_[0-9]

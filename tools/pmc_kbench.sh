#!/bin/bash
# usage (on the GPU box): tools/pmc_kbench.sh <tag> "<kbench filter>" [mode] [cfg] [batch multiplier]
# PMC counters of the implicit-GEMM kernels on single layers (tools/kbench.py), one rocprofv3 --pmc pass per group
# (SQ has 8 slots, TCC 4: MI355X_MICROARCH.md "rocprofv3 PMC slots"); summary -> gpurun_out/pmck_<tag>.txt
tag=$1; flt=$2; mode=${3:-bf16}; cfg=${4:-c3}; scale=${5:-1}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for pass in "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_LDS" \
            "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES" \
            "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
            "TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -k 10 150 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d $R/gpurun_out/pmck_${tag}_$i -o r -- python $R/tools/kbench.py "$flt" $mode $cfg $scale > $R/gpurun_out/pmck_${tag}_$i.log 2>&1
  echo "pass $i rc=$?"
done
cd $R
python - "$tag" <<'PY' > gpurun_out/pmck_$tag.txt
import csv, glob, collections, sys
tag = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(int)
dur = collections.defaultdict(float); nd = collections.defaultdict(int)
csv.field_size_limit(1 << 30)
for f in glob.glob("gpurun_out/pmck_%s_*/**/*counter_collection.csv" % tag, recursive=True):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-40:] + " grid=" + r.get("Grid_Size", "?")
        if "nn16" not in k and "tn16" not in k and "nn_kernel" not in k and "tn_kernel" not in k: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); dur[k] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3; nd[k] += 1
for k, d in sorted(agg.items()):
    print("%s   mean %.1f us under PMC" % (k, dur[k] / max(nd[k], 1)))
    for c, v in sorted(d.items()):
        print("   %-32s %16.0f per launch" % (c, v / max(cnt[(k, c)], 1)))
PY
cat gpurun_out/pmck_$tag.txt
rm -rf gpurun_out/pmck_${tag}_[0-9]

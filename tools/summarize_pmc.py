#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel family.

    python tools/summarize_pmc.py OUT.json [--iterations N] DIR_OR_CSV [DIR_OR_CSV ...]

Every pass (one directory per --pmc pass, see profiles/README.md) contributes its counters; per
kernel family (name up to the first '<' / '(') the script reports launches, the mean of each
counter per launch and, for FETCH_SIZE / WRITE_SIZE, bytes per launch with the corrections of
MI355X_MICROARCH.md "HBM [CDNA4]": both counters are in KiB-like units of 1 KB (rocprofv3
derived metric, KB), FETCH_SIZE on gfx950 reports half of a wide coalesced read and is doubled.
Families that ran MFMA also get mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES) (the guide's
utilisation formula).  With --iterations N (training iterations inside the profiled command) the "_summary" entry
holds the HBM bytes of the GEMM family (implicit-GEMM convolutions, their split-K / reflect folds, attention) per
iteration.
"""
import csv
import json
import os
import re
import sys
from collections import OrderedDict, defaultdict


def family(name):
    name = re.sub(r"^void\s+", "", name)
    m = re.match(r"([A-Za-z0-9_:]+(?:<[^<>()]*>)?)", name)
    return m.group(1) if m else name[:80]


GEMM_FAMILY = r"(\b(nn|tn)(16[xh]?)?_kernel|attn(16)?_|\brgb_|\bthin_|slab_reduce|reflect_fold)"


# the plain GEMMs of the ortho-cosine regulariser (Gram matrices and their gradients: batch-independent, fp32 weights)
REGULARISER = r"tn_kernel_bf16_tr<2>|nn_kernel_bf16<2, 2, false, 2, false, true>|tn16x?_kernel<2[,>]"


def main():
    argv = sys.argv[1:]
    iterations = 0
    if "--iterations" in argv:
        i = argv.index("--iterations")
        iterations = int(argv[i + 1])
        del argv[i:i + 2]
    out, srcs = argv[0], argv[1:]
    files = []
    for s in srcs:
        if os.path.isdir(s):
            for root, _, fs in os.walk(s):
                files += [os.path.join(root, f) for f in fs if f.endswith("counter_collection.csv")]
        else:
            files.append(s)
    csv.field_size_limit(1 << 30)
    agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    dur = defaultdict(lambda: [0, 0.0])
    for f in files:
        seen = set()
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                fam = family(row["Kernel_Name"])
                a = agg[fam][row["Counter_Name"]]
                a[0] += 1
                a[1] += float(row["Counter_Value"])
                key = (row["Dispatch_Id"],)
                if key not in seen:
                    seen.add(key)
                    d = dur[fam]
                    d[0] += 1
                    d[1] += (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3
    res = OrderedDict()
    for fam in sorted(agg, key=lambda k: -dur[k][1]):
        e = OrderedDict(launches=max(v[0] for v in agg[fam].values()),
                        mean_us_under_pmc=round(dur[fam][1] / max(dur[fam][0], 1), 2))
        for cname, (n, tot) in sorted(agg[fam].items()):
            e[cname + "_per_launch"] = tot / n
        if "FETCH_SIZE" in agg[fam]:
            n, tot = agg[fam]["FETCH_SIZE"]
            e["hbm_read_bytes_per_launch"] = 2.0 * 1024.0 * tot / n       # KB -> B, x2 gfx950 correction
        if "WRITE_SIZE" in agg[fam]:
            n, tot = agg[fam]["WRITE_SIZE"]
            e["hbm_write_bytes_per_launch"] = 1024.0 * tot / n
        if agg[fam].get("SQ_VALU_MFMA_BUSY_CYCLES", [0, 0.0])[1] > 0 and "SQ_BUSY_CU_CYCLES" in agg[fam]:
            e["mfma_busy"] = agg[fam]["SQ_VALU_MFMA_BUSY_CYCLES"][1] / (4.0 * agg[fam]["SQ_BUSY_CU_CYCLES"][1])
        res[fam] = e
    if iterations:
        rd = wr = us = 0.0
        fams = []
        for fam, e in res.items():
            if re.search(GEMM_FAMILY, fam) and "hbm_read_bytes_per_launch" in e:
                fams.append(fam)
                rd += e["launches"] * e["hbm_read_bytes_per_launch"]
                wr += e["launches"] * e.get("hbm_write_bytes_per_launch", 0.0)
                us += e["launches"] * e["mean_us_under_pmc"]
        reg = sum(res[f]["launches"] * (res[f]["hbm_read_bytes_per_launch"] + res[f].get("hbm_write_bytes_per_launch", 0.0))
                  for f in fams if re.search(REGULARISER, f)) / iterations
        res["_summary"] = OrderedDict(iterations=iterations, gemm_family_regex=GEMM_FAMILY, gemm_families=fams,
                                      regulariser_regex=REGULARISER,
                                      regulariser_gemm_hbm_bytes_per_iteration=reg,
                                      conv_attention_gemm_hbm_bytes_per_iteration=(rd + wr) / iterations - reg,
                                      gemm_family_hbm_read_bytes_per_iteration=rd / iterations,
                                      gemm_family_hbm_write_bytes_per_iteration=wr / iterations,
                                      gemm_family_hbm_bytes_per_iteration=(rd + wr) / iterations,
                                      gemm_family_us_per_iteration_under_pmc=us / iterations)
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    if "_summary" in res:
        print("_summary", {k: v for k, v in res["_summary"].items() if k != "gemm_families"})
    for fam, e in list(res.items())[:40]:
        print("%-70s %s" % (fam[:70], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()}))


if __name__ == "__main__":
    main()

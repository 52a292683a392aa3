#!/usr/bin/env python
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel family.

    python tools/summarize_pmc.py OUT.json DIR_OR_CSV [DIR_OR_CSV ...]

Every pass (one directory per --pmc pass, see profiles/README.md) contributes its counters; per
kernel family (name up to the first '<' / '(') the script reports launches, the mean of each
counter per launch and, for FETCH_SIZE / WRITE_SIZE, bytes per launch with the corrections of
MI355X_MICROARCH.md "HBM [CDNA4]": both counters are in KiB-like units of 1 KB (rocprofv3
derived metric, KB), FETCH_SIZE on gfx950 reports half of a wide coalesced read and is doubled.
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


def main():
    out, srcs = sys.argv[1], sys.argv[2:]
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
        res[fam] = e
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    for fam, e in list(res.items())[:40]:
        print("%-70s %s" % (fam[:70], {k: (round(v, 1) if isinstance(v, float) else v) for k, v in e.items()}))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Average the rocprofv3 --pmc counters per kernel (last 10 launches) from a counter_collection.csv."""
import collections, csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r["Kernel_Name"].split("(")[0][-60:]
    agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    agg[k]["_dur_ns"].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in agg.items():
    if not any(x in k for x in sys.argv[2:] or ["substeps", "engage"]):
        continue
    print(k, "launches", len(v["_dur_ns"]) // max(1, len(v) - 1))
    for c in sorted(v):
        x = v[c][-10:]
        print(f"   {c:28s} {sum(x)/len(x):14.4g}")

#!/usr/bin/env python3
"""Dev tool: fold rocprofv3 --pmc SQ_* passes (counter_collection.csv files, any number) into one JSON: per kernel the
sum of every counter over its dispatches.   usage: sq_summary.py <out.json> <counter_collection.csv>..."""
import collections, csv, json, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for path in sys.argv[2:]:
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]] += float(r["Counter_Value"])
out = {k: {c: int(v) for c, v in sorted(cs.items())} for k, cs in sorted(agg.items())}
for k, cs in out.items():
    if cs.get("SQ_WAVE_CYCLES"):
        cs["wait_any_over_wave_cycles"] = round(cs.get("SQ_WAIT_ANY", 0) / cs["SQ_WAVE_CYCLES"], 3)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out.get("k_search_fused", {})))

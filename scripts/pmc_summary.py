#!/usr/bin/env python3
"""Dev tool: fold the two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, MI355X_MICROARCH.md §HBM)
into profiles/<tag>_pmc_fetch_write_summary.json: per kernel the dispatch count, the counter sum (KB, as reported)
and MB per dispatch.  FETCH_SIZE is left uncorrected (the guide's x2 applies to wide coalesced streaming reads,
which these kernels are not) — bench.py reads the state-op kernel's figures from the newest summary.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import json
import sys


def fold(path, name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != name:
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: {"dispatches": n, "sum_KB": round(v, 1), "per_dispatch_MB": round(v * 1024 / n / 1e6, 3)} for k, (n, v) in agg.items()}


out = {"FETCH_SIZE": fold(sys.argv[1], "FETCH_SIZE"), "WRITE_SIZE": fold(sys.argv[2], "WRITE_SIZE"),
       "note": "one bench.py step (--steps 1 --warmup 0, D4G_LANES=1); KB as reported by rocprofv3; corrected_per_dispatch_MB = x2 for the "
               "kernels whose loads are 16 bytes per lane and contiguous (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports half of those)"}
for k in ("k_exec_state_ops", "k_search_fused", "k_jump_streams", "k_lz_parse", "k_write"):   # 16-byte-per-lane streaming loads (records / uint4 entries / window staging)
    if k in out["FETCH_SIZE"]:
        out["FETCH_SIZE"][k]["corrected_per_dispatch_MB"] = round(2 * out["FETCH_SIZE"][k]["per_dispatch_MB"], 3)
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: out[k].get("k_exec_state_ops") for k in ("FETCH_SIZE", "WRITE_SIZE")}))

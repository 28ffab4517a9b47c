#!/usr/bin/env python3
"""Print the headline numbers of a bench.py log (the JSON line): usage tools/bench_summary.py <log>..."""
import json
import sys

for path in sys.argv[1:]:
    for line in open(path):
        if not line.startswith("{"):
            continue
        j = json.loads(line)
        r = j["roofline"]
        print(f"{path}: {j['value']:.0f} samples/s, {j['ms_per_step']} ms/step; roofline frac {r['frac']} ({r['ms_per_launch']} ms)")
        print("   per kernel ms:", {k: v["ms_per_launch"] for k, v in r["per_kernel"].items()})
        print("   stages ms:", r.get("block_stage_ms_per_step"))
        for key in ("global_2048", "greedy", "scaled"):
            if key in j:
                print(f"   {key}: {j[key]['value']:.1f} {j[key].get('unit', '')} ({j[key].get('ms_per_step', j[key].get('ms_per_batch'))} ms)")

#!/usr/bin/env python3
"""Per-kernel averages of whatever counters one rocprofv3 --pmc pass collected.
    python tools/pmc_table.py <counter_collection.csv> [min_us]"""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    names = []
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)[:56]
        key = (n, r["Grid_Size"])
        if r["Counter_Name"] not in names:
            names.append(r["Counter_Name"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    print(f"{'kernel':56s} {'grid':>9s} {'us':>8s} " + " ".join(f"{n[-22:]:>22s}" for n in names))
    rows = []
    for key, c in agg.items():
        d = sum(dur[key]) / len(dur[key]) / 1e3
        if d >= min_us:
            rows.append((d, key, [sum(c[n]) / len(c[n]) if n in c else float("nan") for n in names]))
    for d, key, vals in sorted(rows, reverse=True):
        print(f"{key[0]:56s} {key[1]:>9s} {d:8.1f} " + " ".join(f"{v:22.4g}" for v in vals))


if __name__ == "__main__":
    main()

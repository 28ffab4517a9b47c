#!/usr/bin/env python3
"""Last N launches of a rocprofv3 kernel trace csv: start (us, relative), duration, gap to the previous end, name."""
import csv
import re
import sys

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows = rows[-n:]
t0, prev = rows[0][0], rows[0][0]
for s, e, name in rows:
    name = re.sub(r"\(anonymous namespace\)::|^void ", "", name)
    name = re.sub(r"\(.*", "", name)[:60]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {(s - prev) / 1e3:7.1f}  {name}")
    prev = e

#!/usr/bin/env python3
"""Timeline of the last training step in a rocprofv3 kernel trace (the *_kernel_trace.csv of
`rocprofv3 --kernel-trace -- python3 bench.py --steps K --stage-steps 0 ...`: without --stage-steps 0 the trace ends
with bench.py's five untimed steps that bracket every stage with events, whose brackets leave the device idle between
stages): every launch with its start relative to the step's first kernel, its duration, its queue, and the gap since the
previous kernel ended on any queue.
Usage: python tools/step_timeline.py <kernel_trace.csv> [steps back from the end, default 2: the last step may be cut short]"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:70]


def main():
    path = sys.argv[1]
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    # a step ends with its (last) adam_kernel
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 8]
    if len(ends) < 2:
        raise SystemExit("fewer than two steps found")
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    if len(ends) < back + 1:
        raise SystemExit("not enough steps in the trace")
    a, b = ends[-back - 1] + 1, ends[-back] + 1
    b = min(b, len(rows) - 1)
    step = rows[a:b]
    t0 = step[0][0]
    busy_end = t0
    idle = 0
    print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>7s} q  kernel")
    for s, e, n, q in step:
        gap = max(0, s - busy_end)
        idle += gap
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap / 1e3:7.1f} {q:>2s} {short(n)}")
        busy_end = max(busy_end, e)
    print(f"step span {(rows[b][0] - t0) / 1e3:.1f} us, {len(step)} launches, device idle {idle / 1e3 + max(0, rows[b][0] - busy_end) / 1e3:.1f} us, "
          f"sum of kernel durations {sum(e - s for s, e, _, _ in step) / 1e3:.1f} us")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Per-kernel statistics of the TIMED steps of a rocprofv3 kernel trace of bench.py (the *_kernel_trace.csv of
`rocprofv3 --kernel-trace -- python3 bench.py --steps K --stage-steps 0 ...`): the last K training steps of the
trace -- a step ends with its (last) adam_kernel -- so spin-up and warm-up launches, and each kernel's first,
several-times-longer launch in the process, stay out of the averages.

    python tools/trace_steady_stats.py <kernel_trace.csv> <K> [bench log with the JSON line]

With the bench log, the line's roofline.ms_per_launch of the dominant kernel is printed beside the trace's average
of head_fwd_kernel and the two are required to agree within 3 % (exit code 1 otherwise)."""
import collections
import csv
import json
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*", "", name)


def main():
    path, K = sys.argv[1], int(sys.argv[2])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Grid_Size", "")))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    ends = [i for k, i in enumerate(adam) if k + 1 == len(adam) or adam[k + 1] - i > 8]
    if len(ends) < K + 1:
        raise SystemExit(f"{len(ends)} steps in the trace, {K} + 1 needed")
    a, b = ends[-K - 1] + 1, ends[-1] + 1
    steps = rows[a:b]
    span = (steps[-1][1] - steps[0][0]) / 1e6
    agg = collections.defaultdict(list)
    for s, e, n, g in steps:
        agg[n].append((e - s) / 1e3)
    total = sum(sum(v) for v in agg.values())
    print(f"# the last {K} steps of the trace: {len(steps)} launches, {span:.3f} ms from the first kernel's start to the last one's end "
          f"= {span / K:.4f} ms per step; sum of kernel durations {total / 1e3 / K:.4f} ms per step")
    print("kernel,calls,calls_per_step,avg_us,min_us,max_us,total_us_per_step,share_of_kernel_time")
    for n, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"\"{n}\",{len(v)},{len(v) / K:.2f},{sum(v) / len(v):.2f},{min(v):.2f},{max(v):.2f},{sum(v) / K:.2f},{sum(v) / total:.4f}")
    if len(sys.argv) > 3:
        line = None
        for l in open(sys.argv[3]):
            if l.startswith("{"):
                line = json.loads(l)
        if line and line.get("roofline"):
            want = line["roofline"]["ms_per_launch"] * 1e3
            # the training forward is the instantiation with the longest launches (the others are the block's forward products)
            got = sorted((sum(v) / len(v) for n, v in agg.items() if n.startswith("head_fwd_kernel")), reverse=True)
            if got:
                dev = abs(got[0] - want) / want
                print(f"# head_fwd_kernel: trace average {got[0]:.1f} us, bench line roofline.ms_per_launch {want:.1f} us, difference {dev * 100:.1f} %")
                print(f"# step: trace {span / K:.4f} ms, bench line {line['ms_per_step']} ms")
                sys.exit(0 if dev <= 0.03 else 1)


if __name__ == "__main__":
    main()

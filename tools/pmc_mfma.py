#!/usr/bin/env python3
"""MFMA utilisation and wave-state shares per kernel from one rocprofv3 counter pass
(--pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE) over bench.py; per kernel and grid size, the median over its launches.

    python tools/pmc_mfma.py <counter_collection.csv> [min_us]

MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs): GRBM_GUI_ACTIVE is summed
over the 8 XCDs.  Clock = GRBM_GUI_ACTIVE/8 / kernel duration.  The wait / active shares are of
SQ_WAVE_CYCLES (wait_any: parked on s_waitcnt or a barrier; wait_inst: issue stalls)."""
import collections
import csv
import re
import sys


def main():
    path = sys.argv[1]
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)[:60]
        key = (n, r["Grid_Size"])
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    out = []
    for key, c in agg.items():
        if "GRBM_GUI_ACTIVE" not in c:
            continue
        # medians: a kernel's first launch in a process runs several times longer than the rest
        med = lambda v: sorted(v)[len(v) // 2]                                  # noqa: E731
        m = lambda k: med(c[k]) if k in c else float("nan")                     # noqa: E731
        d = med(dur[key]) / 1e3
        if d < min_us:
            continue
        gui = m("GRBM_GUI_ACTIVE") / 8
        wc = m("SQ_WAVE_CYCLES")
        out.append((d, key, m("SQ_VALU_MFMA_BUSY_CYCLES") / (gui * 1024), m("SQ_WAIT_ANY") / wc, m("SQ_WAIT_INST_ANY") / wc,
                    m("SQ_ACTIVE_INST_ANY") / wc, gui / (d * 1e3)))
    print(f"{'kernel':60s} {'grid':>9s} {'us':>8s}  MFMA-busy  wait_any  wait_inst  active   clock")
    for d, key, u, wa, wi, ac, clk in sorted(out, reverse=True):
        print(f"{key[0]:60s} {key[1]:>9s} {d:8.1f}  {u * 100:8.1f}%  {wa * 100:7.0f}%  {wi * 100:8.0f}%  {ac * 100:5.0f}%  {clk:.2f} GHz")


if __name__ == "__main__":
    main()

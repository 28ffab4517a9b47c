#!/usr/bin/env python3
"""Static check of the compiled kernels' matrix loops: no scratch-memory instruction (a register spill, or an array the
compiler could not keep in registers) may sit inside a loop that issues MFMAs.  An accumulator array indexed by a run-time
value goes to scratch memory as a whole -- every MFMA then stores its result and loads it back -- and a spill inside a
ring kernel's main loop is a vector memory operation between the counted LDS-DMA loads.  Both still compute the right
values, so only the ISA shows them.

Usage: python tools/check_mfma_loops.py [--warn-only] <kernel.s>   (the .s comes from hipcc --cuda-device-only -S)
--warn-only: report, exit 0 (kernels whose loops use no counted waits: a spill there costs time, not correctness)."""
import re
import sys


def main():
    warn_only = "--warn-only" in sys.argv
    path = [a for a in sys.argv[1:] if not a.startswith("--")][0]
    txt = open(path).read().splitlines()
    starts = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    bad = 0
    for idx, (i, name) in enumerate(starts):
        end = starts[idx + 1][0] if idx + 1 < len(starts) else len(txt)
        body = txt[i:end]
        labels = {}
        for j, l in enumerate(body):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                labels[m.group(1)] = j
        loops = []
        for j, l in enumerate(body):
            m = re.search(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
            if m and m.group(1) in labels and labels[m.group(1)] < j:
                loops.append((labels[m.group(1)], j))
        # innermost loops only: those that contain no other loop
        inner = [(a, b) for a, b in loops if not any((c > a or d < b) and c >= a and d <= b for c, d in loops if (c, d) != (a, b))]
        for a, b in inner:
            ins = [l.strip() for l in body[a:b + 1] if l.startswith("\t")]
            nm = sum(1 for x in ins if x.startswith("v_mfma"))
            sc = [x for x in ins if x.startswith("scratch_")]
            if nm >= 16 and sc:
                bad += 1
                print(f"{name}: {len(sc)} scratch instruction(s) in a loop of {nm} MFMAs, e.g. {sc[0][:70]}")
    if not bad:
        print(f"{path}: no scratch instruction inside an MFMA loop")
    sys.exit(1 if bad and not warn_only else 0)


if __name__ == "__main__":
    main()

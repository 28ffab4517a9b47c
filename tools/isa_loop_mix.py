#!/usr/bin/env python3
"""Instruction mix of the loops of a kernel in a device ISA dump (hipcc --cuda-device-only -S): for every backward
branch, the instructions between its target and itself by class (MFMA / transcendental / other VALU / LDS / scalar /
memory).  Usage: python tools/isa_loop_mix.py <file.s> <substring of the mangled kernel name> ..."""
import collections
import re
import sys


def main():
    txt = open(sys.argv[1]).read().splitlines()
    wants = sys.argv[2:]
    starts = [(i, re.match(r"^(_Z\w+):", l).group(1)) for i, l in enumerate(txt) if re.match(r"^_Z\w+:", l)]
    for idx, (i, name) in enumerate(starts):
        if wants and not any(w in name for w in wants):
            continue
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
        regs = [l for l in body if "NumVgprs" in l or "NumAgprs" in l or "ScratchSize" in l or "Occupancy" in l]
        print(name[:110])
        for l in regs:
            print("   ", l.strip().lstrip(";").strip())
        for a, b in loops:
            c = collections.Counter()
            for l in body[a:b + 1]:
                if not l.startswith("\t"):
                    continue
                x = l.strip().split()[0] if l.strip() else ""
                if not x or x.startswith((".", ";")):
                    continue
                if x.startswith("v_mfma"): c["mfma"] += 1
                elif x.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt")): c["trans"] += 1
                elif x.startswith("v_"): c["valu"] += 1
                elif x.startswith("ds_"): c["lds"] += 1
                elif x.startswith("s_waitcnt"): c["waitcnt"] += 1
                elif x.startswith("s_"): c["salu"] += 1
                elif x.startswith(("buffer", "global", "scratch", "flat")): c["vmem"] += 1
                else: c["other"] += 1
            if c["mfma"] > 0:
                print(f"    loop of {b - a} lines:", dict(c))


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Turn two rocprofv3 counter passes over `bench.py` (one with --pmc FETCH_SIZE, one with
--pmc WRITE_SIZE; the two cannot share a pass) into per-launch HBM-side byte counts of the four
vocabulary-sized GEMM launches at batch 256.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of a
16-byte-per-lane streaming read at 64 bytes, so it is doubled; WRITE_SIZE is exact for 16-byte
stores."""
import collections
import csv
import json
import re
import sys

# span name in bench.py -> (kernel template arguments, workgroups) at batch 256, V = 50260, d = 768
LAUNCHES = {
    # mode 3: numerators against one reference per row (the training default); mode 0 when that is switched off
    "logits_ce_fwd": [("head_fwd_kernel<2, 4, 8, 4, true, false, 3>", None)],
    "logits_ce_fwd_slot_reference": ("head_fwd_kernel<2, 4, 8, 4, true, false, 0>", None),
    "ce_bias_grad": ("wcolsum_bf16_kernel", None),          # only with --no-ce-bias-in-wgrad
    "logits_ce_bwd_slab": ("gemm_bf16_dma_kernel<false, false, 2, 2, 2, 8, 3>", None),
    # two launches: the whole rounds of 256x256 tiles (170 x 3) and the split-K tail (27 x 3 tiles, 3 slabs)
    "head_wgrad_slab": [("gemm_bf16_pipe_kernel<true, true, 0, 8, 1>", 170 * 3 * 256), ("gemm_bf16_pipe_kernel<true, true, 0, 8, 1>", 27 * 3 * 3 * 256)],
    "head_dgrad_slab": ("gemm_bf16_pipe_kernel<false, true, 0, 8, 0>", 32 * 3 * 8 * 256),
}


def load(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
        agg[(name, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return agg


def pick(agg, pattern, grid):
    vals = [v for (n, g), vs in agg.items() if pattern in n and (grid is None or g == grid) for v in vs]
    return sum(vals) / len(vals) if vals else None


def main():
    fetch, write, out = load(sys.argv[1]), load(sys.argv[2]), sys.argv[3]
    res = {}
    for span, parts in LAUNCHES.items():
        parts = parts if isinstance(parts, list) else [parts]
        fs, wr = [pick(fetch, pt, g) for pt, g in parts], [pick(write, pt, g) for pt, g in parts]
        if any(v is None for v in fs + wr):
            continue
        f, w = sum(fs), sum(wr)
        res[span] = {"fetch_size_kib": round(f), "write_size_kib": round(w),
                     "hbm_bytes_per_launch": int(2 * f * 1024 + w * 1024)}
    doc = {"collected_on": {"per_gpu_batch": 256, "dtype": "bf16", "n_blocks": 1, "patches": 196, "vocab_slab": "whole vocabulary"},
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 "
                     "(FETCH_SIZE doubled: gfx950 tallies 128-byte requests of 16-byte-per-lane streaming reads at 64 bytes); mean over the pass's launches",
           "launches": res}
    json.dump(doc, open(out, "w"), indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()

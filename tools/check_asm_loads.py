#!/usr/bin/env python3
"""Static check of the kernels that read LDS fragments through inline assembly (ds_read_b64_tr_b16 the compiler
cannot see into): between such a read and the next s_waitcnt that covers LDS (lgkmcnt), no other instruction may
read or write the read's destination registers.  The compiler does not know the registers are still in flight: a
copy it inserts there moves stale contents, and a register it re-uses there (for an address, say) is overwritten
when the data lands.

Usage: python tools/check_asm_loads.py <kernel.s> [substring of the kernel names to check]
       (the .s comes from hipcc ... -save-temps)."""
import re
import sys


def regs_of(tok):
    """v[12:13] -> {12, 13}; v7 -> {7}; anything else -> empty."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def waits_lgkm(line):
    if not line.startswith("s_waitcnt"):
        return False
    if "lgkmcnt" in line:
        return True
    m = re.search(r"s_waitcnt\s+(0x[0-9a-fA-F]+|\d+)\s*$", line)      # raw immediate: lgkmcnt is bits 11:8
    if m:
        imm = int(m.group(1), 0)
        return ((imm >> 8) & 0xF) != 0xF
    return False


def check(lines, name):
    pending = {}       # register -> line number of the read that targets it
    bad = []
    for no, raw in lines:
        line = raw.split(";")[0].strip()
        if not line or line.endswith(":") or line.startswith("."):
            continue
        if waits_lgkm(line):
            pending.clear()
            continue
        toks = re.findall(r"v\[\d+:\d+\]|v\d+", line)
        if line.startswith("ds_read_b64_tr_b16"):
            dst = regs_of(toks[0]) if toks else set()
            used = set().union(*[regs_of(t) for t in toks[1:]]) if len(toks) > 1 else set()
            for r in used & set(pending):
                bad.append((no, raw.strip(), pending[r]))
            for r in dst:
                if r in pending:
                    bad.append((no, raw.strip(), pending[r]))
                pending[r] = no
            continue
        touched = set().union(*[regs_of(t) for t in toks]) if toks else set()
        for r in touched & set(pending):
            bad.append((no, raw.strip(), pending[r]))
    return bad


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    text = open(path).read().splitlines()
    kernels, cur, name = [], None, None
    for i, l in enumerate(text, 1):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name, cur = m.group(1), []
            kernels.append((name, cur))
        elif cur is not None:
            cur.append((i, l))
            if "s_endpgm" in l:
                cur = None
    total = 0
    for name, lines in kernels:
        if want not in name or not any("ds_read_b64_tr_b16" in l for _, l in lines):
            continue
        bad = check(lines, name)
        total += len(bad)
        print(f"{name}: {len(bad)} instruction(s) touch a fragment register still in flight")
        for no, ins, src in bad[:12]:
            print(f"   line {no}: {ins}    (read issued at line {src})")
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()

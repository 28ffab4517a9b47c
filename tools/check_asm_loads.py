#!/usr/bin/env python3
"""Static check of the kernels that read LDS fragments through inline assembly (ds_read_b64_tr_b16, and ds_read_b128
between ASMSTART/ASMEND markers, which the compiler cannot see into): between such a read and the next s_waitcnt that
covers LDS (lgkmcnt) on any path of the control flow graph, no other instruction may read or write the read's
destination registers.  The compiler does not know the registers are still in flight: a
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
    """Forward dataflow over the kernel's control flow graph: the set of fragment registers in flight at an instruction
    is the union over its predecessors (fall-through, and every branch that names its label)."""
    ins = []                       # (line number, text, label or None)
    labels = {}
    for no, raw in lines:
        line = raw.split(";")[0].strip()
        if not line or line.startswith("."):
            if line.endswith(":") and line.startswith(".L"):
                labels[line[:-1]] = len(ins)
            continue
        if line.endswith(":"):
            labels[line[:-1]] = len(ins)
            continue
        ins.append((no, line, raw.strip()))
    n = len(ins)
    state_in = [None] * (n + 1)    # register -> line of the read that targets it; None: not reached yet
    state_in[0] = {}
    work = [0]
    bad = {}

    def merge(idx, st):
        if idx > n:
            return
        cur = state_in[idx]
        if cur is None:
            state_in[idx] = dict(st)
            work.append(idx)
            return
        grew = False
        for r, l in st.items():
            if r not in cur:
                cur[r] = l
                grew = True
        if grew:
            work.append(idx)

    while work:
        i = work.pop()
        if i >= n:
            continue
        pending = dict(state_in[i])
        no, line, raw = ins[i]
        if waits_lgkm(line):
            pending.clear()
        else:
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", line)
            if line.startswith("ds_read_b64_tr_b16") or (line.startswith("ds_read_b128") and (no in ASM_LINES)):
                dst = regs_of(toks[0]) if toks else set()
                used = set().union(*[regs_of(t) for t in toks[1:]]) if len(toks) > 1 else set()
                for r in used & set(pending):
                    bad[(no, raw)] = pending[r]
                for r in dst:
                    if r in pending:
                        bad[(no, raw)] = pending[r]
                    pending[r] = no
            else:
                touched = set().union(*[regs_of(t) for t in toks]) if toks else set()
                for r in touched & set(pending):
                    bad[(no, raw)] = pending[r]
        m = re.match(r"s_c?branch\w*\s+(\S+)", line)
        if m and m.group(1) in labels:
            merge(labels[m.group(1)], pending)
        if not (line.startswith("s_branch") or line.startswith("s_endpgm") or line.startswith("s_setpc")):
            merge(i + 1, pending)
    return [(no, raw, src) for (no, raw), src in sorted(bad.items())]


ASM_LINES = set()                  # line numbers of instructions that stand between ;;#ASMSTART and ;;#ASMEND


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    text = open(path).read().splitlines()
    inside = False
    for i, l in enumerate(text, 1):
        if "#ASMSTART" in l:
            inside = True
        elif "#ASMEND" in l:
            inside = False
        elif inside:
            ASM_LINES.add(i)
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
        has_asm_read = any("ds_read_b64_tr_b16" in l or (no in ASM_LINES and l.strip().startswith("ds_read")) for no, l in lines)
        if want not in name or not has_asm_read:
            continue
        bad = check(lines, name)
        total += len(bad)
        print(f"{name}: {len(bad)} instruction(s) touch a fragment register still in flight")
        for no, ins, src in bad[:12]:
            print(f"   line {no}: {ins}    (read issued at line {src})")
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()

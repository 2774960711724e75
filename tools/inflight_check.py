#!/usr/bin/env python3
"""ISA check for kernels whose loads are inline asm with hand-placed s_waitcnt (tools/tune_store.hip, k_mix_pipe): the compiler believes
an asm load's destination register holds its value as soon as the statement has been issued, so it may copy it, or — if it thinks the
value dead — reuse the register, while the load is still in flight.  This walks each kernel's instruction stream (from the top to the last back-edge, then
twice more around every loop), keeps the queue of outstanding vector-memory operations the way vmcnt counts them (in order; `s_waitcnt
vmcnt(N)` retires all but the newest N) and reports every instruction that reads or writes the destination of a load still in flight.

    tools/inflight_check.py file.s [kernel-name-substring]
"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v(\d+)", tok)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def check(name, lines):
    labels = {l[:-1].split(":")[0]: i for i, l in enumerate(lines) if re.match(r"\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(lines):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    # the path walked: from the top to the last back-edge, then twice more around each outermost loop (header .. its last back-edge)
    outer = {}
    for lo, hi in loops:
        outer[lo] = max(outer.get(lo, 0), hi)
    if outer:
        last = max(outer.values())
        order = list(range(last + 1))
        for lo, hi in sorted(outer.items()):
            order += list(range(lo, hi + 1)) * 2
    else:
        order = list(range(len(lines)))
    queue, bad = [], []
    for i in order:
        l = lines[i]
        if l.startswith(".") or l.endswith(":"):
            continue
        op, _, rest = l.partition(" ")
        toks = [t.strip() for t in rest.split(",")]
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", l)
        if m:
            n = int(m.group(1))
            while len(queue) > n:
                queue.pop(0)
            continue
        touched = set()
        for t in toks:
            touched |= regs(t.split(" ")[0])
        inflight = set().union(*[d for d in queue]) if queue else set()
        is_vmem = op.startswith(("global_", "buffer_", "flat_", "scratch_"))
        if touched & inflight:
            bad.append((i, l, sorted(touched & inflight)))
        if is_vmem:
            dest = regs(toks[0].split(" ")[0]) if ("load" in op or ("atomic" in op and "sc0" in l)) else set()
            queue.append(dest)
    return bad


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    total = 0
    for m in re.finditer(r"^(_Z\w+):.*?\n(.*?)\.Lfunc_end", text, re.S | re.M):
        name, body = m.group(1), m.group(2)
        if want not in name:
            continue
        lines = [re.sub(r"\s*;.*$", "", l).strip() for l in body.split("\n")]
        lines = [l for l in lines if l]
        bad = check(name, lines)
        print(f"{name}: {len(lines)} instructions, {len(bad)} accesses to a register with a load in flight")
        for i, l, r in bad[:6]:
            print(f"    [{i}] {l}    <- v{r}")
        total += len(bad)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())

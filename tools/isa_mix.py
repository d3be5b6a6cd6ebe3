#!/usr/bin/env python3
"""Instruction mix of one kernel in a hipcc -save-temps .s file, per basic block and per loop.

usage: isa_mix.py file.s kernel-substring [--blocks]
Classes: mfma, valu (v_*), salu (s_*, no waitcnt/branch), lds (ds_*), vmem (global_/buffer_/scratch_),
         wait (s_waitcnt/s_nop), acc (v_accvgpr_*), branch.
A loop = a label that is the target of a backward branch; its body = lines between label and branch.
"""
import re
import sys
from collections import Counter


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_accvgpr"):
        return "acc"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith("scratch_"):
        return "scratch"
    if op.startswith(("s_waitcnt", "s_nop")):
        return "wait"
    if op.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc", "s_barrier")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def main():
    path, key = sys.argv[1], sys.argv[2]
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\w+:", l) and key in l:
            start = i
            break
    if start is None:
        sys.exit("kernel not found")
    end = start
    while not lines[end].startswith("\t.section") and not lines[end].startswith(".Lfunc_end"):
        end += 1
    body = lines[start:end]
    labels = {}
    ins = []  # (idx, op, text)
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = l.strip()
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        ins.append((op, t))
    total = Counter(classify(op) for op, _ in ins)
    print("kernel total:", dict(total), "n =", len(ins))
    loops = []
    for i, (op, t) in enumerate(ins):
        if op.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] <= i:
                loops.append((labels[tgt], i, tgt))
    loops.sort()
    for a, b, tgt in loops:
        c = Counter(classify(op) for op, _ in ins[a:b + 1])
        vops = Counter(op for op, _ in ins[a:b + 1] if classify(op) == "valu")
        print(f"loop {tgt}: ins[{a}:{b}] n={b - a + 1} {dict(c)}")
        if "--ops" in sys.argv:
            print("   valu ops:", vops.most_common(25))


if __name__ == "__main__":
    main()

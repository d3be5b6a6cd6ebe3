#!/usr/bin/env python3
"""isa_checks.py -- static checks of the compiler's assembly that a functional emulator cannot make (tools/emu/gfx950_emu.py has no timing).

For every v_mfma_f64_4x4x4 whose destination is a VECTOR register (what -amdgpu-mfma-vgpr-form and the 256-register builds produce) the
distance, in wait states, to the first instruction that reads or overwrites that register - along the fall-through path AND across branch
edges - against the wait states LLVM's hazard recognizer requires on gfx90a+ for a DGEMM 4x4x4 result (GCNHazardRecognizer: VALU read /
write 6, memory / LDS / export read 9, MFMA SrcA/B read 6; back-to-back SrcC forwarding is exempt).  A distance below the table would be
a hazard the compiler failed to pad; the histogram of the smallest distances shows the padding is there.  DEV / TEST infrastructure.

usage: isa_checks.py <kernel.s> [<kernel-name-substring>]
"""
import collections
import re
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
import gfx950_emu as E  # noqa: E402

NEED_VALU, NEED_MEM, NEED_MFMA_AB = 6, 9, 6


def vregs(op):
    return set(range(op.n, op.n + op.cnt)) if op.kind == "v" else set()


def check(path, kernel=None, verbose=True):
    insts, labels = E.parse_kernel(path, kernel)
    n = len(insts)

    def succs(i):
        ins = insts[i]
        if ins.op == "s_endpgm":
            return []
        if ins.op == "s_branch":
            return [labels[ins.ops[0].val]]
        out = [i + 1] if i + 1 < n else []
        if ins.op.startswith("s_cbranch"):
            out.append(labels[ins.ops[0].val])
        return out
    findings = []
    hist = collections.Counter()
    n_mfma = n_vdst = 0
    for i, ins in enumerate(insts):
        if not ins.op.startswith("v_mfma"):
            continue
        n_mfma += 1
        d = vregs(ins.ops[0])
        if not d:
            continue
        n_vdst += 1
        stack = [(j, 0) for j in succs(i)]
        best = {}
        while stack:
            j, cnt = stack.pop()
            if j >= n or cnt > 12 or best.get(j, 99) <= cnt:
                continue
            best[j] = cnt
            i2 = insts[j]
            op2 = i2.op
            stop = False
            if not op2.startswith("s_"):
                store = op2.startswith(("global_store", "ds_write", "buffer_store", "scratch_store", "flat_store"))
                mem = op2.startswith(("global_", "flat_", "buffer_", "ds_", "scratch_", "exp"))
                rd, wr = set(), set()
                if i2.ops:
                    if store:
                        for o in i2.ops:
                            rd |= vregs(o)
                    else:
                        wr |= vregs(i2.ops[0])
                        for o in i2.ops[1:]:
                            rd |= vregs(o)
                    if op2.startswith(("v_fmac", "v_writelane")):
                        rd |= vregs(i2.ops[0])
                if op2.startswith("v_mfma"):
                    ab = vregs(i2.ops[1]) | vregs(i2.ops[2])
                    if ab & d:
                        hist[("MFMA A/B read", cnt)] += 1
                        if cnt < NEED_MFMA_AB:
                            findings.append((ins.line, i2.line, cnt, "MFMA A/B read", ins.text, i2.text))
                        stop = True
                    if wr & d:
                        stop = True
                else:
                    if rd & d:
                        need = NEED_MEM if mem else NEED_VALU
                        hist[("memory read" if mem else "VALU read", cnt)] += 1
                        if cnt < need:
                            findings.append((ins.line, i2.line, cnt, "read", ins.text, i2.text))
                        stop = True
                    if wr & d:
                        hist[("overwrite", cnt)] += 1
                        if cnt < NEED_VALU:
                            findings.append((ins.line, i2.line, cnt, "overwrite", ins.text, i2.text))
                        stop = True
            if stop:
                continue
            nc = cnt + ((i2.ops[0].val + 1) if op2 == "s_nop" else 1)
            for k in succs(j):
                stack.append((k, nc))
    if verbose:
        print(f"{Path(path).name}{' [' + kernel + ']' if kernel else ''}: {n} instructions, {n_mfma} MFMAs, {n_vdst} with a vector-register destination")
        for kind in ("VALU read", "memory read", "MFMA A/B read", "overwrite"):
            row = sorted((c, k) for (t, c), k in hist.items() if t == kind)[:6]
            print(f"   smallest distances, {kind:14s}: " + ", ".join(f"{c} ws x{k}" for c, k in row))
        print(f"   below LLVM's DGEMM 4x4x4 table (VALU {NEED_VALU} / memory {NEED_MEM} / MFMA A,B {NEED_MFMA_AB}): {len(findings)}")
        for f in findings[:10]:
            print("     ", f)
    return findings


if __name__ == "__main__":
    sys.exit(1 if check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None) else 0)

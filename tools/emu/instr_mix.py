#!/usr/bin/env python3
"""instr_mix.py -- dynamic instruction mix of one emulated workgroup (what a wave ISSUES, by class), a CPU-side stand-in for SQ_INSTS_* counters.
usage: instr_mix.py <file.s> <kernel> [run_team_kernel options]"""
import sys, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import gfx950_emu as E
import run_team_kernel as R

def klass(op):
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith(("global_load", "global_store", "global_atomic")): return "vmem"
    if op.startswith("ds_"): return "lds"
    if op.startswith("s_nop"): return "s_nop"
    if op.startswith("s_waitcnt"): return "s_waitcnt"
    if op.startswith(("s_cbranch", "s_branch")): return "branch"
    if op.startswith("s_load"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith(("v_accvgpr",)): return "acc move"
    if "f64" in op: return "valu f64"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")): return "lane"
    return "valu 32"

def main():
    sfile, kernel = sys.argv[1], sys.argv[2]
    kw = {}
    it = iter(sys.argv[3:])
    for k in it:
        k = k.lstrip("-")
        if k == "warm": kw["warm"] = True; continue
        v = next(it)
        kw[{"headers": "csrc", "include": "inc", "batch": "B"}.get(k, k)] = int(v) if v.lstrip("-").isdigit() else v
    counts = collections.Counter(); ops = collections.Counter(); nops = [0]
    orig = E.Wave.run
    def run(self):
        insts = self.insts
        while True:
            ins = insts[self.pc]
            self.steps += 1
            counts[klass(ins.op)] += 1; ops[ins.op] += 1
            if ins.op == "s_nop": nops[0] += ins.ops[0].val + 1
            h = E._DISPATCH.get(ins.op) or E._resolve(ins.op)
            E._DISPATCH[ins.op] = h
            nxt = self.pc + 1
            r = h(self, ins)
            if r == "end": return
            self.pc = nxt if r is None else r
    E.Wave.run = run
    r = R.emulate(sfile, kernel, verbose=True, **kw)
    tot = sum(counts.values())
    print(f"{tot} instructions issued by the wave:")
    for k, v in counts.most_common(): print(f"   {k:10s} {v:7d}  {100.0 * v / tot:5.1f} %")
    print(f"   (s_nop wait states in total: {nops[0]})")
    est = counts["mfma"] * 16 + counts["valu f64"] * 8 + (counts["valu 32"] + counts["acc move"] + counts["lane"]) * 8 + nops[0] * 4 + (counts["salu"] + counts["branch"] + counts["s_waitcnt"] + counts["smem"]) * 4 + (counts["lds"] + counts["vmem"]) * 8
    print(f"   issue-time model of a lone wave (MFMA 16, vector 8, scalar 4 cycles, nop wait state 4): {est} cycles = {est / 2.4e3:.1f} us at 2.4 GHz")
    print("   top opcodes:", ", ".join(f"{o} {n}" for o, n in ops.most_common(25)))

if __name__ == "__main__":
    main()

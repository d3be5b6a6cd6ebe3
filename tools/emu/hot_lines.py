#!/usr/bin/env python3
"""hot_lines.py -- per-instruction execution counts of one emulated workgroup, printed as annotated assembly of the hottest loop(s).
usage: hot_lines.py <file.s> <kernel> <min_count> [run_team_kernel options]"""
import sys, collections
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import gfx950_emu as E
import run_team_kernel as R
sfile, kernel, minc = sys.argv[1], sys.argv[2], int(sys.argv[3])
kw = {}
it = iter(sys.argv[4:])
for k in it:
    k = k.lstrip("-")
    if k == "warm": kw["warm"] = True; continue
    v = next(it)
    kw[{"headers": "csrc", "include": "inc", "batch": "B"}.get(k, k)] = int(v) if v.lstrip("-").isdigit() else v
cnt = collections.Counter()
def run(self):
    insts = self.insts
    while True:
        ins = insts[self.pc]
        self.steps += 1
        cnt[self.pc] += 1
        h = E._DISPATCH.get(ins.op) or E._resolve(ins.op)
        E._DISPATCH[ins.op] = h
        nxt = self.pc + 1
        r = h(self, ins)
        if r == "end": break
        self.pc = nxt if r is None else r
    self._insts_ref = insts
E.Wave.run = run
r = R.emulate(sfile, kernel, verbose=False, **kw)
insts, _ = E.parse_kernel(sfile, kernel)
for i, ins in enumerate(insts):
    if cnt[i] >= minc:
        print(f"{cnt[i]:5d}  {ins.line:7d}  {ins.text}")

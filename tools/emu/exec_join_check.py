#!/usr/bin/env python3
"""exec_join_check.py -- static check of the compiler's assembly for ONE class of miscompile: a vector instruction placed at the top of a
join block BEFORE the instruction that restores EXEC.

Divergent control flow on gfx9 is `s_and_saveexec_b64 sX, cond; s_cbranch_execz L; ...; L: s_or_b64 exec, exec, sX`.  Everything between the
label L and the restore runs with the REDUCED mask - with EXEC = 0 when the branch was taken.  The register allocator parks live values in
accumulation registers (v_accvgpr_write_b32) and may only put such a copy behind the restore; a copy in front of it saves the value in
some lanes only, and the reload under the full mask returns garbage in the others.  Found in round 4: with
-amdgpu-sched-strategy=iterative-ilp a build of k_team_as parked its pass counter that way (caught by the emulator's register poison: the
emulated pass counts left the oracle's).  The check lists every vector / memory instruction between a label that a s_cbranch_execz targets
and the next write of EXEC; a clean build has none.  DEV / TEST infrastructure.

usage: exec_join_check.py <file.s> [<kernel-name-substring>]      exit status 1 when something is found
"""
import re
import sys


def kernels(path):
    """[(name, [(line number, text), ...]), ...] of the functions of an assembly file."""
    out, cur, name = [], None, None
    for ln, raw in enumerate(open(path), 1):
        s = raw.split(";")[0].rstrip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            name, cur = m.group(1), []
            out.append((name, cur))
            continue
        if cur is None:
            continue
        t = s.strip()
        if t.startswith(".Lfunc_end") or t.startswith(".end_amdhsa_kernel"):
            cur = None
            continue
        if t:
            cur.append((ln, t))
    return out


WRITES_EXEC = re.compile(r"^s_\w+\s+exec\b|^s_\w*saveexec\w*\s")
VECTOR = re.compile(r"^(v_|global_|flat_|scratch_|ds_|buffer_(load|store|atomic))")


def check(path, kernel=None, verbose=True):
    found = []
    for name, body in kernels(path):
        if kernel and kernel not in name:
            continue
        targets = {m.group(1) for _, t in body for m in [re.match(r"s_cbranch_execz\s+(\S+)", t)] if m}
        for i, (ln, t) in enumerate(body):
            if t.endswith(":") and t[:-1] in targets:
                for ln2, t2 in body[i + 1:]:
                    if t2.endswith(":"):
                        continue                       # (fall-through into the next label: same reduced mask)
                    if WRITES_EXEC.match(t2) or t2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")):
                        break
                    if VECTOR.match(t2) and not t2.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
                        found.append((name, t[:-1], ln2, t2))
        # the same mistake where the compiler left the skip branch out (short bodies): s_and_saveexec sX; body; s_or_b64 exec, exec, sX in one
        # straight line.  A value parked in an accumulation register INSIDE the body, whose vector register was last written BEFORE the
        # mask was reduced, is a copy that belongs behind the restore (a value the body computed itself may be parked under the body's mask)
        open_at, written = None, set()
        for ln, t in body:
            if t.endswith(":") or t.startswith(("s_cbranch", "s_branch", "s_setpc", "s_endpgm")):
                open_at = None
                continue
            m = re.match(r"s_and_saveexec_b64\s+(s\[\d+:\d+\])", t)
            if m:
                open_at, written = m.group(1), set()
                continue
            if open_at:
                if WRITES_EXEC.match(t):
                    open_at = None
                    continue
                m = re.match(r"v_accvgpr_write_b32\s+a\d+,\s*v(\d+)$", t)
                if m and int(m.group(1)) not in written:
                    found.append((name, "(no label) " + open_at, ln, t))
                m = re.match(r"v_\w+\s+v(\d+|\[(\d+):(\d+)\])", t)
                if m and not t.startswith(("v_cmp", "v_accvgpr_write", "v_readlane", "v_readfirstlane")):
                    if m.group(2):
                        written.update(range(int(m.group(2)), int(m.group(3)) + 1))
                    else:
                        written.add(int(m.group(1)))
    if verbose:
        for name, lab, ln, t in found:
            print(f"{path}:{ln}: [{name[:60]}] behind {lab}, before EXEC is restored: {t}")
        print(f"{path}: {len(found)} vector instruction(s) between an execz join label and the EXEC restore")
    return found


if __name__ == "__main__":
    sys.exit(1 if check(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None) else 0)

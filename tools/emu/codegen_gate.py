#!/usr/bin/env python3
"""codegen_gate.py -- build-time gate of the translation units that are compiled with INTERNAL LLVM options (csrc/Makefile: ASFLAGS_HIP =
-amdgpu-mfma-vgpr-form, -amdgpu-sched-strategy=iterative-ilp).

Those options are worth +9 % on the headline, and one of them has miscompiled this code once (round 4: a register-parking copy in front of an
EXEC restore).  The nets that caught it lived in pytest only; a maintainer who builds with another ROCm and does not run the suite got
unguarded code generation.  This script is what `make` runs on the assembly of the flag builds BEFORE it links:
  1. hipcc must be the version the flag builds were validated on (HIPCC_EXPECT);
  2. tools/emu/exec_join_check.py: no vector instruction between a join label and the restore of EXEC;
  3. tools/emu/isa_checks.py: every MFMA result is read no earlier than LLVM's hazard table allows (the kernels the CPU suite scans).
Any finding -> the library is built with -DNMPC_DEFAULT_NOFLAG=1: the default-codegen twins (nmpc_qp.hip, nmpc_block.hip), which are always
in the library and held bit-equal to the flag builds on the GPU, become what runs by default, and nmpc_version() says so.

usage: codegen_gate.py --hipcc <hipcc> --expect <major.minor> --out <file> <nmpc_as.s> <nmpc_qpf.s> <nmpc_blockf.s>
writes two lines to --out: "0" (flag builds are the default) or "1" (default code generation), then the reason.  Always exits 0.
"""
import argparse
import subprocess
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))

HAZARD_SCAN = {"nmpc_as.s": ("k_team_asILb1ELb0ELi1EdEE", "k_team_asILb0ELb1ELi1EdEE"), "nmpc_qpf.s": ("k_team_qpILb1ELb0EdEE",), "nmpc_blockf.s": (None,)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hipcc", required=True)
    ap.add_argument("--expect", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("asm", nargs="+")
    a = ap.parse_args()
    reasons = []
    try:
        ver = subprocess.run([a.hipcc, "--version"], capture_output=True, text=True).stdout
    except OSError as e:
        ver = f"({e})"
    if f"HIP version: {a.expect}" not in ver:
        line = next((ln for ln in ver.splitlines() if "HIP version" in ln), "HIP version: unknown").strip()
        reasons.append(f"hipcc is '{line}', the flag builds were validated on {a.expect}.x")
    if not reasons:
        import exec_join_check as X
        import isa_checks as H
        for f in a.asm:
            found = X.check(f, verbose=False)
            if found:
                reasons.append(f"exec_join_check: {len(found)} vector instruction(s) in front of an EXEC restore in {Path(f).name} ({found[0][0][:40]} line {found[0][2]})")
                continue
            for k in HAZARD_SCAN.get(Path(f).name, (None,)):
                hz = H.check(f, k, verbose=False)
                if hz:
                    reasons.append(f"isa_checks: MFMA result read inside its hazard window in {Path(f).name} {k or ''}: {hz[0]}")
                    break
    mode = "1" if reasons else "0"
    Path(a.out).write_text(mode + "\n" + ("; ".join(reasons) if reasons else "flag builds checked: hipcc version, exec-join scan, MFMA hazard scan") + "\n")
    print(("codegen gate: DEFAULT code generation is what this library runs by default - " + "; ".join(reasons)) if reasons else
          "codegen gate: flag builds pass (hipcc version, exec-join scan, MFMA hazard scan)")


if __name__ == "__main__":
    main()

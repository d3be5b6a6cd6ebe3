#!/usr/bin/env python3
"""what_runs_when.py -- the per-kernel half of DESIGN.md section 0b: registers, spills, scratch and occupancy of every solver kernel, from the
compiler's own report (`make -C rotors_mpc_controller_amd/csrc resource-usage`, i.e. -Rpass-analysis=kernel-resource-usage on the five
translation units exactly as the library builds them).  The dispatch half of that table (which configuration launches which kernel) is
nmpc_capi.hip's launch_f64 / launch_split, restated in DESIGN.md; `nmpc_debug_last_schedule` reports it per solve.

usage: what_runs_when.py [resource_usage.txt]       (without a file it runs the make target itself: minutes)
"""
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def demangle_args(sym):
    """template arguments of the kernel instantiations this library has: k_team_as<SHARED, TRAJ, OCC, TI>, k_team_qp<SHARED, TRAJ, TI>, ..."""
    m = re.search(r"(k_[a-z_]+)I(.*?)E+v", sym)
    if not m:
        m2 = re.search(r"(k_[a-z_]+)", sym)
        return (m2.group(1) if m2 else sym), ""
    name, args = m.group(1), m.group(2)
    out = []
    for a in re.findall(r"Lb([01])E|Li(\d+)E|([df])", args):
        if a[0] != "":
            out.append("1" if a[0] == "1" else "0")
        elif a[1] != "":
            out.append(a[1])
        elif a[2]:
            out.append("f64" if a[2] == "d" else "f32 buffers")
    return name, ", ".join(out)


def main():
    if len(sys.argv) > 1:
        text = Path(sys.argv[1]).read_text()
    else:
        text = subprocess.run(["make", "-C", str(ROOT / "rotors_mpc_controller_amd" / "csrc"), "resource-usage"], capture_output=True, text=True).stderr
    rows, cur = [], None
    for ln in text.splitlines():
        m = re.match(r"(\S+?\.hip):\d+:\d+: remark: Function Name: (\S+)", ln)
        if m:
            cur = dict(tu=m.group(1), sym=m.group(2))
            rows.append(cur)
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", ln)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    print("| kernel<template arguments> | translation unit (build) | VGPR + AGPR | waves / SIMD | scratch B / lane | SGPR / VGPR spills |")
    print("|---|---|---|---|---|---|")
    seen = set()
    for r in rows:
        name, args = demangle_args(r["sym"])
        if not name.startswith("k_"):
            continue
        build = {"nmpc_as.hip": "nmpc_as.hip (flag)", "nmpc_qpf.hip": "nmpc_qpf.hip (flag)", "nmpc_blockf.hip": "nmpc_blockf.hip (flag)"}.get(r["tu"], r["tu"] + " (default)")
        key = (name, args, build)
        if key in seen:
            continue
        seen.add(key)
        print(f"| `{name}<{args}>` | {build} | {r.get('VGPRs', '?')} + {r.get('AGPRs', '?')} | {r.get('Occupancy', '?')} | {r.get('ScratchSize', '?')} | "
              f"{r.get('SGPRs Spill', '?')} / {r.get('VGPRs Spill', '?')} |")


if __name__ == "__main__":
    main()

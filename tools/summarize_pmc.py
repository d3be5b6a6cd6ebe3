#!/usr/bin/env python3
"""Condenses the rocprofv3 --pmc CSVs written by tools/rocprof_capture.sh into profiles/<name>.json.
usage: python tools/summarize_pmc.py gpurun_out/rocprof_<tag> profiles/<name>_pmc_summary.json"""
import collections
import csv
import glob
import json
import sys

base, out_path = sys.argv[1], sys.argv[2]
out = {}
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_sq4", "pmc_sq5"):
    files = glob.glob(f"{base}/{name}/*/*_counter_collection.csv")
    if not files:
        continue
    agg = collections.defaultdict(list)
    meta = {}
    for r in csv.DictReader(open(files[0])):
        kn = r["Kernel_Name"]
        k = next((t for t in ("k_team_as", "k_team_qp_list", "k_team_qp", "k_team_tail", "k_block_sweep_tail", "k_block_scan_tail", "k_block_sweep",
                              "k_block_scan", "k_cond_ipm", "k_ipm", "k_prepare") if t + "<" in kn or t + "(" in kn), None)
        if k:
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
            meta[k] = dict(vgpr=int(r["VGPR_Count"]), agpr=int(r["Accum_VGPR_Count"]), sgpr=int(r["SGPR_Count"]),
                           lds_bytes=int(r["LDS_Block_Size"]), scratch=int(r["Scratch_Size"]),
                           grid=int(r["Grid_Size"]), workgroup=int(r["Workgroup_Size"]))
    for (k, c), v in sorted(agg.items()):
        out.setdefault(k, {})[c] = dict(dispatches=len(v), mean=sum(v) / len(v))
    for k, m in meta.items():
        out.setdefault(k, {})["launch"] = m
import pathlib
import sys as _sys
_sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
from source_hash import source_hash as _sh
out["source_hash"] = _sh()      # bench.py quotes this traffic only on a build with the same hash
json.dump(out, open(out_path, "w"), indent=1)
for k, v in out.items():
    if not isinstance(v, dict):
        continue
    print(k, {c: round(x["mean"], 1) for c, x in v.items() if c != "launch"}, v.get("launch"))

#!/usr/bin/env python3
"""run_team_kernel.py -- emulate ONE workgroup of a kernel of nmpc_team_as.hpp (k_team_qp / k_team_as) from the compiler's assembly.

DEV / TEST infrastructure.  Buffers have exactly the sizes nmpc_create / nmpc_solve_batch allocate (csrc/nmpc_capi.hip: alloc_ws, the
staging buffers) - every global access of the instruction stream is checked against them, every LDS access against the dynamic LDS
size the launch asks for (qp_lds / launch_split) - and the command of the workgroup's instance is compared with the oracle.

usage: run_team_kernel.py <file.s> <kernel-name-substring> [--steps 4] [--polish 0] [--share 0] [--batch 256] [--wg 0] [--seed 8]
                          [--headers <csrc dir>] [--include <include dir>] [--trace out.npz]
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import subprocess
import struct
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(Path(__file__).resolve().parent))
import gfx950_emu as E  # noqa: E402

SYMS = ["_ZN4nmpcL8NMPC_TVQE", "_ZN4nmpcL8NMPC_TQWE", "_ZN4nmpcL8NMPC_TQQE", "_ZN4nmpcL8NMPC_TWWE"]


_EXE = {}


def layout_and_consts(cfg_bytes: bytes, csrc: Path, inc: Path):
    """struct layouts + Consts<double> from the kernel headers of the tree the assembly was built from (tools/emu/mkargs.cpp)."""
    tmp = Path(tempfile.mkdtemp(prefix="nmpc_emu_"))
    key = (str(csrc), str(inc))
    if key not in _EXE:
        exe = tmp / "mkargs"
        subprocess.check_call(["g++", "-std=c++17", "-O1", f"-I{inc}", f"-I{csrc}", "-o", str(exe), str(Path(__file__).resolve().parent / "mkargs.cpp")])
        _EXE[key] = exe
    exe = _EXE[key]
    (tmp / "cfg.bin").write_bytes(cfg_bytes)
    out = subprocess.run([str(exe), str(tmp / "cfg.bin"), str(tmp / "consts.bin")], capture_output=True, text=True, check=True)
    return json.loads(out.stdout), (tmp / "consts.bin").read_bytes()


def hidden_offsets(sfile: str, kernel: str) -> dict:
    """kernarg offsets of the hidden arguments of a kernel, from the metadata the compiler wrote into the assembly."""
    lines = open(sfile).read().split("\n")
    for i, l in enumerate(lines):
        if ".name:" in l and kernel in l:
            j = i
            while not lines[j].startswith("  - .agpr_count"):
                j -= 1
            out = {}
            for k in range(j, i):
                if "value_kind" in lines[k] and "hidden_" in lines[k]:
                    out[lines[k].split()[-1]] = int(lines[k - 2].split()[-1])
            return out
    raise KeyError(kernel)


def private_segment_size(sfile: str, kernel: str) -> int:
    lines = open(sfile).read().split("\n")
    for i, l in enumerate(lines):
        if ".name:" in l and kernel in l:
            for k in range(i, min(i + 12, len(lines))):
                if ".private_segment_fixed_size:" in lines[k]:
                    return int(lines[k].split()[-1])
    raise KeyError(kernel)


def emulate(sfile: str, kernel: str, steps=4, polish=0, share=0, B=256, wg=0, seed=8, dist="aggressive", csrc=None, inc=None, warm=False,
            trace=False, max_steps=40_000_000, kind="qp", verbose=True, lds_overlap=True, cfg_over=None, then=None,
            inplace=True):
    """cfg_over: nmpc_config fields to override (e.g. qp_polish_passes); then = (file.s, kernel): after a k_team_as workgroup, run workgroup 0 of
    that k_team_qp_list build on the SAME memory - the second launch of the default path, on the work list the first workgroup left
    (with inplace=False: the default, inplace=True, makes k_team_as continue its own failed attempts and leaves the list empty)."""
    from rotors_mpc_controller_amd import _lib
    from rotors_mpc_controller_amd.synthetic import AGGRESSIVE, NEAR_HOVER, hover_reference, sample_x0
    csrc = Path(csrc or ROOT / "rotors_mpc_controller_amd" / "csrc")
    inc = Path(inc or ROOT / "include")
    N = 20
    cfg = _lib.default_config(N=N, max_batch=B, sim_num_steps=steps, qp_polish=polish, flags=_lib.FLAG_TEAM_MAPPING | (1 if share else 0))
    if cfg_over:
        cfg.update(**cfg_over)
    lay, consts = layout_and_consts(bytes(cfg)[:10 ** 6], csrc, inc)
    cfg_bytes = bytes(cfg)[:lay["sizeof.config"]]                  # (an older tree's nmpc_config is a prefix of the current one)
    lay, consts = layout_and_consts(cfg_bytes, csrc, inc)
    shared = bool(share) and not warm
    if shared:                                                      # nmpc launch sets c.shared from the call (cold start + flag)
        consts = bytearray(consts); struct.pack_into("<i", consts, lay["C.shared"], 1); consts = bytes(consts)
    NX, NU, NY = 13, 4, 17
    TLM, IV, TAB, TP = lay["TLM_ROWS"], lay["IV_ROWS"], lay["TAB_ROWS"], lay["TP_ROWS"]
    Bp = (B + 63) // 64 * 64
    Bw = Bp + 1
    ck = min(max(cfg.qp_polish_ckpt, 0), N - 1)
    yref, ye = hover_reference(N, cfg.mass * cfg.gravity / 4.0)
    x0 = sample_x0(B, seed, **(AGGRESSIVE if dist == "aggressive" else NEAR_HOVER))
    mem = E.Memory()
    f8 = np.float64
    addr = {}
    def buf(name, nbytes, init=None, writable=True):
        data = np.zeros(nbytes, dtype=np.uint8) if init is None else np.frombuffer(np.ascontiguousarray(init).tobytes(), dtype=np.uint8)
        assert len(data) == nbytes, (name, len(data), nbytes)
        addr[name] = mem.add(name, data, writable)
    # alloc_ws (nmpc_capi.hip), wsz = 8
    AB_ROWS, QR_ROWS = 131, 17
    buf("AB", N * AB_ROWS * Bp * 8); buf("bv", N * NX * Bp * 8); buf("qr", (N * QR_ROWS + NX) * Bp * 8)
    buf("xl", (N + 1) * NX * Bp * 8); buf("ul", N * NU * Bp * 8)
    buf("LM", N * TLM * Bw * 8); buf("iv", N * IV * Bw * 8); buf("tAB", N * TAB * Bw * 8)
    buf("tP", ((ck + 1) * TP * Bw if polish else 1) * 8)
    buf("d_iters", Bp * 4); buf("d_status", Bp * 4); buf("d_npol", Bp * 4); buf("d_wl", (Bp + 2) * 4); buf("d_gbase", (Bp + 1) * 8)
    buf("consts", len(consts), np.frombuffer(consts, dtype=np.uint8), writable=False)
    # staging buffers of nmpc_solve_batch (host entry point), sized by the batch
    buf("x0", B * NX * 8, x0.astype(f8), False); buf("yref", N * NY * 8, yref.astype(f8), False); buf("yref_e", NX * 8, ye.astype(f8), False)
    buf("u0", B * NU * 8); buf("x_out", B * (N + 1) * NX * 8); buf("u_out", B * N * NU * 8); buf("status", B * 4)
    xi = ui = None
    if warm:
        from oracle import oracle as O
        r0 = O.solve_batch(O.default_config(N=N, qp_gamma=0.0, qp_polish=polish, sim_num_steps=steps), x0, yref, ye, want_traj=True, nthreads=8)
        xi, ui = r0["x"], r0["u"]
        buf("x_init", B * (N + 1) * NX * 8, xi.astype(f8), False); buf("u_init", B * N * NU * 8, ui.astype(f8), False)
    ro = E.parse_rodata(sfile, SYMS)
    symbols = {}
    for name, data in ro.items():
        symbols[name] = mem.add(name, np.frombuffer(data, dtype=np.uint8), writable=False)
    # kernel arguments (layout checked against the kernel's metadata: cp 0 | Work 8 | Inputs 120 | Outputs 168 | TeamWork 200 | WorkList 224 | ints 248..)
    ka = bytearray(320)
    def put64(off, v): struct.pack_into("<Q", ka, off, v)
    def put32(off, v): struct.pack_into("<i", ka, off, v)
    put64(0, addr["consts"])
    Wb = 8
    put32(Wb + lay["W.Bp"], Bp)
    for k, nm in (("AB", "AB"), ("bv", "bv"), ("qr", "qr"), ("xl", "xl"), ("ul", "ul"), ("LM", "LM"), ("iv", "iv"), ("iters", "d_iters"),
                  ("status", "d_status"), ("npol", "d_npol"), ("tAB", "tAB"), ("gbase", "d_gbase")):
        put64(Wb + lay[f"W.{k}"], addr[nm])
    put64(Wb + lay["W.prof"], 0)
    Ib = 8 + lay["sizeof.Work"]
    put64(Ib + lay["I.x0"], addr["x0"]); put64(Ib + lay["I.yref"], addr["yref"]); put64(Ib + lay["I.yref_e"], addr["yref_e"])
    put64(Ib + lay["I.x_init"], addr["x_init"] if warm else 0); put64(Ib + lay["I.u_init"], addr["u_init"] if warm else 0)
    put32(Ib + lay["I.yref_bcast"], 1)
    Ob = Ib + lay["sizeof.Inputs"]
    put64(Ob + lay["O.u0"], addr["u0"]); put64(Ob + lay["O.x_out"], addr["x_out"]); put64(Ob + lay["O.u_out"], addr["u_out"])
    put64(Ob + lay["O.status"], addr["status"])
    Tb = Ob + lay["sizeof.Outputs"]
    put64(Tb + lay["T.tLM"], addr["LM"]); put64(Tb + lay["T.tIV"], addr["iv"]); put64(Tb + lay["T.tP"], addr["tP"] if (polish and ck > 0) else 0)
    Lb = Tb + lay["sizeof.TeamWork"]
    put64(Lb, addr["d_wl"]); put64(Lb + 8, addr["d_wl"] + 4); put64(Lb + 16, addr["d_wl"] + 8)
    Sb = Lb + 24
    tpw = 4 if B >= 2048 else (2 if B >= 512 else 1)
    # LDS carve (nmpc_capi.hip: as_lds_base / qp_lds / launch_split)
    A_EV, AS_EV, AS_CH = 312, 56, 8
    carve = (A_EV + AS_EV + (AS_EV if steps > 2 else 0)) if shared else (A_EV + AS_CH * AS_EV)
    base = carve if (shared or not lds_overlap) else A_EV          # as_cache_base: the per-stage cache starts at the (dead) evaluation points
    lm_rows = 88 if kind == "qp" else 80
    per_team = 40960 // 4 // 8
    lstg = max(0, min(N, (per_team - base - 31) // lm_rows))
    stride = max(carve, base + lstg * lm_rows)
    stride += (24 - stride % 32 + 32) % 32
    lds_bytes = 4 * stride * 8
    for i, v in enumerate((B, tpw, stride, lstg, base)):
        put32(Sb + 4 * i, v)
    # the MODE 2 carve (qp_lds: IP_LM_ROWS doubles per cached stage): the work-list launch, or the continuation inside k_team_as
    lstg2 = max(0, min(N, (per_team - base - 31) // 88))
    stride2 = max(carve, base + lstg2 * 88)
    stride2 += (24 - stride2 % 32 + 32) % 32
    if kind == "as":                                               # k_team_as: ..., int pass_cap, double *tail_ts, int cont_stride, int cont_lstg
        put32(Sb + 20, 0); put64(Sb + 24, 0)
        put32(Sb + 32, stride2 if inplace else 0); put32(Sb + 36, lstg2 if inplace else 0)
        lds_bytes = max(lds_bytes, 4 * stride2 * 8) if inplace else lds_bytes
    kaddr = mem.add("kernarg", np.frombuffer(bytes(ka), dtype=np.uint8), writable=False)
    insts, labels = E.parse_kernel(sfile, kernel)
    w = E.Wave(insts, labels, mem, lds_bytes, kaddr, wg, symbols, max_steps=max_steps)
    w.scratch_bytes = private_segment_size(sfile, kernel)
    if trace:
        w.trace_mem = []
    t = time.time()
    err = None
    try:
        w.run()
    except E.EmuError as e:
        err = str(e)
    dt = time.time() - t
    inst = wg * tpw
    second = None
    if then is not None and err is None:
        # the work-list launch (launch_split: grid min((B + 3) / 4, 64), LDS carve of qp_lds with IP_LM_ROWS rows per cached stage)
        sfile2, kernel2 = then
        ka2 = bytearray(512)
        ka2[:248] = ka[:248]
        for i, v in enumerate((B, stride2, lstg2, base)):
            struct.pack_into("<i", ka2, 248 + 4 * i, v)
        hid = hidden_offsets(sfile2, kernel2)
        struct.pack_into("<III", ka2, hid["hidden_block_count_x"], min((B + 3) // 4, 64), 1, 1)
        struct.pack_into("<HHH", ka2, hid["hidden_group_size_x"], 64, 1, 1)
        if "hidden_dynamic_lds_size" in hid:
            struct.pack_into("<I", ka2, hid["hidden_dynamic_lds_size"], 4 * stride2 * 8)
        kaddr2 = mem.add("kernarg2", np.frombuffer(bytes(ka2), dtype=np.uint8), writable=False)
        sym2 = {}
        for name, data in E.parse_rodata(sfile2, SYMS).items():
            sym2[name] = mem.add("second:" + name, np.frombuffer(data, dtype=np.uint8), writable=False)
        insts2, labels2 = E.parse_kernel(sfile2, kernel2)
        w2 = E.Wave(insts2, labels2, mem, 4 * stride2 * 8, kaddr2, 0, sym2, max_steps=max_steps)
        w2.scratch_bytes = private_segment_size(sfile2, kernel2)
        err2 = None
        n_listed = int(mem.view("d_wl", np.int32)[0])
        try:
            w2.run()
        except E.EmuError as e:
            err2 = str(e)
        second = dict(instructions=w2.steps, error=err2, violations=w2.viol, listed=n_listed, lds_bytes=4 * stride2 * 8, lstg=lstg2)
    u0 = mem.view("u0", f8).reshape(B, NU)[inst:inst + tpw].copy()
    st = mem.view("status", np.int32)[inst:inst + tpw].copy()
    its = mem.view("d_iters", np.int32)[inst:inst + tpw].copy()
    wlv = mem.view("d_wl", np.int32)
    listed = sorted(int(x) for x in wlv[2:2 + int(wlv[0])])
    if second is not None:
        listed = sorted(int(x) for x in wlv[2:2 + second["listed"]])
    res = dict(listed=listed, file=sfile, kernel=kernel, instructions=w.steps, seconds=dt, error=err, u0=u0, status=st, iters=its, violations=w.viol,
               lds_bytes=lds_bytes, lstg=lstg, trace=w.trace_mem, second=second, addr=addr, mem=mem, x0=x0, yref=yref, ye=ye, x_init=xi, u_init=ui, inst=inst, tpw=tpw)
    if verbose:
        print(f"{Path(sfile).name} [{kernel}] wg {wg}: {w.steps} instructions in {dt:.1f} s, error {err}, violations {len(w.viol)}; status {st} iters {its}")
        for v in w.viol[:12]:
            print(f"   {v.kind} line {v.line}: {v.text}   lane {v.lane} addr {v.addr:#x} ({v.note})")
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("sfile"); ap.add_argument("kernel")
    ap.add_argument("--steps", type=int, default=4); ap.add_argument("--polish", type=int, default=0); ap.add_argument("--share", type=int, default=0)
    ap.add_argument("--batch", type=int, default=256); ap.add_argument("--wg", type=int, default=0); ap.add_argument("--seed", type=int, default=8)
    ap.add_argument("--dist", default="aggressive"); ap.add_argument("--headers", default=None); ap.add_argument("--include", default=None)
    ap.add_argument("--warm", action="store_true"); ap.add_argument("--kind", default="qp")
    a = ap.parse_args()
    r = emulate(a.sfile, a.kernel, a.steps, a.polish, a.share, a.batch, a.wg, a.seed, a.dist, a.headers, a.include, a.warm, kind=a.kind)
    from oracle import oracle as O
    c = O.default_config(N=20, qp_gamma=0.0, qp_polish=a.polish, sim_num_steps=a.steps)
    i0 = r["inst"]
    sl = slice(i0, i0 + r["tpw"])
    ref = O.solve_batch(c, r["x0"][sl], r["yref"], r["ye"], x_init=None if r["x_init"] is None else r["x_init"][sl],
                        u_init=None if r["u_init"] is None else r["u_init"][sl])
    print("handed to the work list:", r["listed"])
    print("u0 emulated", r["u0"], "\nu0 oracle  ", ref["u0"], "\n|du0| max", np.abs(r["u0"] - ref["u0"]).max(), "iters oracle", ref["iters"], "status oracle", ref["status"])
    return 1 if (r["violations"] or r["error"]) else 0


if __name__ == "__main__":
    sys.exit(main())

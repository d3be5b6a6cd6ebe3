#!/usr/bin/env python3
"""gfx950_emu.py -- a functional emulator for ONE wave of a gfx950 (CDNA4) kernel, read from the compiler's own assembly (.s).

DEV / TEST infrastructure, not the product.  Why it exists: the one GPU memory fault this repository
has recorded came from a build of k_team_qp<per-stage, trajectories> made with the internal LLVM option -amdgpu-mfma-vgpr-form; the
same sources built with the default code generation run clean.  A fault cannot be "tried again" on the GPU pool, so the question
"does that build form a wrong address, and in which instruction?" is answered here, on the CPU: the emulator executes the instruction
stream of either build for one workgroup of the faulting configuration against buffers of exactly the sizes nmpc_create allocates and
checks EVERY global and LDS access against them.  It also gives a GPU-free check of any shipped build (tests/test_isa_emulation.py).

Scope: the ~170 opcodes these kernels use (FP64 VALU, v_mfma_f64_4x4x4_4b_f64, integer VALU, cross-lane, LDS, global, SALU, branches,
EXEC handling).  One wave, one workgroup, no timing: hazards that depend on pipeline timing are outside a functional model (tools/emu/
isa_checks.py scans for those statically).  Unknown opcodes raise - nothing is silently skipped.

Numerics: FP64 through numpy; v_fma / MFMA accumulate through 80-bit long double (one extra rounding in ~2^-11 of the cases: results
agree with the GPU to rounding, which is all an address / control-flow check needs).  v_rcp_f64 is 1/x correctly rounded.
"""
from __future__ import annotations

import re
import struct
import sys
from dataclasses import dataclass, field

import numpy as np

LANES = np.arange(64, dtype=np.uint64)
M32 = 0xFFFFFFFF
M64 = 0xFFFFFFFFFFFFFFFF
np.seterr(all="ignore")


class EmuError(Exception):
    pass


# --------------------------------------------------------------------------------------------------------------------- memory
@dataclass
class Segment:
    name: str
    base: int
    data: np.ndarray          # uint8
    writable: bool = True


@dataclass
class Violation:
    kind: str                 # "global-read" | "global-write" | "lds-read" | "lds-write" | "readonly-write"
    line: int
    text: str
    lane: int
    addr: int
    nbytes: int
    note: str = ""


class Memory:
    """Device address space: named segments at far-apart fake addresses; any access outside all of them is recorded."""

    def __init__(self):
        self.segs: list[Segment] = []
        self.next_base = 0x7F00_0000_0000

    def add(self, name, data, writable=True) -> int:
        if isinstance(data, np.ndarray) and data.dtype == np.uint8 and data.flags.writeable and data.flags.c_contiguous:
            arr = data                                        # (large zero-filled workspaces: no copy)
        else:
            arr = np.frombuffer(bytearray(data.tobytes() if isinstance(data, np.ndarray) else bytes(data)), dtype=np.uint8)
        base = self.next_base
        self.next_base += (len(arr) + (1 << 24)) & ~((1 << 21) - 1)    # neighbours are at least 14 MB apart: an overrun cannot land in one
        self.segs.append(Segment(name, base, arr, writable))
        return base

    def find(self, addr, n):
        for s in self.segs:
            if s.base <= addr and addr + n <= s.base + len(s.data):
                return s
        return None

    def nearest(self, addr):
        best = min(self.segs, key=lambda s: min(abs(addr - s.base), abs(addr - (s.base + len(s.data)))))
        d = addr - best.base
        return f"{best.name}{'+' if d >= 0 else ''}{d} (size {len(best.data)})"

    def view(self, name, dtype):
        for s in self.segs:
            if s.name == name:
                return s.data.view(dtype)
        raise KeyError(name)


# --------------------------------------------------------------------------------------------------------------------- parsing
_REG = re.compile(r"^(-?)(\|?)([vsa])(?:(\d+)|\[(\d+):(\d+)\])(\|?)$")


@dataclass
class Op:
    kind: str                 # v | a | s | vcc | exec | lit | flit | off | label | sym | scc | m0
    n: int = 0
    cnt: int = 1
    val: object = None
    neg: bool = False
    abs_: bool = False


@dataclass
class Inst:
    line: int
    text: str
    op: str
    ops: list
    mods: dict = field(default_factory=dict)


def parse_operand(tok: str) -> Op:
    t = tok.strip()
    m = _REG.match(t)
    if m:
        neg, a1, k, single, lo, hi, a2 = m.groups()
        n = int(single) if single is not None else int(lo)
        cnt = 1 if single is not None else int(hi) - int(lo) + 1
        return Op(k, n, cnt, neg=bool(neg), abs_=bool(a1 and a2))
    if t in ("vcc", "exec", "off", "scc", "m0"):
        return Op(t)
    if t in ("vcc_lo", "vcc_hi", "exec_lo", "exec_hi"):
        return Op(t)
    if "@rel32" in t:
        name, which = t.split("@rel32@")
        return Op("sym", val=(name, which))
    if re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", t):
        return Op("lit", val=int(t, 0))
    if re.fullmatch(r"-?\d+\.\d*(e[-+]?\d+)?", t):
        return Op("flit", val=float(t))
    return Op("label", val=t)


def parse_kernel(path: str, kernel_substr: str | None = None):
    """Instructions of one kernel (the file may hold one kernel - tools/emu/split_kernels.py - or many: first whose label contains kernel_substr)."""
    lines = open(path).read().split("\n")
    start = 0
    if kernel_substr:
        for i, l in enumerate(lines):
            if l.startswith("_Z") and l.split(":")[0].find(kernel_substr) >= 0 and re.match(r"^_Z\w+:", l):
                start = i
                break
        else:
            raise EmuError(f"kernel {kernel_substr} not found in {path}")
    insts, labels = [], {}
    for ln in range(start, len(lines)):
        raw = lines[ln]
        s = raw.split(";")[0].strip()
        if not s:
            continue
        if s.startswith(".end_amdhsa_kernel") or s.startswith(".Lfunc_end") or s.startswith(".section"):
            break
        if s.endswith(":"):
            labels[s[:-1]] = len(insts)
            continue
        if s.startswith("."):
            continue
        ops, mods = [], {}
        mneg = re.search(r"\s+neg:\[(\d),(\d),(\d)\]", s)        # MFMA f64: negate A | B | C
        if mneg:
            mods["neg"] = tuple(int(x) for x in mneg.groups())
            s = s[:mneg.start()] + s[mneg.end():]
        parts = s.split(None, 1)
        op = parts[0]
        if len(parts) > 1:
            for tok in parts[1].split(","):
                tok = tok.strip()
                sub = tok.split()
                if not sub:
                    continue
                ops.append(parse_operand(sub[0]))
                for mod in sub[1:]:
                    if ":" in mod:
                        k, v = mod.split(":", 1)
                        mods[k] = int(v, 0) if re.fullmatch(r"-?(0x[0-9a-fA-F]+|\d+)", v) else v
                    else:
                        if mod == "off":
                            ops.append(Op("off"))
                        else:
                            mods[mod] = True
        insts.append(Inst(ln + 1, s, op, ops, mods))
    return insts, labels


def parse_rodata(path: str, names):
    """Byte contents of the named constant tables (.byte / .short / .long / .quad / .zero directives after their label)."""
    out = {}
    lines = open(path).read().split("\n")
    for name in names:
        for i, l in enumerate(lines):
            if l.startswith(name + ":"):
                buf = bytearray()
                for l2 in lines[i + 1:]:
                    s = l2.split(";")[0].strip()
                    if not s:
                        continue
                    if s.startswith(".byte"):
                        buf += bytes(int(x, 0) & 0xFF for x in s[5:].split(","))
                    elif s.startswith(".short"):
                        for x in s[6:].split(","):
                            buf += struct.pack("<H", int(x, 0) & 0xFFFF)
                    elif s.startswith(".long"):
                        for x in s[5:].split(","):
                            buf += struct.pack("<I", int(x, 0) & M32)
                    elif s.startswith(".quad"):
                        for x in s[5:].split(","):
                            buf += struct.pack("<Q", int(x, 0) & M64)
                    elif s.startswith(".zero"):
                        buf += bytes(int(s[5:].split(",")[0]))
                    elif s.startswith(".ascii") or s.startswith(".asciz"):
                        lit = l2.split('"', 1)[1].rsplit('"', 1)[0]          # C-style escapes: octal \ooo, \n, \t, \", \\
                        i2 = 0
                        while i2 < len(lit):
                            ch = lit[i2]
                            if ch == "\\":
                                nx = lit[i2 + 1]
                                if nx in "01234567":
                                    j2 = i2 + 1
                                    while j2 < len(lit) and j2 < i2 + 4 and lit[j2] in "01234567":
                                        j2 += 1
                                    buf.append(int(lit[i2 + 1:j2], 8) & 0xFF); i2 = j2
                                else:
                                    buf.append({"n": 10, "t": 9, "r": 13, '"': 34, "\\": 92, "b": 8, "f": 12}[nx]); i2 += 2
                            else:
                                buf.append(ord(ch)); i2 += 1
                        if s.startswith(".asciz"):
                            buf.append(0)
                    else:
                        break
                out[name] = bytes(buf)
                break
        else:
            raise EmuError(f"symbol {name} not found")
    return out


# --------------------------------------------------------------------------------------------------------------------- the wave
def _f64(u64):
    return u64.view(np.float64)


def _u64(f):
    return np.ascontiguousarray(f, dtype=np.float64).view(np.uint64)


def fma64(a, b, c):
    ld = np.longdouble
    return (a.astype(ld) * b.astype(ld) + c.astype(ld)).astype(np.float64)


class Wave:
    def __init__(self, insts, labels, mem: Memory, lds_bytes: int, kernarg_addr: int, wg_id: int, symbols: dict | None = None,
                 max_steps: int = 50_000_000, stop_on_violation: bool = False, poison: bool = True):
        self.insts, self.labels, self.mem = insts, labels, mem
        self.R = np.zeros((512, 64), dtype=np.uint32)     # 0..255 VGPR, 256..511 AGPR
        self.S = np.zeros(128, dtype=np.uint64)           # SGPRs as python-int friendly uint64 (only low 32 bits used); 106/107 = vcc
        self.exec_ = M64
        self.scc = 0
        self.lds = np.zeros(65536 + 64, dtype=np.uint8)
        self.lds_bytes = lds_bytes
        self.pc = 0
        self.symbols = symbols or {}
        self.viol: list[Violation] = []
        self.steps = 0
        self.max_steps = max_steps
        self.stop_on_violation = stop_on_violation
        self.scratch_bytes = 0                             # the kernel's private_segment_fixed_size (set by the harness when the kernel spills)
        self.scratch = np.full((64, 4096), 0xFD, dtype=np.uint8)
        self.scratch_set = np.zeros((64, 4096), dtype=bool)
        self.trace_mem = None                              # optional list of (line, kind, lane, addr, nbytes) for address-stream comparison
        if poison:
            # registers and LDS start with garbage on the hardware: a poison pattern (a NaN as a double, a wild pointer as an address)
            # makes any dependence on an unwritten register or LDS word show - as a violation, or as a wrong / NaN result
            self.R[:] = 0x7FF4DEAD
            self.S[:] = 0x7FF4DEAD
            self.lds[:] = 0xFE
        self.S[0] = kernarg_addr & M32
        self.S[1] = kernarg_addr >> 32
        self.S[2] = wg_id
        self.R[0] = np.arange(64, dtype=np.uint32)
        self._mask_cache = (None, None)
        self.opcount = {}

    # ---- exec / masks
    def mask(self):
        if self._mask_cache[0] != self.exec_:
            self._mask_cache = (self.exec_, ((np.uint64(self.exec_) >> LANES) & np.uint64(1)).astype(bool))
        return self._mask_cache[1]

    @staticmethod
    def bits_to_mask(v):
        return ((np.uint64(v) >> LANES) & np.uint64(1)).astype(bool)

    @staticmethod
    def mask_to_bits(m):
        return int(np.bitwise_or.reduce(np.where(m, np.uint64(1) << LANES, np.uint64(0))))

    # ---- scalar register file
    def s32(self, n):
        return int(self.S[n]) & M32

    def s64(self, n):
        return (int(self.S[n]) & M32) | ((int(self.S[n + 1]) & M32) << 32)

    def set_s32(self, n, v):
        self.S[n] = v & M32

    def set_s64(self, n, v):
        self.S[n] = v & M32
        self.S[n + 1] = (v >> 32) & M32

    def rs(self, o: Op, bits=32):
        """scalar source"""
        k = o.kind
        if k == "s":
            return self.s64(o.n) if (bits == 64 or o.cnt == 2) and o.cnt >= 2 else self.s32(o.n)
        if k == "vcc":
            return self.s64(106)
        if k == "vcc_lo":
            return self.s32(106)
        if k == "vcc_hi":
            return self.s32(107)
        if k == "exec":
            return self.exec_
        if k == "exec_lo":
            return self.exec_ & M32
        if k == "exec_hi":
            return self.exec_ >> 32
        if k == "lit":
            return o.val & (M64 if bits == 64 else M32)
        if k == "scc":
            return self.scc
        raise EmuError(f"scalar source {o}")

    def ws(self, o: Op, v, bits=32):
        k = o.kind
        if k == "s":
            if bits == 64:
                self.set_s64(o.n, v)
            else:
                self.set_s32(o.n, v)
        elif k == "vcc":
            self.set_s64(106, v)
        elif k == "vcc_lo":
            self.set_s32(106, v)
        elif k == "vcc_hi":
            self.set_s32(107, v)
        elif k == "exec":
            self.exec_ = v & M64
        else:
            raise EmuError(f"scalar dest {o}")

    # ---- vector operands
    def _row(self, o: Op, i=0):
        return (256 if o.kind == "a" else 0) + o.n + i

    def rv32(self, o: Op):
        k = o.kind
        if k in ("v", "a"):
            return self.R[self._row(o)]
        if k == "lit":
            return np.full(64, o.val & M32, dtype=np.uint32)
        if k == "flit":
            return np.full(64, struct.unpack("<I", struct.pack("<f", o.val))[0], dtype=np.uint32)
        return np.full(64, self.rs(o, 32) & M32, dtype=np.uint32)

    def rv64(self, o: Op):
        k = o.kind
        if k in ("v", "a"):
            r = self._row(o)
            return self.R[r].astype(np.uint64) | (self.R[r + 1].astype(np.uint64) << np.uint64(32))
        if k == "lit":
            return np.full(64, np.int64(o.val).astype(np.uint64) if o.val < 0 else np.uint64(o.val & M64), dtype=np.uint64)
        return np.full(64, self.rs(o, 64) & M64, dtype=np.uint64)

    def rf64(self, o: Op):
        k = o.kind
        if k == "flit":
            v = np.full(64, o.val, dtype=np.float64)
        elif k == "lit":
            # integer inline constants are integer bit patterns; a 32-bit literal is the HIGH dword of the double
            if -16 <= o.val <= 64:
                v = _f64(np.full(64, np.int64(o.val).astype(np.uint64), dtype=np.uint64))
            else:
                v = _f64(np.full(64, (o.val & M32) << 32, dtype=np.uint64))
        else:
            v = _f64(self.rv64(o))
        if o.abs_:
            v = np.abs(v)
        if o.neg:
            v = -v
        return v

    def wv32(self, o: Op, val, masked=True):
        r = self._row(o)
        val = np.asarray(val).astype(np.uint32)
        if masked:
            m = self.mask()
            self.R[r][m] = val[m] if val.shape == (64,) else val
        else:
            self.R[r] = val

    def wv64(self, o: Op, val, masked=True):
        val = np.asarray(val, dtype=np.uint64) if not isinstance(val, np.ndarray) or val.dtype != np.uint64 else val
        lo = (val & np.uint64(M32)).astype(np.uint32)
        hi = (val >> np.uint64(32)).astype(np.uint32)
        r = self._row(o)
        if masked:
            m = self.mask()
            self.R[r][m] = lo[m]
            self.R[r + 1][m] = hi[m]
        else:
            self.R[r] = lo
            self.R[r + 1] = hi

    def wf64(self, o: Op, f, masked=True):
        self.wv64(o, _u64(f), masked)

    # ---- memory access with checking
    def _viol(self, kind, lane, addr, n, note=""):
        ins = self.insts[self.pc]
        self.viol.append(Violation(kind, ins.line, ins.text, int(lane), int(addr), n, note))
        if self.stop_on_violation:
            raise EmuError(f"{kind} line {ins.line}: {ins.text} lane {lane} addr {addr:#x} {note}")

    def gload(self, addrs, n, lanes):
        out = np.zeros((64, n), dtype=np.uint8)
        for l in lanes:
            a = int(addrs[l])
            seg = self.mem.find(a, n)
            if self.trace_mem is not None:
                self.trace_mem.append((self.insts[self.pc].line, "R", int(l), a, n))
            if seg is None:
                self._viol("global-read", l, a, n, self.mem.nearest(a))
                continue
            off = a - seg.base
            out[l] = seg.data[off:off + n]
        return out

    def gstore(self, addrs, data, n, lanes):
        for l in lanes:
            a = int(addrs[l])
            seg = self.mem.find(a, n)
            if self.trace_mem is not None:
                self.trace_mem.append((self.insts[self.pc].line, "W", int(l), a, n))
            if seg is None:
                self._viol("global-write", l, a, n, self.mem.nearest(a))
                continue
            if not seg.writable:
                self._viol("readonly-write", l, a, n, seg.name)
                continue
            off = a - seg.base
            seg.data[off:off + n] = data[l]

    def sload(self, addr, n):
        seg = self.mem.find(addr, n)
        if seg is None:
            self._viol("global-read", -1, addr, n, "scalar load; " + self.mem.nearest(addr))
            return bytes(n)
        off = addr - seg.base
        return seg.data[off:off + n].tobytes()

    def _gaddr(self, ins: Inst, vaddr: Op, saddr: Op):
        off = ins.mods.get("offset", 0)
        if saddr.kind == "off":
            a = self.rv64(vaddr)
        else:
            a = np.uint64(self.rs(saddr, 64)) + self.rv32(vaddr).astype(np.uint64)
        return (a.astype(np.int64) + np.int64(off)).astype(np.uint64)

    def lds_rd(self, addr, n, lanes):
        out = np.zeros((64, n), dtype=np.uint8)
        for l in lanes:
            a = int(addr[l])
            if a + n > self.lds_bytes or a < 0:
                self._viol("lds-read", l, a, n, f"LDS allocation {self.lds_bytes}")
                continue
            out[l] = self.lds[a:a + n]
        return out

    def lds_wr(self, addr, data, n, lanes):
        for l in lanes:
            a = int(addr[l])
            if a + n > self.lds_bytes or a < 0:
                self._viol("lds-write", l, a, n, f"LDS allocation {self.lds_bytes}")
                continue
            self.lds[a:a + n] = data[l]

    # ---- compare helper
    def _cmp_write(self, ins, res):
        res = res & self.mask()
        bits = self.mask_to_bits(res)
        if len(ins.ops) == 3:
            self.ws(ins.ops[0], bits, 64)
        else:
            self.set_s64(106, bits)

    # ---- run
    def run(self):
        insts = self.insts
        while True:
            if self.pc >= len(insts):
                raise EmuError("ran off the end of the kernel")
            ins = insts[self.pc]
            self.steps += 1
            if self.steps > self.max_steps:
                raise EmuError(f"step limit {self.max_steps} reached at line {ins.line}")
            nxt = self.pc + 1
            op = ins.op
            h = _DISPATCH.get(op)
            if h is None:
                h = _resolve(op)
                if h is None:
                    raise EmuError(f"line {ins.line}: opcode not implemented: {ins.text}")
                _DISPATCH[op] = h
            r = h(self, ins)
            if r == "end":
                return
            self.pc = nxt if r is None else r


_DISPATCH = {}


def _resolve(op):
    for pat, fn in _PATTERNS:
        if re.fullmatch(pat, op):
            return fn
    return None


# --------------------------------------------------------------------------------------------------------------------- handlers
def _nop(w, ins):
    return None


def _endpgm(w, ins):
    return "end"


def _branch(w, ins):
    return w.labels[ins.ops[0].val]


def _cbranch(w, ins):
    op = ins.op
    take = {"s_cbranch_scc0": w.scc == 0, "s_cbranch_scc1": w.scc == 1, "s_cbranch_vccz": w.s64(106) == 0, "s_cbranch_vccnz": w.s64(106) != 0,
            "s_cbranch_execz": w.exec_ == 0, "s_cbranch_execnz": w.exec_ != 0}[op]
    return w.labels[ins.ops[0].val] if take else None


def _sgn32(v):
    v &= M32
    return v - (1 << 32) if v & 0x80000000 else v


def _s_mov(w, ins):
    bits = 64 if ins.op.endswith("b64") else 32
    w.ws(ins.ops[0], w.rs(ins.ops[1], bits), bits)


def _simm16(v):
    v &= 0xFFFF
    return v - 0x10000 if v & 0x8000 else v


def _s_movk(w, ins):
    w.ws(ins.ops[0], _simm16(ins.ops[1].val) & M32)


def _s_addk(w, ins):
    a = _sgn32(w.rs(ins.ops[0]))
    r = a + _simm16(ins.ops[1].val)
    w.scc = int(r > 0x7FFFFFFF or r < -0x80000000)
    w.ws(ins.ops[0], r & M32)


def _s_mulk(w, ins):
    w.ws(ins.ops[0], (_sgn32(w.rs(ins.ops[0])) * _simm16(ins.ops[1].val)) & M32)


def _s_alu2(w, ins):
    op = ins.op
    d, a, b = ins.ops
    if op.endswith("_b64"):
        x, y = w.rs(a, 64), w.rs(b, 64)
        base = op[2:-4]
        if base == "and": r = x & y
        elif base == "or": r = x | y
        elif base == "xor": r = x ^ y
        elif base == "andn2": r = x & ~y & M64
        elif base == "orn2": r = (x | ~y) & M64
        elif base == "nor": r = ~(x | y) & M64
        elif base == "nand": r = ~(x & y) & M64
        elif base == "lshl": r = (x << (w.rs(b, 32) & 63)) & M64
        elif base == "lshr": r = x >> (w.rs(b, 32) & 63)
        else: raise EmuError(ins.text)
        w.scc = int(r != 0)
        w.ws(d, r, 64)
        return
    x, y = w.rs(a, 32), w.rs(b, 32)
    base = op[2:]
    if base == "add_i32":
        r = _sgn32(x) + _sgn32(y); w.scc = int(r > 0x7FFFFFFF or r < -0x80000000)
    elif base == "sub_i32":
        r = _sgn32(x) - _sgn32(y); w.scc = int(r > 0x7FFFFFFF or r < -0x80000000)
    elif base == "add_u32":
        r = x + y; w.scc = int(r > M32)
    elif base == "sub_u32":
        r = x - y; w.scc = int(y > x)
    elif base == "addc_u32":
        r = x + y + w.scc; w.scc = int(r > M32)
    elif base == "mul_i32":
        r = _sgn32(x) * _sgn32(y)
    elif base == "mul_hi_i32":
        r = (_sgn32(x) * _sgn32(y)) >> 32
    elif base == "mul_hi_u32":
        r = (x * y) >> 32
    elif base == "and_b32":
        r = x & y; w.scc = int(r != 0)
    elif base == "or_b32":
        r = x | y; w.scc = int((r & M32) != 0)
    elif base == "xor_b32":
        r = x ^ y; w.scc = int(r != 0)
    elif base == "andn2_b32":
        r = x & ~y; w.scc = int((r & M32) != 0)
    elif base == "lshl_b32":
        r = x << (y & 31); w.scc = int((r & M32) != 0)
    elif base == "lshr_b32":
        r = x >> (y & 31); w.scc = int(r != 0)
    elif base == "ashr_i32":
        r = _sgn32(x) >> (y & 31); w.scc = int((r & M32) != 0)
    elif base == "min_i32":
        r = min(_sgn32(x), _sgn32(y)); w.scc = int(_sgn32(x) < _sgn32(y))
    elif base == "max_i32":
        r = max(_sgn32(x), _sgn32(y)); w.scc = int(_sgn32(x) > _sgn32(y))
    elif base == "min_u32":
        r = min(x, y); w.scc = int(x < y)
    elif base == "max_u32":
        r = max(x, y); w.scc = int(x > y)
    else:
        raise EmuError(f"line {ins.line}: {ins.text}")
    w.ws(d, r & M32)


def _s_cselect(w, ins):
    bits = 64 if ins.op.endswith("b64") else 32
    w.ws(ins.ops[0], w.rs(ins.ops[1], bits) if w.scc else w.rs(ins.ops[2], bits), bits)


def _s_cmp(w, ins):
    op = ins.op[6:]
    a, b = ins.ops
    if op.endswith("u64"):
        x, y = w.rs(a, 64), w.rs(b, 64)
    elif op.endswith("i32"):
        x, y = _sgn32(w.rs(a)), _sgn32(w.rs(b))
    else:
        x, y = w.rs(a), w.rs(b)
    c = op.split("_")[0]
    w.scc = int({"eq": x == y, "lg": x != y, "gt": x > y, "ge": x >= y, "lt": x < y, "le": x <= y}[c])


def _saveexec(w, ins):
    base = ins.op[2:-len("_saveexec_b64")]
    old = w.exec_
    src = w.rs(ins.ops[1], 64)
    if base == "and": new = src & old
    elif base == "or": new = src | old
    elif base == "xor": new = src ^ old
    elif base == "andn2": new = src & ~old & M64
    elif base == "andn1": new = ~src & old & M64
    elif base == "orn2": new = (src | ~old) & M64
    else: raise EmuError(ins.text)
    w.ws(ins.ops[0], old, 64)
    w.exec_ = new
    w.scc = int(new != 0)


def _s_load(w, ins):
    n = {"s_load_dword": 1, "s_load_dwordx2": 2, "s_load_dwordx4": 4, "s_load_dwordx8": 8, "s_load_dwordx16": 16}[ins.op]
    base = w.rs(ins.ops[1], 64)
    off = ins.ops[2].val if ins.ops[2].kind == "lit" else w.rs(ins.ops[2])
    raw = w.sload(base + off, 4 * n)
    vals = struct.unpack(f"<{n}I", raw)
    d = ins.ops[0]
    if d.kind not in ("s", "vcc"):
        raise EmuError(f"s_load destination {d}")
    first = 106 if d.kind == "vcc" else d.n          # (the compiler uses vcc as an ordinary register pair where no compare needs it)
    for i, v in enumerate(vals):
        w.set_s32(first + i, v)


def _s_getpc(w, ins):
    w.ws(ins.ops[0], 0, 64)


def _s_sym_add(w, ins):
    # s_add_u32 sX, sX, SYM@rel32@lo+4 / s_addc_u32 sY, sY, SYM@rel32@hi+12 after s_getpc_b64: the pair becomes the symbol's address
    name, which = ins.ops[2].val
    addr = w.symbols[name.split("+")[0]]
    if which.startswith("lo"):
        w.ws(ins.ops[0], addr & M32); w.scc = 0
    else:
        w.ws(ins.ops[0], addr >> 32)


def _s_add_dispatch(w, ins):
    if len(ins.ops) == 3 and ins.ops[2].kind == "sym":
        return _s_sym_add(w, ins)
    return _s_alu2(w, ins)


# ---- vector moves / integer
def _v_mov32(w, ins):
    w.wv32(ins.ops[0], w.rv32(ins.ops[1]))


def _v_mov64(w, ins):
    o = ins.ops[1]
    if o.kind == "flit":
        w.wv64(ins.ops[0], _u64(np.full(64, o.val)))
    elif o.kind == "lit":
        w.wv64(ins.ops[0], np.full(64, o.val & M64 if o.val >= 0 else (o.val + (1 << 64)), dtype=np.uint64))
    else:
        w.wv64(ins.ops[0], w.rv64(o))


def _v_acc(w, ins):
    w.wv32(ins.ops[0], w.rv32(ins.ops[1]))


def _i32(a):
    return a.astype(np.int32)


def _v_int2(w, ins):
    op = ins.op
    base = re.sub(r"_e(32|64)$", "", op)[2:]
    d = ins.ops[0]
    a = w.rv32(ins.ops[1]); b = w.rv32(ins.ops[2])
    if base == "add_u32": r = a + b
    elif base == "sub_u32": r = a - b
    elif base == "subrev_u32": r = b - a
    elif base == "and_b32": r = a & b
    elif base == "or_b32": r = a | b
    elif base == "xor_b32": r = a ^ b
    elif base == "lshlrev_b32": r = b << (a & np.uint32(31))
    elif base == "lshrrev_b32": r = b >> (a & np.uint32(31))
    elif base == "ashrrev_i32": r = (_i32(b) >> (a & np.uint32(31)).astype(np.int32)).astype(np.uint32)
    elif base == "mul_lo_u32": r = (a.astype(np.uint64) * b.astype(np.uint64)).astype(np.uint32)
    elif base == "mul_u32_u24": r = ((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)).astype(np.uint32)
    elif base == "min_u32": r = np.minimum(a, b)
    elif base == "max_u32": r = np.maximum(a, b)
    elif base == "min_i32": r = np.minimum(_i32(a), _i32(b)).astype(np.uint32)
    elif base == "max_i32": r = np.maximum(_i32(a), _i32(b)).astype(np.uint32)
    else: raise EmuError(f"line {ins.line}: {ins.text}")
    w.wv32(d, r)


def _v_int3(w, ins):
    base = re.sub(r"_e(32|64)$", "", ins.op)[2:]
    d = ins.ops[0]
    a, b, c = (w.rv32(o) for o in ins.ops[1:4])
    if base == "lshl_add_u32": r = (a << (b & np.uint32(31))) + c
    elif base == "add_lshl_u32": r = (a + b) << (c & np.uint32(31))
    elif base == "add3_u32": r = a + b + c
    elif base == "or3_b32": r = a | b | c
    elif base == "and_or_b32": r = (a & b) | c
    elif base == "lshl_or_b32": r = (a << (b & np.uint32(31))) | c
    elif base == "min3_i32": r = np.minimum(np.minimum(_i32(a), _i32(b)), _i32(c)).astype(np.uint32)
    elif base == "max3_i32": r = np.maximum(np.maximum(_i32(a), _i32(b)), _i32(c)).astype(np.uint32)
    elif base == "bfe_u32": r = (a >> (b & np.uint32(31))) & ((np.uint32(1) << (c & np.uint32(31))) - np.uint32(1))
    elif base == "bfe_i32":
        wd = (c & np.uint32(31)).astype(np.int64); sh = (b & np.uint32(31)).astype(np.int64)
        v = (a.astype(np.int64) >> sh) & ((np.int64(1) << wd) - 1)
        sign = (v >> (wd - 1)) & 1
        r = np.where((wd > 0) & (sign == 1), v - (np.int64(1) << wd), v).astype(np.int32).astype(np.uint32)
    elif base == "alignbit_b32":
        r = (((a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)) >> (c & np.uint32(31)).astype(np.uint64)).astype(np.uint32)
    elif base == "mad_u32_u24":
        r = ((a & np.uint32(0xFFFFFF)).astype(np.uint64) * (b & np.uint32(0xFFFFFF)).astype(np.uint64)).astype(np.uint32) + c
    else: raise EmuError(f"line {ins.line}: {ins.text}")
    w.wv32(d, r)


def _v_bitop3(w, ins):
    a, b, c = (w.rv32(o) for o in ins.ops[1:4])
    lut = ins.mods["bitop3"]
    r = np.zeros(64, dtype=np.uint32)
    for i in range(8):
        if (lut >> i) & 1:
            ta = a if i & 4 else ~a
            tb = b if i & 2 else ~b
            tc = c if i & 1 else ~c
            r |= ta & tb & tc
    if ins.op.endswith("b16"):
        r = (r & np.uint32(0xFFFF)) | (w.rv32(ins.ops[0]) & np.uint32(0xFFFF0000))
    w.wv32(ins.ops[0], r)


def _v_not(w, ins):
    w.wv32(ins.ops[0], ~w.rv32(ins.ops[1]))


def _v_lshl_add_u64(w, ins):
    a = w.rv64(ins.ops[1]); sh = w.rv32(ins.ops[2]).astype(np.uint64) & np.uint64(7); c = w.rv64(ins.ops[3])
    w.wv64(ins.ops[0], (a << sh) + c)


def _v_lshlrev_b64(w, ins):
    sh = w.rv32(ins.ops[1]).astype(np.uint64) & np.uint64(63)
    w.wv64(ins.ops[0], w.rv64(ins.ops[2]) << sh)


def _v_mad64(w, ins):
    d, sd, a, b, c = ins.ops
    if ins.op == "v_mad_u64_u32":
        r = w.rv32(a).astype(np.uint64) * w.rv32(b).astype(np.uint64) + w.rv64(c)
    else:
        r = (_i32(w.rv32(a)).astype(np.int64) * _i32(w.rv32(b)).astype(np.int64) + w.rv64(c).view(np.int64)).view(np.uint64)
    w.wv64(d, r)
    w.ws(sd, 0, 64)          # carry-out: never consumed by these kernels (checked: no v_addc / s_cbranch_vcc follows on it)


def _v_cndmask(w, ins):
    d, a, b = ins.ops[:3]
    sel = w.bits_to_mask(w.rs(ins.ops[3], 64) if len(ins.ops) > 3 else w.s64(106))
    w.wv32(d, np.where(sel, w.rv32(b), w.rv32(a)))


def _v_mbcnt(w, ins):
    m = w.rs(ins.ops[1], 32) if ins.ops[1].kind != "lit" else (ins.ops[1].val & M32)
    base = w.rv32(ins.ops[2])
    lane = np.arange(64)
    if ins.op.startswith("v_mbcnt_lo"):
        cnt = np.array([bin(m & ((1 << min(l, 32)) - 1)).count("1") for l in lane], dtype=np.uint32)
    else:
        cnt = np.array([bin(m & ((1 << max(l - 32, 0)) - 1)).count("1") for l in lane], dtype=np.uint32)
    w.wv32(ins.ops[0], cnt + base)


def _v_readlane(w, ins):
    lane = w.rs(ins.ops[2]) & 63 if ins.ops[2].kind != "lit" else ins.ops[2].val & 63
    w.ws(ins.ops[0], int(w.rv32(ins.ops[1])[lane]))


def _v_writelane(w, ins):
    lane = w.rs(ins.ops[2]) & 63 if ins.ops[2].kind != "lit" else ins.ops[2].val & 63
    w.R[w._row(ins.ops[0])][lane] = w.rs(ins.ops[1]) & M32


def _v_readfirstlane(w, ins):
    m = w.exec_
    lane = (m & -m).bit_length() - 1 if m else 0
    w.ws(ins.ops[0], int(w.rv32(ins.ops[1])[lane]))


def _v_cmp_int(w, ins):
    m = re.fullmatch(r"v_cmp_(\w+)_([ui])(16|32|64)_e(32|64)", ins.op)
    c, sg, bits, enc = m.groups()
    a, b = (ins.ops[1], ins.ops[2]) if len(ins.ops) == 3 else (ins.ops[0], ins.ops[1])
    if bits == "64":
        x, y = w.rv64(a), w.rv64(b)
        if sg == "i": x, y = x.view(np.int64), y.view(np.int64)
    else:
        x, y = w.rv32(a), w.rv32(b)
        if bits == "16":
            x, y = x & np.uint32(0xFFFF), y & np.uint32(0xFFFF)
        if sg == "i": x, y = _i32(x), _i32(y)
    res = {"eq": x == y, "ne": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}[c]
    w._cmp_write(ins, res)


def _v_cmp_f64(w, ins):
    m = re.fullmatch(r"v_cmp_(\w+)_f64_e(32|64)", ins.op)
    c, enc = m.groups()
    a, b = (ins.ops[1], ins.ops[2]) if len(ins.ops) == 3 else (ins.ops[0], ins.ops[1])
    x, y = w.rf64(a), w.rf64(b)
    un = np.isnan(x) | np.isnan(y)
    tbl = {"lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y, "eq": x == y, "lg": (x < y) | (x > y)}
    if c in tbl: res = tbl[c]
    elif c == "neq": res = ~(x == y)
    elif c == "nlt": res = ~(x < y)
    elif c == "nle": res = ~(x <= y)
    elif c == "ngt": res = ~(x > y)
    elif c == "nge": res = ~(x >= y)
    elif c == "nlg": res = ~((x < y) | (x > y))
    elif c == "u": res = un
    elif c == "o": res = ~un
    else: raise EmuError(ins.text)
    w._cmp_write(ins, res)


# ---- FP64
def _v_f64_2(w, ins):
    base = re.sub(r"_e(32|64)$", "", ins.op)[2:]
    a, b = w.rf64(ins.ops[1]), w.rf64(ins.ops[2])
    if base == "add_f64": r = a + b
    elif base == "mul_f64": r = a * b
    elif base == "max_f64": r = np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.maximum(a, b)))
    elif base == "min_f64": r = np.where(np.isnan(a), b, np.where(np.isnan(b), a, np.minimum(a, b)))
    elif base == "fmac_f64": r = fma64(a, b, w.rf64(ins.ops[0]))
    elif base == "ldexp_f64": r = np.ldexp(a, _i32(w.rv32(ins.ops[2])))
    else: raise EmuError(ins.text)
    w.wf64(ins.ops[0], r)


def _v_fma_f64(w, ins):
    w.wf64(ins.ops[0], fma64(w.rf64(ins.ops[1]), w.rf64(ins.ops[2]), w.rf64(ins.ops[3])))


def _v_rcp_f64(w, ins):
    w.wf64(ins.ops[0], 1.0 / w.rf64(ins.ops[1]))


def _v_rsq_f64(w, ins):
    w.wf64(ins.ops[0], 1.0 / np.sqrt(w.rf64(ins.ops[1])))


def _v_div_scale(w, ins):
    # no scaling is modelled: the quotient is formed exactly in v_div_fixup (the sequence's net effect)
    w.wf64(ins.ops[0], w.rf64(ins.ops[2]))
    w.ws(ins.ops[1], 0, 64)


def _v_div_fmas(w, ins):
    w.wf64(ins.ops[0], fma64(w.rf64(ins.ops[1]), w.rf64(ins.ops[2]), w.rf64(ins.ops[3])))


def _v_div_fixup(w, ins):
    w.wf64(ins.ops[0], w.rf64(ins.ops[3]) / w.rf64(ins.ops[2]))


def _v_cvt_f64_i32(w, ins):
    w.wf64(ins.ops[0], _i32(w.rv32(ins.ops[1])).astype(np.float64))


def _v_cvt_i32_f64(w, ins):
    x = w.rf64(ins.ops[1])
    x = np.where(np.isnan(x), 0.0, np.clip(np.trunc(x), -2147483648.0, 2147483647.0))
    w.wv32(ins.ops[0], x.astype(np.int64).astype(np.int32).astype(np.uint32))


def _v_cvt_f64_u32(w, ins):
    w.wf64(ins.ops[0], w.rv32(ins.ops[1]).astype(np.float64))


_BLK = np.array([[[16 * a + 4 * b + c for c in range(4)] for a in range(4)] for b in range(4)])    # [block][a][c] -> lane


def _v_mfma_f64_4x4x4(w, ins):
    """D = X' Y + C per block; element (a, c) of block b lives in lane 16 a + 4 b + c (probed on gfx950: tools/probe_mfma).  EXEC is ignored."""
    d, xa, yb, cc = ins.ops
    X = _f64(w.rv64(xa)) if xa.kind in "va" else w.rf64(xa)
    Y = _f64(w.rv64(yb)) if yb.kind in "va" else w.rf64(yb)
    C = w.rf64(cc)
    ng = ins.mods.get("neg")
    if ng:
        if ng[0]: X = -X
        if ng[1]: Y = -Y
        if ng[2]: C = -C
    out = np.zeros(64, dtype=np.float64)
    ld = np.longdouble
    for b in range(4):
        L = _BLK[b]
        Xb, Yb, Cb = X[L], Y[L], C[L]                 # [a][c]
        acc = Cb.astype(ld)
        for k in range(4):
            acc = acc + np.outer(Xb[k, :].astype(ld), Yb[k, :].astype(ld))
        out[L.reshape(-1)] = acc.astype(np.float64).reshape(-1)
    w.wf64(d, out, masked=False)


# ---- memory
def _global_load(w, ins):
    n = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "ushort": 2, "sbyte": 1, "ubyte": 1, "sshort": 2}[ins.op[len("global_load_"):]]
    d, vaddr, saddr = ins.ops[0], ins.ops[1], ins.ops[2] if len(ins.ops) > 2 else Op("off")
    addrs = w._gaddr(ins, vaddr, saddr)
    lanes = np.nonzero(w.mask())[0]
    raw = w.gload(addrs, n, lanes)
    if n >= 4:
        words = raw.view(np.uint32).reshape(64, n // 4)
        for i in range(n // 4):
            w.wv32(Op(d.kind, d.n + i), words[:, i])
    else:
        if n == 2:
            v = raw.view(np.uint16).reshape(64).astype(np.uint32)
            if ins.op.endswith("sshort"): v = v.astype(np.uint16).view(np.int16).astype(np.int32).astype(np.uint32)
        else:
            v = raw.reshape(64).astype(np.uint32)
            if ins.op.endswith("sbyte"): v = raw.reshape(64).view(np.int8).astype(np.int32).astype(np.uint32)
        w.wv32(d, v)


def _global_store(w, ins):
    n = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16, "short": 2, "byte": 1}[ins.op[len("global_store_"):]]
    vaddr, data, saddr = ins.ops[0], ins.ops[1], ins.ops[2] if len(ins.ops) > 2 else Op("off")
    addrs = w._gaddr(ins, vaddr, saddr)
    lanes = np.nonzero(w.mask())[0]
    buf = np.zeros((64, max(n, 4)), dtype=np.uint8)
    for i in range(max(n // 4, 1)):
        buf[:, 4 * i:4 * i + 4] = w.rv32(Op(data.kind, data.n + i)).view(np.uint8).reshape(64, 4)
    w.gstore(addrs, buf[:, :n], n, lanes)


def _scratch_addr(w, ins, vaddr, saddr):
    a = np.full(64, ins.mods.get("offset", 0), dtype=np.int64)
    if vaddr.kind != "off":
        a = a + w.rv32(vaddr).astype(np.int64)
    if saddr.kind != "off":
        a = a + np.int64(w.rs(saddr, 32))
    return a


def _scratch_load(w, ins):
    # private (per-lane) memory of the wave: spills.  Bounds = the kernel's private_segment_fixed_size (Wave.scratch_bytes); a read of a byte
    # no store has written is a violation too (a reload without its spill)
    n = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}[ins.op[len("scratch_load_"):]]
    d, vaddr, saddr = ins.ops[0], ins.ops[1], ins.ops[2] if len(ins.ops) > 2 else Op("off")
    a = _scratch_addr(w, ins, vaddr, saddr)
    out = np.zeros((64, n), dtype=np.uint8)
    for l in np.nonzero(w.mask())[0]:
        o = int(a[l])
        if o < 0 or o + n > w.scratch_bytes:
            w._viol("scratch-read", l, o, n, f"private segment {w.scratch_bytes}")
            continue
        if not w.scratch_set[l, o:o + n].all():
            w._viol("scratch-read", l, o, n, "never written")
        out[l] = w.scratch[l, o:o + n]
    words = out.view(np.uint32).reshape(64, n // 4)
    for i in range(n // 4):
        w.wv32(Op(d.kind, d.n + i), words[:, i])


def _scratch_store(w, ins):
    n = {"dword": 4, "dwordx2": 8, "dwordx3": 12, "dwordx4": 16}[ins.op[len("scratch_store_"):]]
    vaddr, data, saddr = ins.ops[0], ins.ops[1], ins.ops[2] if len(ins.ops) > 2 else Op("off")
    a = _scratch_addr(w, ins, vaddr, saddr)
    buf = np.zeros((64, n), dtype=np.uint8)
    for i in range(n // 4):
        buf[:, 4 * i:4 * i + 4] = w.rv32(Op(data.kind, data.n + i)).view(np.uint8).reshape(64, 4)
    for l in np.nonzero(w.mask())[0]:
        o = int(a[l])
        if o < 0 or o + n > w.scratch_bytes:
            w._viol("scratch-write", l, o, n, f"private segment {w.scratch_bytes}")
            continue
        w.scratch[l, o:o + n] = buf[l]
        w.scratch_set[l, o:o + n] = True


def _ds_read(w, ins):
    op = ins.op
    lanes = np.nonzero(w.mask())[0]
    d, a = ins.ops[0], ins.ops[1]
    base = w.rv32(a).astype(np.int64)
    if op in ("ds_read_b64", "ds_read_b32", "ds_read_b128"):
        n = {"ds_read_b32": 4, "ds_read_b64": 8, "ds_read_b128": 16}[op]
        raw = w.lds_rd(base + ins.mods.get("offset", 0), n, lanes).view(np.uint32).reshape(64, n // 4)
        for i in range(n // 4):
            w.wv32(Op(d.kind, d.n + i), raw[:, i])
    elif op in ("ds_read2_b64", "ds_read2_b32"):
        n = 8 if op.endswith("b64") else 4
        for j, key in enumerate(("offset0", "offset1")):
            raw = w.lds_rd(base + ins.mods.get(key, 0) * n, n, lanes).view(np.uint32).reshape(64, n // 4)
            for i in range(n // 4):
                w.wv32(Op(d.kind, d.n + j * (n // 4) + i), raw[:, i])
    else:
        raise EmuError(ins.text)


def _ds_write(w, ins):
    op = ins.op
    lanes = np.nonzero(w.mask())[0]
    a = ins.ops[0]
    base = w.rv32(a).astype(np.int64)

    def pack(o, n):
        buf = np.zeros((64, n), dtype=np.uint8)
        for i in range(n // 4):
            buf[:, 4 * i:4 * i + 4] = w.rv32(Op(o.kind, o.n + i)).view(np.uint8).reshape(64, 4)
        return buf
    if op in ("ds_write_b64", "ds_write_b32", "ds_write_b128"):
        n = {"ds_write_b32": 4, "ds_write_b64": 8, "ds_write_b128": 16}[op]
        w.lds_wr(base + ins.mods.get("offset", 0), pack(ins.ops[1], n), n, lanes)
    elif op in ("ds_write2_b64", "ds_write2_b32"):
        n = 8 if op.endswith("b64") else 4
        w.lds_wr(base + ins.mods.get("offset0", 0) * n, pack(ins.ops[1], n), n, lanes)
        w.lds_wr(base + ins.mods.get("offset1", 0) * n, pack(ins.ops[2], n), n, lanes)
    else:
        raise EmuError(ins.text)


def _ds_bpermute(w, ins):
    d, addr, data = ins.ops[:3]
    src = (((w.rv32(addr).astype(np.int64) + ins.mods.get("offset", 0)) >> 2) & 63).astype(np.int64)
    vals = w.rv32(data)
    act = w.mask()
    w.wv32(d, np.where(act[src], vals[src], np.uint32(0)))



def _sdwa_sel(x, sel, sext=False):
    if sel in (None, "DWORD"):
        return x
    kind, idx = sel.split("_")
    idx = int(idx)
    if kind == "BYTE":
        v = (x >> np.uint32(8 * idx)) & np.uint32(0xFF)
        return v.astype(np.uint8).view(np.int8).astype(np.int32).astype(np.uint32) if sext else v
    v = (x >> np.uint32(16 * idx)) & np.uint32(0xFFFF)
    return v.astype(np.uint16).view(np.int16).astype(np.int32).astype(np.uint32) if sext else v


def _v_cmp_sdwa(w, ins):
    m = re.fullmatch(r"v_cmp_(\w+)_([ui])(16|32)_sdwa", ins.op)
    c, sg, bits = m.groups()
    x = _sdwa_sel(w.rv32(ins.ops[1]), ins.mods.get("src0_sel"))
    y = _sdwa_sel(w.rv32(ins.ops[2]), ins.mods.get("src1_sel"))
    if bits == "16":
        x, y = x & np.uint32(0xFFFF), y & np.uint32(0xFFFF)
    if sg == "i":
        x, y = _i32(x), _i32(y)
    res = {"eq": x == y, "ne": x != y, "lt": x < y, "le": x <= y, "gt": x > y, "ge": x >= y}[c]
    w.ws(ins.ops[0], w.mask_to_bits(res & w.mask()), 64)


def _v_int2_sdwa(w, ins):
    """v_<op>_sdwa vD, src0, src1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:<sel> src1_sel:<sel>: the VOP2 integer operation on a byte /
    word / dword of each source (zero-extended); only a whole-dword destination is written by the compiler in these kernels."""
    if ins.mods.get("dst_sel", "DWORD") != "DWORD":
        raise EmuError(f"line {ins.line}: {ins.text} (dst_sel other than DWORD)")
    base = re.fullmatch(r"v_(\w+)_sdwa", ins.op).group(1)
    a = _sdwa_sel(w.rv32(ins.ops[1]), ins.mods.get("src0_sel"))
    b = _sdwa_sel(w.rv32(ins.ops[2]), ins.mods.get("src1_sel"))
    if base == "add_u32": r = a + b
    elif base == "sub_u32": r = a - b
    elif base == "and_b32": r = a & b
    elif base == "or_b32": r = a | b
    elif base == "lshlrev_b32": r = b << (a & np.uint32(31))
    elif base == "lshrrev_b32": r = b >> (a & np.uint32(31))
    else: raise EmuError(f"line {ins.line}: {ins.text}")
    w.wv32(ins.ops[0], r.astype(np.uint32))


def _v_addsub_co(w, ins):
    d, sd, a, b = ins.ops[:4]
    x, y = w.rv32(a).astype(np.uint64), w.rv32(b).astype(np.uint64)
    if ins.op.startswith("v_add_co"):
        r = x + y; carry = r > np.uint64(M32)
    else:
        r = x - y; carry = y > x
    w.wv32(d, (r & np.uint64(M32)).astype(np.uint32))
    w.ws(sd, w.mask_to_bits(carry & w.mask()), 64)


def _v_mul_hi(w, ins):
    a, b = w.rv32(ins.ops[1]), w.rv32(ins.ops[2])
    if ins.op.startswith("v_mul_hi_i32"):
        r = ((_i32(a).astype(np.int64) * _i32(b).astype(np.int64)) >> 32).astype(np.int32).astype(np.uint32)
    else:
        r = ((a.astype(np.uint64) * b.astype(np.uint64)) >> np.uint64(32)).astype(np.uint32)
    w.wv32(ins.ops[0], r)


def _v_mul_i32_i24(w, ins):
    def s24(x):
        v = (x & np.uint32(0xFFFFFF)).astype(np.int64)
        return np.where(v & 0x800000, v - 0x1000000, v)
    w.wv32(ins.ops[0], (s24(w.rv32(ins.ops[1])) * s24(w.rv32(ins.ops[2]))).astype(np.int32).astype(np.uint32))


def _v_mad_i32_i24(w, ins):
    def s24(x):
        v = (x & np.uint32(0xFFFFFF)).astype(np.int64)
        return np.where(v & 0x800000, v - 0x1000000, v)
    acc = w.rv32(ins.ops[3]).astype(np.int32).astype(np.int64)
    w.wv32(ins.ops[0], ((s24(w.rv32(ins.ops[1])) * s24(w.rv32(ins.ops[2])) + acc) & 0xFFFFFFFF).astype(np.uint32))


def _v_mad_u32_u24(w, ins):
    u24 = lambda x: (x & np.uint32(0xFFFFFF)).astype(np.uint64)
    w.wv32(ins.ops[0], ((u24(w.rv32(ins.ops[1])) * u24(w.rv32(ins.ops[2])) + w.rv32(ins.ops[3]).astype(np.uint64)) & np.uint64(0xFFFFFFFF)).astype(np.uint32))


def _v_bfrev(w, ins):
    x = w.rv32(ins.ops[1])
    r = np.zeros(64, dtype=np.uint32)
    for i in range(32):
        r |= ((x >> np.uint32(i)) & np.uint32(1)) << np.uint32(31 - i)
    w.wv32(ins.ops[0], r)


def _v_cvt_f32_f64(w, ins):
    w.wv32(ins.ops[0], w.rf64(ins.ops[1]).astype(np.float32).view(np.uint32))


def _v_cvt_f64_f32(w, ins):
    w.wf64(ins.ops[0], w.rv32(ins.ops[1]).view(np.float32).astype(np.float64))


def _v_f32_2(w, ins):
    base = re.sub(r"_e(32|64)$", "", ins.op)[2:]
    a, b = w.rv32(ins.ops[1]).view(np.float32), w.rv32(ins.ops[2]).view(np.float32)
    r = {"add_f32": a + b, "mul_f32": a * b, "sub_f32": a - b}[base]
    w.wv32(ins.ops[0], r.astype(np.float32).view(np.uint32))


def _s_misc1(w, ins):
    op = ins.op
    if op == "s_bcnt1_i32_b64":
        r = bin(w.rs(ins.ops[1], 64)).count("1"); w.scc = int(r != 0); w.ws(ins.ops[0], r)
    elif op == "s_brev_b32":
        w.ws(ins.ops[0], int(f"{w.rs(ins.ops[1]) & M32:032b}"[::-1], 2))
    elif op == "s_not_b32":
        r = ~w.rs(ins.ops[1]) & M32; w.scc = int(r != 0); w.ws(ins.ops[0], r)
    elif op == "s_not_b64":
        r = ~w.rs(ins.ops[1], 64) & M64; w.scc = int(r != 0); w.ws(ins.ops[0], r, 64)
    else:
        raise EmuError(ins.text)


def _global_atomic_add(w, ins):
    # global_atomic_add [vdst,] vaddr, vdata, saddr|off [glc]: lanes in lane order (one wave: any order is a valid execution)
    ret = "glc" in ins.mods or "sc0" in ins.mods
    ops = ins.ops
    if ret:
        d, vaddr, data, saddr = ops[0], ops[1], ops[2], ops[3] if len(ops) > 3 else Op("off")
    else:
        d, vaddr, data, saddr = None, ops[0], ops[1], ops[2] if len(ops) > 2 else Op("off")
    addrs = w._gaddr(ins, vaddr, saddr)
    vals = w.rv32(data)
    old = np.zeros(64, dtype=np.uint32)
    for l in np.nonzero(w.mask())[0]:
        raw = w.gload(addrs, 4, [l])
        o = int(raw[l].view(np.uint32)[0])
        old[l] = o
        buf = np.zeros((64, 4), dtype=np.uint8)
        buf[l] = np.frombuffer(struct.pack("<I", (o + int(vals[l])) & M32), dtype=np.uint8)
        w.gstore(addrs, buf, 4, [l])
    if d is not None:
        w.wv32(d, old)


_PATTERNS = [
    (r"s_nop|s_waitcnt|s_barrier|s_setprio|s_sleep|s_waitcnt_depctr|s_setreg_.*|s_sethalt|s_icache_inv|buffer_wbl2.*|buffer_inv.*|s_dcache_wb", _nop),
    (r"s_endpgm", _endpgm),
    (r"s_branch", _branch),
    (r"s_cbranch_\w+", _cbranch),
    (r"s_mov_b(32|64)", _s_mov),
    (r"s_movk_i32", _s_movk),
    (r"s_addk_i32", _s_addk),
    (r"s_mulk_i32", _s_mulk),
    (r"s_(add|addc)_u32", _s_add_dispatch),
    (r"s_(and|or|xor|andn2|orn2|nor|nand|lshl|lshr)_b64", _s_alu2),
    (r"s_(add_i32|sub_i32|sub_u32|mul_i32|mul_hi_i32|mul_hi_u32|and_b32|or_b32|xor_b32|andn2_b32|lshl_b32|lshr_b32|ashr_i32|min_i32|max_i32|min_u32|max_u32)", _s_alu2),
    (r"s_cselect_b(32|64)", _s_cselect),
    (r"s_cmp_\w+", _s_cmp),
    (r"s_\w+_saveexec_b64", _saveexec),
    (r"s_load_dword(x2|x4|x8|x16)?", _s_load),
    (r"s_getpc_b64", _s_getpc),
    (r"v_mov_b32_e32|v_mov_b32_e64", _v_mov32),
    (r"v_mov_b64_e32|v_mov_b64_e64", _v_mov64),
    (r"v_accvgpr_(read|write|mov)_b32", _v_acc),
    (r"v_(add_u32|sub_u32|subrev_u32|and_b32|or_b32|xor_b32|lshlrev_b32|lshrrev_b32|ashrrev_i32|mul_lo_u32|mul_u32_u24|min_u32|max_u32|min_i32|max_i32)(_e32|_e64)?", _v_int2),
    (r"v_(lshl_add_u32|add_lshl_u32|add3_u32|or3_b32|and_or_b32|lshl_or_b32|min3_i32|max3_i32|bfe_u32|bfe_i32|alignbit_b32|mad_u32_u24)(_e64)?", _v_int3),
    (r"v_bitop3_b(16|32)", _v_bitop3),
    (r"v_not_b32_e32", _v_not),
    (r"v_lshl_add_u64", _v_lshl_add_u64),
    (r"v_lshlrev_b64", _v_lshlrev_b64),
    (r"v_mad_u64_u32|v_mad_i64_i32", _v_mad64),
    (r"v_cndmask_b32_e(32|64)", _v_cndmask),
    (r"v_mbcnt_(lo|hi)_u32_b32", _v_mbcnt),
    (r"v_readlane_b32", _v_readlane),
    (r"v_writelane_b32", _v_writelane),
    (r"v_readfirstlane_b32", _v_readfirstlane),
    (r"v_cmp_\w+_[ui](16|32|64)_e(32|64)", _v_cmp_int),
    (r"v_cmp_\w+_f64_e(32|64)", _v_cmp_f64),
    (r"v_(add_f64|mul_f64|max_f64|min_f64|ldexp_f64)(_e32|_e64)?|v_fmac_f64_e(32|64)", _v_f64_2),
    (r"v_fma_f64", _v_fma_f64),
    (r"v_rcp_f64_e32", _v_rcp_f64),
    (r"v_rsq_f64_e32", _v_rsq_f64),
    (r"v_div_scale_f64", _v_div_scale),
    (r"v_div_fmas_f64", _v_div_fmas),
    (r"v_div_fixup_f64", _v_div_fixup),
    (r"v_cvt_f64_i32_e32", _v_cvt_f64_i32),
    (r"v_cvt_f64_u32_e32", _v_cvt_f64_u32),
    (r"v_cvt_i32_f64_e32", _v_cvt_i32_f64),
    (r"v_mfma_f64_4x4x4_4b_f64", _v_mfma_f64_4x4x4),
    (r"global_load_(dword|dwordx2|dwordx3|dwordx4|ushort|sshort|sbyte|ubyte)", _global_load),
    (r"global_store_(dword|dwordx2|dwordx3|dwordx4|short|byte)", _global_store),
    (r"scratch_load_(dword|dwordx2|dwordx3|dwordx4)", _scratch_load),
    (r"scratch_store_(dword|dwordx2|dwordx3|dwordx4)", _scratch_store),
    (r"ds_read2?_b(32|64|128)", _ds_read),
    (r"ds_write2?_b(32|64|128)", _ds_write),
    (r"ds_bpermute_b32", _ds_bpermute),
    (r"v_cmp_\w+_[ui](16|32)_sdwa", _v_cmp_sdwa),
    (r"v_(add_u32|sub_u32|and_b32|or_b32|lshlrev_b32|lshrrev_b32)_sdwa", _v_int2_sdwa),
    (r"v_(add|sub)_co_u32_e64", _v_addsub_co),
    (r"v_mul_hi_(i32|u32)(_e64)?", _v_mul_hi),
    (r"v_mul_i32_i24(_e32|_e64)?", _v_mul_i32_i24),
    (r"v_mad_i32_i24", _v_mad_i32_i24),
    (r"v_mad_u32_u24", _v_mad_u32_u24),
    (r"v_bfrev_b32_e32", _v_bfrev),
    (r"v_cvt_f32_f64_e32", _v_cvt_f32_f64),
    (r"v_cvt_f64_f32_e32", _v_cvt_f64_f32),
    (r"v_(add|mul|sub)_f32(_e32|_e64)?", _v_f32_2),
    (r"s_bcnt1_i32_b64|s_brev_b32|s_not_b(32|64)", _s_misc1),
    (r"global_atomic_add", _global_atomic_add),
]

"""Minimal symbolic tracer standing in for the parts of CasADi that the reference's
controller.py:267-355 touches.  Written from scratch; see tests/levelb/README.md."""
from __future__ import annotations

import numpy as np


class SX:
    """Scalar expression node or a column of them."""

    def __init__(self, op, args=(), name=None, value=None, items=None):
        self.op, self.args, self.name, self.value, self.items = op, args, name, value, items

    # -- construction --------------------------------------------------------------------
    @staticmethod
    def sym(name, n=1):
        if n == 0:
            return SX("vec", items=[])
        return SX("vec", items=[SX("sym", name=f"{name}_{i}") for i in range(n)])

    @staticmethod
    def _lift(v):
        if isinstance(v, SX):
            return v
        return SX("const", value=float(v))

    # -- vectors --------------------------------------------------------------------------
    def __getitem__(self, i):
        if self.op != "vec":
            raise TypeError("indexing a scalar expression")
        return self.items[i]

    def __len__(self):
        return len(self.items) if self.op == "vec" else 1

    def _elem(self):
        return self.items if self.op == "vec" else [self]

    # -- arithmetic (element-wise on vectors of equal length, scalar broadcast) --------------
    def _bin(self, other, op, swap=False):
        a, b = (SX._lift(other), self) if swap else (self, SX._lift(other))
        ea, eb = a._elem(), b._elem()
        if len(ea) == 1 and len(eb) > 1:
            ea = ea * len(eb)
        if len(eb) == 1 and len(ea) > 1:
            eb = eb * len(ea)
        if len(ea) != len(eb):
            raise ValueError("shape mismatch")
        out = [SX(op, (x, y)) for x, y in zip(ea, eb)]
        return out[0] if (a.op != "vec" and b.op != "vec") else SX("vec", items=out)

    def __add__(self, o): return self._bin(o, "+")
    def __radd__(self, o): return self._bin(o, "+", True)
    def __sub__(self, o): return self._bin(o, "-")
    def __rsub__(self, o): return self._bin(o, "-", True)
    def __mul__(self, o): return self._bin(o, "*")
    def __rmul__(self, o): return self._bin(o, "*", True)
    def __truediv__(self, o): return self._bin(o, "/")
    def __rtruediv__(self, o): return self._bin(o, "/", True)
    def __neg__(self): return SX._lift(0.0) - self

    def __matmul__(self, o):
        if isinstance(self, Matrix):
            return Matrix.matvec(self, o)
        raise TypeError("only Matrix @ vector is traced")

    # -- evaluation -------------------------------------------------------------------------
    def eval(self, env):
        if self.op == "vec":
            return np.array([e.eval(env) for e in self.items], dtype=float)
        if self.op == "const":
            return self.value
        if self.op == "sym":
            return env[self.name]
        a, b = self.args[0].eval(env), self.args[1].eval(env)
        return a + b if self.op == "+" else a - b if self.op == "-" else a * b if self.op == "*" else a / b


class Matrix(SX):
    """Row-major matrix of scalar expressions (result of vertcat of hcat rows)."""

    def __init__(self, rows):
        super().__init__("mat")
        self.rows = rows

    @staticmethod
    def matvec(m, v):
        ev = SX._lift(v)._elem()
        out = []
        for row in m.rows:
            acc = SX._lift(0.0)
            for a, b in zip(row, ev):
                acc = acc + a * b
            out.append(acc)
        return SX("vec", items=out)

    def __matmul__(self, o):
        return Matrix.matvec(self, o)


class _Row(SX):
    def __init__(self, items):
        super().__init__("row")
        self.items = items


def hcat(items):
    return _Row([SX._lift(i) for i in items])


def vertcat(*items):
    if items and all(isinstance(i, _Row) for i in items):
        return Matrix([r.items for r in items])
    out = []
    for i in items:
        out.extend(SX._lift(i)._elem())
    return SX("vec", items=out)


def sum1(v):
    acc = SX._lift(0.0)
    for e in SX._lift(v)._elem():
        acc = acc + e
    return acc


def dot(a, b):
    acc = SX._lift(0.0)
    for x, y in zip(SX._lift(a)._elem(), SX._lift(b)._elem()):
        acc = acc + x * y
    return acc


def DM(values):
    arr = np.asarray(values, dtype=float).reshape(-1)
    return SX("vec", items=[SX("const", value=float(v)) for v in arr])

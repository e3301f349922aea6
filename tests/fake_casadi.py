"""Test double for the slice of casadi that optas_amd.casadi_tape touches: an SX-like scalar recorder whose ``Function`` exposes the
instruction-introspection methods of ``casadi.Function`` (n_instructions, instruction_id/_input/_output/_constant, sparsity_out) over a
work vector with slot reuse, like the SX virtual machine.  Opcode integers are deliberately arbitrary: the walker must read them from
the module it is handed.  No casadi code or data is involved: casadi is not installed in this image."""
import math

import numpy as np

OP_ASSIGN, OP_ADD, OP_SUB, OP_MUL, OP_DIV, OP_NEG, OP_SIN, OP_COS, OP_SQRT, OP_SQ, OP_TWICE, OP_INV, OP_ATAN2, OP_CONSTPOW, OP_TAN = range(101, 116)
OP_CONST, OP_INPUT, OP_OUTPUT, OP_FABS, OP_POW = 201, 202, 203, 204, 205
OP_ASIN, OP_FMIN, OP_FMAX, OP_LT, OP_LE, OP_EQ, OP_NE, OP_NOT, OP_AND, OP_OR, OP_IF_ELSE_ZERO = range(301, 312)
OP_EXP, OP_LOG, OP_ACOS, OP_ATAN, OP_TANH, OP_SINH, OP_COSH, OP_ASINH, OP_ACOSH, OP_ATANH, OP_LOG1P, OP_EXPM1, OP_SIGN = range(401, 414)
OP_ERF = 501  # an opcode the tape has no counterpart for


class SX:
    """Scalar expression node."""

    def __init__(self, op, args=(), const=None, inp=None):
        self.op, self.args, self.const, self.inp = op, tuple(args), const, inp

    @staticmethod
    def wrap(v):
        return v if isinstance(v, SX) else SX(OP_CONST, const=float(v))

    def __add__(self, o): return SX(OP_ADD, (self, SX.wrap(o)))
    def __radd__(self, o): return SX(OP_ADD, (SX.wrap(o), self))
    def __sub__(self, o): return SX(OP_SUB, (self, SX.wrap(o)))
    def __rsub__(self, o): return SX(OP_SUB, (SX.wrap(o), self))
    def __mul__(self, o): return SX(OP_MUL, (self, SX.wrap(o)))
    def __rmul__(self, o): return SX(OP_MUL, (SX.wrap(o), self))
    def __truediv__(self, o): return SX(OP_DIV, (self, SX.wrap(o)))
    def __rtruediv__(self, o): return SX(OP_DIV, (SX.wrap(o), self))
    def __neg__(self): return SX(OP_NEG, (self,))
    def __pow__(self, e): return SX(OP_POW if isinstance(e, SX) else OP_CONSTPOW, (self, SX.wrap(e)))
    def __lt__(self, o): return SX(OP_LT, (self, SX.wrap(o)))
    def __le__(self, o): return SX(OP_LE, (self, SX.wrap(o)))
    def __gt__(self, o): return SX(OP_LT, (SX.wrap(o), self))  # casadi keeps LT / LE only and swaps the operands
    def __ge__(self, o): return SX(OP_LE, (SX.wrap(o), self))


def sin(a): return SX(OP_SIN, (a,))
def cos(a): return SX(OP_COS, (a,))
def tan(a): return SX(OP_TAN, (a,))
def sqrt(a): return SX(OP_SQRT, (a,))
def sq(a): return SX(OP_SQ, (a,))
def twice(a): return SX(OP_TWICE, (a,))
def inv(a): return SX(OP_INV, (a,))
def fabs(a): return SX(OP_FABS, (a,))
def asin(a): return SX(OP_ASIN, (a,))
def exp(a): return SX(OP_EXP, (a,))
def log(a): return SX(OP_LOG, (a,))
def acos(a): return SX(OP_ACOS, (a,))
def atan(a): return SX(OP_ATAN, (a,))
def tanh(a): return SX(OP_TANH, (a,))
def sinh(a): return SX(OP_SINH, (a,))
def cosh(a): return SX(OP_COSH, (a,))
def asinh(a): return SX(OP_ASINH, (a,))
def acosh(a): return SX(OP_ACOSH, (a,))
def atanh(a): return SX(OP_ATANH, (a,))
def log1p(a): return SX(OP_LOG1P, (a,))
def expm1(a): return SX(OP_EXPM1, (a,))
def sign(a): return SX(OP_SIGN, (a,))
def erf(a): return SX(OP_ERF, (a,))
def power(a, b): return SX(OP_POW, (SX.wrap(a), SX.wrap(b)))
def fmin(a, b): return SX(OP_FMIN, (SX.wrap(a), SX.wrap(b)))
def fmax(a, b): return SX(OP_FMAX, (SX.wrap(a), SX.wrap(b)))
def logic_not(a): return SX(OP_NOT, (a,))
def logic_and(a, b): return SX(OP_AND, (a, b))
def logic_or(a, b): return SX(OP_OR, (a, b))
def if_else_zero(c, x): return SX(OP_IF_ELSE_ZERO, (c, SX.wrap(x)))
def if_else(c, x, y): return if_else_zero(c, x) + if_else_zero(logic_not(c), y)  # what casadi builds for SX
def atan2(a, b): return SX(OP_ATAN2, (a, SX.wrap(b)))
def sym(i, n): return [SX(OP_INPUT, inp=(i, j)) for j in range(n)]


class Sparsity:
    def __init__(self, m, n, rows, cols):
        self._m, self._n, self._rows, self._cols = m, n, rows, cols

    def size1(self): return self._m
    def size2(self): return self._n
    def row(self): return list(self._rows)
    def get_col(self): return list(self._cols)
    def nnz(self): return len(self._rows)


class Function:
    """Function(name, outputs): each output is a list of (row, SX) pairs of an m x 1 column (missing rows = structural zeros)."""

    def __init__(self, name, outputs, sizes):
        self._name, self._ins, self._sp = name, [], []
        free, slot_of, uses = [], {}, {}
        order = []

        def visit(e):
            if id(e) in uses:
                uses[id(e)] += 1
                return
            uses[id(e)] = 1
            for a in e.args:
                visit(a)
            order.append(e)

        for out in outputs:
            for _, e in out:
                visit(e)
        n_slots = 0

        def release(e):
            uses[id(e)] -= 1
            if uses[id(e)] == 0:
                free.append(slot_of[id(e)])

        for e in order:
            ins = [slot_of[id(a)] for a in e.args]
            for a in e.args:
                release(a)
            if free:
                s = free.pop()  # slot reuse: the walker must track which register currently lives in a work slot
            else:
                s, n_slots = n_slots, n_slots + 1
            slot_of[id(e)] = s
            if e.op == OP_CONST:
                self._ins.append((OP_CONST, [s], [], e.const))
            elif e.op == OP_INPUT:
                self._ins.append((OP_INPUT, [s], list(e.inp), None))
            else:
                self._ins.append((e.op, [s], ins, None))
        # outputs last would defeat slot reuse above (a released slot might be overwritten before it is output), so outputs reference
        # expressions that were kept alive by their extra use count from visit(): emit them now
        for j, (out, m) in enumerate(zip(outputs, sizes)):
            rows = [r for r, _ in out]
            self._sp.append(Sparsity(m, 1, rows, [0] * len(rows)))
            for nz, (_, e) in enumerate(out):
                self._ins.append((OP_OUTPUT, [j, nz], [slot_of[id(e)]], None))

    def name(self): return self._name
    def n_out(self): return len(self._sp)
    def n_instructions(self): return len(self._ins)
    def instruction_id(self, k): return self._ins[k][0]
    def instruction_output(self, k): return self._ins[k][1]
    def instruction_input(self, k): return self._ins[k][2]
    def instruction_constant(self, k): return self._ins[k][3]
    def sparsity_out(self, j): return self._sp[j]
    def is_a(self, what): return what == "SXFunction"

    def __call__(self, x, p):
        """Direct evaluation of the instruction list with python floats (the check the tape is compared with)."""
        w, args = {}, [x, p]
        outs = [np.zeros(sp.size1()) for sp in self._sp]
        for op, o, i, c in self._ins:
            if op == OP_CONST: w[o[0]] = c
            elif op == OP_INPUT: w[o[0]] = float(args[i[0]][i[1]])
            elif op == OP_OUTPUT: outs[o[0]][self._sp[o[0]].row()[o[1]]] = w[i[0]]
            elif op == OP_ASSIGN: w[o[0]] = w[i[0]]
            elif op == OP_ADD: w[o[0]] = w[i[0]] + w[i[1]]
            elif op == OP_SUB: w[o[0]] = w[i[0]] - w[i[1]]
            elif op == OP_MUL: w[o[0]] = w[i[0]] * w[i[1]]
            elif op == OP_DIV: w[o[0]] = w[i[0]] / w[i[1]]
            elif op == OP_NEG: w[o[0]] = -w[i[0]]
            elif op == OP_SIN: w[o[0]] = math.sin(w[i[0]])
            elif op == OP_COS: w[o[0]] = math.cos(w[i[0]])
            elif op == OP_TAN: w[o[0]] = math.tan(w[i[0]])
            elif op == OP_SQRT: w[o[0]] = math.sqrt(w[i[0]])
            elif op == OP_SQ: w[o[0]] = w[i[0]] * w[i[0]]
            elif op == OP_TWICE: w[o[0]] = 2.0 * w[i[0]]
            elif op == OP_INV: w[o[0]] = 1.0 / w[i[0]]
            elif op == OP_ATAN2: w[o[0]] = math.atan2(w[i[0]], w[i[1]])
            elif op in (OP_CONSTPOW, OP_POW): w[o[0]] = w[i[0]] ** w[i[1]]
            elif op == OP_FABS: w[o[0]] = abs(w[i[0]])
            elif op == OP_ASIN: w[o[0]] = math.asin(w[i[0]])
            elif op == OP_EXP: w[o[0]] = math.exp(w[i[0]])
            elif op == OP_LOG: w[o[0]] = math.log(w[i[0]])
            elif op == OP_ACOS: w[o[0]] = math.acos(w[i[0]])
            elif op == OP_ATAN: w[o[0]] = math.atan(w[i[0]])
            elif op == OP_TANH: w[o[0]] = math.tanh(w[i[0]])
            elif op == OP_SINH: w[o[0]] = math.sinh(w[i[0]])
            elif op == OP_COSH: w[o[0]] = math.cosh(w[i[0]])
            elif op == OP_ASINH: w[o[0]] = math.asinh(w[i[0]])
            elif op == OP_ACOSH: w[o[0]] = math.acosh(w[i[0]])
            elif op == OP_ATANH: w[o[0]] = math.atanh(w[i[0]])
            elif op == OP_LOG1P: w[o[0]] = math.log1p(w[i[0]])
            elif op == OP_EXPM1: w[o[0]] = math.expm1(w[i[0]])
            elif op == OP_SIGN: w[o[0]] = float((w[i[0]] > 0) - (w[i[0]] < 0))
            elif op == OP_FMIN: w[o[0]] = min(w[i[0]], w[i[1]])
            elif op == OP_FMAX: w[o[0]] = max(w[i[0]], w[i[1]])
            elif op == OP_LT: w[o[0]] = float(w[i[0]] < w[i[1]])
            elif op == OP_LE: w[o[0]] = float(w[i[0]] <= w[i[1]])
            elif op == OP_EQ: w[o[0]] = float(w[i[0]] == w[i[1]])
            elif op == OP_NE: w[o[0]] = float(w[i[0]] != w[i[1]])
            elif op == OP_NOT: w[o[0]] = float(w[i[0]] == 0.0)
            elif op == OP_AND: w[o[0]] = float(w[i[0]] != 0.0 and w[i[1]] != 0.0)
            elif op == OP_OR: w[o[0]] = float(w[i[0]] != 0.0 or w[i[1]] != 0.0)
            elif op == OP_IF_ELSE_ZERO: w[o[0]] = w[i[1]] if w[i[0]] != 0.0 else 0.0
            elif op == OP_FABS: w[o[0]] = abs(w[i[0]])
            else: raise ValueError(op)
        return outs


class DM(np.ndarray):
    def __new__(cls, a):
        return np.asarray(a, dtype=np.float64).view(cls)

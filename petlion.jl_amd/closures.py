"""Input closures `I = (t, Y, YP, p) -> ...` (reference input_methods.jl:159-176, scalar_residual.jl:169-170) for the device.

A closure cannot cross the C ABI, its expression can: `trace(f, p)` calls the Python function ONCE with tracer objects for t, Y, YP and p and records the arithmetic as the
postfix program of include/petlion_hip.h (PLH_VAL_EXPR / PLH_OP_*), the way the reference traces the same closure with Symbolics to differentiate it
(scalar_residual.jl:248-274).  Inside the closure use Python arithmetic, comparisons combined through `where(cond, a, b)` (= Julia's ifelse), and the functions of this
module or of numpy (np.sin(x) dispatches to x.sin()): sin cos exp log sqrt tanh abs minimum maximum.  `Y[i]` / `YP[i]` index the state vector (0-based; p.ind gives the
sections, negative indices count from the end), `p.θ[:key]` reads a model parameter of the CELL being integrated (per-cell in an ensemble), `calc_V(Y, p)`, `calc_I(Y, p)` are
provided as in the reference.  A closure that branches on a traced value (`if t < 100:`) cannot be traced and raises TraceError -- pass a table for those (reference
`tdiscon` semantics), or write the branch with `where`.

Accepted signatures, as redefine_func (scalar_residual.jl:228-246): f(t), f(t, p), f(t, Y, p), f(t, Y, YP, p)."""
import inspect
import math

import numpy as np

OPS = dict(CONST=0, T=1, Y=2, YP=3, THETA=4, ADD=5, SUB=6, MUL=7, DIV=8, NEG=9, SIN=10, COS=11, EXP=12, LOG=13, SQRT=14, POW=15, ABS=16, MIN=17, MAX=18,
           LT=19, LE=20, GT=21, GE=22, SELECT=23, TANH=24)
STACK = 16


class TraceError(TypeError):
    pass


class X:
    """a traced value: a tuple-tree (op, operand, children)"""
    __array_priority__ = 1000

    def __init__(self, op, arg=0.0, kids=()):
        self.op, self.arg, self.kids = op, float(arg), tuple(kids)

    @staticmethod
    def lift(v):
        if isinstance(v, X):
            return v
        if isinstance(v, (bool, np.bool_)):
            return X("CONST", 1.0 if v else 0.0)
        return X("CONST", float(v))

    def _bin(self, op, other, swap=False):
        a, b = (X.lift(other), self) if swap else (self, X.lift(other))
        return X(op, 0.0, (a, b))

    def __add__(self, o): return self._bin("ADD", o)
    def __radd__(self, o): return self._bin("ADD", o, True)
    def __sub__(self, o): return self._bin("SUB", o)
    def __rsub__(self, o): return self._bin("SUB", o, True)
    def __mul__(self, o): return self._bin("MUL", o)
    def __rmul__(self, o): return self._bin("MUL", o, True)
    def __truediv__(self, o): return self._bin("DIV", o)
    def __rtruediv__(self, o): return self._bin("DIV", o, True)
    def __pow__(self, o): return self._bin("POW", o)
    def __rpow__(self, o): return self._bin("POW", o, True)
    def __neg__(self): return X("NEG", 0.0, (self,))
    def __pos__(self): return self
    def __abs__(self): return X("ABS", 0.0, (self,))
    def __lt__(self, o): return self._bin("LT", o)
    def __le__(self, o): return self._bin("LE", o)
    def __gt__(self, o): return self._bin("GT", o)
    def __ge__(self, o): return self._bin("GE", o)
    # == / != as (a <= b) * (a >= b) and 1 - that (no EQ opcode is needed); without these Python falls back to identity comparison and the closure would be traced
    # with a CONSTANT branch
    def __eq__(self, o): return self._bin("LE", o)._bin("MUL", self._bin("GE", o))
    def __ne__(self, o): return X.lift(1.0)._bin("SUB", self.__eq__(o))
    __hash__ = object.__hash__

    def __bool__(self):
        raise TraceError("the closure branches on a traced value (if / and / or / min() / max() on t, Y, ...): write the branch as where(cond, a, b), or pass the input as a table")

    def __float__(self):
        raise TraceError("the closure converts a traced value to float (math.sin(t)?): use the functions of petlion.jl_amd.closures or numpy (np.sin)")

    # numpy ufuncs on an object call these methods: np.sin(x) -> x.sin()
    def sin(self): return X("SIN", 0.0, (self,))
    def cos(self): return X("COS", 0.0, (self,))
    def exp(self): return X("EXP", 0.0, (self,))
    def log(self): return X("LOG", 0.0, (self,))
    def sqrt(self): return X("SQRT", 0.0, (self,))
    def tanh(self): return X("TANH", 0.0, (self,))


def _un(op, pyf):
    def f(x):
        return X(op, 0.0, (x,)) if isinstance(x, X) else pyf(x)
    return f


sin, cos, exp, log, sqrt, tanh = (_un("SIN", math.sin), _un("COS", math.cos), _un("EXP", math.exp), _un("LOG", math.log), _un("SQRT", math.sqrt), _un("TANH", math.tanh))


def minimum(a, b):
    return X.lift(a)._bin("MIN", b) if isinstance(a, X) or isinstance(b, X) else min(a, b)


def maximum(a, b):
    return X.lift(a)._bin("MAX", b) if isinstance(a, X) or isinstance(b, X) else max(a, b)


def where(c, a, b):
    """ifelse(c, a, b)"""
    if not any(isinstance(v, X) for v in (c, a, b)):
        return a if c else b
    return X("SELECT", 0.0, (X.lift(c), X.lift(a), X.lift(b)))


class _StateVec:
    def __init__(self, op, n):
        self.op, self.n = op, n

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(self.n))]
        i = int(i)
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return X(self.op, i)


class _Theta:
    def __init__(self, p):
        self.p = p

    def __getitem__(self, k):
        k = k.lstrip(":") if isinstance(k, str) else k
        if k in self.p.θ_keys:
            return X("THETA", self.p.θ_keys.index(k))
        if k == "I1C":
            raise TraceError("p.θ[:I1C] is derived per cell on the device and is not a theta entry: express the input in C-rates")
        if k in self.p.θ:
            return float(self.p.θ[k])                    # a parameter the model does not read (user-added): the handle's value, a constant of the program
        raise KeyError(k)


class _P:
    """what a closure may read of the model: p.θ[key], p.ind, p.N"""
    def __init__(self, p):
        self.θ = self.theta = _Theta(p)
        self.ind, self.N = p.ind, p.N


def calc_V(Y, p):
    """calc_V, scalar_residual.jl:86: Phi_s[1] - Phi_s[end]"""
    sl = p.ind["Φ_s"]
    return Y[sl.start] - Y[sl.stop - 1]


def calc_I(Y, p):
    return Y[len(Y) - 1]


def compile_tree(x):
    """postfix program (opcodes, operands) of a traced value; checks the stack depth of the device interpreter"""
    ops, args = [], []

    def emit(n):
        for k in n.kids:
            emit(k)
        ops.append(float(OPS[n.op])); args.append(n.arg)
    emit(X.lift(x))
    sp = deepest = 0
    for o in ops:
        o = int(o)
        pop = 0 if o <= OPS["THETA"] else 3 if o == OPS["SELECT"] else 1 if o in (OPS["NEG"], OPS["SIN"], OPS["COS"], OPS["EXP"], OPS["LOG"], OPS["SQRT"], OPS["ABS"], OPS["TANH"]) else 2
        sp += 1 - pop; deepest = max(deepest, sp)
    if deepest > STACK:
        raise TraceError("the closure needs %d stack slots, the device interpreter has %d: simplify the expression" % (deepest, STACK))
    return np.array(ops, dtype=np.float64), np.array(args, dtype=np.float64)


def _is_const(x, v=None):
    return x.op == "CONST" and (v is None or x.arg == v)


def _add(a, b):
    if a is None: return b
    if b is None: return a
    return a._bin("ADD", b)


def _sub(a, b):
    if b is None: return a
    if a is None: return X("NEG", 0.0, (b,))
    return a._bin("SUB", b)


def _mul(a, b):
    """a * b with a or b possibly None (= 0) and the unit factor folded"""
    if a is None or b is None or _is_const(a, 0.0) or _is_const(b, 0.0): return None
    if _is_const(a, 1.0): return b
    if _is_const(b, 1.0): return a
    return a._bin("MUL", b)


def diff(x, kind, idx):
    """d x / d Y[idx] (kind = "Y") or d x / d YP[idx] (kind = "YP") of a traced value as another traced value; None = identically zero.  The rules are the ones Symbolics
    applies to the same closure in the reference (scalar_residual.jl:289-291, sparsejacobian of the traced control row): ifelse / min / max / abs differentiate branch-wise,
    comparisons are constants."""
    one = X("CONST", 1.0)
    op, k = x.op, x.kids
    if op in ("CONST", "T", "THETA"): return None
    if op in ("Y", "YP"): return one if (op == kind and int(x.arg) == idx) else None
    if op in ("LT", "LE", "GT", "GE"): return None
    d = [diff(c, kind, idx) for c in k]
    if all(v is None for v in d): return None
    if op == "ADD": return _add(d[0], d[1])
    if op == "SUB": return _sub(d[0], d[1])
    if op == "NEG": return X("NEG", 0.0, (d[0],))
    if op == "MUL": return _add(_mul(d[0], k[1]), _mul(k[0], d[1]))
    if op == "DIV":                                                          # a' / b - (a / b) b' / b
        t1 = None if d[0] is None else d[0]._bin("DIV", k[1])
        t2 = None if d[1] is None else _mul(x, d[1])._bin("DIV", k[1])
        return _sub(t1, t2)
    if op == "SIN": return _mul(X("COS", 0.0, (k[0],)), d[0])
    if op == "COS": return X("NEG", 0.0, (_mul(X("SIN", 0.0, (k[0],)), d[0]),))
    if op == "EXP": return _mul(x, d[0])
    if op == "LOG": return d[0]._bin("DIV", k[0])
    if op == "SQRT": return d[0]._bin("DIV", X("CONST", 2.0)._bin("MUL", x))
    if op == "TANH": return _mul(one._bin("SUB", x._bin("MUL", x)), d[0])
    if op == "ABS": return X("SELECT", 0.0, (k[0]._bin("GE", 0.0), d[0], X("NEG", 0.0, (d[0],))))
    if op == "POW":
        out = None
        if d[0] is not None:                                                 # b a^(b-1) a'
            bm1 = X("CONST", k[1].arg - 1.0) if _is_const(k[1]) else k[1]._bin("SUB", 1.0)
            out = _mul(_mul(k[1], k[0]._bin("POW", bm1)), d[0])
        if d[1] is not None:                                                 # a^b log(a) b'
            out = _add(out, _mul(_mul(x, X("LOG", 0.0, (k[0],))), d[1]))
        return out
    z = lambda v: X("CONST", 0.0) if v is None else v
    if op == "MIN": return X("SELECT", 0.0, (k[0]._bin("LT", k[1]), z(d[0]), z(d[1])))
    if op == "MAX": return X("SELECT", 0.0, (k[0]._bin("GT", k[1]), z(d[0]), z(d[1])))
    if op == "SELECT": return X("SELECT", 0.0, (k[0], z(d[1]), z(d[2]))) if (d[1] is not None or d[2] is not None) else None
    raise TraceError("no derivative rule for " + op)


def _refs(x, kind, acc):
    if x.op == kind:
        acc.add(int(x.arg))
    for c in x.kids:
        _refs(c, kind, acc)
    return acc


def row_derivatives(x, n_tot=None, n_diff=None):
    """what the reference's differentiate_residual_func (scalar_residual.jl:276-416) builds for a closure of the state: the entries of the control row, as
    (columns, [program per column]).  Column c < n_tot: d f / d Y[c]; column n_tot + i: d f / d YP[i] for a DIFFERENTIAL state i (the device multiplies it by cj in the
    integration row and, in the consistent-initialisation row, chains it through the differential equation of state i, as the reference substitutes YP -> rhs there,
    scalar_residual.jl:335-362).  None -- the reference's no-differentiation fallback (scalar_residual.jl:248-274) -- when the closure reads no state, reads YP of an
    ALGEBRAIC state, or a derivative does not fit the device interpreter's stack."""
    x = X.lift(x)
    cols = sorted(_refs(x, "Y", set()))
    pcols = sorted(_refs(x, "YP", set()))
    if pcols and (n_tot is None or n_diff is None or pcols[-1] >= n_diff):
        return None
    out_c, out_p = [], []
    for kind, cc in (("Y", cols), ("YP", pcols)):
        for c in cc:
            d = diff(x, kind, c)
            if d is None:
                continue
            try:
                out_p.append(compile_tree(d))
            except TraceError:
                return None
            out_c.append(c if kind == "Y" else n_tot + c)
    return (out_c, out_p) if out_c else None


def trace(f, p, with_tree=False):
    """postfix program of the input closure f for model p"""
    n = len(inspect.signature(f).parameters)
    t, Y, YP, P = X("T"), _StateVec("Y", p.N.tot), _StateVec("YP", p.N.tot), _P(p)
    if n == 1:
        out = f(t)
    elif n == 2:
        out = f(t, P)
    elif n == 3:
        out = f(t, Y, P)
    elif n == 4:
        out = f(t, Y, YP, P)
    else:
        raise TraceError("Input function must have one to four arguments (t, Y, YP, p)")
    return (compile_tree(out), X.lift(out)) if with_tree else compile_tree(out)


def evaluate(prog, t, Y=None, YP=None, theta=None):
    """host-side evaluation of a program (tests, plotting)"""
    ops, args = prog
    st = []
    for o, a in zip(ops.astype(int), args):
        nm = [k for k, v in OPS.items() if v == o][0]
        if nm == "CONST": st.append(a)
        elif nm == "T": st.append(t)
        elif nm == "Y": st.append(Y[int(a)])
        elif nm == "YP": st.append(YP[int(a)])
        elif nm == "THETA": st.append(theta[int(a)])
        elif nm == "SELECT":
            b, x, c = st.pop(), st.pop(), st.pop(); st.append(x if c != 0 else b)
        elif nm in ("NEG", "SIN", "COS", "EXP", "LOG", "SQRT", "ABS", "TANH"):
            x = st.pop()
            st.append({"NEG": lambda v: -v, "SIN": math.sin, "COS": math.cos, "EXP": math.exp, "LOG": math.log, "SQRT": math.sqrt, "ABS": abs, "TANH": math.tanh}[nm](x))
        else:
            y, x = st.pop(), st.pop()
            st.append({"ADD": x + y, "SUB": x - y, "MUL": x * y, "DIV": x / y if nm == "DIV" else 0, "POW": x ** y if nm == "POW" else 0, "MIN": min(x, y), "MAX": max(x, y),
                       "LT": float(x < y), "LE": float(x <= y), "GT": float(x > y), "GE": float(x >= y)}[nm])
    assert len(st) == 1
    return st[0]

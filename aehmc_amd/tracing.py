"""A Python ``logprob_fn`` as the reference takes it (README.md:27-36, aehmc/hmc.py:16-40: any function of the
position): traced ONCE with proxy objects and emitted as the ``aehmc_logp`` template that ``targets.Custom`` /
``targets.CustomJoint`` compile with hipRTC and the engine differentiates (csrc/dual.cuh).

    logprob_fn = lambda y: (-0.5 * y**2 - 0.5 * np.log(2 * np.pi)).sum()       # coordinate-wise  -> targets.Custom
    logprob_fn = lambda y: -0.5 * (y - loc) @ (P @ (y - loc))                  # joint            -> targets.CustomJoint
    kernel = nuts.new_kernel(srng, logprob_fn); state = nuts.new_state(q, logprob_fn)

What a traced function may do with its argument (a scalar for a scalar position, else a vector of ``dim`` entries):
``+ - * / **`` and unary ``-`` with numbers, numpy arrays (captured as device parameter arrays) and other traced values;
numpy ufuncs ``exp exp2 log log2 log10 log1p expm1 sqrt sin cos tanh sinh cosh arctan abs square power reciprocal negative
maximum minimum logaddexp``, ``scipy.special.erf``, ``erfc``, ``expit`` and ``gammaln``; ``softplus``, ``logsumexp`` (up to 64 terms) and ``where`` from this module; comparisons (inside ``where`` only);
``.sum()`` / ``np.sum`` / ``.mean()`` / ``var`` / ``std``, ``@`` / ``np.dot`` (vector . vector, constant matrix @ vector, vector @ constant
matrix), ``np.linalg.norm``, ``np.diff``, ``np.clip``, ``np.sign``, ``max`` / ``min`` / ``np.concatenate`` / ``np.stack`` of up to 64 entries,
``np.logaddexp.reduce``, ``.T`` / ``.reshape(-1)`` / ``.copy()`` of a vector; indexing and slicing with static bounds, gathers through a constant integer array (``theta[group]``);
iteration over a vector.  Anything else -- Python ``if`` on a traced
value, ``float()``, ``math.exp``, fancy indexing -- raises ``TypeError`` at trace time and says what it was.
"""
from __future__ import annotations

import math
import numbers

import numpy as np

__all__ = ["trace", "where", "softplus", "logsumexp", "TraceError"]


class TraceError(TypeError):
    pass


# ------------------------------------------------------------------------------------------------------ indices
class Idx:
    """An index: const + sum coef * loop variable + sum coef * LOOKUP(k, inner index) -- the last form is a gather through
    a constant integer array (parameter array k, stored as doubles): theta[group] with `group` a captured numpy array."""

    __slots__ = ("c", "terms", "ind")

    def __init__(self, c=0, terms=(), ind=()):
        self.c, self.terms = int(c), tuple(sorted((v, k) for v, k in terms if k != 0))
        self.ind = tuple((int(a), int(k), i) for a, k, i in ind if a != 0)

    @staticmethod
    def var(v):
        return Idx(0, ((v, 1),))

    def __add__(self, o):
        o = o if isinstance(o, Idx) else Idx(o)
        d = dict(self.terms)
        for v, k in o.terms:
            d[v] = d.get(v, 0) + k
        return Idx(self.c + o.c, d.items(), self.ind + o.ind)

    def __mul__(self, k):
        return Idx(self.c * k, ((v, c * k) for v, c in self.terms), ((a * k, p, i) for a, p, i in self.ind))

    def key(self):
        return (self.c, self.terms, tuple((a, k, i.key()) for a, k, i in self.ind))

    def is_var(self, v):
        return self.c == 0 and self.terms == ((v, 1),) and not self.ind

    def vars(self):
        out = frozenset(v for v, _ in self.terms)
        for _, _, i in self.ind:
            out |= i.vars()
        return out

    def code(self, names, par=None):
        par = par or (lambda k: f"prm[{k}]")
        parts = [f"{names[v]}" if k == 1 else f"{k} * {names[v]}" for v, k in self.terms]
        for a, k, i in self.ind:
            look = f"(int){par(k)}[{i.code(names, par)}]"
            parts.append(look if a == 1 else f"{a} * {look}")
        if self.c or not parts:
            parts.append(str(self.c))
        return " + ".join(parts)


# ------------------------------------------------------------------------------------------------------ scalars
_UNARY = {"exp": "exp", "log": "log", "log1p": "log1p", "expm1": "expm1", "sqrt": "sqrt", "sin": "sin", "cos": "cos",
          "tanh": "tanh", "absolute": "fabs", "fabs": "fabs", "erf": "erf", "softplus": "softplus", "gammaln": "lgamma", "lgamma": "lgamma", "expit": "logistic",
          "logistic": "logistic", "arctan": "atan", "sinh": "sinh", "cosh": "cosh", "erfc": "erfc"}
_CMP = {"less": "<", "greater": ">", "less_equal": "<=", "greater_equal": ">=", "equal": "==", "not_equal": "!="}


def _lit(x):
    x = float(x)
    if x != x:
        return "NAN"
    if x in (float("inf"), float("-inf")):
        return "INFINITY" if x > 0 else "(-INFINITY)"
    r = repr(x)
    return r if ("." in r or "e" in r or "n" in r) else r + ".0"


class S:
    """A traced scalar: a node of the expression tree.  ``t``: depends on the position (its C++ type is the template's
    arithmetic type; otherwise plain double).  ``b``: a comparison (usable in ``where`` only)."""

    __array_priority__ = 1000
    __slots__ = ("op", "args", "t", "b", "ctx")

    def __init__(self, ctx, op, args, t, b=False):
        self.ctx, self.op, self.args, self.t, self.b = ctx, op, args, t, b

    # -- construction helpers
    def _lift(self, x):
        return _lift(self.ctx, x)

    def _bin(self, op, a, b):
        a, b = self._lift(a), self._lift(b)
        if isinstance(a, V) or isinstance(b, V):
            return V._ew2(self.ctx, lambda x, y: x._bin(op, x, y), a, b)
        if a.b or b.b:
            raise TraceError("arithmetic on a comparison result: use where(cond, a, b)")
        if a.op == "const" and b.op == "const":  # folded in Python: what numpy itself would have computed
            x, y = a.args[0], b.args[0]
            with np.errstate(all="ignore"):
                return S(self.ctx, "const", (float({"+": np.add, "-": np.subtract, "*": np.multiply, "/": np.divide}[op](x, y)),), False)
        return S(self.ctx, "bin", (op, a, b), a.t or b.t)

    def __add__(self, o): return self._bin("+", self, o)
    def __radd__(self, o): return self._bin("+", o, self)
    def __sub__(self, o): return self._bin("-", self, o)
    def __rsub__(self, o): return self._bin("-", o, self)
    def __mul__(self, o): return self._bin("*", self, o)
    def __rmul__(self, o): return self._bin("*", o, self)
    def __truediv__(self, o): return self._bin("/", self, o)
    def __rtruediv__(self, o): return self._bin("/", o, self)

    def __neg__(self):
        if self.op == "const":
            return S(self.ctx, "const", (-self.args[0],), False)
        return S(self.ctx, "neg", (self,), self.t)

    def __pos__(self): return self

    def __pow__(self, o):
        return _pow(self.ctx, self, o)

    def __rpow__(self, o):
        return _pow(self.ctx, o, self)

    def __abs__(self): return _unary(self.ctx, "fabs", self)

    def _cmp(self, name, o):
        o = self._lift(o)
        if isinstance(o, V):
            return V._ew2(self.ctx, lambda x, y: x._cmp(name, y), self, o)
        return S(self.ctx, "cmp", (_CMP[name], self, o), self.t or o.t, b=True)

    def __lt__(self, o): return self._cmp("less", o)
    def __gt__(self, o): return self._cmp("greater", o)
    def __le__(self, o): return self._cmp("less_equal", o)
    def __ge__(self, o): return self._cmp("greater_equal", o)
    __hash__ = None

    def __eq__(self, o): return self._cmp("equal", o)
    def __ne__(self, o): return self._cmp("not_equal", o)

    def __bool__(self):
        raise TraceError("the truth value of a traced quantity was asked for (an `if`, `and`, `or`, `max()` ... on a value that "
                         "depends on the position): Python control flow cannot be traced -- use aehmc_amd.tracing.where(cond, a, b)")

    def __float__(self):
        raise TraceError("float() of a traced quantity (math.exp / math.log / float(...) on a value that depends on the "
                         "position): use the numpy functions (np.exp, np.log, ...), which trace")

    __int__ = __index__ = __float__

    def sum(self): return self
    def mean(self): return self

    @property
    def shape(self): return ()

    @property
    def ndim(self): return 0

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        return _ufunc(self.ctx, ufunc, method, inputs, kw)

    def __array_function__(self, func, types, args, kwargs):
        return _array_function(self.ctx, func, args, kwargs)


def _const(ctx, x):
    return S(ctx, "const", (float(x),), False)


def _lift(ctx, x):
    if isinstance(x, (S, V, M)):
        return x
    if isinstance(x, (numbers.Real, np.floating, np.integer, np.bool_)):
        return _const(ctx, x)
    if isinstance(x, (list, tuple)) or (isinstance(x, np.ndarray) and x.dtype == object and x.ndim == 1):
        items = [_lift(ctx, y) for y in x]  # a Python list of traced scalars / numbers: a vector with static entries
        if any(not isinstance(y, S) for y in items):
            raise TraceError("a list used as a vector must hold scalars")
        if all(y.op == "const" for y in items):
            return _lift(ctx, np.array([y.args[0] for y in items]))

        def at(i, items=items):
            if i.terms:
                raise TraceError("internal: a vector built from a Python list was indexed by a loop variable")
            return items[i.c]
        return V(ctx, len(items), at, unroll=True)
    a = np.asarray(x)
    if a.dtype == object or not (np.issubdtype(a.dtype, np.number) or a.dtype == bool):
        raise TraceError(f"cannot use a {type(x).__name__} in a traced logprob_fn (numbers, numpy arrays and traced values only)")
    if a.ndim == 0:
        return _const(ctx, a.item())
    if a.ndim == 1:
        k = ctx.param(a)
        return V(ctx, a.shape[0], lambda i, k=k: S(ctx, "par", (k, i), False))
    if a.ndim == 2:
        return M(ctx, a)
    raise TraceError(f"arrays of {a.ndim} dimensions are not supported in a traced logprob_fn")


def _unary(ctx, name, x):
    x = _lift(ctx, x)
    if isinstance(x, V):
        return V(ctx, x.n, lambda i: _unary(ctx, name, x.at(i)), x.unroll)
    if isinstance(x, M):
        raise TraceError(f"{name} of a matrix is not supported")
    if x.b:
        raise TraceError(f"{name} of a comparison result")
    if x.op == "const":
        with np.errstate(all="ignore"):
            f = {"fabs": np.fabs, "erf": _erf_np, "softplus": lambda z: np.logaddexp(0.0, z), "logistic": lambda z: 1.0 / (1.0 + np.exp(-z)),
                 "atan": np.arctan, "erfc": lambda z: math.erfc(float(z)),
                 "lgamma": lambda z: __import__("math").lgamma(float(z))}.get(name) or getattr(np, name)
            return _const(ctx, f(x.args[0]))
    return S(ctx, "un", (name, x), x.t)


def _erf_np(z):
    import math
    return math.erf(float(z))


def _pow(ctx, a, b):
    a, b = _lift(ctx, a), _lift(ctx, b)
    if isinstance(a, V) or isinstance(b, V):
        return V._ew2(ctx, lambda x, y: _pow(ctx, x, y), a, b)
    if b.op == "const":
        p = b.args[0]
        if p == 2.0:
            return S(ctx, "un", ("square", a), a.t) if a.op != "const" else _const(ctx, a.args[0] ** 2)
        if p == 1.0:
            return a
        if p == 0.5:
            return _unary(ctx, "sqrt", a)
        if p == -1.0:
            return _const(ctx, 1.0) / a
        if p == 3.0:
            return a * S(ctx, "un", ("square", a), a.t) if a.op != "const" else _const(ctx, a.args[0] ** 3)
    if a.op == "const" and b.op == "const":
        return _const(ctx, a.args[0] ** b.args[0])
    if a.op == "const" and a.args[0] > 0:  # c ** x = exp(x log c)
        return _unary(ctx, "exp", b * float(np.log(a.args[0])))
    return S(ctx, "pow", (a, b), a.t or b.t)


def where(cond, a, b):
    """``a`` where ``cond`` holds, else ``b`` (the traceable form of an ``if`` on a value that depends on the position)."""
    ctx = next((x.ctx for x in (cond, a, b) if isinstance(x, (S, V))), None)
    if ctx is None:
        return np.where(cond, a, b)
    cond, a, b = _lift(ctx, cond), _lift(ctx, a), _lift(ctx, b)
    if any(isinstance(x, V) for x in (cond, a, b)):
        n = next(x.n for x in (cond, a, b) if isinstance(x, V))
        at = lambda x, i: x.at(i) if isinstance(x, V) else x
        for x in (cond, a, b):
            if isinstance(x, V) and x.n != n:
                raise TraceError(f"where: vectors of lengths {n} and {x.n}")
        return V(ctx, n, lambda i: where(at(cond, i), at(a, i), at(b, i)), any(isinstance(x, V) and x.unroll for x in (cond, a, b)))
    if cond.op == "const":
        return a if cond.args[0] else b
    if not cond.b:
        raise TraceError("where: the condition must be a comparison")
    return S(ctx, "where", (cond, a, b), cond.t or a.t or b.t)


def logsumexp(x):
    """log(sum(exp(x))) of up to 64 traced terms (a list of scalars or of equally long vectors -- elementwise then --, or the
    entries of one traced vector): the mixture-model building block, m + log(sum(exp(x_k - m))) with m the largest term."""
    if isinstance(x, V):
        items = [x.at(k) for k in range(x.n)]
    else:
        items = list(x)
    ctx = next((y.ctx for y in items if isinstance(y, (S, V))), None)  # (items may be vectors: elementwise over them)
    if ctx is None:
        a = np.asarray(items, dtype=np.float64)  # (plain numbers / arrays: the same function over axis 0)
        m = a.max(axis=0)
        return m + np.log(np.exp(a - m).sum(axis=0))
    if not 1 <= len(items) <= 64:
        raise TraceError(f"logsumexp over {len(items)} terms: 1 ... 64 are supported")
    items = [_lift(ctx, y) for y in items]
    m = items[0]
    for y in items[1:]:
        m = where(y > m, y, m)
    tot = _unary(ctx, "exp", items[0] - m)
    for y in items[1:]:
        tot = tot + _unary(ctx, "exp", y - m)
    return m + _unary(ctx, "log", tot)


def softplus(x):
    """log(1 + exp(x)) without overflow (the logistic log-likelihood's building block)."""
    if isinstance(x, (S, V)):
        return _unary(x.ctx, "softplus", x)
    return np.logaddexp(0.0, x)


# ------------------------------------------------------------------------------------------------------ vectors
class V:
    """A traced vector of static length: a function from an index to a traced scalar (nothing is materialised)."""

    __array_priority__ = 1000
    __slots__ = ("ctx", "n", "_at", "unroll", "_memo")

    def __init__(self, ctx, n, at, unroll=False):
        # unroll: built from a Python list of traced scalars (np.array([a, b, c])): its entries exist at static indices
        # only, so reductions over it are written out term by term instead of as a loop
        self.ctx, self.n, self._at, self.unroll = ctx, int(n), at, bool(unroll)
        self._memo = {}

    def at(self, i):
        # the entry at an index is ONE node however often it is asked for (z = X @ q used twice in a row's term: its
        # value is computed once in the emitted code)
        i = i if isinstance(i, Idx) else Idx(i)
        k = i.key()
        if k not in self._memo:
            self._memo[k] = self._at(i)
        return self._memo[k]

    @staticmethod
    def _ew2(ctx, f, a, b):
        a, b = _lift(ctx, a), _lift(ctx, b)
        if isinstance(a, M) or isinstance(b, M):
            raise TraceError("elementwise arithmetic with a matrix is not supported (use M @ v)")
        n = a.n if isinstance(a, V) else b.n
        if isinstance(a, V) and isinstance(b, V) and a.n != b.n:
            if a.n == 1:
                a = a.at(0)
            elif b.n == 1:
                b = b.at(0)
            else:
                raise TraceError(f"operands of lengths {a.n} and {b.n} do not broadcast")
            n = a.n if isinstance(a, V) else b.n
        un = (isinstance(a, V) and a.unroll) or (isinstance(b, V) and b.unroll)
        return V(ctx, n, lambda i: f(a.at(i) if isinstance(a, V) else a, b.at(i) if isinstance(b, V) else b), un)

    def _bin(self, op, a, b):
        return V._ew2(self.ctx, lambda x, y: x._bin(op, x, y), a, b)

    def __add__(self, o): return self._bin("+", self, o)
    def __radd__(self, o): return self._bin("+", o, self)
    def __sub__(self, o): return self._bin("-", self, o)
    def __rsub__(self, o): return self._bin("-", o, self)
    def __mul__(self, o): return self._bin("*", self, o)
    def __rmul__(self, o): return self._bin("*", o, self)
    def __truediv__(self, o): return self._bin("/", self, o)
    def __rtruediv__(self, o): return self._bin("/", o, self)
    def __neg__(self): return V(self.ctx, self.n, lambda i: -self.at(i), self.unroll)
    def __pos__(self): return self
    def __abs__(self): return _unary(self.ctx, "fabs", self)
    def __pow__(self, o): return _pow(self.ctx, self, o)
    def __rpow__(self, o): return _pow(self.ctx, o, self)

    def _cmp(self, name, o):
        return V._ew2(self.ctx, lambda x, y: x._cmp(name, y), self, o)

    def __lt__(self, o): return self._cmp("less", o)
    def __gt__(self, o): return self._cmp("greater", o)
    def __le__(self, o): return self._cmp("less_equal", o)
    def __ge__(self, o): return self._cmp("greater_equal", o)
    __hash__ = None

    def __eq__(self, o): return self._cmp("equal", o)
    def __ne__(self, o): return self._cmp("not_equal", o)

    def __bool__(self):
        raise TraceError("the truth value of a traced vector was asked for: Python control flow cannot be traced -- use "
                         "aehmc_amd.tracing.where(cond, a, b)")

    def __len__(self): return self.n

    @property
    def shape(self): return (self.n,)

    @property
    def ndim(self): return 1

    @property
    def size(self): return self.n

    def __iter__(self):
        return (self.at(k) for k in range(self.n))

    def __getitem__(self, k):
        if isinstance(k, (int, np.integer)):
            k = int(k)
            if not -self.n <= k < self.n:
                raise IndexError(f"index {k} out of range for a vector of {self.n}")
            return self.at(k % self.n)
        if isinstance(k, slice):
            start, stop, step = k.indices(self.n)
            m = len(range(start, stop, step))
            return V(self.ctx, m, lambda i: self.at(i * step + start), self.unroll)
        if k is Ellipsis:
            return self
        if isinstance(k, (np.ndarray, list)) and not self.unroll:
            ia = np.asarray(k)
            if ia.ndim == 1 and ia.dtype.kind in "iu" and ia.size:
                if ia.min() < -self.n or ia.max() >= self.n:
                    raise IndexError(f"index array out of range for a vector of {self.n}")
                kk = self.ctx.param((ia % self.n).astype(np.float64))  # a gather through a constant index array: theta[group]
                return V(self.ctx, ia.size, lambda i: self.at(Idx(0, (), ((1, kk, i),))))
        raise TraceError(f"indexing a traced vector with {type(k).__name__}: integers, slices with static bounds and constant "
                         "integer arrays (gathers) are supported")

    def sum(self, axis=None):
        if self.n == 0:
            return _const(self.ctx, 0.0)
        if self.unroll:
            if self.n > 64:
                raise TraceError(f"a reduction over a Python list of {self.n} traced scalars: build long vectors from the position by "
                                 "slicing and arithmetic instead")
            out = self.at(0)
            for k in range(1, self.n):
                out = out + self.at(k)
            return out
        v = self.ctx.new_var()
        body = self.at(Idx.var(v))
        if body.b:
            raise TraceError("sum of comparison results")
        return S(self.ctx, "sum", (v, self.n, body), body.t)

    def mean(self, axis=None):
        return self.sum() / float(self.n)

    def var(self, axis=None, ddof=0):
        d = self - self.mean()
        return (d * d).sum() / float(self.n - ddof)

    def std(self, axis=None, ddof=0):
        return _unary(self.ctx, "sqrt", self.var(ddof=ddof))

    def _extreme(self, big):  # max / min of a short vector, written out as nested selects (<= 64 entries)
        if self.n == 0 or self.n > 64:
            raise TraceError(f"max / min over {self.n} traced entries: supported for 1 ... 64 (use tracing.logsumexp or where for a "
                             "smooth or elementwise form)")
        out = self.at(0)
        for k in range(1, self.n):
            x = self.at(k)
            out = where(x > out if big else x < out, x, out)
        return out

    def max(self, axis=None): return self._extreme(True)
    def min(self, axis=None): return self._extreme(False)

    @property
    def T(self): return self

    def copy(self): return self
    def ravel(self): return self
    def flatten(self): return self

    def astype(self, dtype, **kw):
        if np.dtype(dtype).kind != "f":
            raise TraceError(f"astype({dtype}) of a traced vector")
        return self

    def reshape(self, *shape):
        shape = shape[0] if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else shape
        if tuple(shape) not in ((-1,), (self.n,)):
            raise TraceError(f"reshape{tuple(shape)} of a traced vector of {self.n}: traced values are scalars and vectors")
        return self

    def dot(self, o):
        return self @ o

    def __matmul__(self, o):
        o = _lift(self.ctx, o)
        if isinstance(o, V):
            if o.n != self.n:
                raise TraceError(f"dot of vectors of lengths {self.n} and {o.n}")
            return (self * o).sum()
        if isinstance(o, M):
            return o.rmatvec(self)
        raise TraceError("vector @ scalar")

    def __rmatmul__(self, o):
        o = _lift(self.ctx, o)
        if isinstance(o, M):
            return o.matvec(self)
        if isinstance(o, V):
            return o @ self
        raise TraceError("scalar @ vector")

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        return _ufunc(self.ctx, ufunc, method, inputs, kw)

    def __array_function__(self, func, types, args, kwargs):
        return _array_function(self.ctx, func, args, kwargs)


class M:
    """A constant matrix (a captured 2-D numpy array): only products with traced vectors."""

    def __init__(self, ctx, a):
        self.ctx, self.rows, self.cols = ctx, a.shape[0], a.shape[1]
        self.k = ctx.param(np.ascontiguousarray(a, dtype=np.float64).reshape(-1))

    def matvec(self, v):
        if v.n != self.cols:
            raise TraceError(f"matrix [{self.rows}, {self.cols}] @ vector of {v.n}")
        ctx, k, m = self.ctx, self.k, self.cols
        return V(ctx, self.rows, lambda i: V(ctx, m, lambda j: S(ctx, "par", (k, i * m + j), False) * v.at(j), v.unroll).sum())

    def rmatvec(self, v):
        if v.n != self.rows:
            raise TraceError(f"vector of {v.n} @ matrix [{self.rows}, {self.cols}]")
        ctx, k, m = self.ctx, self.k, self.cols
        return V(ctx, self.cols, lambda j: V(ctx, self.rows, lambda i: v.at(i) * S(ctx, "par", (k, i * m + j), False), v.unroll).sum())


def _ufunc(ctx, ufunc, method, inputs, kw):
    name = ufunc.__name__
    if method != "__call__" or kw.get("out") is not None:
        if method == "reduce" and name == "add" and len(inputs) == 1:
            return _lift(ctx, inputs[0]).sum()
        if method == "reduce" and name == "logaddexp" and len(inputs) == 1:
            return logsumexp(_lift(ctx, inputs[0]))
        if method == "reduce" and name in ("maximum", "minimum") and len(inputs) == 1:
            return _lift(ctx, inputs[0])._extreme(name == "maximum")
        raise TraceError(f"numpy.{name}.{method} is not supported in a traced logprob_fn")
    x = [_lift(ctx, i) for i in inputs]
    if name in _UNARY:
        return _unary(ctx, _UNARY[name], x[0])
    if name == "square":
        return _pow(ctx, x[0], 2.0)
    if name in ("log2", "log10"):
        return _unary(ctx, "log", x[0]) * (1.0 / math.log(2.0 if name == "log2" else 10.0))
    if name == "exp2":
        return _unary(ctx, "exp", x[0] * math.log(2.0))
    if name == "sign":  # (piecewise constant: no gradient flows through it)
        return where(x[0] > 0.0, 1.0, where(x[0] < 0.0, -1.0, 0.0))
    if name == "negative":
        return -x[0]
    if name == "positive":
        return x[0]
    if name == "reciprocal":
        return 1.0 / x[0]
    if name in ("add", "subtract", "multiply", "divide", "true_divide"):
        a, b = x
        return {"add": lambda: a + b, "subtract": lambda: a - b, "multiply": lambda: a * b}.get(name, lambda: a / b)()
    if name in ("power", "float_power"):
        return _pow(ctx, x[0], x[1])
    if name in _CMP:
        a, b = x
        return (a if isinstance(a, (S, V)) else _lift(ctx, a))._cmp(name, b)
    if name in ("maximum", "fmax"):
        return where(x[0] >= x[1], x[0], x[1])
    if name in ("minimum", "fmin"):
        return where(x[0] <= x[1], x[0], x[1])
    if name == "logaddexp":  # numpy's own formulation
        a, b = x
        d = a - b
        return where(d > 0, a + _unary(ctx, "log1p", _unary(ctx, "exp", -d)), b + _unary(ctx, "log1p", _unary(ctx, "exp", d)))
    if name == "matmul":
        a, b = x
        return a @ b if isinstance(a, V) else b.__rmatmul__(a)
    raise TraceError(f"numpy.{name} is not supported in a traced logprob_fn (supported: + - * / **, exp log log1p expm1 sqrt "
                     "log2 log10 exp2 sin cos tanh sinh cosh arctan abs square power reciprocal maximum minimum logaddexp, scipy.special.erf erfc "
                     "expit gammaln, sum, dot, where)")


def _array_function(ctx, func, args, kwargs):
    name = getattr(func, "__name__", str(func))
    if name == "sum" and len(args) == 1 and kwargs.get("axis") in (None, 0, -1):
        return _lift(ctx, args[0]).sum()
    if name == "mean" and len(args) == 1 and kwargs.get("axis") in (None, 0, -1):
        return _lift(ctx, args[0]).mean()
    if name in ("dot", "inner", "vdot", "matmul") and len(args) == 2:
        a, b = _lift(ctx, args[0]), _lift(ctx, args[1])
        if isinstance(a, M):
            return a.matvec(b)
        return a @ b
    if name == "where" and len(args) == 3:
        return where(*args)
    if name in ("var", "std") and len(args) == 1 and kwargs.get("axis") in (None, 0, -1):
        return getattr(_lift(ctx, args[0]), name)(ddof=kwargs.get("ddof", 0))
    if name in ("max", "amax", "min", "amin") and len(args) == 1 and kwargs.get("axis") in (None, 0, -1):
        return _lift(ctx, args[0])._extreme(name in ("max", "amax"))
    if name == "norm" and len(args) == 1 and kwargs.get("ord") in (None, 2) and kwargs.get("axis") in (None, 0, -1):
        x = _lift(ctx, args[0])
        return _unary(ctx, "sqrt", (x * x).sum() if isinstance(x, V) else x * x)
    if name == "diff" and len(args) == 1 and kwargs.get("n", 1) == 1 and kwargs.get("axis", -1) in (0, -1):
        x = _lift(ctx, args[0])
        return x[1:] - x[:-1]
    if name == "clip" and len(args) == 3 and kwargs.get("out") is None:
        x, lo, hi = args
        if lo is not None:
            x = where(x < lo, lo, x)
        if hi is not None:
            x = where(x > hi, hi, x)
        return x
    if name in ("zeros_like", "ones_like") and len(args) == 1:
        x = _lift(ctx, args[0])
        c = 0.0 if name == "zeros_like" else 1.0
        return np.full(x.n, c) if isinstance(x, V) else c
    if name in ("concatenate", "hstack", "stack") and len(args) == 1 and kwargs.get("axis", 0) in (0, -1):
        parts = [_lift(ctx, a) for a in args[0]]
        if name == "stack" and any(isinstance(a, V) for a in parts):
            raise TraceError("numpy.stack of traced vectors would be a matrix: traced values are scalars and vectors")
        items = [x for a in parts for x in (list(a) if isinstance(a, V) else [a])]
        if len(items) > 64:
            raise TraceError(f"numpy.{name} to {len(items)} traced entries: supported up to 64 (index / slice the position instead)")
        return _lift(ctx, items)
    if name in ("ravel", "copy") and len(args) == 1:
        return _lift(ctx, args[0])
    if name == "reshape" and len(args) == 2:
        return _lift(ctx, args[0]).reshape(args[1])
    raise TraceError(f"numpy.{name} is not supported in a traced logprob_fn (supported: sum, mean, var, std, dot, where, clip, diff, "
                     "linalg.norm, max / min of up to 64 entries and the ufuncs listed in aehmc_amd.tracing)")


# ------------------------------------------------------------------------------------------------------ tracing
class _Ctx:
    def __init__(self):
        self.params, self._ids, self.nvars = [], {}, 0

    def param(self, a):
        a = np.ascontiguousarray(np.asarray(a, dtype=np.float64).reshape(-1))
        key = a.tobytes()
        if key not in self._ids:
            self._ids[key] = len(self.params)
            self.params.append(a.copy())
        return self._ids[key]

    def new_var(self):
        self.nvars += 1
        return self.nvars - 1


class Traced:
    """The result of tracing: ``source`` (HIP source defining ``aehmc_logp``), ``params`` (the captured arrays, in
    ``prm[k]`` order), ``elementwise`` (a sum of per-coordinate terms: ``targets.Custom``) or a joint density
    (``targets.CustomJoint``), and ``dim``."""

    def __init__(self, source, params, elementwise, dim):
        self.source, self.params, self.elementwise, self.dim = source, params, elementwise, dim


class _Gen:
    def __init__(self, elem_var=None, scalar=False):
        self.lines, self.ind, self.ntmp = [], 1, 0
        self.names = {}            # loop variable -> C++ name
        self.elem_var, self.scalar = elem_var, scalar
        self.par_local = None

    def put(self, s):
        self.lines.append("  " * self.ind + s)

    def par(self, k):
        if self.par_local is not None:
            self.par_local.add(k)
            return f"prm{k}"
        return f"prm[{k}]"

    def q(self, idx):
        if self.scalar or (self.elem_var is not None and idx.is_var(self.elem_var)):
            return "q"
        return f"q[{idx.code(self.names, self.par)}]"

    def ex(self, e):
        op, a = e.op, e.args
        if op == "const":
            return _lit(a[0])
        if op == "par":
            if self.par_local is not None:  # (the reverse-mode program reads its parameter arrays through local restrict pointers)
                self.par_local.add(a[0])
                return f"prm{a[0]}[{a[1].code(self.names, self.par)}]"
            return f"prm[{a[0]}][{a[1].code(self.names)}]"
        if op == "q":
            return self.q(a[0])
        if op == "neg":
            return f"(-{self.ex(a[0])})"
        if op == "un":
            return f"{a[0]}({self.ex(a[1])})"
        if op == "bin":
            return f"({self.ex(a[1])} {a[0]} {self.ex(a[2])})"
        if op == "pow":
            base = self.ex(a[0])
            return f"pow({base if a[0].t or not e.t else 'T(' + base + ')'}, {self.ex(a[1])})"
        if op == "cmp":
            return f"({self.ex(a[1])} {a[0]} {self.ex(a[2])})"
        if op == "where":
            c, x, y = self.ex(a[0]), self.ex(a[1]), self.ex(a[2])
            if e.t:
                return f"({c} ? T({x}) : T({y}))"
            return f"({c} ? {x} : {y})"
        if op == "ref":
            return f"h{a[0]}"
        if op == "sum":
            v, n, body = a[:3]
            for k, x in (a[3] if len(a) > 3 else ()):  # loop-invariant sub-expressions (tracing._hoist): once, before the loop
                self.put(f"const T h{k} = T({self.ex(x)});")
            acc = f"s{self.ntmp}"
            self.ntmp += 1
            iv = f"i{v}"
            self.names[v] = iv
            self.put(f"{'T' if e.t else 'double'} {acc} = {'T(0.0)' if e.t else '0.0'};")
            self.put(f"for (int {iv} = 0; {iv} < {n}; {iv}++) {{")
            self.ind += 1
            b = self.ex(body)
            self.put(f"{acc} += {b};")
            self.ind -= 1
            self.put("}")
            return acc
        raise AssertionError(op)


def _elementwise_var(root, dim, scalar):
    """The loop variable of a root that is ONE sum over the coordinates of per-coordinate terms (q read at the loop index
    only, no inner reduction that reads q), else None."""
    if scalar or root.op != "sum" or root.args[1] != dim:
        return None
    v = root.args[0]

    def ok(e):
        if e.op == "q":
            return e.args[0].is_var(v)
        if e.op == "par":
            return e.args[1].terms in ((), ((v, 1),)) and not e.args[1].ind
        if e.op == "sum":
            return False
        return all(ok(x) for x in e.args if isinstance(x, S))

    return v if ok(root.args[2]) else None


# ------------------------------------------------------------------------------------------------------ reverse mode
# The gradient of a joint density in ONE sweep (the reference differentiates in reverse mode: aesara.grad,
# aehmc/hmc.py:33-34, integrators.py:61-65 -- one sweep whatever the dimension).  The recorded expression is a tree whose
# only repetition is its `sum` nodes, so the adjoint program is written down directly: a forward sweep that names every
# intermediate value, then the tree walked from the root with the adjoint of each node; a sum is a loop in the forward
# sweep and a loop again in the backward sweep (its body's values are recomputed there).  The chain's WAVEFRONT runs it:
#   * sub-expressions of a loop body that do not depend on the loop index (exp(-v) in a funnel) are hoisted -- computed
#     once before the loop, their adjoint collected in an accumulator during the backward loop;
#   * a loop at the outermost level whose body reads the position only at indices a * i + c (a != 0) is DISTRIBUTED over
#     the 64 lanes (i = lane, lane + 64, ...): partial sums and the accumulators of hoisted values are reduced across the
#     wavefront behind the loop (a xor butterfly: every lane ends with the same bits), the gradient entries written by
#     the lane that owns the iteration -- O(dim / 64) per gradient;
#   * everything else runs on all lanes alike (same values), lane 0 adding to the gradient row.
_UN_BWD = {"exp": "{a} * {v}", "log": "{a} / {x}", "log1p": "{a} / (1.0 + {x})", "expm1": "{a} * exp({x})",
           "sqrt": "0.5 * {a} / {v}", "sin": "{a} * cos({x})", "cos": "-({a} * sin({x}))", "tanh": "{a} * (1.0 - {v} * {v})",
           "fabs": "({x} < 0 ? -{a} : {a})", "erf": "{a} * 1.1283791670955126 * exp(-{x} * {x})",
           "softplus": "{a} * aehmc::ad::logistic({x})", "logistic": "{a} * {v} * (1.0 - {v})", "atan": "{a} / (1.0 + {x} * {x})", "sinh": "{a} * cosh({x})", "cosh": "{a} * sinh({x})",
           "erfc": "-({a} * 1.1283791670955126 * exp(-{x} * {x}))", "square": "2.0 * {x} * {a}", "lgamma": "{a} * aehmc::ad::digamma({x})"}


# (functions of the generated program that are not the device library's: dual.cuh has why log / log1p are)
_REV_FN = {"square": "aehmc_sq", "softplus": "aehmc_softplus", "log": "aehmc::ad::log_fast", "log1p": "aehmc::ad::log1p_fast", "lgamma": "aehmc::ad::lgamma_fast"}


def _free_vars(e):
    """loop variables an expression reads (those of sums inside it are bound there)"""
    if e.op in ("q",):
        return e.args[0].vars()
    if e.op == "par":
        return e.args[1].vars()
    if e.op == "sum":
        inner = _free_vars(e.args[2]) - {e.args[0]}
        for _, x in (e.args[3] if len(e.args) > 3 else ()):
            inner |= _free_vars(x)
        return inner
    out = frozenset()
    for x in e.args:
        if isinstance(x, S):
            out |= _free_vars(x)
    return out


def _hoist(e, counter, memo=None):
    """The expression with, in every sum, the position-dependent sub-expressions that do not read its loop variable replaced
    by `ref` nodes; the sum node gains a fourth argument: the list of (k, hoisted expression).  Shared nodes stay shared."""
    memo = {} if memo is None else memo
    if id(e) in memo:
        return memo[id(e)]
    if e.op == "sum":
        v, n, body = e.args[:3]
        lets = []
        local = {}

        def go(x):
            if not isinstance(x, S) or not x.t:
                return x
            if id(x) in local:
                return local[id(x)]
            if x.op != "ref" and v not in _free_vars(x):
                k = counter[0]
                counter[0] += 1
                lets.append((k, _hoist(x, counter, memo)))
                r = S(x.ctx, "ref", (k,), True)
            elif x.op in ("q", "ref"):
                r = x
            elif x.op == "sum":
                r = _hoist(x, counter, memo)
            else:
                r = S(x.ctx, x.op, tuple(go(y) for y in x.args), x.t, x.b)
            local[id(x)] = r
            return r

        out = S(e.ctx, "sum", (v, n, go(body), lets), e.t)
    elif e.op in ("const", "par", "q", "ref"):
        out = e
    else:
        out = S(e.ctx, e.op, tuple(_hoist(x, counter, memo) if isinstance(x, S) else x for x in e.args), e.t, e.b)
    memo[id(e)] = out
    return out


PRIVATE_MAX = 64  # an inner reduction up to this long may read the position inside a distributed loop (private accumulators:
                  # registers up to ~32 terms, beyond that partly scratch -- logistic regression with 40 / 64 coefficients runs at 0.34 /
                  # 0.08 of CustomGLM's GEMM path, tools/debug/logistic_many_coefficients.py, against ~0.01 on one lane)


def _private_leaves(e):
    """For a sum that is to be distributed over the lanes: {id(q leaf): (inner sum node, a, c)} of the position reads that
    do not depend on ITS variable but on the variable of ONE inner sum of at most PRIVATE_MAX terms (index a * j + c) -- a
    data row's product with the coefficients, sum_j X[i, j] q[j].  Their adjoints collect in per-lane arrays indexed by j,
    reduced over the wavefront behind the loop.  None if some read fits neither this form nor a * i + c."""
    v = e.args[0]
    out = {}

    def walk(x, inner):
        if x.op == "q":
            t = x.args[0].terms
            if x.args[0].ind:  # a gather: added atomically
                return True
            if len(t) == 1 and t[0][0] == v and t[0][1] != 0:
                return True
            if len(t) == 1 and t[0][0] in inner and t[0][1] != 0 and inner[t[0][0]].args[1] <= PRIVATE_MAX:
                out[id(x)] = (inner[t[0][0]], t[0][1], x.args[0].c)
                return True
            return False
        if x.op == "sum":
            inner2 = dict(inner)
            inner2[x.args[0]] = x
            return all(walk(h, inner) for _, h in x.args[3]) and walk(x.args[2], inner2)
        return all(walk(y, inner) for y in x.args if isinstance(y, S))

    return out if walk(e.args[2], {}) else None


def _mixed_owners(e):
    """Does a distributed loop write gradient entries through more than one index map a * i + c?  (x[1:] - x[:-1]: lane i
    owns entry i + 1 for one read and entry i for the other, which is lane i - 1's for the first.)  Then no lane owns an
    entry and the additions are atomic."""
    v = e.args[0]
    maps = set()

    def walk(x):
        if x.op == "q":
            t = x.args[0].terms
            if not x.args[0].ind and len(t) == 1 and t[0][0] == v:
                maps.add((t[0][1], x.args[0].c))
        elif x.op == "sum":
            for _, h in x.args[3]:
                walk(h)
            walk(x.args[2])
        else:
            for y in x.args:
                if isinstance(y, S):
                    walk(y)

    walk(e.args[2])
    return len(maps) > 1


def _distributable(e):
    """a sum whose body reads (and so writes the gradient of) the position at indices a * i + c, a != 0, or through short
    inner reductions (_private_leaves)"""
    return _private_leaves(e) is not None


def _spine(root):
    """sums joined to the root by + / - / unary minus only: id -> their adjoint as a literal"""
    out = {}

    def go(e, sign):
        if e.op == "sum":
            out[id(e)] = "1.0" if sign > 0 else "-1.0"
        elif e.op == "neg":
            go(e.args[0], -sign)
        elif e.op == "bin" and e.args[0] in "+-":
            go(e.args[1], sign)
            go(e.args[2], sign if e.args[0] == "+" else -sign)

    go(root, 1)
    return out


class _RevGen:
    def __init__(self, spine=None):
        self.lines, self.ind, self.n = [], 1, 0
        self.names = {}  # loop variable -> C++ name
        self.depth = 0   # loop nesting
        self.spine, self.done = spine or {}, set()
        self.used_params = set()
        self.warned = set()
        self.uses, self.cond = {}, set()
        self.private = {}     # id(q leaf) -> (accumulator array, inner sum node)
        self.unrolled = set()  # ids of inner sums whose loops are written out (their accumulator arrays stay in registers)

    def put(self, s):
        self.lines.append("  " * self.ind + s)

    def tmp(self, prefix="t"):
        self.n += 1
        return f"{prefix}{self.n}"

    def let(self, expr, prefix="t"):
        name = self.tmp(prefix)
        self.put(f"const double {name} = {expr};")
        return name

    # ---- forward: the value of e as a C++ expression; every position-dependent inner node gets a name in env
    def fwd(self, e, env):
        op, a = e.op, e.args
        if id(e) in env:  # (a shared node: its value has a name already)
            return env[id(e)]
        if op == "const":
            return _lit(a[0])
        if op == "par":
            self.used_params.add(a[0])
            return f"prm{a[0]}[{a[1].code(self.names, self.par)}]"
        if op == "q":
            return f"q[{a[0].code(self.names, self.par)}]"
        if op == "ref":
            return env[("let", a[0])]
        if not e.t and op != "sum":  # parameters and constants only: inlined
            g = _Gen()
            g.names = self.names
            g.par_local = self.used_params
            return g.ex(e)
        if op == "neg":
            r = f"(-{self.fwd(a[0], env)})"
        elif op == "un":
            r = f"{_REV_FN.get(a[0], a[0])}({self.fwd(a[1], env)})"
        elif op == "bin":
            r = f"({self.fwd(a[1], env)} {a[0]} {self.fwd(a[2], env)})"
        elif op == "pow":
            r = f"pow({self.fwd(a[0], env)}, {self.fwd(a[1], env)})"
        elif op == "cmp":
            return f"({self.fwd(a[1], env)} {a[0]} {self.fwd(a[2], env)})"
        elif op == "where":
            c, x, y = self.fwd(a[0], env), self.fwd(a[1], env), self.fwd(a[2], env)
            r = f"({c} ? {x} : {y})"
        elif op == "sum":
            return self.fwd_sum(e, env)
        else:
            raise AssertionError(op)
        name = self.let(r)
        env[id(e)] = name
        return name

    def loop_head(self, e, dist):
        v, n = e.args[0], e.args[1]
        iv = f"i{v}"
        self.names[v] = iv
        if dist and n >= 512 and not self.has_inner_sum(e.args[2]):  # (a long sweep over data: four iterations' loads in flight;
            self.put("#pragma unroll 4")                               #  with an inner reduction that one is written out instead)
        if id(e) in self.unrolled:
            self.put("#pragma unroll")
        return f"for (int {iv} = {'lane' if dist else '0'}; {iv} < {n}; {iv} {'+= AEHMC_LANES' if dist else '++'}) {{"

    def par(self, k):
        self.used_params.add(k)
        return f"prm{k}"

    def has_inner_sum(self, e):
        return isinstance(e, S) and (e.op == "sum" or any(self.has_inner_sum(x) for x in e.args if isinstance(x, S)))

    def fwd_sum(self, e, env):
        v, n, body, lets = e.args
        dist = self.depth == 0 and _distributable(e)
        if self.depth == 0 and not dist and n >= 1024 and id(e) not in self.warned:
            self.warned.add(id(e))
            import warnings
            warnings.warn(f"aehmc_amd.tracing: a reduction over {n} terms reads the position through an inner reduction of more "
                          f"than {PRIVATE_MAX} terms (or at indices that mix loop variables): on the device it runs on ONE lane per "
                          "chain.  For a regression with many coefficients targets.CustomGLM puts the products with the data matrix "
                          "on the matrix cores", stacklevel=6)
        env[("dist", id(e))] = dist
        for k, x in lets:
            env[("let", k)] = self.fwd(x, env)
        acc = self.tmp("s")
        self.put(f"double {acc} = 0.0;")
        # a sum on the additive spine of the density (log-density = term + term + ...) has the adjoint +1 / -1 whatever the
        # other terms are: its backward sweep rides in the forward loop -- one pass over the data instead of two
        fused = self.spine.get(id(e)) if self.depth == 0 else None
        priv = self.private_begin(e) if dist else []
        if fused:
            for k, x in lets:
                self.put(f"double ah{k} = 0.0;")
            self.private_declare(priv)
        if fused and dist:
            self.put("AEHMC_SYNC();")  # (gradient entries change owner between loops when several wavefronts run the program)
        self.put(self.loop_head(e, dist))
        self.ind += 1
        self.depth += 1
        benv = dict(env)
        self.put(f"{acc} += {self.fwd(body, benv)};")
        if fused:
            outer_owned, self.lane_owned = self.lane_owned, dist
            outer_atomic, self.atomic_loop = self.atomic_loop, dist and _mixed_owners(e)
            self.bwd(body, fused, benv)
            self.lane_owned, self.atomic_loop = outer_owned, outer_atomic
        self.depth -= 1
        self.ind -= 1
        self.put("}")
        if fused and dist:
            self.put("AEHMC_SYNC();")
        if dist:
            self.put(f"{acc} = AEHMC_WSUM({acc});")
        if fused:
            self.private_reduce(priv)
            for k, x in reversed(lets):
                if dist:
                    self.put(f"ah{k} = AEHMC_WSUM(ah{k});")
                self.bwd(x, f"ah{k}", env)
            self.done.add(id(e))
        env[("priv", id(e))] = priv
        env[id(e)] = acc
        return acc

    def val(self, e, env):
        """the forward value of a child as it was named (or its inlined expression)"""
        if e.op == "ref":
            return env[("let", e.args[0])]
        if id(e) in env:
            return env[id(e)]
        return self.fwd(e, env)  # (leaves and parameter-only expressions: no statements)

    def count_uses(self, root):
        """how many parents every position-dependent inner node has (a shared node's adjoint is collected from all of them
        and propagated ONCE), and which nodes are reached through a branch of a `where` (those are never merged: the
        statement that would propagate the merged adjoint might sit in the branch that is not taken)"""
        self.uses, self.cond = {}, set()

        def go(e, cond):
            if not isinstance(e, S) or not e.t or e.op in ("q", "ref", "const", "par", "cmp"):
                return  # (a comparison passes no adjoint on: its operands' uses there do not count)
            if cond:
                self.cond.add(id(e))
            self.uses[id(e)] = self.uses.get(id(e), 0) + 1
            if self.uses[id(e)] > 1 and not cond:
                return
            if e.op == "where":
                go(e.args[1], True)
                go(e.args[2], True)
            elif e.op == "sum":
                for _, x in e.args[3]:
                    go(x, cond)
                go(e.args[2], cond)
            else:
                for x in e.args:
                    go(x, cond)

        go(root, False)

    # ---- backward: propagate the adjoint `adj` (a name or a literal) of e into the position
    def bwd(self, e, adj, env):
        if not e.t:
            return
        op, a = e.op, e.args
        if op not in ("q", "ref") and self.uses.get(id(e), 1) > 1 and id(e) not in self.cond and id(e) in env:
            # a shared node: collect, propagate behind the last parent
            key = ("adj", id(e))
            seen = env.get(("nadj", id(e)), 0) + 1
            env[("nadj", id(e))] = seen
            if seen == 1:
                env[key] = self.tmp("ad")
                self.put(f"double {env[key]} = {adj};")
            else:
                self.put(f"{env[key]} += {adj};")
            if seen < self.uses[id(e)]:
                return
            adj = env[key]
        if op == "q":
            if id(e) in self.private:  # (read through a short inner reduction inside a distributed loop: per-lane accumulator)
                arr, inner = self.private[id(e)]
                self.put(f"{arr}[{self.names[inner.args[0]]}] += {adj};")
                return
            tgt = f"g[{a[0].code(self.names, self.par)}]"
            if (a[0].ind or self.atomic_loop) and self.depth > 0 and self.lane_owned:
                # a gather, or a loop that writes entries through several index maps, inside a distributed loop: several lanes
                # may hold the same entry -- an atomic add into the LDS row (the order of the additions of one wavefront
                # instruction is the hardware's: rounding-level effects only)
                self.put(f"AEHMC_ATOMIC_ADD(&{tgt}, {adj});")
            else:
                self.put(f"{tgt} += {adj};" if self.depth > 0 and self.lane_owned else f"if (lane == 0) {tgt} += {adj};")
        elif op == "ref":
            self.put(f"ah{a[0]} += {adj};")
        elif op == "neg":
            self.bwd(a[0], self.let(f"-{adj}", "a"), env)
        elif op == "un":
            x = a[1]
            expr = _UN_BWD[a[0]].format(a=adj, v=env[id(e)], x=self.val(x, env))
            self.bwd(x, self.let(expr, "a"), env)
        elif op == "bin":
            o, x, y = a
            if o == "+":
                self.bwd(x, adj, env)
                self.bwd(y, adj, env)
            elif o == "-":
                self.bwd(x, adj, env)
                if y.t:
                    self.bwd(y, self.let(f"-{adj}", "a"), env)
            elif o == "*":
                if x.t:
                    self.bwd(x, self.let(f"{adj} * {self.val(y, env)}", "a"), env)
                if y.t:
                    self.bwd(y, self.let(f"{adj} * {self.val(x, env)}", "a"), env)
            else:
                if x.t:
                    self.bwd(x, self.let(f"{adj} / {self.val(y, env)}", "a"), env)
                if y.t:
                    self.bwd(y, self.let(f"-({adj} * {env[id(e)]} / {self.val(y, env)})", "a"), env)
        elif op == "pow":
            x, y = a
            if x.t:
                self.bwd(x, self.let(f"{adj} * {self.val(y, env)} * pow({self.val(x, env)}, {self.val(y, env)} - 1.0)", "a"), env)
            if y.t:
                self.bwd(y, self.let(f"{adj} * {env[id(e)]} * log({self.val(x, env)})", "a"), env)
        elif op == "where":
            c, x, y = a
            self.put(f"if ({self.fwd(c, env)}) {{")
            self.ind += 1
            self.bwd(x, adj, env)
            self.ind -= 1
            self.put("} else {")
            self.ind += 1
            self.bwd(y, adj, env)
            self.ind -= 1
            self.put("}")
        elif op == "sum":
            if id(e) in self.done:  # (its backward sweep rode in its forward loop)
                return
            v, n, body, lets = a
            dist = env[("dist", id(e))]
            for k, x in lets:
                self.put(f"double ah{k} = 0.0;")
            self.private_declare(env.get(("priv", id(e)), []))
            if adj[0] not in "at" or not adj[1:].isdigit():  # (an expression: named once, outside the loop)
                adj = self.let(adj, "a")
            if dist:
                self.put("AEHMC_SYNC();")
            self.put(self.loop_head(e, dist))
            self.ind += 1
            self.depth += 1
            outer_owned, outer_atomic = self.lane_owned, self.atomic_loop
            if dist:
                self.lane_owned = True
                self.atomic_loop = _mixed_owners(e)
            benv = dict(env)
            self.fwd(body, benv)  # (the body's values again)
            self.bwd(body, adj, benv)
            self.lane_owned, self.atomic_loop = outer_owned, outer_atomic
            self.depth -= 1
            self.ind -= 1
            self.put("}")
            if dist:
                self.put("AEHMC_SYNC();")
            self.private_reduce(env.get(("priv", id(e)), []))
            for k, x in reversed(lets):
                if dist:
                    self.put(f"ah{k} = AEHMC_WSUM(ah{k});")
                self.bwd(x, f"ah{k}", env)
        else:
            raise AssertionError(op)

    lane_owned = False  # inside a distributed loop: the lane owns the gradient entries its iterations touch
    atomic_loop = False  # ... unless the loop writes them through several index maps (_mixed_owners): atomic additions

    # ---- per-lane accumulators of position reads through short inner reductions (_private_leaves)
    def private_begin(self, e):
        """name one accumulator array per (inner sum, stride, offset) of the distributed sum e; returns [(array, n, a, c)]"""
        arrays, out = {}, []
        for leaf, (inner, a, c) in (_private_leaves(e) or {}).items():
            key = (id(inner), a, c)
            if key not in arrays:
                arrays[key] = self.tmp("gp")
                out.append((arrays[key], inner.args[1], a, c))
            self.private[leaf] = (arrays[key], inner)
            self.unrolled.add(id(inner))
        return out

    def private_declare(self, priv):
        for arr, n, a, c in priv:
            self.put(f"double {arr}[{n}];")
            self.put("#pragma unroll")
            self.put(f"for (int j = 0; j < {n}; j++) {arr}[j] = 0.0;")

    def private_reduce(self, priv):
        for arr, n, a, c in priv:
            self.put("#pragma unroll")
            self.put(f"for (int j = 0; j < {n}; j++) {{")
            self.put(f"  const double r = AEHMC_WSUM({arr}[j]);")
            self.put(f"  if (lane == 0) g[{a} * j + {c}] += r;")
            self.put("}")


_REV_PRELUDE = """
#ifndef AEHMC_LANES  /* (plain C++ builds for the CPU tests define AEHMC_LANES 1, AEHMC_WSUM(x) (x), AEHMC_SYNC(), AEHMC_ATOMIC_ADD(p, v) (*(p) += (v))) */
// AEHMC_W wavefronts run the program together (1: the chain's wavefront; 8: a workgroup per chain, engine.cuh k_nuts_joint_wg)
template <int W> __device__ inline double aehmc_wsum_(double x) {  // xor butterfly: a + b == b + a bit for bit, so every lane ends with the same bits
  for (int m = 32; m >= 1; m >>= 1) x += __shfl_xor(x, m);
  if (W > 1) {  // the wavefronts' sums through LDS, added in wavefront order by every thread
    __shared__ double aehmc_part[16];
    __syncthreads();  // (the reads of the previous reduction are done)
    if ((threadIdx.x & 63) == 0) aehmc_part[threadIdx.x >> 6] = x;
    __syncthreads();
    x = aehmc_part[0];
    for (int w = 1; w < W; w++) x += aehmc_part[w];
  }
  return x;
}
#define AEHMC_LANES (64 * AEHMC_W)
#define AEHMC_WSUM(x) aehmc_wsum_<AEHMC_W>(x)
#define AEHMC_SYNC() do { if (AEHMC_W > 1) __syncthreads(); } while (0)
#define AEHMC_ATOMIC_ADD(p, v) atomicAdd((p), (v))
#endif
__device__ inline double aehmc_sq(double x) { return x * x; }
__device__ inline double aehmc_softplus(double x) { return aehmc::ad::softplus(x); }
#define AEHMC_JOINT_GRAD 1
// log-density and its gradient in one reverse sweep, run by AEHMC_W wavefronts together (`lane` = the thread's index among
// their 64 AEHMC_W lanes): q = the position row, g = the gradient row (both in LDS; g zeroed by the caller); every lane
// returns the same log-density
template <int AEHMC_W> __device__ double aehmc_logp_grad_t(const double *q, double *g, int lane, const double *const *prm) {
"""
_REV_EPILOGUE = """
__device__ inline double aehmc_logp_grad(const double *q, double *g, int lane, const double *const *prm) {
  return aehmc_logp_grad_t<1>(q, g, lane, prm);
}
"""


def _distributed_terms(e, top=True):
    """how many loop iterations the reverse-mode program spreads over the lanes (outermost distributable sums)"""
    if e.op == "sum":
        if top and _distributable(e):
            return e.args[1] + sum(_distributed_terms(x) for _, x in e.args[3])
        return sum(_distributed_terms(x, top) for _, x in e.args[3])  # (what it hoisted is evaluated outside it)
    return sum(_distributed_terms(x, top) for x in e.args if isinstance(x, S))


def _reverse_source(root, dim):
    tree = _hoist(root, [0])
    gen = _RevGen(_spine(tree))
    gen.count_uses(tree)
    env = {}
    val = gen.fwd(tree, env)
    if not isinstance(val, str):
        raise AssertionError
    gen.bwd(tree, "1.0", env)
    ptrs = "".join(f"  const double *__restrict__ const prm{k} = prm[{k}];\n" for k in sorted(gen.used_params))
    src = _REV_PRELUDE + ptrs + "\n".join(gen.lines) + f"\n  return {val};\n}}\n" + _REV_EPILOGUE
    # how many loop iterations the program spreads over its lanes: with few chains and long sweeps the engine gives a
    # chain a whole workgroup (engine.hip: joint_wg_wanted)
    src += f"#define AEHMC_JOINT_SWEEP_TERMS {_distributed_terms(tree)}\n"
    # up to 64 coordinates the engine's one-launch kernels differentiate in forward mode, every lane running the whole
    # density for its own coordinate: right for a funnel, 64 times too much for a sum over 10^4 data rows
    if dim <= 64 and _distributed_terms(tree) >= 256:
        src += "#define AEHMC_JOINT_GRAD_SMALL 1\n"
    return src


def trace(fn, dim, scalar=False, args=(), reverse="auto"):
    """Call ``fn`` once on a proxy of the position (a scalar if ``scalar``, else a vector of ``dim`` entries) and emit
    the ``aehmc_logp`` template.  ``args``: further constant arguments handed to ``fn`` (numbers / numpy arrays).
    ``reverse`` (joint densities): "auto" -- the reverse-mode program is emitted and taken above 64 coordinates, or below
    when the density's reductions are long; True -- taken at every size; False -- not emitted (forward mode only)."""
    dim = int(dim)
    ctx = _Ctx()
    if scalar:
        if dim != 1:
            raise ValueError("a scalar position has dim 1")
        arg = S(ctx, "q", (Idx(0),), True)
    else:
        arg = V(ctx, dim, lambda i: S(ctx, "q", (i,), True))
    out = fn(arg, *[_lift(ctx, a) if isinstance(a, np.ndarray) else a for a in args])
    if isinstance(out, V):
        if out.n == 1:
            out = out.at(0)
        else:
            raise TraceError(f"logprob_fn returned a vector of {out.n} entries: it must return a scalar (call .sum() on it?)")
    if not isinstance(out, S):
        raise TraceError(f"logprob_fn returned a {type(out).__name__} that does not depend on the position")
    if out.b:
        raise TraceError("logprob_fn returned a comparison result")
    if not out.t:
        raise TraceError("logprob_fn returned a value that does not depend on the position")
    def has_sum(e):
        return e.op == "sum" or any(has_sum(x) for x in e.args if isinstance(x, S))

    if dim == 1 and not scalar and not has_sum(out):  # a vector of one entry used entry by entry: the scalar form
        scalar = True
    ev = _elementwise_var(out, dim, scalar)
    if scalar or ev is not None:
        g = _Gen(elem_var=ev, scalar=scalar)
        if ev is not None:
            g.names[ev] = "i"
            body = g.ex(out.args[2])
        else:
            body = g.ex(out)
        src = ("template <class T> __device__ T aehmc_logp(T q, long long i, const double *const *prm) {\n"
               + "\n".join(g.lines) + ("\n" if g.lines else "") + f"  return T({body});\n}}\n")
        return Traced(src, ctx.params, True, dim)
    g = _Gen()
    body = g.ex(_hoist(out, [0]))
    src = ("template <class V> __device__ auto aehmc_logp(const V &q, const double *const *prm) {\n"
           "  typedef decltype(q[0]) T;\n" + "\n".join(g.lines) + ("\n" if g.lines else "") + f"  return T({body});\n}}\n")
    tr = Traced(src, ctx.params, False, dim)
    tr.grad_source = None
    if reverse is not False:  # (above 64 coordinates the engine takes this instead of ceil(dim / 64) forward passes)
        tr.grad_source = _reverse_source(out, dim)
        if reverse is True and "#define AEHMC_JOINT_GRAD_SMALL" not in tr.grad_source:
            tr.grad_source += "#define AEHMC_JOINT_GRAD_SMALL 1\n"
    return tr

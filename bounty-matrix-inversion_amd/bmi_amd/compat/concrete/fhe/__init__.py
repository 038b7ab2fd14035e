"""`concrete.fhe`-compatible front end of the MI355X engine (see bmi_amd/compat/__init__.py for the supported surface and how to
put it on the import path): the reference's UNMODIFIED base_p_arrays.py / qfloat.py / qfloat_matrix_inversion.py / main.py trace
into this package's circuit IR (bmi_amd.circuit) and run on LWE ciphertexts, every look-up a programmable bootstrap on the GPU.

How a function is compiled (the reference is data-oblivious, so its sequence of operations is the same on every input):
  1. MEASURE: the function runs on the inputset; every encrypted scalar carries the vector of its sample values, and
     every non-linear operation (comparison, //, %, abs, sign, bit-wise op, ciphertext product, fhe.univariate) records
     the range its operand takes - what Concrete's compiler does with its inputset (main.py:41-66).
  2. BUILD: the function runs again on symbolic values (Lin); the k-th non-linear operation claims the range measured
     for it (`Lin.assume`, verified by `Circuit.simulate`) and becomes a table look-up of the circuit.
  3. PARAMETERS: the circuit is frozen into a Program (pruned, scheduled) and the first parameter set of the engine whose error
     budget for THIS circuit is at most `Configuration(global_p_error=...)` (default 1e-5, Concrete's) is chosen
     (bmi_amd/error_budget.py); `Configuration(security_level=128)` restricts the choice to the 128-bit-secure set.
Operator surface: SURVEY.md section 8b.  Fixture generation (tools/gen_ref_traced.py, tools/gen_ref_fhe_tests.py) runs this same
front end with BMI_COMPAT_BACKEND=simulate in the build container, where the reference exists and a GPU does not."""
import operator as _op

import numpy as np

from .tracing.tracer import Tracer
from . import tracing  # noqa: F401

_S = {"mode": None, "circuit": None, "ranges": [], "site": 0, "margin": 0, "fuse": False}


def _site(vals):
    """one operand of one non-linear operation: records (measure) or returns (build) its range"""
    k = _S["site"]
    _S["site"] += 1
    if _S["mode"] == "measure":
        lo, hi = int(np.min(vals)), int(np.max(vals))
        if k == len(_S["ranges"]):
            _S["ranges"].append([lo, hi])
        else:
            r = _S["ranges"][k]
            r[0], r[1] = min(r[0], lo), max(r[1], hi)
        return None
    lo, hi = _S["ranges"][k]
    if _S.get("bitwidth"):
        # Concrete derives a BIT WIDTH from the inputset, not an interval: an intermediate seen in [7, 17] is a 5-bit
        # unsigned value and 5 or 31 are as good as 12.  Round the measured interval out to its bit width.
        if lo >= 0:
            lo, hi = 0, (1 << max(hi.bit_length(), 1)) - 1
        else:
            b = max((-lo - 1).bit_length(), hi.bit_length()) + 1
            lo, hi = -(1 << (b - 1)), (1 << (b - 1)) - 1
    return [lo, hi]


class Def:
    """Build phase, lazy look-up fusion (trace(..., fuse=True)): a value known as a univariate function of ONE linear
    combination `base` of ciphertexts - its table over base's interval - and not yet computed.  Further univariate
    operations, constants, and values deferred on the same base compose into the table for free; a look-up (one PBS)
    is emitted only when the value meets a different base in a linear combination or a product, or is an output.  This
    is the fusion the restated layer in bmi_amd/ does by hand (e.g. the reference's abs -> // base -> * sign ->
    c - carry * base chain of qfloat.py:617-621 becomes two look-ups on c)."""
    __slots__ = ("base", "tab", "_lin")

    def __init__(self, base, tab):
        self.base, self.tab, self._lin = base, tab, None

    @property
    def key(self):
        return _lin_key(self.base)

    def lin(self):
        if self._lin is None:
            lo, tab = self.base.lo, self.tab
            self._lin = _S["circuit"].lut(self.base, lambda t: tab[t - lo])
        return self._lin


def _lin_key(lin):
    return (tuple(sorted(lin.terms.items())), lin.const)


def _mat(v):
    return v.lin() if isinstance(v, Def) else v


def _as_def(v):
    """v as a deferred univariate value, if it is one (or a narrow linear combination: the identity on itself)"""
    if isinstance(v, Def):
        return v
    if hasattr(v, "terms") and v.terms and v.hi - v.lo + 1 <= (1 << _S["circuit"].msg_bits):
        return Def(v, list(range(v.lo, v.hi + 1)))
    return None


def _fused(a, b, op):
    """op(a, b) without a PBS when a and b are univariate in the same base (or one is a constant); None otherwise"""
    if not _S["fuse"]:
        return None
    ia, ib = isinstance(a, (int, np.integer)), isinstance(b, (int, np.integer))
    if ia and isinstance(b, Def):
        return Def(b.base, [int(op(int(a), y)) for y in b.tab])
    if ib and isinstance(a, Def):
        return Def(a.base, [int(op(x, int(b))) for x in a.tab])
    if not (isinstance(a, Def) or isinstance(b, Def)):
        return None
    da, db = _as_def(a), _as_def(b)
    if da is None or db is None or da.key != db.key:
        return None
    lo, hi = max(da.base.lo, db.base.lo), min(da.base.hi, db.base.hi)   # both interval claims hold
    if lo > hi:
        return None
    base = da.base if (da.base.lo, da.base.hi) == (lo, hi) else (db.base if (db.base.lo, db.base.hi) == (lo, hi) else da.base.assume(lo, hi))
    return Def(base, [int(op(da.tab[t - da.base.lo], db.tab[t - db.base.lo])) for t in range(lo, hi + 1)])


class Enc(Tracer):
    """one encrypted scalar: sample vector (measure), Lin (build) - plain ints stay ints"""
    __slots__ = ("v",)
    __array_priority__ = 1000

    def __init__(self, v):
        self.v = v

    # ---- helpers
    @staticmethod
    def _raw(x):
        return x.v if isinstance(x, Enc) else (int(x) if isinstance(x, (int, np.integer, bool, np.bool_)) else x)

    @staticmethod
    def _wrap(v):
        if isinstance(v, (int, np.integer)):
            return int(v)
        return Enc(v)

    def _lut(self, fn):
        """univariate non-linear operation on this value"""
        if _S["mode"] == "measure":
            _site(self.v)
            return Enc(np.array([int(fn(int(t))) for t in self.v], dtype=np.int64))
        lo, hi = _site(None)
        c = _S["circuit"]
        if isinstance(self.v, Def):
            return Enc(Def(self.v.base, [int(fn(int(y))) for y in self.v.tab]))
        x = self.v.assume(lo, hi) if hasattr(self.v, "assume") else self.v
        if _S["fuse"] and hasattr(x, "terms") and x.terms and x.hi - x.lo + 1 <= (1 << c.msg_bits):
            return Enc(Def(x, [int(fn(t)) for t in range(x.lo, x.hi + 1)]))
        out = c.lut(x, lambda t: int(fn(int(t))))
        return Enc(out)   # stays encrypted even when the look-up folds to a constant: both phases must see the same sites

    # ---- linear
    @staticmethod
    def _linear(a, b, op):
        if _S["mode"] == "build":
            r = _fused(a, b, op)
            if r is not None:
                return Enc(r)
            a, b = _mat(a), _mat(b)
        return Enc._wrap(op(a, b))

    def __add__(self, o):
        if isinstance(o, (np.ndarray, EncArray)):
            return NotImplemented
        return Enc._linear(self.v, Enc._raw(o), _op.add)
    __radd__ = __add__

    def __sub__(self, o):
        if isinstance(o, (np.ndarray, EncArray)):
            return NotImplemented
        return Enc._linear(self.v, Enc._raw(o), _op.sub)

    def __rsub__(self, o):
        return Enc._linear(Enc._raw(o), self.v, _op.sub)

    def __neg__(self):
        return Enc._linear(0, self.v, _op.sub)

    def __pos__(self):
        return self

    def __mul__(self, o):
        if isinstance(o, (np.ndarray, EncArray)):
            return NotImplemented
        if not isinstance(o, Enc):
            return Enc._linear(self.v, int(o), _op.mul)
        if _S["mode"] == "measure":
            _site(self.v)
            _site(o.v)
            return Enc(self.v * o.v)
        ra, rb = _site(None), _site(None)
        c = _S["circuit"]
        r = _fused(self.v, o.v, _op.mul)
        if r is not None:
            return Enc(r)
        out = c.mul(_mat(self.v).assume(*ra), _mat(o.v).assume(*rb))
        return Enc(out)   # stays encrypted even when the look-up folds to a constant: both phases must see the same sites
    __rmul__ = __mul__

    # ---- non-linear, univariate on a linear combination
    def _cmp(self, o, fn):
        d = self - o
        return fn(int(d)) if not isinstance(d, Enc) else d._lut(fn)

    def __lt__(self, o):
        return self._cmp(o, lambda t: int(t < 0))

    def __le__(self, o):
        return self._cmp(o, lambda t: int(t <= 0))

    def __gt__(self, o):
        return self._cmp(o, lambda t: int(t > 0))

    def __ge__(self, o):
        return self._cmp(o, lambda t: int(t >= 0))

    def __eq__(self, o):  # noqa: PLW1641 (hash not needed: Enc lives in object arrays)
        return self._cmp(o, lambda t: int(t == 0))

    def __ne__(self, o):
        return self._cmp(o, lambda t: int(t != 0))

    def __floordiv__(self, k):
        return self._lut(lambda t, k=int(k): t // k)

    def __mod__(self, k):
        return self._lut(lambda t, k=int(k): t % k)

    def __rshift__(self, k):
        return self._lut(lambda t, k=int(k): t >> k)

    def __abs__(self):
        return self._lut(abs)

    def sign(self):
        return self._lut(lambda t: (t > 0) - (t < 0))

    # bit-wise operations: both operands are bits (what the reference uses them for) -> a look-up on their sum
    def _bits(self, o, fn):
        if not isinstance(o, Enc):
            o = int(o)
            return self._lut(lambda t, o=o: fn(t, o))
        return (self + o)._lut(lambda s: fn(int(s >= 1), int(s >= 2)) if fn is not _xor else int(s == 1))

    def __and__(self, o):
        return self._bits(o, lambda a, b: a & b)
    __rand__ = __and__

    def __or__(self, o):
        return self._bits(o, lambda a, b: a | b)
    __ror__ = __or__

    def __xor__(self, o):
        return self._bits(o, _xor)
    __rxor__ = __xor__

    # ---- numpy-facing
    def reshape(self, *shape):
        a = np.empty(1, dtype=object)
        a[0] = self
        return EncArray(a).reshape(*shape)

    @property
    def size(self):
        return 1

    @property
    def shape(self):
        return ()

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        if method != "__call__":
            return NotImplemented
        if ufunc is np.sign:
            return self.sign()
        if ufunc is np.absolute:
            return abs(self)
        table = {np.add: lambda a, b: a + b, np.subtract: lambda a, b: a - b, np.multiply: lambda a, b: a * b,
                 np.negative: lambda a: -a, np.less: lambda a, b: a < b, np.greater: lambda a, b: a > b,
                 np.less_equal: lambda a, b: a <= b, np.greater_equal: lambda a, b: a >= b,
                 np.equal: lambda a, b: a == b, np.not_equal: lambda a, b: a != b,
                 np.floor_divide: lambda a, b: a // b, np.remainder: lambda a, b: a % b,
                 np.bitwise_and: lambda a, b: a & b, np.bitwise_or: lambda a, b: a | b, np.bitwise_xor: lambda a, b: a ^ b}
        if ufunc not in table:
            return NotImplemented
        inputs = tuple(int(x) if isinstance(x, (np.integer, np.bool_)) else x for x in inputs)   # numpy scalar (op) Enc
        if any(isinstance(x, (np.ndarray, EncArray)) for x in inputs):
            return _elementwise(table[ufunc], *inputs)
        return table[ufunc](*inputs)


def _xor(a, b):
    return a ^ b


import operator as _op

_UFUNC_OPS = {np.add: _op.add, np.subtract: _op.sub, np.multiply: _op.mul, np.negative: _op.neg, np.positive: _op.pos,
              np.less: _op.lt, np.greater: _op.gt, np.less_equal: _op.le, np.greater_equal: _op.ge, np.equal: _op.eq,
              np.not_equal: _op.ne, np.floor_divide: _op.floordiv, np.remainder: _op.mod, np.bitwise_and: _op.and_,
              np.bitwise_or: _op.or_, np.bitwise_xor: _op.xor, np.right_shift: _op.rshift, np.absolute: abs,
              np.sign: lambda t: t.sign() if isinstance(t, Enc) else int((t > 0) - (t < 0))}


def _unwrap(x):
    if isinstance(x, EncArray):
        return x._a
    if isinstance(x, (list, tuple)):
        return type(x)(_unwrap(e) for e in x)
    return x


def _wrap_result(r):
    if isinstance(r, np.ndarray) and r.dtype == object:
        return EncArray(r)
    if isinstance(r, (list, tuple)):
        return type(r)(_wrap_result(e) for e in r)
    return r


def _elementwise(fn, *operands):
    """fn applied element by element with numpy broadcasting; the elements' own operators build the circuit"""
    arrs = [np.asarray(_unwrap(x), dtype=object) if isinstance(x, (EncArray, np.ndarray)) else None for x in operands]
    shape = np.broadcast(*[a for a in arrs if a is not None]).shape
    bc = [np.broadcast_to(a, shape) if a is not None else None for a in arrs]
    out = np.empty(shape, dtype=object)
    for idx in np.ndindex(shape):
        out[idx] = fn(*[b[idx] if b is not None else x for b, x in zip(bc, operands)])
    return EncArray(out)


class EncArray(Tracer):
    """An encrypted tensor as the reference sees it: NOT an np.ndarray (qfloat.py:278 treats ndarrays as cleartext),
    but with the array surface the reference uses - shape/size/len, indexing and slice assignment, reshape/flatten,
    element-wise arithmetic, comparisons and bit-wise operations, and the numpy functions sum / concatenate / abs /
    sign / ... through the __array_function__ / __array_ufunc__ protocols.  Elements are Enc scalars or plain ints."""
    __array_priority__ = 2000

    def __init__(self, a):
        self._a = np.asarray(a, dtype=object)

    shape = property(lambda self: self._a.shape)
    size = property(lambda self: self._a.size)
    ndim = property(lambda self: self._a.ndim)
    T = property(lambda self: EncArray(self._a.T))

    def __len__(self):
        return len(self._a)

    def __iter__(self):
        return (self[i] for i in range(len(self._a)))

    def __getitem__(self, k):
        r = self._a[_unwrap(k)]
        if isinstance(r, np.ndarray):
            return EncArray(r)
        return np.int64(r) if isinstance(r, int) else r   # a cleartext element still answers .reshape() etc.

    def __setitem__(self, k, v):
        v = _unwrap(v)
        if isinstance(v, np.ndarray) and v.dtype != object:
            v = v.astype(object)
        self._a[_unwrap(k)] = v

    def reshape(self, *shape):
        return EncArray(self._a.reshape(*shape))

    def flatten(self):
        return EncArray(self._a.flatten())

    def copy(self):
        return EncArray(self._a.copy())

    def astype(self, _dtype):
        return self

    def __array_ufunc__(self, ufunc, method, *inputs, **kw):
        if method != "__call__" or ufunc not in _UFUNC_OPS:
            return NotImplemented
        return _elementwise(_UFUNC_OPS[ufunc], *inputs)

    def __array_function__(self, func, types, args, kwargs):
        return _wrap_result(func(*_unwrap(args), **{k: _unwrap(v) for k, v in kwargs.items()}))

    def _bin(fn, swap=False):  # noqa: N805
        def method(self, o):
            return _elementwise((lambda a, b: fn(b, a)) if swap else fn, self, o)
        return method

    __add__, __radd__ = _bin(_op.add), _bin(_op.add, True)
    __sub__, __rsub__ = _bin(_op.sub), _bin(_op.sub, True)
    __mul__, __rmul__ = _bin(_op.mul), _bin(_op.mul, True)
    __floordiv__, __mod__, __rshift__ = _bin(_op.floordiv), _bin(_op.mod), _bin(_op.rshift)
    __and__, __rand__ = _bin(_op.and_), _bin(_op.and_, True)
    __or__, __ror__ = _bin(_op.or_), _bin(_op.or_, True)
    __xor__, __rxor__ = _bin(_op.xor), _bin(_op.xor, True)
    __lt__, __le__, __gt__, __ge__ = _bin(_op.lt), _bin(_op.le), _bin(_op.gt), _bin(_op.ge)
    __eq__, __ne__ = _bin(_op.eq), _bin(_op.ne)

    def __neg__(self):
        return _elementwise(_op.neg, self)

    def __abs__(self):
        return _elementwise(abs, self)


def zeros(shape):
    if _S["mode"] is None:   # outside a trace Concrete's constructors are numpy's (the reference's plaintext path)
        return np.zeros(shape, dtype=np.int64)
    a = np.empty(shape, dtype=object)
    a.fill(0)
    return EncArray(a)


def ones(shape):
    if _S["mode"] is None:
        return np.ones(shape, dtype=np.int64)
    a = np.empty(shape, dtype=object)
    a.fill(1)
    return EncArray(a)


def univariate(f):
    """fhe.univariate(f)(x): an arbitrary table look-up (base_p_arrays.py:365)"""
    def apply(x):
        g = lambda t: int(f(np.int64(t)))  # noqa: E731
        if isinstance(x, Enc):
            return x._lut(g)
        if isinstance(x, np.ndarray) and x.dtype != object:
            return np.array([g(t) for t in x.reshape(-1)], dtype=np.int64).reshape(x.shape)
        if isinstance(x, (np.ndarray, EncArray)):
            return _elementwise(lambda t: t._lut(g) if isinstance(t, Enc) else g(t), x)
        return g(x)
    return apply


# ------------------------------------------------------------------------------------------ the two-phase tracer
def trace(fn, input_ranges, inputset, msg_bits=6, fuse=False, bitwidth=False):
    """fn(*arrays) -> array/scalar/tuple of them; input_ranges: per argument a list of (lo, hi) per element;
    inputset: list of argument tuples (lists of ints).  Returns (circuit, n_outputs)."""
    import os
    import sys
    pkg_root = os.path.abspath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "..", ".."))
    if pkg_root not in sys.path:          # <pkg root>/bmi_amd/compat/concrete/fhe/__init__.py
        sys.path.insert(0, pkg_root)
    from bmi_amd.circuit import Circuit

    def flat(res):
        if isinstance(res, tuple):
            return [e for r in res for e in flat(r)]
        if isinstance(res, (np.ndarray, EncArray)):
            return list(np.asarray(_unwrap(res), dtype=object).reshape(-1))
        return [res]

    # 1. measure
    _S.update(mode="measure", ranges=[], site=0, circuit=None)
    samples = np.array([[v for arg in one for v in arg] for one in inputset], dtype=np.int64)   # [S][n_in]
    args, col = [], 0
    for rng in input_ranges:
        a = np.empty(len(rng), dtype=object)
        for i in range(len(rng)):
            a[i] = Enc(samples[:, col].copy())
            col += 1
        args.append(EncArray(a))
    _S["site"] = 0
    fn(*args)
    n_sites = _S["site"]
    # 2. build
    c = Circuit(msg_bits=msg_bits)
    _S.update(mode="build", site=0, circuit=c, fuse=bool(fuse), bitwidth=bool(bitwidth))
    args = []
    for rng in input_ranges:
        a = np.empty(len(rng), dtype=object)
        for i, (lo, hi) in enumerate(rng):
            a[i] = Enc(c.input(lo, hi))
        args.append(EncArray(a))
    outs = flat(fn(*args))
    assert _S["site"] == n_sites, "the function is not data-oblivious: different operation sequence"
    c.set_outputs([_mat(Enc._raw(o)) for o in outs])
    _S.update(mode=None, fuse=False, bitwidth=False)
    return c, len(outs)


# ------------------------------------------------------------------- the Compiler / Circuit surface of concrete.fhe
# What the reference drives (main.py:53-86, qfloat_matrix_inversion.py:989-1053, tests/test_qfloat_fhe.py:136-175):
# fhe.Compiler(fn, {"x": "encrypted", ...}).compile(inputset, configuration, verbose) -> a circuit with keygen /
# encrypt / run / decrypt / simulate.  Here the circuit is this repo's IR; `run` evaluates it with the plaintext
# simulator (this shim only exists where the reference does, i.e. off the GPU box) and, with ENCSHIM_RECORD=<file>,
# every compiled circuit and every input it was run on are written out as data, for the GPU suite to replay on
# ciphertexts (tests/golden/ref_own_fhe_tests.json.gz, tools/gen_ref_fhe_tests.py).
class Configuration:
    """fhe.Configuration: `p_error` / `global_p_error` bound the probability that noise makes a result wrong (global: the whole
    circuit, which is what this front end budgets - a per-look-up `p_error` is turned into a global one by the look-up count);
    `security_level=128` restricts the parameter choice to the 128-bit-secure set.  Concrete's other options (key cache, dataflow,
    graph printing, precision strategy: qfloat_matrix_inversion.py:995-1002) are accepted and have no counterpart here."""

    def __init__(self, p_error=None, global_p_error=None, security_level=None, **options):
        self.p_error, self.global_p_error, self.security_level = p_error, global_p_error, security_level
        self.options = options


_RECORDED = []
_ENGINES = {}


def _backend():
    """gpu (default): encrypt / run / decrypt on LWE ciphertexts through the engine; simulate: plaintext evaluation of the same
    circuit (BMI_COMPAT_BACKEND, or the older ENCSHIM_BACKEND)"""
    return os.environ.get("BMI_COMPAT_BACKEND") or os.environ.get("ENCSHIM_BACKEND") or "gpu"


def _inner_name(fn):
    """the name of the function a `lambda x, y: circuit_function(x, y, params)` wrapper closes over"""
    for cell in (fn.__closure__ or ()):
        try:
            v = cell.cell_contents
        except ValueError:
            continue
        if callable(v) and hasattr(v, "__name__") and v.__name__ != "<lambda>":
            return v.__name__
    return getattr(fn, "__name__", "circuit")


class Circuit:
    def __init__(self, fn, inputset, fuse=False, configuration=None):
        self.configuration = configuration or Configuration()
        samples = [[np.asarray(a) for a in one] for one in inputset]
        self.shapes = [a.shape for a in samples[0]]
        flat_set = [tuple([int(v) for v in a.reshape(-1)] for a in one) for one in samples]
        ranges = []
        for k in range(len(self.shapes)):
            cols = np.array([one[k] for one in flat_set], dtype=np.int64)
            lo, hi = cols.min(axis=0), cols.max(axis=0)
            # like Concrete, the bit width of an input comes from the inputset; values in {-1, 0, 1} (digits of base 2,
            # signs) always get the whole of it, so that a test input of the other sign is inside the claim
            sign_like = cols.min() < 0 and cols.max() <= 1 and cols.min() >= -1
            ranges.append([(-1, 1) if sign_like else (min(int(a), 0), max(int(b), 1)) for a, b in zip(lo, hi)])
        self._out_shape = None

        def flat_fn(*flat_args):
            res = fn(*[EncArray(a._a.reshape(shape)) for a, shape in zip(flat_args, self.shapes)])
            if isinstance(res, (np.ndarray, EncArray)):
                self._out_shape = res.shape
            return res
        probe, _ = trace(flat_fn, ranges, flat_set, msg_bits=8, fuse=fuse, bitwidth=True)
        bits = max([4] + [p for p, _ in probe.luts])
        self.circuit, self.n_out = trace(flat_fn, ranges, flat_set, msg_bits=bits, fuse=fuse, bitwidth=True)
        self.name = _inner_name(fn)
        self.runs = []
        if os.environ.get("ENCSHIM_RECORD"):
            _RECORDED.append(self)

    # ---- back ends.  Default: encrypt / run / decrypt work on LWE ciphertexts through the engine (bmi_amd.tfhe.Engine + executor:
    # every look-up a programmable bootstrap on the MI355X).  The parameter set is the first one of the chosen modulus whose error
    # budget for this circuit meets the configuration's global_p_error (4-bit look-ups: N = 1024, or N = 2048 where the look-up
    # count needs the margin; 5 bits: N = 2048; 6 bits: N = 4096; with security_level=128 the secure128_torus sets, N = 4096 for 5 bits).  BMI_COMPAT_BACKEND=simulate: plaintext.
    def program(self):
        if getattr(self, "_prog", None) is None:
            from bmi_amd.program import Program
            self._prog = Program.from_circuit(self.circuit, meta={"kind": "compat", "function": self.name})
        return self._prog

    def _p_error_target(self):
        cfg = self.configuration
        if cfg.global_p_error is not None:
            return float(cfg.global_p_error)
        if cfg.p_error is not None:          # per look-up -> whole circuit
            return min(1.0, float(cfg.p_error) * max(len(self.circuit.nodes), 1))
        return 1e-5                          # Concrete's default global_p_error

    def choose_parameters(self):
        """(tfhe.Params, error-budget report) for this circuit under its configuration"""
        from bmi_amd import error_budget, tfhe
        qb = int(os.environ.get("BMI_COMPAT_Q_BITS", str(tfhe.TORUS64)))
        prog = self.program()
        secure = self.configuration.security_level is not None
        if secure and int(self.configuration.security_level) != 128:
            raise ValueError("security_level: only 128 has a parameter set (secure128_torus / secure128)")
        return error_budget.choose_params(prog, self._p_error_target(), q_bits=qb, secure=secure)

    def _gpu(self):
        if getattr(self, "_ex", None) is None:
            from bmi_amd import tfhe
            from bmi_amd.executor import Executor
            P, self.error_budget = self.choose_parameters()
            key = (int(P.q_bits), int(P.log_N), int(P.n), float(P.lwe_noise))
            if key not in _ENGINES:
                eng = tfhe.Engine(P)
                seed = os.environ.get("BMI_COMPAT_KEY_SEED") or os.environ.get("ENCSHIM_KEY_SEED")
                eng.keygen(int(seed) if seed else None)        # CSPRNG keys unless the test-only seed is given
                _ENGINES[key] = eng
            self._eng = _ENGINES[key]
            self._ex = Executor(self.program(), self._eng)
            self._dl = self._eng.delta_log(self.circuit.msg_bits)
        return self._ex

    def keygen(self, *a, **k):
        if _backend() == "gpu":
            self._gpu()

    def encrypt(self, *args):
        flat = [int(v) for a in args for v in np.asarray(a).reshape(-1)]
        if _backend() == "gpu":
            self._gpu()
            enc = PublicArguments([self._eng.encrypt(flat, self._dl)])
            enc.plain = flat          # kept beside the ciphertexts only for ENCSHIM_RECORD
            return enc
        return PublicArguments(flat)

    def run(self, enc):
        if _backend() == "gpu":
            res = PublicResult([self._gpu().run(enc[0])])
            res.plain_inputs = getattr(enc, "plain", None)
            return res
        out = self.circuit.simulate(list(enc))
        self.runs.append({"inputs": list(enc), "outputs": [int(v) for v in out]})
        return PublicResult(out)

    def decrypt(self, res):
        if _backend() == "gpu":
            r = np.array(self._eng.decrypt(res[0], self._dl), dtype=np.int64)
            if getattr(res, "plain_inputs", None) is not None:
                self.runs.append({"inputs": res.plain_inputs, "outputs": [int(v) for v in r]})
        else:
            r = np.array(list(res), dtype=np.int64)
        return r.reshape(self._out_shape) if self._out_shape is not None else r

    def simulate(self, *args):
        flat = [int(v) for a in args for v in np.asarray(a).reshape(-1)]
        r = np.array(self.circuit.simulate(flat), dtype=np.int64)
        return r.reshape(self._out_shape) if self._out_shape is not None else r

    def encrypt_run_decrypt(self, *args):
        return self.simulate(*args)


class Compiler:
    def __init__(self, function, parameter_encryption_statuses):
        self.function, self.statuses = function, dict(parameter_encryption_statuses)

    def compile(self, inputset, configuration=None, verbose=False, **_):
        return Circuit(self.function, list(inputset), configuration=configuration)


def _dump_recorded():
    path = os.environ.get("ENCSHIM_RECORD")
    if not path or not _RECORDED:
        return
    import gzip
    import json
    cases = [{"function": c.name, "msg_bits": c.circuit.msg_bits, "pbs": len(c.circuit.nodes), "depth": len(c.circuit.levels()),
              "input_shapes": [list(s) for s in c.shapes], "output_shape": list(c._out_shape or ()),
              "circuit": c.circuit.to_dict(), "runs": c.runs} for c in _RECORDED if c.runs]
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(json.dumps({"cases": cases}).encode())


import atexit  # noqa: E402
import os  # noqa: E402

atexit.register(_dump_recorded)


class PublicArguments(list):
    """what circuit.encrypt returns (main.py:73-76 annotates with it)"""


class PublicResult(list):
    """what circuit.run returns (main.py:78-81)"""

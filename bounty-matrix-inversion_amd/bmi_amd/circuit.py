"""Circuit IR: the host-side scheduler that turns QFloat arithmetic into batched PBS levels.

The reference traces its Python code with concrete-python's `Tracer` and lets Concrete compile and run the
graph (matrix_inversion/main.py:53-66,81).  Here the same role is played by:

  * `Lin`     — a symbolic encrypted scalar: an integer linear combination  const + sum coef_i * leaf_i  of
                *leaves* (circuit inputs and PBS outputs) with a guaranteed value interval [lo, hi].
                + - and constant * stay linear (no PBS), exactly as on a Tracer.
  * `Circuit` — records every table look-up (`lut`, `lut2`, `mul`) as a PBS node whose input is a `Lin`,
                assigns each node its ASAP level (1 + the deepest leaf it reads) and exposes, per level, the
                CSR description (row_ptr / idx / coef / const) the GPU executor feeds to
                bmi_lincomb_batch + bmi_pbs_batch.  Identical look-ups are shared (CSE).
  * `simulate`— plaintext evaluation of the recorded graph (the analogue of `circuit.simulate`,
                main.py:107); it also checks every interval claim, so it doubles as the range checker.

Encoding: every ciphertext carries a signed message m at scale 2^(q_bits - 1 - MSG_BITS) (MSG_BITS = 4: a PBS input
must lie in [-8, 7]; between PBS, values may range over [-16, 15]).  A look-up whose input interval is
narrower than 16 values is evaluated with a coarser message space (input multiplied by 2^(4-p)), which
widens the decision boxes and makes mod-switch failures vanishingly rare.
"""
from __future__ import annotations

import numpy as np

MSG_BITS = 4
DELTA_LOG = 63 - MSG_BITS  # 59 for the 64-bit modulus; engines report theirs (tfhe.Engine.delta_log())


class RangeError(ValueError):
    pass


class Lin:
    __slots__ = ("c", "terms", "const", "lo", "hi")

    def __init__(self, c, terms, const, lo, hi):
        self.c, self.terms, self.const, self.lo, self.hi = c, terms, const, lo, hi

    # ---- linear algebra (no PBS) -----------------------------------------------------------------
    def _coerce(self, o):
        if isinstance(o, Lin):
            return o
        if isinstance(o, (int, np.integer, bool, np.bool_)):
            return Lin(self.c, {}, int(o), int(o), int(o))
        return None

    def __add__(self, o):
        o = self._coerce(o)
        if o is None:
            return NotImplemented
        if not o.terms:
            return Lin(self.c, self.terms, self.const + o.const, self.lo + o.const, self.hi + o.const)
        if not self.terms:
            return Lin(self.c, o.terms, self.const + o.const, o.lo + self.const, o.hi + self.const)
        a, b = (self.terms, o.terms) if len(self.terms) >= len(o.terms) else (o.terms, self.terms)
        t = dict(a)
        for k, v in b.items():
            nv = t.get(k, 0) + v
            if nv:
                t[k] = nv
            else:
                del t[k]
        r = Lin(self.c, t, self.const + o.const, self.lo + o.lo, self.hi + o.hi)
        if not t:
            r.lo = r.hi = r.const
        return r

    __radd__ = __add__

    def __neg__(self):
        return Lin(self.c, {k: -v for k, v in self.terms.items()}, -self.const, -self.hi, -self.lo)

    def __sub__(self, o):
        o = self._coerce(o)
        if o is None:
            return NotImplemented
        return self + (-o)

    def __rsub__(self, o):
        return (-self) + o

    def __mul__(self, o):
        if isinstance(o, Lin):
            return self.c.mul(self, o)
        if isinstance(o, (int, np.integer, bool, np.bool_)):
            k = int(o)
            if k == 0:
                return self.c.const(0)
            if k == 1:
                return self
            lo, hi = (self.lo * k, self.hi * k) if k > 0 else (self.hi * k, self.lo * k)
            return Lin(self.c, {t: v * k for t, v in self.terms.items()}, self.const * k, lo, hi)
        return NotImplemented

    __rmul__ = __mul__

    # ---- helpers ---------------------------------------------------------------------------------
    @property
    def is_const(self):
        return not self.terms

    def assume(self, lo, hi):
        """Tighten the interval with knowledge the interval arithmetic cannot derive (checked by simulate)."""
        lo, hi = max(lo, self.lo), min(hi, self.hi)
        if lo > hi:
            raise RangeError(f"empty interval after assume: [{lo}, {hi}]")
        if self.is_const:
            return self
        r = Lin(self.c, self.terms, self.const, lo, hi)
        self.c.claims.append((self.c._snapshot(r), lo, hi))
        return r

    def lut(self, fn):
        return self.c.lut(self, fn)

    def lt0(self):
        return self.c.lut(self, _LT0)

    def ge0(self):
        return self.c.lut(self, _GE0)

    def gt0(self):
        return self.c.lut(self, _GT0)

    def eq0(self):
        return self.c.lut(self, _EQ0)

    def ne0(self):
        return self.c.lut(self, _NE0)

    def __repr__(self):
        return f"Lin(const={self.const}, {len(self.terms)} terms, [{self.lo},{self.hi}])"


def _LT0(v):
    return int(v < 0)


def _GE0(v):
    return int(v >= 0)


def _GT0(v):
    return int(v > 0)


def _EQ0(v):
    return int(v == 0)


def _NE0(v):
    return int(v != 0)


_SCALARS = (int, bool, str, float, tuple, frozenset, type(None))


def _fn_key(fn, depth=0):
    """Hashable identity of a look-up function by VALUE: code object + closure cells + defaults (plain scalars or nested
    functions of the same kind); None when anything else is captured.  Two lambdas created by the same line with the same
    captured numbers are the same function of v, so their tables can be shared without calling them 16 times each."""
    import types
    if not isinstance(fn, types.FunctionType) or depth > 3:    # bound methods, partials, callables with state: no identity by value
        return None
    code = fn.__code__
    cells = []
    for cell in fn.__closure__ or ():
        try:
            v = cell.cell_contents
        except ValueError:
            return None
        if isinstance(v, _SCALARS):
            cells.append(v)
        elif callable(v):
            k = _fn_key(v, depth + 1)
            if k is None:
                return None
            cells.append(k)
        else:
            return None
    kwd = tuple(sorted((fn.__kwdefaults__ or {}).items()))
    for v in tuple(fn.__defaults__ or ()) + tuple(x for _, x in kwd):
        if not isinstance(v, _SCALARS):
            return None
    key = (code, tuple(cells), fn.__defaults__, kwd)
    try:
        hash(key)            # a tuple cell may hold something unhashable: no memo then
    except TypeError:
        return None
    return key


class Circuit:
    def __init__(self, msg_bits=MSG_BITS):
        """msg_bits: message bits of every ciphertext of this circuit (look-ups take 2^msg_bits values, lut_odd twice
        that).  4 is what N = 1024 carries at 6.2 sigma (the QFloat layer uses it); the N = 2048 / 4096 parameter sets
        carry 5 / 6 at the same margin."""
        self.msg_bits = int(msg_bits)
        self.n_inputs = 0
        self.leaf_level = []     # per leaf
        self.leaf_lo = []
        self.leaf_hi = []
        self.nodes = []          # PBS nodes, in creation order: (terms tuple, const, lut_index, out_leaf)
        self.luts = []           # (p, table tuple)
        self._lut_index = {}
        self._cse = {}
        self.claims = []         # (snapshot, lo, hi) interval claims to be verified by simulate()
        self.outputs = []        # snapshots of output Lins
        self.stats = {"pbs": 0, "cse_hits": 0, "const_folds": 0}
        self.wide_leaves = set()  # outputs of lut_odd() / lut_neg(): their PBS input may use the whole torus
        self.half_leaves = set()  # outputs of lut_neg(): the ciphertext carries (bit - 1/2) * Delta (see lut_neg)
        self.half_luts = set()    # ... and their table (-+1) is registered at half the output scale
        self._table_memo = {}     # (function key, lo, hi) -> evaluated table data of lut()

    # ---- construction ----------------------------------------------------------------------------
    def input(self, lo, hi):
        leaf = len(self.leaf_level)
        if leaf != self.n_inputs:
            raise ValueError("declare all inputs before the first look-up")
        self.n_inputs += 1
        self.leaf_level.append(0)
        self.leaf_lo.append(lo)
        self.leaf_hi.append(hi)
        return Lin(self, {leaf: 1}, 0, lo, hi)

    def const(self, v):
        return Lin(self, {}, int(v), int(v), int(v))

    def _snapshot(self, x):
        return (tuple(sorted(x.terms.items())), x.const)

    def level_of(self, x):
        return max((self.leaf_level[t] for t in x.terms), default=0)

    def lut(self, x, fn):
        """Univariate table look-up f(x): one PBS (or a constant fold)."""
        if not isinstance(x, Lin):
            return self.const(fn(int(x)))
        if x.is_const:
            self.stats["const_folds"] += 1
            return self.const(fn(x.const))
        width = x.hi - x.lo + 1
        if width > (1 << self.msg_bits):
            raise RangeError(f"look-up input interval [{x.lo}, {x.hi}] wider than {1 << self.msg_bits} values")
        fk = _fn_key(fn)
        memo = self._table_memo.get((fk, x.lo, x.hi)) if fk is not None else None
        if memo is None:
            vals = [int(fn(v)) for v in range(x.lo, x.hi + 1)]
            olo, ohi = min(vals), max(vals)
            p = 1
            while (1 << p) < width:
                p += 1
            half = 1 << (p - 1)
            # table over m = x - off in [-half, half); entries outside the reachable interval repeat the nearest reachable value
            table = tuple(vals[min(max(m + half, 0), width - 1)] for m in range(-half, half))
            memo = (olo, ohi, p, table)
            if fk is not None:
                self._table_memo[(fk, x.lo, x.hi)] = memo
        olo, ohi, p, table = memo
        if olo == ohi:
            self.stats["const_folds"] += 1
            return self.const(olo)
        if not (-(1 << self.msg_bits) <= olo and ohi < (1 << self.msg_bits)):
            raise RangeError(f"look-up output interval [{olo}, {ohi}] does not fit the message space")
        half = 1 << (p - 1)
        off = x.lo + half
        key = (p, table)
        li = self._lut_index.get(key)
        if li is None:
            li = len(self.luts)
            self.luts.append(key)
            self._lut_index[key] = li
        scale = 1 << (self.msg_bits - p)
        pin = (x - off) * scale
        snap = self._snapshot(pin)
        ck = (snap, li)
        leaf = self._cse.get(ck)
        if leaf is None:
            leaf = len(self.leaf_level)
            self.leaf_level.append(self.level_of(x) + 1)
            self.leaf_lo.append(olo)
            self.leaf_hi.append(ohi)
            self.nodes.append((snap[0], snap[1], li, leaf))
            self._cse[ck] = leaf
            self.stats["pbs"] += 1
        else:
            self.stats["cse_hits"] += 1
        return Lin(self, {leaf: 1}, 0, olo, ohi)

    def lut_odd(self, x, fn):
        """Look-up on a (msg_bits + 1)-bit input for functions with f(v - 2^msg_bits) = -f(v) (sign-like functions).

        A msg_bits-bit message occupies half of the torus; the negacyclic test polynomial returns -f(v - 16) for an
        input v in [8, 16) and -f(v + 16) for v in [-16, -8).  For a function with exactly that symmetry the look-up is
        therefore correct on the whole interval [-15, 15] with the ordinary 16-entry table, the same box width and
        the same noise margin as any 4-bit look-up.  This packs FOUR binary digit differences (8 d3 + 4 d2 + 2 d1 + d0)
        or THREE ternary look-ahead signals (9 s2 + 3 s1 + s0) into one PBS.  The symmetry is verified here on every
        reachable value; `simulate` evaluates with the wrap-around rule."""
        if not isinstance(x, Lin):
            return self.const(fn(int(x)))
        if x.is_const:
            self.stats["const_folds"] += 1
            return self.const(fn(x.const))
        half = 1 << (self.msg_bits - 1)
        period = 1 << self.msg_bits      # f(v - period) = -f(v); the torus holds 2 * period boxes
        if x.lo < -(period - 1) or x.hi > period - 1:
            raise RangeError(f"wide look-up input interval [{x.lo}, {x.hi}] outside [-{period - 1}, {period - 1}]")
        if -half <= x.lo and x.hi < half:
            return self.lut(x, fn)  # fits the ordinary message space
        table = [None] * (2 * half)
        for v in range(x.lo, x.hi + 1):
            fv = int(fn(v))
            if -half <= v < half:
                m, want = v, fv
            elif v >= half:
                m, want = v - period, -fv          # slot of v - 16, which must hold -f(v)
            else:
                m, want = v + period, -fv
            if table[m + half] is None:
                table[m + half] = want
            elif table[m + half] != want:
                raise RangeError(f"lut_odd: f({v}) = {fv} breaks f(v - {period}) = -f(v)")
        known = [t for t in table if t is not None]
        vals = [int(fn(v)) for v in range(x.lo, x.hi + 1)]
        olo, ohi = min(vals), max(vals)
        if olo == ohi:
            self.stats["const_folds"] += 1
            return self.const(olo)
        last = known[0]
        for i in range(len(table)):       # unreachable slots repeat a neighbouring reachable value
            if table[i] is None:
                table[i] = last
            else:
                last = table[i]
        key = (self.msg_bits, tuple(table))
        li = self._lut_index.get(key)
        if li is None:
            li = len(self.luts)
            self.luts.append(key)
            self._lut_index[key] = li
        snap = self._snapshot(x)
        ck = (snap, li, "wide")
        leaf = self._cse.get(ck)
        if leaf is None:
            leaf = len(self.leaf_level)
            self.leaf_level.append(self.level_of(x) + 1)
            self.leaf_lo.append(olo)
            self.leaf_hi.append(ohi)
            self.nodes.append((snap[0], snap[1], li, leaf))
            self._cse[ck] = leaf
            self.wide_leaves.add(leaf)
            self.stats["pbs"] += 1
        else:
            self.stats["cse_hits"] += 1
        return Lin(self, {leaf: 1}, 0, olo, ohi)

    def lut_neg(self, x):
        """[x < 0] in {0, 1} for x anywhere in [-(2^msg_bits - 1), 2^msg_bits - 1], ONE PBS.

        lut_odd() covers the whole torus only for functions with f(v - 16) = -f(v), which a bit is not.  But
        [v < 0] - 1/2 is: +1/2 on [-16, 0), -1/2 on [0, 16) - the constant test polynomial of the classic sign bootstrap.  The
        PBS therefore returns (bit - 1/2) Delta (a table of -+1 at HALF the output scale) and the missing Delta / 2 is a
        constant, which every consumer's linear combination absorbs into its own constant (executor.py; Program keeps
        which leaves are of this kind).  On the tracer's side the leaf simply has the value of the bit.  Same box width and
        noise margin as any msg_bits-bit look-up (the decision boundaries are v = -1 | 0 and the wrap at +-16, where the
        inputs +-15 are two boxes apart).  This turns 'signal, then convert the signal to a bit' (two look-up levels) into
        one level wherever a borrow / carry / comparison bit comes out of a packed window or of packed look-ahead signals."""
        if not isinstance(x, Lin):
            return self.const(int(int(x) < 0))
        if x.is_const:
            self.stats["const_folds"] += 1
            return self.const(int(x.const < 0))
        half = 1 << (self.msg_bits - 1)
        period = 1 << self.msg_bits
        if x.lo < -(period - 1) or x.hi > period - 1:
            raise RangeError(f"lut_neg input interval [{x.lo}, {x.hi}] outside [-{period - 1}, {period - 1}]")
        if x.lo >= 0 or x.hi < 0:
            self.stats["const_folds"] += 1
            return self.const(int(x.hi < 0))
        if -half <= x.lo and x.hi < half:
            return self.lut(x, _LT0)  # fits the ordinary message space (and possibly a coarser one: wider boxes)
        table = tuple([1] * half + [-1] * half)     # in half units: (bit - 1/2) * 2 on m = -half .. half - 1
        # a half-scale table has its own identity: an ordinary look-up with the same entries (f(v) = 1 if v < 0 else -1) must not
        # share the index, or the executor would register it at half the output scale too
        key = (self.msg_bits, table, "half")
        li = self._lut_index.get(key)
        if li is None:
            li = len(self.luts)
            self.luts.append((self.msg_bits, table))
            self._lut_index[key] = li
        self.half_luts.add(li)
        snap = self._snapshot(x)
        ck = (snap, li, "neg")
        leaf = self._cse.get(ck)
        if leaf is None:
            leaf = len(self.leaf_level)
            self.leaf_level.append(self.level_of(x) + 1)
            self.leaf_lo.append(0)
            self.leaf_hi.append(1)
            self.nodes.append((snap[0], snap[1], li, leaf))
            self._cse[ck] = leaf
            self.wide_leaves.add(leaf)
            self.half_leaves.add(leaf)
            self.stats["pbs"] += 1
        else:
            self.stats["cse_hits"] += 1
        return Lin(self, {leaf: 1}, 0, 0, 1)

    def lut2(self, x, y, fn):
        """Bivariate look-up f(x, y) as ONE PBS on the packed value (x - xlo) * ny + (y - ylo)."""
        if not isinstance(x, Lin):
            x = self.const(x)
        if not isinstance(y, Lin):
            y = self.const(y)
        if x.is_const and y.is_const:
            return self.const(fn(x.const, y.const))
        if x.is_const:
            return self.lut(y, lambda v, a=x.const: fn(a, v))
        if y.is_const:
            return self.lut(x, lambda v, b=y.const: fn(v, b))
        nx, ny = x.hi - x.lo + 1, y.hi - y.lo + 1
        if nx * ny > (1 << self.msg_bits):
            raise RangeError(f"bivariate look-up needs {nx}x{ny} > {1 << self.msg_bits} packed values")
        xlo, ylo = x.lo, y.lo
        z = (x - xlo) * ny + (y - ylo)
        z.lo, z.hi = 0, nx * ny - 1
        return self.lut(z, lambda v: fn(v // ny + xlo, v % ny + ylo))

    def mul(self, x, y):
        """ciphertext x ciphertext product: packed bivariate PBS when the ranges allow it, otherwise the
        quarter-square identity xy = floor((x+y)^2/4) - floor((x-y)^2/4) (two PBS)."""
        if not isinstance(x, Lin):
            return y * x
        if not isinstance(y, Lin):
            return x * y
        if x.is_const:
            return y * x.const
        if y.is_const:
            return x * y.const
        nx, ny = x.hi - x.lo + 1, y.hi - y.lo + 1
        if nx * ny <= (1 << self.msg_bits):
            return self.lut2(x, y, lambda a, b: a * b)
        if nx + ny - 1 <= (1 << self.msg_bits):
            return self.lut(x + y, lambda s: (s * s) // 4) - self.lut(x - y, lambda d: (d * d) // 4)
        raise RangeError(f"product of intervals [{x.lo},{x.hi}] x [{y.lo},{y.hi}] does not fit")

    def select(self, bit, x, y):
        """bit ? x : y  (bit in {0,1}) = y + bit * (x - y), one PBS."""
        r = y + self.mul(bit, x - y)
        lo = min(_lo(x), _lo(y))
        hi = max(_hi(x), _hi(y))
        return r.assume(lo, hi) if isinstance(r, Lin) else r

    # ---- (de)serialisation: a circuit as plain data (fixtures of traced functions, tests/golden/) -------------------
    def to_dict(self):
        return {"msg_bits": self.msg_bits, "n_inputs": self.n_inputs,
                "leaf_level": list(self.leaf_level), "leaf_lo": list(self.leaf_lo), "leaf_hi": list(self.leaf_hi),
                "nodes": [[[list(t) for t in terms], const, li, leaf] for terms, const, li, leaf in self.nodes],
                "luts": [[p, list(tab)] for p, tab in self.luts],
                "outputs": [[[list(t) for t in terms], const] for terms, const in self.outputs],
                "claims": [[[[list(t) for t in snap[0]], snap[1]], lo, hi] for snap, lo, hi in self.claims],
                "wide_leaves": sorted(self.wide_leaves), "half_leaves": sorted(self.half_leaves), "half_luts": sorted(self.half_luts)}

    @classmethod
    def from_dict(cls, d):
        c = cls(msg_bits=d["msg_bits"])
        c.n_inputs = d["n_inputs"]
        c.leaf_level, c.leaf_lo, c.leaf_hi = list(d["leaf_level"]), list(d["leaf_lo"]), list(d["leaf_hi"])
        c.nodes = [(tuple(tuple(t) for t in terms), const, li, leaf) for terms, const, li, leaf in d["nodes"]]
        c.luts = [(p, tuple(tab)) for p, tab in d["luts"]]
        half = set(d.get("half_luts", ()))
        c._lut_index = {(k + ("half",) if i in half else k): i for i, k in enumerate(c.luts)}
        c.outputs = [(tuple(tuple(t) for t in terms), const) for terms, const in d["outputs"]]
        c.claims = [((tuple(tuple(t) for t in snap[0]), snap[1]), lo, hi) for snap, lo, hi in d["claims"]]
        c.wide_leaves = set(d["wide_leaves"])
        c.half_leaves = set(d.get("half_leaves", ()))
        c.half_luts = set(d.get("half_luts", ()))
        c.stats["pbs"] = len(c.nodes)
        return c

    def set_outputs(self, lins):
        self.outputs = [self._snapshot(x if isinstance(x, Lin) else self.const(x)) for x in lins]
        self.out_ranges = [(_lo(x), _hi(x)) for x in lins]

    # ---- schedule --------------------------------------------------------------------------------
    # Capacity steps of one level on one MI355X: the latency kernel runs one workgroup per ciphertext
    # on 256 CUs, so a level costs one round per started 256 ciphertexts up to 512; wider levels take the throughput
    # kernel, whose round covers 1,024 ciphertexts.
    ROUND = 256
    WIDE_ROUND = 1024

    def asap_levels(self):
        """PBS nodes grouped by ASAP level: list (level 1..D) of lists of node indices."""
        depth = max(self.leaf_level, default=0)
        out = [[] for _ in range(depth)]
        for i, (_, _, _, leaf) in enumerate(self.nodes):
            out[self.leaf_level[leaf] - 1].append(i)
        return out

    def levels(self, balance=True):
        """The level schedule the executor runs: same depth as ASAP, but width-aware.  A level's cost is a step
        function of its width (kernel rounds), so nodes with slack (ALAP level later than ASAP) are moved out of levels
        that would otherwise spill into one more round: list scheduling in level order, critical nodes (ALAP = now)
        first, then the remaining ready nodes by ALAP while the rounds the critical ones need anyway have room.
        Measured shapes: 3x3 inverse 2.97 s -> 2.56 s estimated, 4x4 7.2 s -> 5.8 s, depth unchanged."""
        if not balance:
            return self.asap_levels()
        key = (len(self.nodes), len(self.leaf_level))
        if getattr(self, "_levels_cache", None) and self._levels_cache[0] == key:
            return self._levels_cache[1]
        import heapq
        nodes = self.nodes
        nn = len(nodes)
        if nn == 0:
            return []
        producer = {}
        for i, (_, _, _, leaf) in enumerate(nodes):
            producer[leaf] = i
        preds = [[producer[t] for t, _ in nodes[i][0] if t in producer] for i in range(nn)]
        succs = [[] for _ in range(nn)]
        for i, ps in enumerate(preds):
            for q in ps:
                succs[q].append(i)
        asap = [self.leaf_level[nodes[i][3]] for i in range(nn)]
        depth = max(asap)
        alap = [depth] * nn
        for i in sorted(range(nn), key=lambda j: -asap[j]):   # successors have larger ASAP levels: done first
            for sc in succs[i]:
                if alap[sc] - 1 < alap[i]:
                    alap[i] = alap[sc] - 1
        indeg = [len(ps) for ps in preds]
        # ready nodes bucketed by ALAP level.  Every level takes ALL ready nodes whose ALAP is the current level (a
        # node is never ready later than its ALAP), so the critical set of level t is exactly bucket[t].
        bucket = {}
        keys = []            # heap of ALAP values that have a non-empty bucket
        for i in range(nn):
            if indeg[i] == 0:
                if alap[i] not in bucket:
                    bucket[alap[i]] = []
                    heapq.heappush(keys, alap[i])
                bucket[alap[i]].append(i)
        out = []
        done = 0
        t = 0
        while done < nn:
            t += 1
            must = len(bucket.get(t, ()))
            if must <= 2 * self.ROUND:
                cap = max(1, -(-must // self.ROUND)) * self.ROUND
            else:
                cap = -(-must // self.WIDE_ROUND) * self.WIDE_ROUND
            level = []
            while keys and len(level) < cap:
                a = keys[0]
                b = bucket[a]
                room = cap - len(level)
                if len(b) <= room:
                    level.extend(b)
                    del bucket[a]
                    heapq.heappop(keys)
                else:
                    level.extend(b[-room:])
                    del b[-room:]
            released = []
            for i in level:
                for sc in succs[i]:
                    indeg[sc] -= 1
                    if indeg[sc] == 0:
                        released.append(sc)
            for sc in released:   # ready from the next level on
                a = alap[sc]
                if a not in bucket:
                    bucket[a] = []
                    heapq.heappush(keys, a)
                bucket[a].append(sc)
            done += len(level)
            level.sort()
            out.append(level)
        assert len(out) == depth, (len(out), depth)
        self._levels_cache = (key, out)
        return out

    def summary(self):
        lv = self.levels()
        widths = [len(x) for x in lv]
        return {"inputs": self.n_inputs, "pbs": len(self.nodes), "depth": len(lv), "luts": len(self.luts),
                "max_width": max(widths, default=0), "mean_width": float(np.mean(widths)) if widths else 0.0,
                "median_width": float(np.median(widths)) if widths else 0.0, **self.stats}

    # ---- plaintext evaluation (the analogue of circuit.simulate, main.py:107) -----------------------
    def simulate(self, inputs, check=True):
        inputs = [int(v) for v in inputs]
        if len(inputs) != self.n_inputs:
            raise ValueError("wrong number of inputs")
        val = [0] * len(self.leaf_level)
        for i, v in enumerate(inputs):
            if check and not (self.leaf_lo[i] <= v <= self.leaf_hi[i]):
                raise RangeError(f"input {i} = {v} outside its declared interval [{self.leaf_lo[i]}, {self.leaf_hi[i]}]")
            val[i] = v
        half_space = 1 << (self.msg_bits - 1)
        for terms, const, li, leaf in self.nodes:
            p, table = self.luts[li]
            x = const + sum(cf * val[t] for t, cf in terms)
            scale = 1 << (self.msg_bits - p)
            wide = leaf in self.wide_leaves
            if check and not ((-2 * half_space < x < 2 * half_space) if wide else (-half_space <= x < half_space)):
                raise RangeError(f"PBS input {x} outside the message space")
            if check and x % scale:
                raise RangeError("PBS input not a multiple of its scale")
            m = x // scale
            if wide and m >= half_space:        # the other half of the torus: negacyclic wrap-around
                out = -table[m - 2 * half_space + (1 << (p - 1))]
            elif wide and m < -half_space:
                out = -table[m + 2 * half_space + (1 << (p - 1))]
            else:
                out = table[m + (1 << (p - 1))]
            if leaf in self.half_leaves:        # lut_neg: the table holds (bit - 1/2) * 2
                out = (out + 1) // 2
            if check and not (self.leaf_lo[leaf] <= out <= self.leaf_hi[leaf]):
                raise RangeError("look-up output outside its interval")
            val[leaf] = out
        if check:
            for (terms, const), lo, hi in self.claims:
                x = const + sum(cf * val[t] for t, cf in terms)
                if not (lo <= x <= hi):
                    raise RangeError(f"interval claim [{lo}, {hi}] violated by value {x}")
        return [const + sum(cf * val[t] for t, cf in terms) for terms, const in self.outputs]


def _lo(x):
    return x.lo if isinstance(x, Lin) else int(x)


def _hi(x):
    return x.hi if isinstance(x, Lin) else int(x)

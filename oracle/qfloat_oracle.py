"""ORACLE (test infrastructure, not product code) — CPU restatement of the
reference's plaintext QFloat / LU-inverse algorithm (layers L1-L3 of SURVEY.md).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It is pinned against tests/golden/*.json, which were produced by
the reference itself (tools/gen_golden.py) — parity at the QFloat level is
therefore PINNED.  (Ciphertext-level parity with Concrete is unpinned: see
oracle/tfhe_oracle.c.)

All arithmetic is integer (numpy int64).  A number is a `Q` record:
digits (most-significant first), ints, base, sign in {-1, 0, +1}, tidy flag.
Each function cites the reference lines (relative to /root/reference) whose
behaviour it restates, quirks included.
"""
from __future__ import annotations

import numbers
import numpy as np

I64 = np.int64


# =============================================================================
# L1: digit-array primitives  (matrix_inversion/base_p_arrays.py)
# =============================================================================
def digits_to_int(d, p):
    """base_p_arrays.py:11-21 — MSD-first digits (may be signed) -> integer."""
    v = 0
    for x in np.asarray(d, dtype=object):
        v = v * p + int(x)
    return v


def int_to_digits(v, n, p):
    """base_p_arrays.py:24-48 — the top digit is NOT reduced mod p."""
    if n == 0:
        return np.zeros(0, I64)
    s = (v > 0) - (v < 0)
    v = abs(int(v))
    out = np.zeros(n, I64)
    for k in range(n):  # k = index from the left; weight p**(n-1-k)
        w = p ** (n - 1 - k)
        out[k] = v // w
        v -= int(out[k]) * w
    return out * s


def frac_to_digits(f, n, p):
    """base_p_arrays.py:62-81 — |f| < 1, greedy digit extraction in floating point."""
    s = float(np.sign(f))
    f = abs(f)
    assert 0 <= f < 1
    out = []
    while f and len(out) < n:
        f *= p
        dg = int(f)
        if dg > 0:
            f -= dg
        out.append(dg)
    out += [0] * (n - len(out))
    return s * np.array(out, dtype=float)  # reference returns float * array


def digits_to_frac(d, p):
    """base_p_arrays.py:51-59."""
    f = 0.0
    for k, x in enumerate(d):
        f += x * (p ** -(k + 1))
    return f


def sub_digits(a, b, p, overflow=False):
    """base_p_arrays.py:108-139 — borrow-chain a-b aligned on the right.

    Returns digits of size a.size (and, with overflow=True, the flag a<b computed
    as in :130-137 for unequal sizes)."""
    a = np.asarray(a, I64)
    b = np.asarray(b, I64)
    m = min(a.size, b.size)
    out = np.zeros(a.size, I64)
    borrow = 0
    for k in range(1, m + 1):
        t = int(a[-k]) - int(b[-k]) - borrow
        borrow = 1 if t < 0 else 0
        out[-k] = t + p * borrow
    if not overflow:
        return out
    extra = b.size - a.size
    if extra == 0:
        lt = borrow
    elif extra < 0:
        lt = borrow & int(np.sum(a[:-extra]) == 0)
        out[:-extra] = a[:-extra]
    else:
        lt = borrow | int(np.sum(b[:extra]) > 0)
    return out, lt


def div_digits(dividend, divisor, p):
    """base_p_arrays.py:173-203 — restoring long division, MSD first.

    Division by zero yields all (p-1) digits (falls out of the algorithm)."""
    dividend = np.asarray(dividend, I64)
    divisor = np.asarray(divisor, I64)
    quo = np.zeros(dividend.size, I64)
    rem = dividend[:1].copy()
    for k in range(dividend.size):
        if k > 0:
            drop = 1 if rem.size > divisor.size else 0
            rem = np.concatenate((rem[drop:], dividend[k:k + 1]))
        for _ in range(p - 1):
            diff, lt = sub_digits(rem, divisor, p, True)
            ge = 1 - lt
            rem = diff * ge + rem * lt
            quo[k] += ge
    return quo


def ge_digits(a, b):
    """base_p_arrays.py:245-260."""
    m = min(len(a), len(b))
    borrow = 0
    for k in range(1, m + 1):
        borrow = 1 if int(a[-k]) - int(b[-k]) - borrow < 0 else 0
    return 1 - borrow


def eq_digits(a, b):
    """base_p_arrays.py:276-280."""
    a = np.asarray(a)
    return int((a.size - int(np.sum(a == np.asarray(b)))) == 0)


# =============================================================================
# L2: number types  (matrix_inversion/qfloat.py)
# =============================================================================
class Zero:
    """qfloat.py:14-118 — compile-time zero."""

    def copy(self):
        return self

    def to_float(self):
        return 0.0

    def __add__(self, o):
        return self if isinstance(o, Zero) else o

    __radd__ = __add__

    def __sub__(self, o):
        return self if isinstance(o, Zero) else -o

    def __rsub__(self, o):
        return o

    def __mul__(self, o):
        return self

    __rmul__ = __mul__

    def __truediv__(self, o):
        if isinstance(o, Zero):
            raise ValueError("division by Zero")
        return self

    def __rtruediv__(self, o):
        raise ValueError("division by Zero")

    def __neg__(self):
        return self

    def neg(self):
        return self

    def __abs__(self):
        return self


class SignedBinary:
    """qfloat.py:120-243 — a value known to be in {-1, 0, +1}."""

    def __init__(self, value):
        self.value = value

    encrypted = False

    def copy(self):
        return SignedBinary(self.value)

    def to_float(self):
        return float(self.value)

    def __add__(self, o):
        if isinstance(o, SignedBinary):
            return self.value + o.value
        if isinstance(o, Q):
            return o.__add__(self)
        return self.value + o

    def __sub__(self, o):
        if isinstance(o, SignedBinary):
            return self.value - o.value
        if isinstance(o, Q):
            return o.__rsub__(self)
        return self.value - o

    def __mul__(self, o):
        if isinstance(o, SignedBinary):
            return SignedBinary(self.value * o.value)
        if isinstance(o, Q):
            return o.__mul__(self)
        return self.value * o

    def __truediv__(self, o):
        if isinstance(o, SignedBinary):
            return SignedBinary(self.value // o.value)
        if isinstance(o, Q):
            return o.__rtruediv__(self)
        return self.value / o

    def __neg__(self):
        return SignedBinary(-1 * self.value)

    def neg(self):
        self.value *= -1
        return self

    def __abs__(self):
        return SignedBinary(abs(self.value))


class Q:
    """qfloat.py:245-1376 — sign-magnitude fixed point; value = sign * sum d_i p^(ints-1-i)."""

    ADD = 0
    MUL = 0
    DIV = 0

    def __init__(self, digits, ints=None, base=2, tidy=True, sign=1):
        d = np.array(digits).astype(I64)  # qfloat.py:278-288 (copy)
        if d.ndim != 1:
            raise ValueError("array must be one dimension")
        if not (isinstance(base, (int, np.integer)) and base > 1):
            raise ValueError("base must be a int >1")
        if ints is None:
            ints = d.size // 2
        elif not (isinstance(ints, (int, np.integer)) and 0 <= ints <= d.size):
            raise ValueError("ints must be in range [0,array.size]")
        self.d, self.ints, self.base = d, int(ints), int(base)
        self.sign = int(sign) if isinstance(sign, (float, np.floating)) else sign
        self.tidy_flag = tidy
        if not tidy:
            self.base_tidy()

    # -- stats (qfloat.py:262-326) --------------------------------------------
    @classmethod
    def reset_stats(cls):
        cls.ADD = cls.MUL = cls.DIV = 0

    @classmethod
    def stats(cls):
        return [cls.ADD, cls.MUL, cls.DIV]

    # -- plaintext codecs (qfloat.py:336-410) ---------------------------------
    @classmethod
    def from_float(cls, f, length=10, ints=None, base=2):
        if ints is None:
            ints = length // 2
        ip = int(f)
        arr = np.zeros(length, I64)
        arr[:ints] = int_to_digits(ip, ints, base)
        arr[ints:] = frac_to_digits(f - ip, length - ints, base)  # float -> int64 truncation, as in the reference
        return cls(np.abs(arr), ints, base, True, np.sign(f) or 1)

    def to_float(self):
        ip = digits_to_int(self.d[:self.ints], self.base)
        fp = digits_to_frac(self.d[self.ints:], self.base)
        return (ip + fp) * self.sign

    def to_str(self, tidy=True):
        if tidy:
            self.base_tidy()
        nz = 1 if self.sign != 0 else 0
        a = [int(x) * nz for x in self.d[:self.ints]]
        b = [int(x) * nz for x in self.d[self.ints:]]
        if self.base <= 10:
            a, b = "".join(map(str, a)), "".join(map(str, b))
        else:
            a, b = str(np.array(a)), str(np.array(b))
        return ("" if self.sign >= 0 else "-") + a + "." + b

    __str__ = to_str

    # -- structure ------------------------------------------------------------
    def __len__(self):
        return self.d.size

    def copy(self):
        return Q(self.d.copy(), self.ints, self.base, self.tidy_flag, self.sign)

    def to_array(self):
        return self.d.copy()

    def set_len_ints(self, newlen, newints):
        """qfloat.py:565-589 — may truncate either part."""
        if self.ints != newints:
            if newints > self.ints:
                self.d = np.concatenate((np.zeros(newints - self.ints, I64), self.d))
            else:
                self.d = self.d[self.ints - newints:]
            self.ints = int(newints)
        extra = int(newlen - self.d.size)
        if extra > 0:
            self.d = np.concatenate((self.d, np.zeros(extra, I64)))
        elif extra < 0:
            self.d = self.d[:extra]
        return self

    def _compatible(self, o):
        if not isinstance(o, Q):
            raise ValueError("Object must also be a QFloat")
        if self.base != o.base:
            raise ValueError("bases are different")
        if len(self) != len(o):
            raise ValueError("different length")
        if self.ints != o.ints:
            raise ValueError("different dot index")

    # -- carry / sign normalisation ------------------------------------------
    def base_tidy(self):
        """qfloat.py:607-626 — carries truncate toward zero; carry out of digit 0 dropped."""
        if self.tidy_flag:
            return
        p = self.base
        carry = 0
        for k in range(self.d.size - 1, -1, -1):
            c = int(self.d[k]) + carry
            carry = (abs(c) // p) * ((c > 0) - (c < 0))
            self.d[k] = c - carry * p
        self.tidy_flag = True

    def tidy(self):
        """qfloat.py:648-673 — non-negative digits + sign (zero gets sign +1)."""
        self.base_tidy()
        pos = self.d * (self.d >= 0)
        neg = -1 * (self.d * (self.d < 0))
        pmn, isneg = sub_digits(pos, neg, self.base, True)
        nmp = sub_digits(neg, pos, self.base)
        self.d = (1 - isneg) * pmn + isneg * nmp
        self.sign = 2 * (1 - isneg) - 1

    # -- comparisons (qfloat.py:681-764) --------------------------------------
    def __eq__(self, o):
        self._compatible(o)
        if not (self.tidy_flag and o.tidy_flag):
            raise Exception("cannot compare QFloats that are not tidy")
        return eq_digits(self.d, o.d) & int(self.sign == o.sign)

    __hash__ = None

    def __gt__(self, o):
        self._compatible(o)
        self.base_tidy()
        o.base_tidy()
        same = int(self.sign == o.sign)
        mag_gt = 1 - ge_digits(o.d, self.d)
        flip = int(self.sign < 0) & (1 - eq_digits(self.d, o.d))
        return same * (mag_gt ^ flip) + (1 - same) * int(self.sign > o.sign)

    def __lt__(self, o):
        return o > self

    def __le__(self, o):
        return 1 - (self > o)

    def __ge__(self, o):
        return 1 - (o > self)

    def __abs__(self):
        r = self.copy()
        r.sign *= r.sign
        return r

    def abs(self):
        self.sign *= self.sign
        return self

    def __neg__(self):
        r = self.copy()
        r.sign *= -1
        return r

    def neg(self):
        self.sign *= -1
        return self

    # -- addition (qfloat.py:766-850) -----------------------------------------
    def __iadd__(self, o):
        if isinstance(o, Zero):
            return None  # reference quirk: `return` with no value (qfloat.py:803-804)
        Q.ADD += 1
        self.d = self.d * self.sign
        if isinstance(o, numbers.Integral):
            self.d[self.ints - 1] += o
        elif isinstance(o, SignedBinary):
            self.d[self.ints - 1] += o.value
        else:
            self._compatible(o)
            self.d = self.d + o.d * o.sign
        self.tidy_flag = False
        self.sign = None
        self.tidy()
        return self

    def __add__(self, o):
        r = self.copy()
        r += o
        return r

    __radd__ = __add__

    def __sub__(self, o):
        r = -o
        r += self
        return r

    def __rsub__(self, o):
        r = -self
        r += o
        return r

    # -- multiplication (qfloat.py:852-1021) ----------------------------------
    def __imul__(self, o):
        if isinstance(o, numbers.Integral):
            s = int(np.sign(o))
            self.d = self.d * (o * s)
            self.sign *= s
            self.tidy_flag = False
            self.base_tidy()
        elif isinstance(o, SignedBinary):
            self.sign *= o.value
        else:
            Q.MUL += 1
            self.base_tidy()
            o.base_tidy()
            self._compatible(o)
            n, it = len(self), self.ints
            rows = np.zeros((n, n), I64)
            for k in range(it):  # integer part: shift left (qfloat.py:890-893)
                rows[k, 0:n - (it - 1 - k)] = self.d[k] * o.d[it - 1 - k:]
            for k in range(it, n):  # fractional part: shift right (qfloat.py:895-898)
                rows[k, 1 + k - it:] = self.d[k] * o.d[0:n - (k - it) - 1]
            self.d = rows.sum(axis=0)
            self.sign = self.sign * o.sign
            self.tidy_flag = False
            self.base_tidy()
        return self

    def __mul__(self, o):
        if isinstance(o, Zero):
            return Zero()
        r = self.copy()
        r *= o
        return r

    __rmul__ = __mul__

    @classmethod
    def from_mul(cls, a, b, newlength=None, newints=None):
        """qfloat.py:955-1021 — product into a requested (length, ints) format;
        partial products outside the window are dropped (truncation, not rounding)."""
        if newlength is None:
            newlength = len(a)
        if newints is None:
            newints = a.ints
        if isinstance(a, Zero) or isinstance(b, Zero):
            return Zero()
        if isinstance(a, SignedBinary) or isinstance(b, SignedBinary):
            if isinstance(a, SignedBinary) and isinstance(b, SignedBinary):
                return a * b
            r = a * b
            r.set_len_ints(newlength, newints)
            return r
        cls.MUL += 1
        assert a.tidy_flag and b.tidy_flag
        if a.base != b.base:
            raise ValueError("bases are different")
        cols = np.zeros(newlength, I64)
        for k in range(len(a)):
            off = newints - a.ints + k + 1 - b.ints  # column of b[0] for row k (qfloat.py:1000)
            lo = 0 if off >= 0 else -off
            hi = min(len(b), newlength - off)
            if hi > lo:
                cols[off + lo:off + hi] += b.d[lo:hi] * a.d[k]
        return cls(cols, newints, a.base, False, a.sign * b.sign)

    @classmethod
    def multi_from_mul(cls, la, lb, newlength=None, newints=None):
        """qfloat.py:1023-1181 — numerically identical to per-pair from_mul; the
        stats double count when exactly one pair is QFloat x QFloat is reproduced."""
        a0 = next((x for x in la if isinstance(x, cls)), None)
        b0 = next((x for x in lb if isinstance(x, cls)), None)
        if newlength is None:
            newlength = len(a0) if a0 is not None else (len(b0) if b0 is not None else None)
        if newints is None:
            newints = a0.ints if a0 is not None else (b0.ints if b0 is not None else None)
        assert len(la) == len(lb)
        out = [None] * len(la)
        todo = []
        for k, (a, b) in enumerate(zip(la, lb)):
            if isinstance(a, Zero) or isinstance(b, Zero):
                out[k] = Zero()
            elif isinstance(a, SignedBinary) or isinstance(b, SignedBinary):
                out[k] = a * b
                out[k].set_len_ints(newlength, newints)
            else:
                todo.append(k)
        cls.MUL += len(todo)
        if len(todo) == 1:
            out[todo[0]] = cls.from_mul(la[todo[0]], lb[todo[0]], newlength, newints)  # counts again
        else:
            for k in todo:
                before = cls.MUL
                out[k] = cls.from_mul(la[k], lb[k], newlength, newints)
                cls.MUL = before
        return out

    # -- division (qfloat.py:1183-1376) ---------------------------------------
    def __itruediv__(self, o):
        if isinstance(o, Zero):
            raise ValueError("division by Zero")
        if isinstance(o, SignedBinary):
            z = int(o.value == 0)
            self.d = (1 - z) * self.d + z * np.ones(len(self), I64) * (self.base - 1)
            self.sign = (1 - z) * o.value + z * self.sign
            return self
        assert o.tidy_flag
        Q.DIV += 1
        self._compatible(o)
        assert self.tidy_flag
        fp = len(self) - self.ints
        quo = div_digits(np.concatenate((self.d, np.zeros(fp, I64))), o.d, self.base)
        self.sign = self.sign * o.sign
        self.d = quo[fp:]
        return self

    def __truediv__(self, o):
        r = self.copy()
        r /= o
        return r

    def __rtruediv__(self, o):
        if isinstance(o, Zero):
            return Zero()
        if isinstance(o, SignedBinary):
            return self.invert(o.value, len(self), self.ints)
        if isinstance(o, Q):
            return o / self
        raise ValueError("Unknown class for other")

    def invert(self, sign=1, newlength=None, newints=None):
        """qfloat.py:1263-1309 — 1/x into a new format via long division of
        [1, 0 x (fp_self + fp_new)] by the digit array."""
        if not (isinstance(sign, SignedBinary) or (isinstance(sign, numbers.Integral) and abs(sign) == 1)):
            raise ValueError("sign must be a SignedBinary or a signed binary scalar")
        Q.DIV += 1
        assert self.tidy_flag
        if newlength is None:
            newlength = len(self)
        if newints is None:
            newints = self.ints
        fp_new = newlength - newints
        fp_old = len(self) - self.ints
        dividend = np.concatenate((np.ones(1, I64), np.zeros(fp_old + fp_new, I64)))
        quo = div_digits(dividend, self.d, self.base)
        extra = newlength - quo.size
        quo = np.concatenate((np.zeros(extra, I64), quo)) if extra > 0 else quo[-extra:]
        return Q(quo, newints, self.base, True, sign * self.sign)

    @classmethod
    def multi_invert(cls, qs, sign=1, newlength=None, newints=None):
        """qfloat.py:1311-1376 — same numbers as invert(), one per element."""
        return [q.invert(sign, newlength, newints) for q in qs]


# =============================================================================
# L3: matrix inverse  (matrix_inversion/qfloat_matrix_inversion.py)
# =============================================================================
def float_matrix_to_qfloat_arrays(M, ln, ints, base):
    """qfloat_matrix_inversion.py:222-236."""
    qs = [Q.from_float(f, ln, ints, base) for f in np.asarray(M).flatten()]
    return (np.array([q.to_array() for q in qs], I64).reshape(len(qs), ln),
            np.array([q.sign for q in qs], I64))


def arrays_to_matrix(arrays, signs, ints, base):
    """qfloat_matrix_inversion.py:239-262."""
    n = int(np.sqrt(arrays.shape[0]))
    return [[Q(arrays[r * n + c], ints, base, True, signs[r * n + c]) for c in range(n)] for r in range(n)]


def arrays_to_float_matrix(arrays, ints, base):
    """qfloat_matrix_inversion.py:265-283."""
    arrays = np.asarray(arrays)
    n = int(np.sqrt(arrays.shape[0]))
    return np.array([Q(arrays[k, :-1], ints, base, True, arrays[k, -1]).to_float()
                     for k in range(n * n)]).reshape(n, n)


def matrix_to_arrays(M, ln, ints, base):
    """qfloat_matrix_inversion.py:286-309 — (n^2, len+1), sign in the last column."""
    n = len(M)
    out = np.zeros((n * n, ln + 1), I64)
    for r in range(n):
        for c in range(n):
            x, k = M[r][c], r * n + c
            if isinstance(x, Q):
                out[k, :ln] = x.to_array()
                out[k, ln] = x.sign
            elif isinstance(x, SignedBinary):
                out[k, ints - 1] = x.value
                out[k, ln] = x.value
            elif isinstance(x, Zero):
                pass
            else:
                out[k, ints - 1] = x
                out[k, ln] = np.sign(x)
    return out


def dot(l1, l2, tensorize=False):
    """qfloat_matrix_inversion.py:183-200."""
    if len(l1) != len(l2):
        raise ValueError("Lists should have the same length.")
    if tensorize:
        prods = Q.multi_from_mul(l1, l2, None, None)
        acc = prods[0]
        for m in prods[1:]:
            acc += m
        return acc
    acc = l1[0] * l2[0]
    for k in range(1, len(l1)):
        acc += l1[k] * l2[k]
    return acc


def argmax(indices, qs):
    """qfloat_matrix_inversion.py:317-328."""
    best = qs[0].copy()
    besti = indices[0]
    for k in range(1, len(indices)):
        gt = qs[k] > best
        best.d = gt * qs[k].d + (1 - gt) * best.d
        besti = gt * indices[k] + (1 - gt) * besti
    return besti


def pivot_matrix(M):
    """qfloat_matrix_inversion.py:331-369 — oblivious row swaps of an identity."""
    n = len(M)
    piv = np.eye(n, dtype=I64)
    for j in range(n - 1):
        r = argmax(list(range(j, n)), [abs(M[i][j]) for i in range(j, n)])
        tmp = piv.copy()
        acc = tmp[j, :] * int(j == r)
        for i in range(j + 1, n):
            acc = acc + tmp[i, :] * int(i == r)
        piv[j, :] = acc
        for jj in range(j + 1, n):
            e = int(jj == r)
            piv[jj, :] = (1 - e) * tmp[jj, :] + e * tmp[j, :]
    return piv


def lu_decomposition(M, ln, ints, true_division=False, tensorize=False):
    """qfloat_matrix_inversion.py:377-453 — Doolittle LU of P*M."""
    n = len(M)
    Lm = [[Zero() for _ in range(n)] for _ in range(n)]
    U = [[Zero() for _ in range(n)] for _ in range(n)]
    pm = pivot_matrix(M)
    P = [[SignedBinary(pm[i, j]) for j in range(n)] for i in range(n)]
    PM = [[dot(P[i], [M[k][j] for k in range(n)]) for j in range(n)] for i in range(n)]
    for j in range(n):
        Lm[j][j] = SignedBinary(1)
        for i in range(j + 1):
            if i > 0:
                s1 = dot([U[k][j] for k in range(i)], [Lm[i][k] for k in range(i)], tensorize)
                U[i][j] = PM[i][j] + s1.neg()
            else:
                U[i][j] = PM[i][j].copy()
        if not true_division:
            inv = U[j][j].invert(1, ln, 0)
        for i in range(j + 1, n):
            if j > 0:
                s2 = dot([U[k][j] for k in range(j)], [Lm[i][k] for k in range(j)], tensorize)
                num = PM[i][j] + s2.neg()
            else:
                num = PM[i][j]
            Lm[i][j] = (num / U[j][j]) if true_division else Q.from_mul(num, inv, ln, ints)
    P = [list(r) for r in zip(*P)]
    return P, Lm, U


def lu_inverse(P, Lm, U, ln, ints, true_division=False, tensorize=False):
    """qfloat_matrix_inversion.py:461-518 — forward / back substitution."""
    n = len(Lm)
    Y = [[Zero() for _ in range(n)] for _ in range(n)]
    for i in range(n):
        Y[i][0] = P[i][0].copy()
        for j in range(1, n):
            Y[i][j] = P[i][j] - dot([Lm[j][k] for k in range(j)], [Y[i][k] for k in range(j)], tensorize)
    X = [[Zero() for _ in range(n)] for _ in range(n)]
    if not true_division:
        if tensorize:
            inv = Q.multi_invert([U[j][j] for j in range(n)], 1, ln, 0)
        else:
            inv = [U[j][j].invert(1, ln, 0) for j in range(n)]
    for i in range(n - 1, -1, -1):
        X[i][-1] = (Y[i][-1] / U[-1][-1]) if true_division else Q.from_mul(Y[i][-1], inv[-1], ln, ints)
        for j in range(n - 2, -1, -1):
            t = Y[i][j] - dot([U[j][k] for k in range(j + 1, n)], [X[i][k] for k in range(j + 1, n)], tensorize)
            X[i][j] = (t / U[j][j]) if true_division else Q.from_mul(t, inv[j], ln, ints)
    return [list(r) for r in zip(*X)]


def inverse_2x2(M, ln, ints, tensorize=False):
    """qfloat_matrix_inversion.py:526-584 — adj(M)/det with det in format (2*ints+3, 2*ints)."""
    (a, b), (c, d) = M
    if tensorize:
        ad, bc = Q.multi_from_mul([a, b], [d, c], 2 * ints + 3, 2 * ints)
    else:
        ad = Q.from_mul(a, d, 2 * ints + 3, 2 * ints)
        bc = Q.from_mul(b, c, 2 * ints + 3, 2 * ints)
    det = ad + bc.neg()
    di = det.invert(1, ln, 0)
    if tensorize:
        ma, mb, mc, md = Q.multi_from_mul([a, b, c, d], [di] * 4, ln, ints)
    else:
        md, mb, mc, ma = (Q.from_mul(x, di, ln, ints) for x in (d, b, c, a))
    return [[md, mb.neg()], [mc.neg(), ma]]


def qfloat_matrix_inverse(arrays, signs, n, ln, ints, base, true_division, tensorize=False):
    """qfloat_matrix_inversion.py:672-720 — the circuit body, in plaintext."""
    arrays = np.asarray(arrays)
    assert n * n == arrays.shape[0] and ln == arrays.shape[1]
    M = arrays_to_matrix(arrays, np.asarray(signs), ints, base)
    if n == 2:
        Minv = inverse_2x2(M, ln, ints, tensorize)
    else:
        P, Lm, U = lu_decomposition(M, ln, ints, true_division, tensorize)
        Minv = lu_inverse(P, Lm, U, ln, ints, true_division, tensorize)
    return matrix_to_arrays(Minv, ln, ints, base)

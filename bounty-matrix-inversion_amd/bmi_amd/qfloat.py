"""QFloat on encrypted digits — host-side scheduler counterpart of the reference's matrix_inversion/qfloat.py.

Same class names, methods, argument meaning and error behaviour as the reference (`QFloat`, `SignedBinary`,
`Zero`; `from_float`, `to_float`, `base_tidy`, `tidy`, `+ - * /`, `neg`, `abs`, comparisons, `from_mul`,
`multi_from_mul`, `invert`, `multi_invert`), but a digit is either a plain int (unencrypted QFloat, as the
reference's NumPy mode) or a `circuit.Lin` (encrypted: every non-linear step is recorded as a PBS in the
circuit and later executed on the GPU).  Decrypted results are identical to the reference's plaintext
QFloat results (tests/golden/*.json).  Digits are stored most-significant first; value = sign * sum d_i p^(ints-1-i).
"""
from __future__ import annotations

import numbers

import numpy as np

from . import base_p_arrays as bpa
from .circuit import Circuit, Lin


def _is_enc(x):
    return isinstance(x, Lin)


class Zero:
    """reference qfloat.py:14-118 — a value known at trace time to be zero."""

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
    """reference qfloat.py:120-243 — a value known to lie in {-1, 0, +1}, encrypted or not."""

    def __init__(self, value):
        self._value = value

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        self._value = v

    @property
    def encrypted(self):
        return _is_enc(self._value)

    def copy(self):
        return SignedBinary(self._value)

    def to_float(self):
        return float(self._value)

    def __add__(self, o):
        if isinstance(o, SignedBinary):
            return self._value + o._value
        if isinstance(o, QFloat):
            return o.__add__(self)
        return self._value + o

    def __sub__(self, o):
        if isinstance(o, SignedBinary):
            return self._value - o._value
        if isinstance(o, QFloat):
            return o.__rsub__(self)
        return self._value - o

    def __mul__(self, o):
        if isinstance(o, SignedBinary):
            return SignedBinary(_mul(self._value, o._value))
        if isinstance(o, QFloat):
            return o.__mul__(self)
        return self._value * o

    def __truediv__(self, o):
        if isinstance(o, SignedBinary):
            if _is_enc(self._value) or _is_enc(o._value):
                raise ValueError("SignedBinary / SignedBinary is only defined on unencrypted values")
            return SignedBinary(self._value // o._value)
        if isinstance(o, QFloat):
            return o.__rtruediv__(self)
        return self._value / o

    def __neg__(self):
        return SignedBinary(-1 * self._value)

    def neg(self):
        self._value = -1 * self._value
        return self

    def __abs__(self):
        v = self._value
        return SignedBinary(_circ(v).lut(v, abs) if _is_enc(v) else abs(v))


def _circ(*xs):
    for x in xs:
        if isinstance(x, Lin):
            return x.c
        if isinstance(x, (list, tuple)):
            for y in x:
                if isinstance(y, Lin):
                    return y.c
    return None


PAIR_PRODUCTS = True    # _product: two bit x bit partial products of a column per look-up (False: one each, as in round 1)


def _two_ands(v):
    """v = 8 x + 4 y + 2 x' + y' with all four bits: x y + x' y'"""
    return ((v >> 3) & (v >> 2) & 1) + ((v >> 1) & v & 1)


def _mul(a, b):
    """product of two scalars, each an int or a Lin"""
    if _is_enc(a) and _is_enc(b):
        return a.c.mul(a, b)
    return a * b


class QFloat:
    """reference qfloat.py:245-1376."""

    ADDITIONS = 0
    MULTIPLICATION = 0
    DIVISION = 0

    def __init__(self, array, ints=None, base=2, is_base_tidy=True, sign=1):
        if isinstance(array, np.ndarray):
            if array.ndim != 1:
                raise ValueError("array must be one dimension")
            array = [int(x) for x in array.astype("int")]
        elif isinstance(array, (list, tuple)):
            array = list(array)
        else:
            raise ValueError("array must be np.ndarray or a list of encrypted digits")
        if not (isinstance(base, (int, np.integer)) and base > 1):
            raise ValueError("base must be a int >1")
        if ints is None:
            ints = len(array) // 2
        elif not (isinstance(ints, (int, np.integer)) and 0 <= ints <= len(array)):
            raise ValueError("ints must be in range [0,array.size]")
        self._array = array
        self._ints = int(ints)
        self._base = int(base)
        self._sign = int(sign) if isinstance(sign, (float, np.floating, np.integer)) else sign
        self._is_base_tidy = is_base_tidy
        if not is_base_tidy:
            self.base_tidy()

    # ---- statistics (reference qfloat.py:262-326) --------------------------------------------------
    @classmethod
    def reset_stats(cls):
        cls.ADDITIONS = cls.MULTIPLICATION = cls.DIVISION = 0

    @classmethod
    def show_stats(cls):
        print(f"\nQFloat statistics :\n======================\nAdditions       : {cls.ADDITIONS}\n"
              f"Multiplications : {cls.MULTIPLICATION}\nDivisions       : {cls.DIVISION}\n\n")

    # ---- properties --------------------------------------------------------------------------------
    ints = property(lambda self: self._ints)
    base = property(lambda self: self._base)
    is_base_tidy = property(lambda self: self._is_base_tidy)
    array = property(lambda self: self._array)
    sign = property(lambda self: self._sign)

    @property
    def encrypted(self):
        return any(_is_enc(x) for x in self._array) or _is_enc(self._sign)

    def _c(self, *others) -> Circuit:
        c = _circ(self._array, self._sign, *others)
        if c is None:
            raise ValueError("internal: no circuit for an unencrypted operation")
        return c

    def _check_unencrypted(self):
        if self.encrypted:
            raise ValueError("This function does not work on encrypted QFloats")

    # ---- plaintext codecs (reference qfloat.py:336-410) -----------------------------------------------
    def to_str(self, tidy=True):
        self._check_unencrypted()
        if tidy:
            self.base_tidy()
        nz = 1 if self._sign != 0 else 0
        a = [int(x) * nz for x in self._array[:self._ints]]
        b = [int(x) * nz for x in self._array[self._ints:]]
        if self._base <= 10:
            a, b = "".join(map(str, a)), "".join(map(str, b))
        else:
            a, b = str(np.array(a)), str(np.array(b))
        return ("" if self._sign >= 0 else "-") + a + "." + b

    __str__ = to_str

    @classmethod
    def from_float(cls, f, length=10, ints=None, base=2):
        if ints is None:
            ints = length // 2
        ip = int(f)
        arr = np.zeros(length, dtype=np.int64)
        arr[:ints] = bpa.int_to_base_p(ip, ints, base)
        arr[ints:] = bpa.float_to_base_p(f - ip, length - ints, base)
        return cls(np.abs(arr), ints, base, True, np.sign(f) or 1)

    def to_float(self):
        self._check_unencrypted()
        ip = bpa.base_p_to_int(self._array[:self._ints], self._base)
        fp = bpa.base_p_to_float(self._array[self._ints:], self._base)
        return (ip + fp) * self._sign

    # ---- structure ---------------------------------------------------------------------------------
    def __len__(self):
        return len(self._array)

    def copy(self):
        return QFloat(list(self._array), self._ints, self._base, self._is_base_tidy, self._sign)

    def to_array(self):
        return list(self._array)

    def set_len_ints(self, newlen, newints):
        """reference qfloat.py:565-589."""
        if self._ints != newints:
            if newints > self._ints:
                self._array = [0] * int(newints - self._ints) + self._array
            else:
                self._array = self._array[self._ints - newints:]
            self._ints = int(newints)
        extra = int(newlen - len(self))
        if extra > 0:
            self._array = self._array + [0] * extra
        elif extra < 0:
            self._array = self._array[:extra]
        return self

    def check_compatibility(self, other):
        if not isinstance(other, QFloat):
            raise ValueError("Object must also be a " + str(QFloat))
        if self._base != other.base:
            raise ValueError(str(QFloat) + "s bases are different")
        if len(self) != len(other):
            raise ValueError(str(QFloat) + "s have different length")
        if self._ints != other.ints:
            raise ValueError(str(QFloat) + "s have different dot index")

    # ---- carry / sign normalisation (reference qfloat.py:607-673) -----------------------------------------
    def base_tidy(self):
        if self._is_base_tidy:
            return
        c = _circ(self._array)
        if c is not None and all(bpa.lo_of(x) >= 0 for x in self._array):
            self._array = bpa.carry_propagate_nonneg(c, [[x] for x in self._array], self._base)
        else:
            self._array = bpa.carry_propagate_signed(c, self._array, self._base)
        self._is_base_tidy = True

    def tidy(self):
        """digits -> non-negative digits + sign.  With P / N the positive / negative parts of the digits,
        P - N is the digit array itself, so both borrow chains of the reference (qfloat.py:666-671) run
        directly on d and -d; the result is selected by the overflow flag of the first."""
        if not self._is_base_tidy:
            self.base_tidy()
        d = self._array
        c = _circ(d)
        p = self._base
        if c is None:
            arr = np.array(d, dtype=np.int64)
            pos, neg = arr * (arr >= 0), -1 * (arr * (arr < 0))
            pmn, isneg = _plain_sub(pos, neg, p)
            nmp, _ = _plain_sub(neg, pos, p)
            self._array = [int(x) for x in ((1 - isneg) * pmn + isneg * nmp)]
            self._sign = 2 * (1 - isneg) - 1
            return
        zeros = [0] * len(d)
        pmn, isneg = bpa.base_p_subtraction(c, d, zeros, p, True)
        if not _is_enc(isneg):
            self._array = pmn if not isneg else bpa.base_p_subtraction(c, [-x for x in d], zeros, p)
            self._sign = 1 - 2 * isneg
            return
        nmp = bpa.base_p_subtraction(c, [-x for x in d], zeros, p)
        self._array = [c.select(isneg, n, q) for n, q in zip(nmp, pmn)]
        self._sign = 1 - 2 * isneg

    # ---- comparisons (reference qfloat.py:681-764) --------------------------------------------------------
    def __eq__(self, other):
        self.check_compatibility(other)
        if not (self._is_base_tidy and other._is_base_tidy):
            raise Exception("cannot compare QFloats that are not tidy")
        c = _circ(self._array, other._array, self._sign, other._sign)
        if c is None:
            return int(list(self._array) == list(other._array)) & int(self._sign == other._sign)
        eq = bpa.is_equal(c, self._array, other._array)
        seq = bpa.lut(c, self._sign - other._sign, lambda v: int(v == 0))
        return c.lut2(eq, seq, lambda a, b: a & b)

    __hash__ = None

    def __gt__(self, other):
        self.check_compatibility(other)
        self.base_tidy()
        other.base_tidy()
        c = _circ(self._array, other._array, self._sign, other._sign)
        if c is None:
            same = int(self._sign == other._sign)
            mag_gt = 1 - _plain_ge(other._array, self._array)
            flip = int(self._sign < 0) & (1 - int(list(self._array) == list(other._array)))
            return same * (mag_gt ^ flip) + (1 - same) * int(self._sign > other._sign)
        sdiff = self._sign - other._sign
        same = bpa.lut(c, sdiff, lambda v: int(v == 0))
        sgt = bpa.lut(c, sdiff, lambda v: int(v > 0))
        mag_gt = 1 - bpa.is_greater_or_equal(c, other._array, self._array)
        ne = 1 - bpa.is_equal(c, self._array, other._array)
        flip = c.lut2(bpa.lut(c, self._sign, lambda v: int(v < 0)), ne, lambda a, b: a & b)
        x = c.lut2(mag_gt, flip, lambda a, b: a ^ b)
        # same * x + (1 - same) * sgt
        return c.select(same, x, sgt)

    def __lt__(self, other):
        return other > self

    def __le__(self, other):
        return 1 - (self > other)

    def __ge__(self, other):
        return 1 - (other > self)

    def __abs__(self):
        r = self.copy()
        r._sign = _mul(r._sign, r._sign)
        return r

    def abs(self):
        self._sign = _mul(self._sign, self._sign)
        return self

    def __neg__(self):
        r = self.copy()
        r._sign = -1 * r._sign
        return r

    def neg(self):
        self._sign = -1 * self._sign
        return self

    # ---- addition (reference qfloat.py:766-850) --------------------------------------------------------------
    def __iadd__(self, other):
        if isinstance(other, Zero):
            return None  # reference quirk (qfloat.py:803-804): `return` without a value
        QFloat.ADDITIONS += 1
        arr = [_mul(x, self._sign) for x in self._array]
        if isinstance(other, Lin) or isinstance(other, numbers.Integral):
            arr[self._ints - 1] = arr[self._ints - 1] + other
        elif isinstance(other, SignedBinary):
            arr[self._ints - 1] = arr[self._ints - 1] + other.value
        else:
            self.check_compatibility(other)
            arr = [x + _mul(y, other._sign) for x, y in zip(arr, other._array)]
        self._array = arr
        self._is_base_tidy = False
        self._sign = None
        self.tidy()
        return self

    def __add__(self, other):
        r = self.copy()
        r += other
        return r

    __radd__ = __add__

    def __sub__(self, other):
        r = -other
        r += self
        return r

    def __rsub__(self, other):
        r = -self
        r += other
        return r

    # ---- multiplication (reference qfloat.py:852-1181) ----------------------------------------------------------
    def __imul__(self, other):
        if isinstance(other, numbers.Integral):
            s = int(np.sign(other))
            self._array = [x * (other * s) for x in self._array]
            self._sign = self._sign * s
            self._is_base_tidy = False
            self.base_tidy()
        elif isinstance(other, Lin):
            # an encrypted integer (a Tracer in the reference, qfloat.py:858-865): sign and magnitude are one look-up
            # each, every digit x magnitude one packed product, then the carries
            c = other.c
            s = c.lut(other, lambda v: (v > 0) - (v < 0))
            mag = c.lut(other, abs)
            self._array = [_mul(x, mag) for x in self._array]
            self._sign = _mul(self._sign, s)
            self._is_base_tidy = False
            self.base_tidy()
        elif isinstance(other, SignedBinary):
            self._sign = _mul(self._sign, other.value)
        else:
            QFloat.MULTIPLICATION += 1
            self.base_tidy()
            other.base_tidy()
            self.check_compatibility(other)
            prod = _product(self, other, len(self), self._ints)
            self._array, self._sign = prod._array, prod._sign
            self._is_base_tidy = True
        return self

    def __mul__(self, other):
        if isinstance(other, Zero):
            return Zero()
        r = self.copy()
        r *= other
        return r

    __rmul__ = __mul__

    @classmethod
    def from_mul(cls, a, b, newlength=None, newints=None):
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
        cls.MULTIPLICATION += 1
        assert a.is_base_tidy
        assert b.is_base_tidy
        if a.base != b.base:
            raise ValueError("bases are different")
        return _product(a, b, newlength, newints)

    @classmethod
    def multi_from_mul(cls, list_a, list_b, newlength=None, newints=None):
        """reference qfloat.py:1023-1181.  The scheduler batches independent look-ups by itself, so the
        'tensorised' form needs no separate lowering; the reference's statistics quirk (double count
        when exactly one pair is QFloat x QFloat) is reproduced."""
        a0 = next((x for x in list_a if isinstance(x, cls)), None)
        b0 = next((x for x in list_b if isinstance(x, cls)), None)
        if newlength is None:
            newlength = len(a0) if a0 is not None else (len(b0) if b0 is not None else None)
        if newints is None:
            newints = a0.ints if a0 is not None else (b0.ints if b0 is not None else None)
        assert len(list_a) == len(list_b)
        out = [None] * len(list_a)
        todo = []
        for i, (a, b) in enumerate(zip(list_a, list_b)):
            if isinstance(a, Zero) or isinstance(b, Zero):
                out[i] = Zero()
            elif isinstance(a, SignedBinary) or isinstance(b, SignedBinary):
                out[i] = a * b
                out[i].set_len_ints(newlength, newints)
            else:
                todo.append(i)
        cls.MULTIPLICATION += len(todo)
        for i in todo:
            before = cls.MULTIPLICATION
            out[i] = cls.from_mul(list_a[i], list_b[i], newlength, newints)
            if len(todo) != 1:
                cls.MULTIPLICATION = before
        return out

    # ---- division (reference qfloat.py:1183-1376) ------------------------------------------------------------------
    def __itruediv__(self, other):
        if isinstance(other, Zero):
            raise ValueError("division by Zero")
        if isinstance(other, SignedBinary):
            v = other.value
            if _is_enc(v):
                c = v.c
                z = c.lut(v, lambda x: int(x == 0))
                top = self._base - 1
                self._array = [c.select(z, top, x) for x in self._array]
                self._sign = c.select(z, self._sign, v)
            else:
                z = int(v == 0)
                self._array = [(1 - z) * x + z * (self._base - 1) for x in self._array]
                self._sign = (1 - z) * v + z * self._sign
            return self
        assert other.is_base_tidy
        QFloat.DIVISION += 1
        self.check_compatibility(other)
        assert self._is_base_tidy
        fp = len(self) - self._ints
        quo = _divide(self._array + [0] * fp, other._array, self._base)
        self._sign = _mul(self._sign, other._sign)
        self._array = quo[fp:]
        return self

    def __truediv__(self, other):
        r = self.copy()
        r /= other
        return r

    def __rtruediv__(self, other):
        if isinstance(other, Zero):
            return Zero()
        if isinstance(other, SignedBinary):
            return self.invert(other.value, len(self), self._ints)
        if isinstance(other, QFloat):
            return other / self
        raise ValueError("Unknown class for other")

    def invert(self, sign=1, newlength=None, newints=None):
        if not (isinstance(sign, (SignedBinary, Lin)) or (isinstance(sign, numbers.Integral) and abs(sign) == 1)):
            raise ValueError("sign must be a SignedBinary or a signed binary scalar")
        if isinstance(sign, SignedBinary):
            sign = sign.value
        QFloat.DIVISION += 1
        assert self._is_base_tidy
        if newlength is None:
            newlength = len(self)
        if newints is None:
            newints = self._ints
        fp_new = newlength - newints
        fp_old = len(self) - self._ints
        quo = _divide([1] + [0] * (fp_old + fp_new), self._array, self._base)
        extra = newlength - len(quo)
        quo = [0] * extra + quo if extra > 0 else quo[-extra:]
        return QFloat(quo, newints, self._base, True, _mul(sign, self._sign))

    @classmethod
    def multi_invert(cls, list_qfloats, sign=1, newlength=None, newints=None):
        q0 = list_qfloats[0]
        for q in list_qfloats:
            assert isinstance(q, cls) and q.is_base_tidy
            assert len(q) == len(q0) and q.base == q0.base and q.ints == q0.ints
        return [q.invert(sign, newlength, newints) for q in list_qfloats]


# ---------------------------------------------------------------------------------------- helpers
def _plain_sub(a, b, p):
    """plaintext borrow chain on equal-size numpy arrays -> (digits, a<b)"""
    out = np.zeros(a.size, dtype=np.int64)
    borrow = 0
    for k in range(1, a.size + 1):
        t = int(a[-k]) - int(b[-k]) - borrow
        borrow = 1 if t < 0 else 0
        out[-k] = t + p * borrow
    return out, borrow


def _plain_ge(a, b):
    borrow = 0
    for k in range(1, min(len(a), len(b)) + 1):
        borrow = 1 if int(a[-k]) - int(b[-k]) - borrow < 0 else 0
    return 1 - borrow


def _divide(dividend, divisor, p):
    c = _circ(dividend, divisor)
    if c is None:
        from .circuit import Circuit as _C  # constant folding runs the same code on plain ints
        c = _C()
    return bpa.base_p_division(c, list(dividend), list(divisor), p)


def _product(a, b, newlength, newints):
    """Schoolbook product of two base-tidy QFloats into the format (newlength, newints): partial products
    a_i * b_j land in column newints - a.ints + i + 1 - b.ints + j and those outside [0, newlength) are
    dropped (reference qfloat.py:890-898 for equal formats, :998-1010 in general), then base-p carry
    propagation (qfloat.py:908, 1019)."""
    p = a.base
    c = _circ(a._array, b._array)
    cols = [[] for _ in range(newlength)]
    # Two partial products of the same column in ONE look-up when all four digits are encrypted bits: the packed value
    # 8 x + 4 y + 2 x' + y' spans the 16 entries of a 4-bit table, whose output x y + x' y' (0..2) goes into the column
    # sum as one term.  Halves the look-ups of the schoolbook product (the widest levels of the inverse); the column sums,
    # hence the digits, are unchanged.
    waiting = {}

    def is_bit(v):
        return _is_enc(v) and v.lo >= 0 and v.hi <= 1

    for i in range(len(a)):
        off = newints - a.ints + i + 1 - b.ints
        for j in range(len(b)):
            col = off + j
            if 0 <= col < newlength:
                x, y = a._array[i], b._array[j]
                if PAIR_PRODUCTS and is_bit(x) and is_bit(y):
                    first = waiting.pop(col, None)
                    if first is None:
                        waiting[col] = (x, y)
                    else:
                        cols[col].append(c.lut(first[0] * 8 + first[1] * 4 + x * 2 + y, _two_ands))
                    continue
                t = _mul(x, y)
                if not (not _is_enc(t) and t == 0):
                    cols[col].append(t)
    for col, (x, y) in waiting.items():
        cols[col].append(_mul(x, y))
    sign = _mul(a._sign, b._sign)
    if c is None:
        return QFloat(np.array([sum(col) for col in cols], dtype=np.int64), newints, p, False, sign)
    if all(bpa.lo_of(t) >= 0 for col in cols for t in col):
        digits = bpa.carry_propagate_nonneg(c, cols, p)
    else:
        sums = []
        for col in cols:
            s = 0
            for t in col:
                s = s + t
            sums.append(s)
        digits = bpa.carry_propagate_signed(c, sums, p)
    return QFloat(digits, newints, p, True, sign)

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

    # ---- sign bookkeeping ---------------------------------------------------------------------------
    # `_sign` is the reference's field (in {-1, 0, +1}; 0 = the value is zero whatever the digits say).  Beside it `_neg` is
    # kept whenever the sign is KNOWN to be non-zero and available as a bit: _sign == 1 - 2 * _neg.  The outputs of tidy and
    # of the fused addition have it, products / quotients of operands that have it keep it; a sign multiplied by a ternary
    # SignedBinary loses it.  The fused addition (base_p_arrays.signed_add_binary) works on sign bits.
    @property
    def _sign(self):
        return self._sgn

    @_sign.setter
    def _sign(self, v):
        self._sgn = v
        self._neg = int(v < 0) if (v is not None and not _is_enc(v) and v != 0) else None

    def _set_neg(self, neg):
        self._sgn = 1 - 2 * neg
        self._neg = neg

    @staticmethod
    def _xor_neg(na, nb):
        """sign bit of a product of two non-zero signs"""
        if not _is_enc(na):
            return nb if not na else 1 - nb
        if not _is_enc(nb):
            return na if not nb else 1 - na
        return na.c.lut(na + nb, lambda v: int(v == 1))

    @staticmethod
    def _sign_product(x, y):
        """(sign, neg) of the product of the signs of two QFloats (neg is None when either sign may be zero)"""
        if x._neg is not None and y._neg is not None:
            n = QFloat._xor_neg(x._neg, y._neg)
            return 1 - 2 * n, n
        return _mul(x._sign, y._sign), None

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
        r = QFloat(list(self._array), self._ints, self._base, self._is_base_tidy, self._sign)
        r._neg = self._neg
        return r

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
        self._set_neg(isneg)

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
        return r.abs()

    def abs(self):
        if self._neg is not None:      # a non-zero sign squared
            self._sign = 1
        else:
            self._sign = _mul(self._sign, self._sign)
        return self

    def __neg__(self):
        r = self.copy()
        return r.neg()

    def neg(self):
        if self._neg is not None:
            self._set_neg(1 - self._neg)
        else:
            self._sign = -1 * self._sign
        return self

    # ---- addition (reference qfloat.py:766-850) --------------------------------------------------------------
    def __iadd__(self, other):
        if isinstance(other, Zero):
            return None  # reference quirk (qfloat.py:803-804): `return` without a value
        QFloat.ADDITIONS += 1
        if self._fused_add(other):
            return self
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

    def _fused_add(self, other):
        """`+=` of base-2 operands by base_p_arrays.signed_add_binary (same integers as the reference's digit-times-sign
        sum, carry chain and tidy; four look-up levels).  Returns False when the operands are not of that shape (other bases,
        digits that are not tidy, a scalar beyond [-1, 1], plaintext): the caller then runs the reference's steps."""
        units = self._ints - 1                         # the reference adds scalars at array[ints - 1] (the last digit when ints = 0)
        if isinstance(other, QFloat):
            b_digits, b_sign, b_neg = other._array, other._sign, other._neg
            if not (other._is_base_tidy and other._base == 2 and len(other) == len(self) and other._ints == self._ints):
                return False
        else:
            v = other.value if isinstance(other, SignedBinary) else other
            if isinstance(v, numbers.Integral):
                if abs(int(v)) > 1:
                    return False
                mag, b_neg = abs(int(v)), int(v < 0)
            elif _is_enc(v) and v.lo >= -1 and v.hi <= 1:
                mag, b_neg = v.c.lut(v, abs), v.c.lut(v, lambda x: int(x < 0))
            else:
                return False
            b_digits = [0] * len(self)
            b_digits[units] = mag
            b_sign = None
        c = _circ(self._array, self._sign, b_digits, b_sign, b_neg)
        if c is None or self._base != 2 or not self._is_base_tidy or len(self) == 0:
            return False

        def shaped(digits):
            return all(bpa.lo_of(x) >= 0 and bpa.hi_of(x) <= (3 if i == 0 else 1) for i, x in enumerate(digits))

        def operand(digits, sign, neg):
            """(digits, sign bit); a sign that may be zero is folded into the digits first (one look-up per digit)"""
            if neg is not None:
                return list(digits), neg
            if not _is_enc(sign):
                return ([0] * len(digits), 0) if int(sign) == 0 else (list(digits), int(sign < 0))
            if sign.lo < -1 or sign.hi > 1:
                return None
            return [bpa.lut2(c, sign, x, lambda s_, d: d if s_ != 0 else 0) for x in digits], c.lut(sign, lambda s_: int(s_ < 0))

        if not (shaped(self._array) and shaped(b_digits)):
            return False
        A = operand(self._array, self._sign, self._neg)
        B = operand(b_digits, b_sign, b_neg)
        if A is None or B is None:
            return False
        digits, neg = bpa.signed_add_binary(c, A[0], A[1], B[0], B[1])
        self._array = digits
        self._is_base_tidy = True
        self._set_neg(neg)
        return True

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
            self._neg = prod._neg
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
        sign, neg = QFloat._sign_product(self, other)
        self._sign = sign
        self._neg = neg
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
        r = QFloat(quo, newints, self._base, True, _mul(sign, self._sign))
        if self._neg is not None and isinstance(sign, numbers.Integral):     # +-1 times a non-zero sign
            r._set_neg(self._neg if sign > 0 else 1 - self._neg)
        return r

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
    sign, neg = QFloat._sign_product(a, b)
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
    r = QFloat(digits, newints, p, True, sign)
    r._neg = neg
    return r

"""Digit-array primitives on encrypted digits — host-side scheduler counterpart of the reference's
matrix_inversion/base_p_arrays.py.  Arrays are Python lists of `circuit.Lin` (or plain ints for
compile-time constants), most-significant digit first, exactly as in the reference.

Each function produces the SAME integers as the reference function it names (tests compare decrypted /
simulated results with the reference-generated golden vectors), but lowers to look-ups differently:
single-input chains are fused into one table, selects are one packed bivariate PBS, and wide sums are
reduced with narrow (4-bit) look-ups so that the whole path runs at the N = 1024 parameter set.
"""
from __future__ import annotations

import numpy as np

from .circuit import Lin, RangeError, MSG_BITS

CAP = (1 << MSG_BITS) - 1  # largest non-negative value a look-up input interval [0, CAP] can span


# ---------------------------------------------------------------------------- plaintext codecs
def int_to_base_p(integer, n, p):
    """reference base_p_arrays.py:24-48 (the top digit is not reduced mod p)."""
    if n == 0:
        return np.zeros(0, dtype=np.int64)
    s = (integer > 0) - (integer < 0)
    v = abs(int(integer))
    out = np.zeros(n, dtype=np.int64)
    for k in range(n):
        w = p ** (n - 1 - k)
        out[k] = v // w
        v -= int(out[k]) * w
    return out * s


def float_to_base_p(f, precision, p):
    """reference base_p_arrays.py:62-81."""
    s = float(np.sign(f))
    f = abs(f)
    assert 0 <= f < 1, "Input should be a float between 0 and 1 (exclusive)"
    digits = []
    while f and len(digits) < precision:
        f *= p
        d = int(f)
        if d > 0:
            f -= d
        digits.append(d)
    digits += [0] * (precision - len(digits))
    return s * np.array(digits, dtype=float)


def base_p_to_int(arr, p):
    """reference base_p_arrays.py:11-21."""
    v = 0
    for x in arr:
        v = v * p + int(x)
    return v


def base_p_to_float(arr, p):
    """reference base_p_arrays.py:51-59."""
    f = 0.0
    for k, x in enumerate(arr):
        f += x * (p ** -(k + 1))
    return f


# ------------------------------------------------------------------------------------ helpers
def lo_of(x):
    return x.lo if isinstance(x, Lin) else int(x)


def hi_of(x):
    return x.hi if isinstance(x, Lin) else int(x)


def lut(c, x, fn):
    return c.lut(x, fn) if isinstance(x, Lin) else fn(int(x))


def lut2(c, x, y, fn):
    if isinstance(x, Lin) or isinstance(y, Lin):
        return c.lut2(x, y, fn)
    return fn(int(x), int(y))


def sum_is_positive(c, xs):
    """1 if sum(xs) > 0 else 0 (replaces `np.sum(...) > 0`, base_p_arrays.py:137).  Non-negative terms:
    an OR tree over chunks whose sums fit one look-up.  Mixed-sign terms (only arise from non-binary leading
    digits): a single look-up on the sum, which must then fit the message space."""
    return _sum_test(c, xs, lambda v: int(v > 0))


def sum_is_zero(c, xs):
    """1 if sum(xs) == 0 else 0 (replaces `np.sum(...) == 0`, base_p_arrays.py:134)."""
    return _sum_test(c, xs, lambda v: int(v == 0))


def _sum_test(c, xs, test):
    xs = [x for x in xs if not (not isinstance(x, Lin) and int(x) == 0)]
    if not xs:
        return test(0)
    if any(lo_of(x) < 0 for x in xs):
        s = 0
        for x in xs:
            s = s + x
        return lut(c, s, test)  # raises RangeError if the interval is wider than one look-up
    any_pos = any_positive(c, xs)
    # for non-negative terms: sum > 0 <=> any term > 0 ; sum == 0 <=> no term > 0
    return any_pos if test(1) else 1 - any_pos


def any_positive(c, xs):
    """OR tree: 1 if any of the non-negative xs is > 0."""
    xs = [x for x in xs if not (not isinstance(x, Lin) and int(x) == 0)]
    if not xs:
        return 0
    for x in xs:
        if lo_of(x) < 0:
            raise RangeError("any_positive needs non-negative terms")
    while True:
        chunks, cur, cur_hi = [], 0, 0
        for x in xs:
            h = hi_of(x)
            if h > CAP:
                raise RangeError("term too wide")
            if cur_hi + h > CAP:
                chunks.append(cur)
                cur, cur_hi = 0, 0
            cur = cur + x
            cur_hi += h
        chunks.append(cur)
        flags = [lut(c, s, lambda v: int(v > 0)) for s in chunks]
        if len(flags) == 1:
            return flags[0]
        xs = flags


def base_p_subtraction(c, a, b, p, overflow=False):
    """reference base_p_arrays.py:108-139: borrow-chain a - b, right-aligned; one look-up per digit
    (the borrow); the digit itself is linear: t + p * borrow."""
    m = min(len(a), len(b))
    out = [0] * len(a)
    borrow = 0
    for k in range(1, m + 1):
        t = a[-k] - b[-k] - borrow
        borrow = lut(c, t, lambda v: int(v < 0))
        d = t + p * borrow
        if isinstance(d, Lin) and isinstance(t, Lin):
            vals = [v + p * (v < 0) for v in range(t.lo, t.hi + 1)]
            d = d.assume(min(vals), max(vals))
        out[-k] = d
    if not overflow:
        return out
    extra = len(b) - len(a)
    if extra == 0:
        lt = borrow
    elif extra < 0:
        # a < b  <=>  borrow and the extra leading digits of a are all zero
        zero = sum_is_zero(c, a[:-extra])
        lt = lut2(c, borrow, zero, lambda bo, z: bo & z)
        out[:-extra] = a[:-extra]
    else:
        pos = sum_is_positive(c, b[:extra])
        lt = lut2(c, borrow, pos, lambda bo, z: bo | z)
    return out, lt


def base_p_division(c, dividend, divisor, p):
    """reference base_p_arrays.py:173-203: restoring long division, MSD first.  Per step and trial:
    one borrow chain with overflow flag, then a select per remainder digit (one packed PBS each)."""
    quo = [0] * len(dividend)
    rem = [dividend[0]]
    for k in range(len(dividend)):
        if k > 0:
            drop = 1 if len(rem) > len(divisor) else 0
            rem = rem[drop:] + [dividend[k]]
        for _ in range(p - 1):
            diff, lt = base_p_subtraction(c, rem, divisor, p, True)
            ge = 1 - lt
            rem = [c.select(lt, r, d) if isinstance(lt, Lin) else (r if lt else d) for r, d in zip(rem, diff)]
            quo[k] = quo[k] + ge
    return quo


def is_greater_or_equal(c, a, b):
    """reference base_p_arrays.py:245-260."""
    m = min(len(a), len(b))
    borrow = 0
    for k in range(1, m + 1):
        borrow = lut(c, a[-k] - b[-k] - borrow, lambda v: int(v < 0))
    return 1 - borrow


def is_equal(c, a, b):
    """reference base_p_arrays.py:276-280: all digits equal (an AND tree instead of a wide sum)."""
    ne = [lut(c, x - y, lambda v: int(v != 0)) for x, y in zip(a, b)]
    return 1 - any_positive(c, ne)


# ----------------------------------------------------------------------- carry propagation
def _pack_bins(terms, cap):
    """greedy first-fit packing of (expr, hi) terms into bins whose hi-sums stay <= cap"""
    bins = []
    for x in sorted(terms, key=lambda t: -hi_of(t)):
        h = hi_of(x)
        if h > cap:
            raise RangeError(f"single term with range up to {h} exceeds look-up capacity")
        for b in bins:
            if b[1] + h <= cap:
                b[0].append(x)
                b[1] += h
                break
        else:
            bins.append([[x], h])
    return bins


def carry_propagate_nonneg(c, columns, p):
    """Base-p carry propagation of NON-NEGATIVE column sums (reference qfloat.py:607-626 applied to the
    column sums of a product, qfloat.py:901/1015): digit_i = (sum of column i + carry_{i+1}) mod p,
    carry_i = floor(. / p), carry out of column 0 dropped.

    `columns[i]` is the list of terms of column i (each a Lin/int with lo >= 0).  Because every term is
    non-negative, truncation toward zero equals floor and the result is the plain base-p representation of
    sum_i column_i * p^(L-1-i) mod p^L, so the sum may be reassociated freely: columns whose total exceeds
    what one 4-bit look-up can hold are first compressed in parallel (each bin of terms -> its base-p
    digits, pushed to the columns on the left), then one sequential chain of L look-ups finishes."""
    L = len(columns)
    cols = [[t for t in col if not (not isinstance(t, Lin) and int(t) == 0)] for col in columns]
    for col in cols:
        for t in col:
            if lo_of(t) < 0:
                raise RangeError("carry_propagate_nonneg needs non-negative terms")
    # chain capacity: c = column + carry <= CAP with carry <= floor(CAP / p)
    carry_cap = CAP // p
    col_cap = CAP - carry_cap
    while True:
        over = [i for i in range(L) if sum(hi_of(t) for t in cols[i]) > col_cap]
        if not over:
            break
        new_cols = [list(col) if i not in over else [] for i, col in enumerate(cols)]
        for i in over:
            for terms, h in _pack_bins(cols[i], CAP):
                s = 0
                for t in terms:
                    s = s + t
                if len(terms) == 1 and h <= p - 1:
                    new_cols[i].append(s)
                    continue
                j, w = 0, 1
                while w <= h:  # digit j of the bin sum goes to column i - j
                    if i - j >= 0:
                        new_cols[i - j].append(lut(c, s, lambda v, w=w: (v // w) % p))
                    j += 1
                    w *= p
        cols = new_cols
    out = [0] * L
    carry = 0
    for i in range(L - 1, -1, -1):
        s = carry
        for t in cols[i]:
            s = s + t
        carry = lut(c, s, lambda v: v // p) if i > 0 else 0
        if isinstance(s, Lin):
            d = (s - p * carry) if i > 0 else lut(c, s, lambda v: v % p)
            out[i] = d.assume(0, p - 1) if isinstance(d, Lin) else d
        else:
            out[i] = s % p
            carry = (s // p) if i > 0 else 0
    return out


def carry_propagate_signed(c, digits, p):
    """reference QFloat.base_tidy (qfloat.py:607-626) on mixed-sign digits: carry = trunc(c / p) toward
    zero, digit = c - carry * p; one fused look-up per digit (abs, //, sign and the product are all
    functions of the same c).  Used where the inputs are narrow (sums of two digit arrays)."""
    out = list(digits)
    carry = 0
    for i in range(len(out) - 1, -1, -1):
        s = out[i] + carry
        if isinstance(s, Lin):
            carry = lut(c, s, lambda v: (abs(v) // p) * ((v > 0) - (v < 0)))
            d = s - p * carry
            vals = [v - p * ((abs(v) // p) * ((v > 0) - (v < 0))) for v in range(s.lo, s.hi + 1)]
            out[i] = d.assume(min(vals), max(vals)) if isinstance(d, Lin) else d
        else:
            carry = (abs(s) // p) * ((s > 0) - (s < 0))
            out[i] = s - p * carry
    return out

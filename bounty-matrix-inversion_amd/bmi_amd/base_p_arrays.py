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
def _find_circuit(*xs):
    for x in xs:
        if isinstance(x, Lin):
            return x.c
        if isinstance(x, (list, tuple, np.ndarray)):
            c = _find_circuit(*x)
            if c is not None:
                return c
    return None


def reference_signature(fn):
    """Lets a digit-array primitive be called exactly like its namesake in the reference's base_p_arrays.py
    (operands first: `base_p_subtraction(a, b, p, overflow)`), the circuit being found from the encrypted operands, as
    well as with the circuit as a leading argument (how the QFloat layer of this package calls it)."""
    import functools
    from .circuit import Circuit as _Circuit

    @functools.wraps(fn)
    def wrapper(*args, **kw):
        if args and (args[0] is None or isinstance(args[0], _Circuit)):
            return fn(*args, **kw)
        args = tuple(list(a) if isinstance(a, np.ndarray) else a for a in args)
        return fn(_find_circuit(*args), *args, **kw)
    return wrapper


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


def _sign3(v):
    return (v > 0) - (v < 0)


def _comb(h, l):
    """borrow-lookahead operator on signals in {-1: generate, 0: propagate, +1: kill}: the more significant
    signal wins unless it propagates"""
    return h if h else l


def _prefix_borrows(c, sig, first_bit):
    """Kogge-Stone parallel prefix over signals sig[0..m-1] (index 0 = least significant).  Returns
    bits[i] = 1 iff a borrow leaves position i, i.e. the combined signal of positions 0..i is 'generate'.
    Depth ceil(log2 m) look-up levels instead of m; every node is one packed 3x3 bivariate look-up."""
    m = len(sig)
    bits = [None] * m
    bits[0] = first_bit
    if WIDE_LOOKAHEAD:
        return _lookahead_bits(c, list(sig), 1, bits)
    S = list(sig)
    d = 1
    while d < m:
        newS = list(S)
        for i in range(d, m):
            hi, lo = S[i], S[i - d]
            if i < 2 * d:  # positions d..2d-1 become final at this level
                bits[i] = lut2(c, hi, lo, lambda h, l: int(_comb(h, l) == -1))
            if i + 2 * d < m or (i >= 2 * d):  # still needed as an operand (or not final yet)
                newS[i] = lut2(c, hi, lo, _comb)
        S = newS
        d *= 2
    return bits


def _window3_borrows(c, deltas):
    """Borrow look-ahead for digit differences in {-1, 0, 1} (binary operands): the first level reads windows
    of THREE adjacent positions in one look-up each - sign(4 d_i + 2 d_{i-1} + d_{i-2}) is the combined signal
    of positions i-2..i and the argument spans 15 values - then Kogge-Stone doubling with distances 3, 6, 12...
    One level fewer than signals-then-prefix.  Returns bits[i] = borrow out of position i."""
    m = len(deltas)
    S = [None] * m
    bits = [None] * m
    for i in range(m):
        w = deltas[i] * 4
        if i >= 1:
            w = w + deltas[i - 1] * 2
        if i >= 2:
            w = w + deltas[i - 2]
        if i <= 2:
            bits[i] = lut(c, w, lambda v: int(v < 0))   # the window reaches position 0: final
        if i + 3 < m or i > 2:
            S[i] = lut(c, w, _sign3)
    d = 3
    while d < m:
        newS = list(S)
        for i in range(d, m):
            if bits[i] is not None and not (i + 2 * d < m):
                continue
            hi, lo = S[i], S[i - d]
            if i < 2 * d and bits[i] is None:  # window now reaches position 0
                bits[i] = lut2(c, hi, lo, lambda h, l: int(_comb(h, l) == -1))
            if i >= 2 * d or i + 2 * d < m:
                newS[i] = lut2(c, hi, lo, _comb)
        S = newS
        d *= 2
    return bits


WIDE_LOOKAHEAD = True  # use the 5-bit look-ups of Circuit.lut_odd in the look-ahead networks (False: 4-bit forms only)


def _comb3(v):
    """combined signal of THREE look-ahead signals packed as 9 s2 + 3 s1 + s0 (s2 most significant): the most
    significant non-propagating one.  Odd in v, and f(v - 16) = -f(v) on [-13, 13]: one Circuit.lut_odd look-up."""
    s2 = (v + 13) // 9 - 1            # balanced ternary digits of v
    r = v - 9 * s2
    s1 = (r + 4) // 3 - 1
    s0 = r - 3 * s1
    return s2 if s2 else (s1 if s1 else s0)


def _neg(c, v):
    """[v < 0] for v in [-15, 15] in one look-up (Circuit.lut_neg); plain values pass through"""
    return c.lut_neg(v) if isinstance(v, Lin) else int(int(v) < 0)


def _wide_borrows(c, deltas):
    """Borrow look-ahead for digit differences in {-1, 0, 1} with 5-bit look-ups: the first level reads sliding windows
    of FOUR positions (sign(8 d_i + 4 d_{i-1} + 2 d_{i-2} + d_{i-3}), Circuit.lut_odd), later levels combine THREE
    signals at a time (spans 4, 12, 36, ...).  A borrow BIT is [window or packed signals < 0], which Circuit.lut_neg
    reads off the same 5-bit argument in the same level - so the bits of positions 0..3 come out of level 1, those below
    12 out of level 2, below 36 out of level 3: a 33-digit subtraction has all its borrows after THREE levels (four with
    a separate signal-to-bit conversion, five with window-3 + doubling).  Returns bits[i] = borrow out of position i."""
    m = len(deltas)
    S = [None] * m
    bits = [None] * m
    sign = lambda v: (v > 0) - (v < 0)  # noqa: E731
    for i in range(m):
        w = deltas[i] * 8 if i >= 3 else deltas[i] * (1 << i)
        for t in range(1, min(i, 3) + 1):
            w = w + deltas[i - t] * (1 << ((3 if i >= 3 else i) - t))
        if i <= 3:
            bits[i] = _neg(c, w)                          # the window reaches position 0: final
        if any(j < m for j in (i + 4, i + 8)) or i > 3:   # still an operand of a later combine
            S[i] = (c.lut_odd(w, sign) if isinstance(w, Lin) else sign(int(w)))
    return _lookahead_bits(c, S, 4, bits)


def _lookahead_bits(c, S, span, bits):
    """Finishes a look-ahead: S[i] is the combined signal of positions max(0, i - span + 1) .. i (None where it is never
    read); fills the missing bits[i] = [combined signal of positions 0..i is 'generate'].  Per level: a position whose
    two (three) packed signals reach position 0 gets its bit from that packed value directly (Circuit.lut_neg: negative
    <=> the most significant non-propagating signal generates); the signals later positions still need are widened
    three-way (Circuit.lut_odd), spans x 3 per level."""
    m = len(S)
    while any(b is None for b in bits):
        newS = [None] * m
        for i in range(m):
            if bits[i] is None:
                if i < span:                               # the signal already reaches position 0: convert
                    bits[i] = lut(c, S[i], lambda v: int(v == -1))
                elif i < 2 * span:                         # two signals reach position 0 together
                    bits[i] = _neg(c, S[i] * 3 + S[i - span])
                elif i < 3 * span:                         # three signals
                    bits[i] = _neg(c, S[i] * 9 + S[i - span] * 3 + S[i - 2 * span])
            # the signal of span 3 x span ending at i is read next level by positions i (if >= 3 span), i + 3 span, i + 6 span
            if any(j < m and j >= 3 * span for j in (i, i + 3 * span, i + 6 * span)):
                if i >= 2 * span:
                    packed = S[i] * 9 + S[i - span] * 3 + S[i - 2 * span]
                    newS[i] = c.lut_odd(packed, _comb3) if isinstance(packed, Lin) else _comb3(int(packed))
                elif i >= span:
                    newS[i] = lut2(c, S[i], S[i - span], _comb)
                else:
                    newS[i] = S[i]
        S = newS
        span *= 3
    return bits


@reference_signature
def base_p_subtraction(c, a, b, p, overflow=False):
    """reference base_p_arrays.py:108-139: a - b with borrows, right-aligned, and (overflow=True) the flag a < b
    as defined there for unequal sizes.  Same integers as the reference's sequential borrow chain, computed
    by borrow look-ahead: per digit the signal sign(a_i - b_i) (a borrow leaves digit i iff the signals of
    digits 0..i combine to 'generate'), a log-depth parallel prefix, then digit_i = a_i - b_i - bin_i + p*bout_i
    (linear).  Valid for any integer digits (non-binary leading digits included)."""
    m = min(len(a), len(b))
    out = [0] * len(a)
    extra = len(b) - len(a)
    deltas = [a[-k] - b[-k] for k in range(1, m + 1)]  # least significant first
    if all(not isinstance(d, Lin) for d in deltas) and (not overflow or extra == 0 or all(
            not isinstance(x, Lin) for x in (a[:-extra] if extra < 0 else b[:extra]))):
        # compile-time constants: the plain chain
        borrow = 0
        for k in range(m):
            t = deltas[k] - borrow
            borrow = int(t < 0)
            out[-k - 1] = t + p * borrow
        if not overflow:
            return out
        if extra == 0:
            return out, borrow
        if extra < 0:
            out[:-extra] = a[:-extra]
            return out, borrow & int(sum(int(x) for x in a[:-extra]) == 0)
        return out, borrow | int(sum(int(x) for x in b[:extra]) > 0)
    pseudo = None
    if overflow and extra < 0:
        # a < b  <=>  borrow out and the extra leading digits of a sum to zero: a 'propagate / kill' position on top
        pseudo = 1 - sum_is_zero(c, a[:-extra])        # 0 = propagate (all zero), 1 = kill
        out[:-extra] = a[:-extra]
    elif overflow and extra > 0:
        pseudo = -1 * sum_is_positive(c, b[:extra])     # 0 = propagate, -1 = generate
    if pseudo is None and all(-1 <= lo_of(d) and hi_of(d) <= 1 for d in deltas):
        bits = (_wide_borrows if WIDE_LOOKAHEAD else _window3_borrows)(c, deltas)
    else:
        sig = [lut(c, d, _sign3) for d in deltas]
        if pseudo is not None:
            sig = sig + [pseudo]
        bits = _prefix_borrows(c, sig, lut(c, deltas[0], lambda v: int(v < 0)))
    for k in range(m):
        bin_k = bits[k - 1] if k > 0 else 0
        t = deltas[k] - bin_k
        d = t + p * bits[k]
        if isinstance(d, Lin):
            tlo, thi = lo_of(t), hi_of(t)
            vals = [v + p * (v < 0) for v in range(tlo, thi + 1)]
            d = d.assume(min(vals), max(vals))
        out[-k - 1] = d
    if not overflow:
        return out
    return out, bits[-1]


@reference_signature
def base_p_division(c, dividend, divisor, p):
    """reference base_p_arrays.py:173-203: restoring long division, MSD first; quotient has dividend.size digits.

    For canonical binary operands (every digit in [0, 1], which is what the inverse feeds it: the operands of
    every division are outputs of `tidy`/`invert`) the quotient digits are the binary expansion of
    floor(dividend / divisor) (all ones when the divisor is zero), so any exact division algorithm returns the
    same array; the radix-4 form below retires two quotient bits per step and halves the circuit depth.
    Anything else (other bases, non-binary leading digits) runs the reference's bit-serial algorithm."""
    if p == 2 and all(0 <= lo_of(x) and hi_of(x) <= 1 for x in list(dividend) + list(divisor)) and len(divisor) >= 2 \
            and any(isinstance(x, Lin) for x in list(dividend) + list(divisor)):
        return _division_radix4(c, list(dividend), list(divisor))
    return _division_bitserial(c, dividend, divisor, p)


def _division_bitserial(c, dividend, divisor, p):
    """The reference's algorithm, step for step (remainder window of divisor.size + 1 digits, p - 1 trial
    subtractions per digit); per trial: one borrow look-ahead, then one packed select per remainder digit."""
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


def _add_binary(c, a, b):
    """a + b for equal-length canonical binary digit lists (MSD first), carry out of the top dropped"""
    cols = [[x, y] for x, y in zip(a, b)]
    return carry_propagate_nonneg(c, cols, 2)


def _two_ands(v):
    """v = 8 x + 4 y + 2 x' + y' with all four bits: x y + x' y'"""
    return ((v >> 3) & (v >> 2) & 1) + ((v >> 1) & v & 1)


DIVISION_BITS = 2  # quotient bits retired per step of the binary division (2: radix 4, 3: radix 8)


def _division_radix4(c, dividend, divisor):
    return _division_radix(c, dividend, divisor, DIVISION_BITS)


def _division_radix(c, dividend, divisor, kbits):
    """floor(dividend / divisor) in binary, `kbits` quotient bits per step (radix R = 2^kbits): the partial
    remainder (always < divisor, so it fits m digits) is extended by kbits dividend digits and compared with
    D, 2D, ..., (R-1)D in parallel (R - 1 borrow look-aheads); sel = number of failed comparisons picks the new
    remainder with one R-way mux per digit (sel and a candidate bit pack into one 4-bit look-up for R <= 8).
    A leading group of n mod kbits digits is retired first with the smaller radix.  A zero divisor makes every
    comparison succeed: all quotient bits are 1, as in the reference."""
    n, m = len(dividend), len(divisor)
    R = 1 << kbits
    w = m + kbits
    shifted = lambda sh: [0] * (kbits - sh) + divisor + [0] * sh   # D * 2^sh as w digits  # noqa: E731
    mult = {1 << sh: shifted(sh) for sh in range(kbits)}
    for j in range(3, R):
        if j not in mult:
            hi = 1 << (j.bit_length() - 1)
            if j == R - 1 and kbits >= 3:
                # (R - 1) D = R D - D: ONE subtraction of given operands (4 D + 3 D would have to wait for 3 D); R D = D followed
                # by kbits zeros is exactly w digits
                mult[j] = base_p_subtraction(c, divisor + [0] * kbits, mult[1], 2)
            else:
                mult[j] = _add_binary(c, mult[hi], mult[j - hi])       # j D < 2^w: no carry out
    quo = [0] * n
    rem = [0] * m  # m digits, MSD first

    def step(rem, digs):
        r = len(digs)                       # this step's radix is 2^r (r <= kbits)
        Rr = 1 << r
        ext = rem + digs                    # 2^r * rem + digits, m + r digits
        diffs, lts = {}, []
        for j in range(1, Rr):
            dj, lt = base_p_subtraction(c, ext, mult[j][kbits - r:], 2, True)   # j D < 2^(m + r): leading zeros dropped
            diffs[j] = dj
            lts.append(lt)
        sel = sum(lts[1:], lts[0])          # failed comparisons: 0 -> quotient digit Rr - 1, ..., Rr - 1 -> 0
        if isinstance(sel, Lin):
            sel = sel.assume(0, Rr - 1)
            bits = [c.lut(sel, lambda s, b=b: ((Rr - 1 - s) >> b) & 1) for b in range(r - 1, -1, -1)]
            # The comparisons are monotone (ext < j D implies ext < (j + 1) D), so the flags [sel == kk] are DIFFERENCES of
            # consecutive comparison bits - linear, no look-up - and the new remainder digit sum_kk [sel == kk] cand_kk is a sum of
            # bit x bit products, two of them per 4-bit look-up (x y + x' y' of 8 x + 4 y + 2 x' + y'): Rr / 2 look-ups per
            # digit instead of Rr packed selections (a radix-8 step of a 40-digit division drops from 344 to 172 look-ups in
            # this level: one kernel round instead of two).
            onehot = []
            for kk in range(Rr):             # sel == kk  <=>  exactly kk comparisons failed: lts[Rr-2-kk+1..] ... as differences
                # lts[j-1] = [ext < j D]; failed comparisons are the LARGEST multiples: sel == kk <=> lt_{Rr-kk} = 1 and lt_{Rr-1-kk} = 0
                hi_flag = lts[Rr - 1 - kk] if kk >= 1 else 0          # [ext < (Rr - kk) D]   (kk = 0: no comparison failed)
                lo_flag = lts[Rr - 2 - kk] if kk <= Rr - 2 else 1     # [ext < (Rr - 1 - kk) D]   (kk = Rr - 1: all failed)
                f = (1 - lo_flag) if kk == 0 else (hi_flag - lo_flag if kk <= Rr - 2 else hi_flag)
                onehot.append(f.assume(0, 1) if isinstance(f, Lin) else f)
            new = []
            for i in range(r, m + r):       # the new remainder is < D: only its low m digits can be non-zero
                cands = [diffs[j][i] for j in range(Rr - 1, 0, -1)] + [ext[i]]
                acc = 0
                for kk in range(0, Rr, 2):
                    x, y, x2, y2 = onehot[kk], cands[kk], onehot[kk + 1], cands[kk + 1]
                    if all(isinstance(t, Lin) for t in (x, y, x2, y2)):
                        acc = acc + c.lut(x * 8 + y * 4 + x2 * 2 + y2, _two_ands)
                    else:
                        acc = acc + lut2(c, x, y, lambda a_, b_: a_ & b_) + lut2(c, x2, y2, lambda a_, b_: a_ & b_)
                new.append(acc.assume(0, 1) if isinstance(acc, Lin) else acc)
            return bits, new
        qv = Rr - 1 - sel
        cands = [diffs[j] for j in range(Rr - 1, 0, -1)] + [ext]
        return [(qv >> b) & 1 for b in range(r - 1, -1, -1)], list(cands[sel][r:])

    k = 0
    if n % kbits:
        lead = n % kbits
        bits, rem = step(rem, list(dividend[:lead]))
        quo[:lead] = bits
        k = lead
    while k < n:
        bits, rem = step(rem, list(dividend[k:k + kbits]))
        quo[k:k + kbits] = bits
        k += kbits
    return quo


@reference_signature
def is_greater_or_equal(c, a, b):
    """reference base_p_arrays.py:245-260: 1 - (borrow out of a - b), by a tree over look-ahead signals: windows of four
    digit differences first (binary operands), three signals per look-up after that, and the last look-up returns the
    borrow bit itself (Circuit.lut_neg on the packed signals): 36 digits in three levels."""
    m = min(len(a), len(b))
    deltas = [a[-k] - b[-k] for k in range(1, m + 1)]
    if all(not isinstance(d, Lin) for d in deltas):
        borrow = 0
        for d in deltas:
            borrow = int(d - borrow < 0)
        return 1 - borrow
    sign = lambda v: (v > 0) - (v < 0)  # noqa: E731
    if WIDE_LOOKAHEAD and all(-1 <= lo_of(d) and hi_of(d) <= 1 for d in deltas):
        packed = []
        for g in range(0, m, 4):
            w = 0
            for t, d in enumerate(deltas[g:g + 4]):
                w = w + d * (1 << t)
            packed.append(w)
        if len(packed) == 1:
            return 1 - _neg(c, packed[0])
        sig = [(c.lut_odd(w, sign) if isinstance(w, Lin) else sign(int(w))) for w in packed]
    else:
        if m == 1:
            return 1 - lut(c, deltas[0], lambda v: int(v < 0))
        sig = [lut(c, d, _sign3) for d in deltas]
    while len(sig) > 3:
        nxt = []
        for i in range(0, len(sig), 3):
            grp = sig[i:i + 3]
            if len(grp) == 3 and WIDE_LOOKAHEAD:
                v = grp[2] * 9 + grp[1] * 3 + grp[0]
                nxt.append(c.lut_odd(v, _comb3) if isinstance(v, Lin) else _comb3(int(v)))
            elif len(grp) >= 2:
                nxt.append(lut2(c, grp[1], grp[0], _comb))
                if len(grp) == 3:
                    nxt.append(grp[2])
            else:
                nxt.append(grp[0])
        sig = nxt
    if len(sig) == 1:
        return 1 - lut(c, sig[0], lambda v: int(v == -1))
    if len(sig) == 2 or not WIDE_LOOKAHEAD:
        if len(sig) == 3:
            sig = [lut2(c, sig[1], sig[0], _comb), sig[2]]
        return 1 - _neg(c, sig[1] * 3 + sig[0])
    return 1 - _neg(c, sig[2] * 9 + sig[1] * 3 + sig[0])


@reference_signature
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
    sum_i column_i * p^(L-1-i) mod p^L, so the sum may be reassociated freely:
      1. columns are compressed in parallel (each bin of terms whose sum fits one 4-bit look-up -> its base-p
         digits, pushed to the columns on the left) until every column sums to at most 2(p-1), i.e. two
         operands remain;
      2. the final two-operand addition is a carry look-ahead: signal per column (>= p: generate, == p-1:
         propagate, else kill), log-depth parallel prefix, digit_i = col_i + cin_i - p * cout_i (linear).
    Depth ~ (3-4 compression rounds) + 1 + log2(L) instead of L."""
    L = len(columns)
    cols = [[t for t in col if not (not isinstance(t, Lin) and int(t) == 0)] for col in columns]
    for col in cols:
        for t in col:
            if lo_of(t) < 0:
                raise RangeError("carry_propagate_nonneg needs non-negative terms")
    if all(not isinstance(t, Lin) for col in cols for t in col):
        out, carry = [0] * L, 0
        for i in range(L - 1, -1, -1):
            s = carry + sum(int(t) for t in cols[i])
            out[i], carry = s % p, s // p
        return out
    target = 2 * (p - 1)
    for _round in range(12):
        if all(sum(hi_of(t) for t in col) <= target for col in cols):
            break
        if p == 2 and max(sum(hi_of(t) for t in col) for col in cols) <= 4:
            # Last round, two columns per look-up: with every column sum at most 4, the value 2 s_i + s_{i+1} of a pair of
            # adjacent columns is at most 12 - one 4-bit look-up per output bit - and its four bits go to columns i+1, i, i-1,
            # i-2.  A column then holds its own pair's bit and ONE bit of the pair to its right: at most 2, the target.  (The
            # single-column rounds need two more rounds from here: 4 -> 3 -> 2.)
            new_cols = [[] for _ in range(L)]
            i = L - 2
            while i >= -1:
                hi_col = cols[i] if i >= 0 else []
                lo_col = cols[i + 1]
                hs, ls = sum(hi_of(t) for t in hi_col), sum(hi_of(t) for t in lo_col)
                if hs <= 1 and ls <= 1:        # nothing to compress: the terms stay where they are
                    if i >= 0:
                        new_cols[i].extend(hi_col)
                    new_cols[i + 1].extend(lo_col)
                else:
                    v = 0
                    for t in hi_col:
                        v = v + t * 2
                    for t in lo_col:
                        v = v + t
                    top = 2 * hs + ls
                    for k in range(top.bit_length()):
                        col = i + 1 - k
                        if col >= 0:
                            new_cols[col].append(lut(c, v, lambda x, k=k: (x >> k) & 1))
                i -= 2
            cols = new_cols
            continue
        # Compress every column that holds more than one digit's worth: a column left alone at the target would
        # be pushed over it again by the digits arriving from its right neighbour (a ripple of one column per
        # round); compressing all of them together is the carry-save step and converges in ~log rounds.
        over = [i for i in range(L) if sum(hi_of(t) for t in cols[i]) > p - 1]
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
                while w <= h:  # digit j of the bin sum goes to column i - j (dropped beyond column 0)
                    if i - j >= 0:
                        new_cols[i - j].append(lut(c, s, lambda v, w=w: (v // w) % p))
                    j += 1
                    w *= p
        cols = new_cols
    else:
        raise RangeError("column compression did not converge")
    sums = []
    for col in cols:
        s = 0
        for t in col:
            s = s + t
        sums.append(s)
    # carry look-ahead over positions L-1 (least significant) .. 0
    lsb_first = sums[::-1]
    if p == 2 and all(0 <= lo_of(x) and hi_of(x) <= 2 for x in lsb_first):
        # carry out of a window <=> the most significant non-'propagate' column generates: with e = 1 - col in
        # {-1: generate, 0: propagate, 1: kill} this is exactly the borrow look-ahead on e (windows of three)
        bits = (_wide_borrows if WIDE_LOOKAHEAD else _window3_borrows)(c, [1 - x for x in lsb_first])
    else:
        sig = [lut(c, x, lambda v: -1 if v >= p else (0 if v == p - 1 else 1)) for x in lsb_first]
        bits = _prefix_borrows(c, sig, lut(c, lsb_first[0], lambda v: int(v >= p)))
    out = [0] * L
    for k in range(L):
        cin = bits[k - 1] if k > 0 else 0
        d = lsb_first[k] + cin - p * bits[k]
        if isinstance(d, Lin):
            d = d.assume(0, p - 1)
        out[L - 1 - k] = d
    return out


def _signed_chain(c, digits_lsb_first, p, carry_in):
    """sequential truncating carry chain over a block (least significant first); returns (carries, carry_out)
    where carries[i] is the carry OUT of position i"""
    carries = []
    carry = carry_in
    for d in digits_lsb_first:
        s = d + carry
        carry = lut(c, s, lambda v: (abs(v) // p) * ((v > 0) - (v < 0)))
        carries.append(carry)
    return carries


def _mux3(c, sel, cands):
    """cands[sel + 1] for an encrypted sel in {-1, 0, 1}: three packed bivariate look-ups, one level"""
    if not isinstance(sel, Lin):
        return cands[int(sel) + 1]
    acc = 0
    for k, x in zip((-1, 0, 1), cands):
        acc = acc + lut2(c, sel, x, lambda s_, v, k=k: v if s_ == k else 0)
    lo = min(lo_of(x) for x in cands)
    hi = max(hi_of(x) for x in cands)
    return acc.assume(lo, hi) if isinstance(acc, Lin) else acc


SIGNED_BLOCK = 6


def carry_propagate_signed(c, digits, p):
    """reference QFloat.base_tidy (qfloat.py:607-626) on mixed-sign digits: carry = trunc(c / p) toward zero,
    digit = c - carry * p (abs, //, sign and the product are all functions of the same c: one fused look-up).
    Same integers as the reference's right-to-left loop, evaluated as a carry-SELECT adder: the digits are cut
    into blocks; every block runs its short chain for each possible incoming carry (-1, 0, +1) in parallel, then
    the true carries ripple block to block through 3-way muxes.  Depth ~ block + #blocks instead of L.  Falls
    back to the plain chain when a digit is too wide for the carry to stay in {-1, 0, 1}."""
    L = len(digits)
    if c is None or all(not isinstance(d, Lin) for d in digits):
        out = list(digits)
        carry = 0
        for i in range(L - 1, -1, -1):
            s = out[i] + carry
            carry = (abs(s) // p) * ((s > 0) - (s < 0))
            out[i] = s - p * carry
        return out
    lsb = list(digits[::-1])  # position 0 = least significant
    narrow = all(-(2 * p - 2) <= lo_of(d) and hi_of(d) <= 2 * p - 2 for d in lsb[:-1])
    carries = [None] * L
    if not narrow or L <= SIGNED_BLOCK + 2:
        carries = _signed_chain(c, lsb, p, 0)
    else:
        # positions 0 .. L-2 by carry-select; the leading digit (possibly wider: non-binary leading digits of the
        # inputs) is finished with one ordinary step - its carry out is dropped anyway
        body = L - 1
        blocks = [list(range(a, min(a + SIGNED_BLOCK, body))) for a in range(0, body, SIGNED_BLOCK)]
        cin = 0
        for bi, blk in enumerate(blocks):
            seg = [lsb[i] for i in blk]
            if bi == 0:
                cs = _signed_chain(c, seg, p, 0)
                for i, v in zip(blk, cs):
                    carries[i] = v
            else:
                cand = [_signed_chain(c, seg, p, k) for k in (-1, 0, 1)]  # independent of cin: scheduled early
                for j, i in enumerate(blk):
                    carries[i] = _mux3(c, cin, [cand[0][j], cand[1][j], cand[2][j]])
            cin = carries[blk[-1]]
        carries[L - 1] = _signed_chain(c, [lsb[L - 1]], p, cin)[0]
    out = [0] * L
    for i in range(L):
        cin_i = carries[i - 1] if i > 0 else 0
        s = lsb[i] + cin_i
        d = s - p * carries[i]
        if isinstance(d, Lin):
            slo, shi = lo_of(s), hi_of(s)
            vals = [v - p * ((abs(v) // p) * ((v > 0) - (v < 0))) for v in range(slo, shi + 1)]
            d = d.assume(min(vals), max(vals))
        out[L - 1 - i] = d
    return out


def signed_add_binary(c, a, na, b, nb):
    """The reference's  QFloat += QFloat  (qfloat.py:798-834: digits * sign + digits * sign, then base_tidy :607-626 and tidy
    :648-673) for base-2 operands in sign-magnitude form, WITHOUT walking the signed carry chain.

    a, b: equal-length digit lists (MSD first), every digit in [0, 1] except the leading ones, which may reach 3 (from_float does
    not reduce the leading digit, base_p_arrays.py:42-46); na, nb: sign bits (1 = negative; a zero-signed operand is passed as
    all-zero digits).  Returns (digits, neg) with digits in [0, 1] and sign = 1 - 2 neg.

    What the reference computes, case by case (d = s_a a + s_b b are the mixed-sign digits its truncating chain walks):
      * equal signs s: every digit of d has the sign s, truncation toward zero is floor on magnitudes, so the chain is the plain
        binary addition of the magnitudes with the carry out of the leading digit dropped; tidy returns that magnitude with the
        sign s (+1 if the magnitude is zero).  A leading-digit sum of up to 7 keeps (a_L + b_L + carry) mod 2.
      * opposite signs: the digits are in {-1, 0, 1} below the leading one, and a truncating chain that starts with carry 0 never
        produces a carry from such digits (|d + 0| <= 1); the leading digit delta = a_L - b_L (|delta| <= 3) becomes
        rho = delta - 2 trunc(delta / 2) in {-1, 0, 1} (its carry is dropped), and tidy returns |X|, sign s_a sign(X) for
        X = A' - B' with A' = ([rho = 1], a body), B' = ([rho = -1], b body) - an ordinary binary subtraction.
    So three carry / borrow look-aheads (A + B, A' - B', B' - A') run in parallel (three levels for up to 36 digits), one
    packed three-way selection per digit picks the magnitude, and the sign follows two levels later: four levels per addition
    of a chain instead of sixteen (sign products, carry-select chain, two subtractions, selection)."""
    L = len(a)
    assert L == len(b) and L >= 1
    lead_wide = hi_of(a[0]) > 1 or hi_of(b[0]) > 1
    if lead_wide:
        body = _add_binary(c, [0] + list(a[1:]), [0] + list(b[1:]))          # leading digit of the result: the carry out of the body
        SUM = [lut(c, a[0] + b[0] + body[0], lambda v: v & 1)] + body[1:]
        rho = lambda v: ((v > 0) - (v < 0)) * (abs(v) & 1)  # noqa: E731
        delta = a[0] - b[0]
        A2 = [lut(c, delta, lambda v: int(rho(v) == 1))] + list(a[1:])
        B2 = [lut(c, delta, lambda v: int(rho(v) == -1))] + list(b[1:])
    else:
        SUM = _add_binary(c, list(a), list(b))                                # carry out of the leading digit dropped
        A2, B2 = list(a), list(b)
    D1, lt = base_p_subtraction(c, A2, B2, 2, True)
    D2, gt = base_p_subtraction(c, B2, A2, 2, True)
    same = lut(c, na + nb, lambda v: int(v != 1))
    w = same * 2 + lt                      # 0: opposite signs, A' >= B' -> D1;  1: opposite signs, A' < B' -> D2;  2, 3: equal signs -> SUM
    digits = []
    for x1, x2, xs in zip(D1, D2, SUM):
        acc = lut2(c, w, x1, lambda s_, v: v if s_ == 0 else 0) + lut2(c, w, x2, lambda s_, v: v if s_ == 1 else 0) \
            + lut2(c, w, xs, lambda s_, v: v if s_ >= 2 else 0)
        digits.append(acc.assume(0, 1) if isinstance(acc, Lin) else acc)
    # sign.  equal signs: negative iff s = -1 and the magnitude is not zero; opposite: negative iff s_a (A' - B') < 0
    chunks, cur, n = [], 0, 0
    for x in SUM:
        if n == CAP:
            chunks.append(cur)
            cur, n = 0, 0
        cur = cur + x
        n += 1
    chunks.append(cur)
    flags = [lut(c, s, lambda v: int(v > 0)) for s in chunks]
    while len(flags) > 3:                  # (not reached below 46 digits)
        flags = [any_positive(c, flags[i:i + CAP]) for i in range(0, len(flags), CAP)]
    nzsum = 0
    for f in flags:
        nzsum = nzsum + f
    g_same = lut2(c, na, nzsum, lambda n_, z: int(n_ == 1 and z > 0))
    g_diff = lut2(c, na, lt - gt, lambda n_, t: int((n_ == 0 and t == 1) or (n_ == 1 and t == -1)))
    neg = c.select(same, g_same, g_diff) if isinstance(same, Lin) else (g_same if same else g_diff)
    if isinstance(neg, Lin):
        neg = neg.assume(0, 1)
    return digits, neg


# ------------------------------------------------------------------------------- tensorised twins
# The reference's multi_* functions (base_p_arrays.py:142-170, 206-242, 263-273) apply the same primitive to every row
# of a 2-D array so that Concrete sees one wide tensor operation.  Here every look-up of every row is an independent
# node of the circuit and the level scheduler batches them whatever the call structure, so the twins are row maps:
# same signatures, same results, and the same PBS levels as a "tensorised" formulation would produce.
def _rows(x):
    return [list(r) for r in x]


def multi_base_p_subtraction(a_arrays, b_arrays, p, overflow=False):
    res = [base_p_subtraction(a, b, p, overflow) for a, b in zip(_rows(a_arrays), _rows(b_arrays))]
    if not overflow:
        return res
    return [d for d, _ in res], [lt for _, lt in res]


def multi_base_p_division(dividends, divisors, p):
    return [base_p_division(a, b, p) for a, b in zip(_rows(dividends), _rows(divisors))]


def multi_is_greater_or_equal(a_arrays, b_arrays):
    return [is_greater_or_equal(a, b) for a, b in zip(_rows(a_arrays), _rows(b_arrays))]


def insert_array_at_index(a, B, i, j):
    """a into row i of B from column j on, clipped on both sides (reference :326-338); B is a list of rows (or a 2-D array)"""
    a = list(a)
    if j < 0:
        a, j = a[-j:], 0
    n = min(len(B[i]) - j, len(a))
    for t in range(max(n, 0)):
        B[i][j + t] = a[t]


def insert_array_at_index_3D(A, B, i, j):
    """the same for every matrix of a stack: rows of A into B[m][i] (reference :341-354)"""
    for m, a in enumerate(A):
        insert_array_at_index(a, B[m], i, j)

"""LU / 2x2 matrix inverse on QFloats — host-side scheduler counterpart of the reference's
matrix_inversion/qfloat_matrix_inversion.py:131-720 (same function names and argument meaning).
Running `qfloat_matrix_inverse` on encrypted digits (circuit.Lin) records the whole inverse as a PBS graph;
running it on plain ints computes the plaintext result directly."""
from __future__ import annotations

import numpy as np

from . import base_p_arrays as bpa
from .circuit import Lin
from .qfloat import QFloat, SignedBinary, Zero, _circ


# ------------------------------------------------------------------------------------------- utils
# 2-D "list matrices" (lists of rows) of QFloat / SignedBinary / Zero entries, as in the reference (:145-219).
def transpose_2D_list(list2D):
    width = len(list2D[0]) if list2D else 0
    return [[row[j] for row in list2D] for j in range(width)]


def matrix_column(M, j):
    return transpose_2D_list(M)[j]


def map_2D_list(list2D, function):
    return [list(map(function, row)) for row in list2D]


def binary_list_matrix(M):
    """n x n 0/1 scalars (ints or encrypted) -> SignedBinary entries"""
    return map_2D_list([list(row) for row in M], SignedBinary)


def zero_list_matrix(n):
    return [[Zero()] * n for _ in range(n)]   # Zero is immutable: sharing one object per row is safe


def _accumulate(terms):
    """left-to-right in-place sum t0 += t1 += ... (the accumulation order fixes the digits: `+=` truncates)"""
    total = terms[0]
    for t in terms[1:]:
        total += t
    return total


def qfloat_list_dot_product(list1, list2, tensorize=False):
    """sum_k list1[k] * list2[k], products formed one by one or (tensorize) by QFloat.multi_from_mul."""
    if len(list1) != len(list2):
        raise ValueError("Lists should have the same length.")
    if tensorize:
        return _accumulate(QFloat.multi_from_mul(list1, list2, None, None))
    return _accumulate([x * y for x, y in zip(list1, list2)])


def qfloat_list_matrix_multiply(matrix1, matrix2):
    columns = transpose_2D_list(matrix2)
    return [[qfloat_list_dot_product(row, col) for col in columns] for row in matrix1]


# ------------------------------------------------------------------------------------- marshalling
def float_matrix_to_qfloat_arrays(M, qfloat_len, qfloat_ints, qfloat_base):
    """reference :222-236 — float matrix -> (n^2, len) digit array + (n^2,) sign array."""
    qs = [QFloat.from_float(f, qfloat_len, qfloat_ints, qfloat_base) for f in np.asarray(M).flatten()]
    arrays = np.array([q.to_array() for q in qs], dtype=np.int64).reshape(len(qs), qfloat_len)
    signs = np.array([q.sign for q in qs], dtype=np.int64)
    return arrays, signs


def qfloat_arrays_to_qfloat_matrix(qfloat_arrays, qfloat_signs, qfloat_ints, qfloat_base):
    """reference :239-262 — rows may be numpy ints or lists of encrypted digits."""
    n = int(np.sqrt(len(qfloat_arrays)))
    return [[QFloat(qfloat_arrays[r * n + c] if isinstance(qfloat_arrays[r * n + c], np.ndarray)
                    else list(qfloat_arrays[r * n + c]), qfloat_ints, qfloat_base, True, qfloat_signs[r * n + c])
             for c in range(n)] for r in range(n)]


def qfloat_and_signs_arrays_to_float_matrix(qfloat_arrays, qfloat_ints, qfloat_base):
    """reference :265-283."""
    qfloat_arrays = np.asarray(qfloat_arrays)
    n = int(np.sqrt(qfloat_arrays.shape[0]))
    return np.array([QFloat(qfloat_arrays[k, :-1], qfloat_ints, qfloat_base, True, qfloat_arrays[k, -1]).to_float()
                     for k in range(n * n)]).reshape(n, n)


def qfloat_matrix_to_arrays_and_signs(M, qfloat_len, qfloat_ints, qfloat_base):
    """reference :286-309 — returns an (n^2) x (len + 1) list of scalars, sign in the last column."""
    n = len(M)
    assert n == len(M[0])
    out = [[0] * (qfloat_len + 1) for _ in range(n * n)]
    for i in range(n):
        for j in range(n):
            x, row = M[i][j], out[i * n + j]
            if isinstance(x, QFloat):
                row[:qfloat_len] = x.to_array()
                row[qfloat_len] = x.sign
            elif isinstance(x, SignedBinary):
                row[qfloat_ints - 1] = x.value
                row[qfloat_len] = x.value
            elif isinstance(x, Zero):
                pass
            else:
                row[qfloat_ints - 1] = x
                row[qfloat_len] = np.sign(x)
    return out


# ------------------------------------------------------------------------------------------ pivot
def _select(c, bit, x, y):
    if isinstance(bit, Lin):
        return c.select(bit, x, y)
    return x if bit else y


def _argmax_flags(qfloats):
    """One flag per candidate, exactly one of them 1: the FIRST maximum (what the reference's running `>` comparison
    selects, :317-328).  Keeping the position as flags instead of one encrypted integer lets it grow with n (an index in
    [0, 9] does not fit a 4-bit bivariate select; ten flags do) and hands the pivot its row masks without a look-up."""
    max_qf = qfloats[0].copy()
    flags = [1]
    for i in range(1, len(qfloats)):
        is_gt = qfloats[i] > max_qf
        c = _circ([is_gt], max_qf.array, qfloats[i].array)
        max_qf._array = [_select(c, is_gt, a, b) for a, b in zip(qfloats[i].array, max_qf.array)]
        keep = 1 - is_gt
        flags = [f * keep if not (isinstance(f, Lin) and isinstance(keep, Lin)) else c.mul(f, keep) for f in flags] + [is_gt]
    return flags


def qfloat_argmax(indices, qfloats):
    """reference :317-328 - the index of the (first) largest QFloat: sum of index x flag, linear in the flags"""
    total = 0
    for idx, f in zip(indices, _argmax_flags(qfloats)):
        total = total + f * idx
    return total.assume(min(indices), max(indices)) if isinstance(total, Lin) else total


def _bitmul(row, bit):
    return [x * bit if not (isinstance(x, Lin) and isinstance(bit, Lin)) else x.c.mul(x, bit) for x in row]


def qfloat_pivot_matrix(M):
    """reference :331-369 - oblivious row swaps of an identity matrix: column by column, the row holding the largest
    |entry| at or below the diagonal is exchanged with the diagonal row."""
    assert len(M) == len(M[0])
    n = len(M)
    piv = [[int(i == j) for j in range(n)] for i in range(n)]
    for j in range(n - 1):
        is_row = _argmax_flags([abs(M[i][j]) for i in range(j, n)])     # is_row[i - j] = [row i holds the maximum]
        tmp = [list(row) for row in piv]
        acc = _bitmul(tmp[j], is_row[0])
        for i in range(j + 1, n):
            acc = [a + b for a, b in zip(acc, _bitmul(tmp[i], is_row[i - j]))]
        acc = [a.assume(0, 1) if isinstance(a, Lin) else a for a in acc]
        piv[j] = acc
        for jj in range(j + 1, n):
            e = is_row[jj - j]
            c = _circ([e], tmp[jj], tmp[j])
            piv[jj] = [_select(c, e, b, a) for a, b in zip(tmp[jj], tmp[j])]
    return piv


# --------------------------------------------------------------------------------- LU decomposition
class _Quotient:
    """`numerator / pivot` for one pivot, in the mode the caller asked for: a true division per quotient, or ONE
    reciprocal of the pivot into the pure-fraction format (len, 0) shared by every quotient of that pivot (the
    reference's faster, less precise default: qfloat_matrix_inversion.py:421-425, 490-495)."""

    def __init__(self, pivot, qfloat_len, qfloat_ints, true_division, reciprocal=None):
        self.pivot, self.fmt, self.true_division = pivot, (qfloat_len, qfloat_ints), true_division
        self.reciprocal = reciprocal
        if not true_division and reciprocal is None:
            self.reciprocal = pivot.invert(1, qfloat_len, 0)

    def __call__(self, numerator):
        if self.true_division:
            return numerator / self.pivot
        return QFloat.from_mul(numerator, self.reciprocal, *self.fmt)


def _minus_dot(head, left, right, tensorize):
    """head - <left, right> as the reference forms it inside the factorisation: head + (-dot)   (:412-417, 428-438)"""
    return head + qfloat_list_dot_product(left, right, tensorize).neg()


def qfloat_lu_decomposition(M, qfloat_len, qfloat_ints, true_division=False, tensorize=False):
    """Doolittle factorisation of the row-permuted matrix, P M = L U (L unit lower triangular), column by column:
    the entries of U above and on the diagonal depend on each other top-down; the entries of L below the diagonal
    are mutually independent and share the pivot's reciprocal.  Returns (P transposed, L, U) like the reference
    (:377-453), so that M = P L U."""
    n = len(M)
    assert all(len(row) == n for row in M)
    perm = binary_list_matrix(qfloat_pivot_matrix(M))
    A = qfloat_list_matrix_multiply(perm, M)
    lower, upper = zero_list_matrix(n), zero_list_matrix(n)
    for col in range(n):
        lower[col][col] = SignedBinary(1)
        upper[0][col] = A[0][col].copy()
        for row in range(1, col + 1):
            upper[row][col] = _minus_dot(A[row][col], [upper[k][col] for k in range(row)], lower[row][:row], tensorize)
        if col == n - 1:
            if not true_division:      # the reference takes this reciprocal too (its last one is unused)
                upper[col][col].invert(1, qfloat_len, 0)
            break
        over_pivot = _Quotient(upper[col][col], qfloat_len, qfloat_ints, true_division)
        for row in range(col + 1, n):
            rest = A[row][col] if col == 0 else \
                _minus_dot(A[row][col], [upper[k][col] for k in range(col)], lower[row][:col], tensorize)
            lower[row][col] = over_pivot(rest)
    return transpose_2D_list(perm), lower, upper


# --------------------------------------------------------------------------------------- LU inverse
def _forward_row(rhs, lower, tensorize):
    """one right-hand side of L y = rhs (unit diagonal: no division)   (reference :476-485)"""
    y = [rhs[0].copy()]
    for j in range(1, len(rhs)):
        y.append(rhs[j] - qfloat_list_dot_product(lower[j][:j], y[:j], tensorize))
    return y


def _backward_row(y, upper, over_pivot, tensorize):
    """one right-hand side of U x = y, last unknown first   (reference :496-513)"""
    n = len(y)
    x = [None] * n
    x[n - 1] = over_pivot[n - 1](y[n - 1])
    for j in range(n - 2, -1, -1):
        x[j] = over_pivot[j](y[j] - qfloat_list_dot_product(upper[j][j + 1:], x[j + 1:], tensorize))
    return x


def qfloat_lu_inverse(P, L, U, qfloat_len, qfloat_ints, true_division=False, tensorize=False, debug=False):
    """M^-1 from M = P L U: for every row of P as right-hand side, a forward solve against L then a backward solve
    against U; the n right-hand sides are independent of each other, so the scheduler sees them as one n-wide batch.
    The diagonal of U is inverted once (all n reciprocals are independent) unless true_division is set."""
    n = len(L)
    Y = [_forward_row(P[i], L, tensorize) for i in range(n)]
    if true_division:
        recips = [None] * n
    elif tensorize:
        recips = QFloat.multi_invert([U[j][j] for j in range(n)], 1, qfloat_len, 0)
    else:
        recips = [U[j][j].invert(1, qfloat_len, 0) for j in range(n)]
    over_pivot = [_Quotient(U[j][j], qfloat_len, qfloat_ints, true_division, recips[j]) for j in range(n)]
    X = [None] * n
    for i in reversed(range(n)):
        X[i] = _backward_row(Y[i], U, over_pivot, tensorize)
    inverse = transpose_2D_list(X)
    return (inverse, Y, X) if debug else inverse


# ------------------------------------------------------------------------------- 2x2 closed formula
def _adjugate_over_det(entries, det_recip, qfloat_len, qfloat_ints, tensorize):
    """[[d, -b], [-c, a]] * (1 / det) in the output format"""
    a, b, c, d = entries
    if tensorize:
        pa, pb, pc, pd = QFloat.multi_from_mul([a, b, c, d], [det_recip] * 4, qfloat_len, qfloat_ints)
    else:
        pd, pb, pc, pa = (QFloat.from_mul(e, det_recip, qfloat_len, qfloat_ints) for e in (d, b, c, a))
    return [[pd, pb.neg()], [pc.neg(), pa]]


def _inverse_2x2(qfloat_M, qfloat_len, qfloat_ints, tensorize):
    """adj(M) / det(M).  The determinant is formed in the wider format (2 ints + 3 digits, 2 ints of them integer)
    so that a d - b c cannot overflow, then inverted into a pure fraction of `len` digits (reference :526-584)."""
    (a, b), (c, d) = qfloat_M
    wide = (2 * qfloat_ints + 3, 2 * qfloat_ints)
    if tensorize:
        ad, bc = QFloat.multi_from_mul([a, b], [d, c], *wide)
    else:
        ad, bc = QFloat.from_mul(a, d, *wide), QFloat.from_mul(b, c, *wide)
    det = ad + bc.neg()
    return _adjugate_over_det((a, b, c, d), det.invert(1, qfloat_len, 0), qfloat_len, qfloat_ints, tensorize)


def qfloat_inverse_2x2(qfloat_M, qfloat_len, qfloat_ints):
    return _inverse_2x2(qfloat_M, qfloat_len, qfloat_ints, tensorize=False)


def qfloat_inverse_2x2_multi(qfloat_M, qfloat_len, qfloat_ints):
    return _inverse_2x2(qfloat_M, qfloat_len, qfloat_ints, tensorize=True)


# --------------------------------------------------------------------------------- circuit bodies
def _matrix_from_args(qfloat_arrays, qfloat_signs, n, qfloat_len, qfloat_ints, qfloat_base):
    if len(qfloat_arrays) != n * n or len(qfloat_arrays[0]) != qfloat_len:
        raise AssertionError("expected an (n^2, len) digit array")
    return qfloat_arrays_to_qfloat_matrix(qfloat_arrays, qfloat_signs, qfloat_ints, qfloat_base)


def qfloat_pivot(qfloat_arrays, qfloat_signs, params):
    """partial circuit: the pivot matrix only (reference :592-609); params = [n, len, ints, base, ...]"""
    n, qfloat_len, qfloat_ints, qfloat_base = params[:4]
    return qfloat_pivot_matrix(_matrix_from_args(qfloat_arrays, qfloat_signs, n, qfloat_len, qfloat_ints, qfloat_base))


def _lu_factor(qfloat_arrays, qfloat_signs, params, which):
    n, qfloat_len, qfloat_ints, qfloat_base, true_division = params[:5]
    M = _matrix_from_args(qfloat_arrays, qfloat_signs, n, qfloat_len, qfloat_ints, qfloat_base)
    factors = qfloat_lu_decomposition(M, qfloat_len, qfloat_ints, true_division)
    return qfloat_matrix_to_arrays_and_signs(factors[which], qfloat_len, qfloat_ints, qfloat_base)


def qfloat_lu_L(qfloat_arrays, qfloat_signs, params):
    """partial circuit: L of P M = L U (reference :612-639)"""
    return _lu_factor(qfloat_arrays, qfloat_signs, params, 1)


def qfloat_lu_U(qfloat_arrays, qfloat_signs, params):
    """partial circuit: U of P M = L U (reference :642-669)"""
    return _lu_factor(qfloat_arrays, qfloat_signs, params, 2)


def qfloat_matrix_inverse(qfloat_arrays, qfloat_signs, n, qfloat_len, qfloat_ints, qfloat_base, true_division,
                          tensorize=False):
    """The circuit body (reference :672-720).  Input: (n^2, len) digits (>= 0, most significant first) and (n^2,)
    signs; output: (n^2) x (len + 1) scalars, sign in the last column.  n = 2 takes the closed formula, larger
    matrices the pivoted LU factorisation and n pairs of triangular solves."""
    M = _matrix_from_args(qfloat_arrays, qfloat_signs, n, qfloat_len, qfloat_ints, qfloat_base)
    if n == 2:
        inverse = _inverse_2x2(M, qfloat_len, qfloat_ints, tensorize)
    else:
        factors = qfloat_lu_decomposition(M, qfloat_len, qfloat_ints, true_division, tensorize)
        inverse = qfloat_lu_inverse(*factors, qfloat_len, qfloat_ints, true_division, tensorize)
    return qfloat_matrix_to_arrays_and_signs(inverse, qfloat_len, qfloat_ints, qfloat_base)

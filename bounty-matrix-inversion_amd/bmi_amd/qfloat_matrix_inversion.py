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
def matrix_column(M, j):
    return [row[j] for row in M]


def transpose_2D_list(list2D):
    return [list(row) for row in zip(*list2D)]


def map_2D_list(list2D, function):
    return [[function(f) for f in row] for row in list2D]


def binary_list_matrix(M):
    """reference :166-173 — M is an n x n list of 0/1 scalars (ints or encrypted)"""
    return [[SignedBinary(x) for x in row] for row in M]


def zero_list_matrix(n):
    return [[Zero() for _ in range(n)] for _ in range(n)]


def qfloat_list_dot_product(list1, list2, tensorize=False):
    """reference :183-200."""
    if len(list1) != len(list2):
        raise ValueError("Lists should have the same length.")
    if tensorize:
        multiplications = QFloat.multi_from_mul(list1, list2, None, None)
        result = multiplications[0]
        for m in multiplications[1:]:
            result += m
        return result
    result = list1[0] * list2[0]
    for i in range(1, len(list1)):
        result += list1[i] * list2[i]
    return result


def qfloat_list_matrix_multiply(matrix1, matrix2):
    """reference :203-219."""
    return [[qfloat_list_dot_product(matrix1[i], matrix_column(matrix2, j)) for j in range(len(matrix2[0]))]
            for i in range(len(matrix1))]


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


def qfloat_argmax(indices, qfloats):
    """reference :317-328 — encrypted arg-max (first maximum wins)."""
    max_qf = qfloats[0].copy()
    maxi = indices[0]
    for i in range(1, len(indices)):
        is_gt = qfloats[i] > max_qf
        c = _circ([is_gt], max_qf.array, qfloats[i].array)
        max_qf._array = [_select(c, is_gt, a, b) for a, b in zip(qfloats[i].array, max_qf.array)]
        maxi = _select(c, is_gt, indices[i], maxi)
    return maxi


def _eq_index(r, i):
    if isinstance(r, Lin):
        return r.c.lut(r - i, lambda v: int(v == 0))
    return int(r == i)


def _bitmul(row, bit):
    return [x * bit if not (isinstance(x, Lin) and isinstance(bit, Lin)) else x.c.mul(x, bit) for x in row]


def qfloat_pivot_matrix(M):
    """reference :331-369 — oblivious row swaps of an identity matrix."""
    assert len(M) == len(M[0])
    n = len(M)
    piv = [[int(i == j) for j in range(n)] for i in range(n)]
    for j in range(n - 1):
        r = qfloat_argmax(list(range(j, n)), [abs(M[i][j]) for i in range(j, n)])
        tmp = [list(row) for row in piv]
        acc = _bitmul(tmp[j], _eq_index(r, j))
        for i in range(j + 1, n):
            acc = [a + b for a, b in zip(acc, _bitmul(tmp[i], _eq_index(r, i)))]
        acc = [a.assume(0, 1) if isinstance(a, Lin) else a for a in acc]
        piv[j] = acc
        for jj in range(j + 1, n):
            e = _eq_index(r, jj)
            c = _circ([e], tmp[jj], tmp[j])
            piv[jj] = [_select(c, e, b, a) for a, b in zip(tmp[jj], tmp[j])]
    return piv


# --------------------------------------------------------------------------------- LU decomposition
def qfloat_lu_decomposition(M, qfloat_len, qfloat_ints, true_division=False, tensorize=False):
    """reference :377-453 — Doolittle LU of P*M; returns P (transposed), L, U."""
    assert len(M) == len(M[0])
    n = len(M)
    L = zero_list_matrix(n)
    U = zero_list_matrix(n)
    P = binary_list_matrix(qfloat_pivot_matrix(M))
    PM = qfloat_list_matrix_multiply(P, M)
    for j in range(n):
        L[j][j] = SignedBinary(1)
        for i in range(j + 1):
            if i > 0:
                s1 = qfloat_list_dot_product([U[k][j] for k in range(i)], [L[i][k] for k in range(i)], tensorize)
                U[i][j] = PM[i][j] + s1.neg()
            else:
                U[i][j] = PM[i][j].copy()
        if not true_division:
            inv_Ujj = U[j][j].invert(1, qfloat_len, 0)
        for i in range(j + 1, n):
            if j > 0:
                s2 = qfloat_list_dot_product([U[k][j] for k in range(j)], [L[i][k] for k in range(j)], tensorize)
                num = PM[i][j] + s2.neg()
            else:
                num = PM[i][j]
            L[i][j] = (num / U[j][j]) if true_division else QFloat.from_mul(num, inv_Ujj, qfloat_len, qfloat_ints)
    return transpose_2D_list(P), L, U


def qfloat_lu_inverse(P, L, U, qfloat_len, qfloat_ints, true_division=False, tensorize=False, debug=False):
    """reference :461-518 — forward and back substitution."""
    n = len(L)
    Y = zero_list_matrix(n)
    for i in range(n):
        Y[i][0] = P[i][0].copy()
        for j in range(1, n):
            Y[i][j] = P[i][j] - qfloat_list_dot_product([L[j][k] for k in range(j)], [Y[i][k] for k in range(j)],
                                                        tensorize)
    X = zero_list_matrix(n)
    if not true_division:
        if tensorize:
            Ujj_inv = QFloat.multi_invert([U[j][j] for j in range(n)], 1, qfloat_len, 0)
        else:
            Ujj_inv = [U[j][j].invert(1, qfloat_len, 0) for j in range(n)]
    for i in range(n - 1, -1, -1):
        X[i][-1] = (Y[i][-1] / U[-1][-1]) if true_division else QFloat.from_mul(Y[i][-1], Ujj_inv[-1], qfloat_len,
                                                                              qfloat_ints)
        for j in range(n - 2, -1, -1):
            temp = Y[i][j] - qfloat_list_dot_product([U[j][k] for k in range(j + 1, n)],
                                                     [X[i][k] for k in range(j + 1, n)], tensorize)
            X[i][j] = (temp / U[j][j]) if true_division else QFloat.from_mul(temp, Ujj_inv[j], qfloat_len, qfloat_ints)
    if not debug:
        return transpose_2D_list(X)
    return transpose_2D_list(X), Y, X


def qfloat_inverse_2x2(qfloat_M, qfloat_len, qfloat_ints):
    """reference :526-555 — adj(M) / det(M), det in format (2*ints+3, 2*ints)."""
    [a, b] = qfloat_M[0]
    [c, d] = qfloat_M[1]
    ad = QFloat.from_mul(a, d, 2 * qfloat_ints + 3, 2 * qfloat_ints)
    bc = QFloat.from_mul(b, c, 2 * qfloat_ints + 3, 2 * qfloat_ints)
    det = ad + bc.neg()
    det_inv = det.invert(1, qfloat_len, 0)
    mul = lambda x, y: QFloat.from_mul(x, y, qfloat_len, qfloat_ints)  # noqa: E731
    return [[mul(d, det_inv), mul(b, det_inv).neg()], [mul(c, det_inv).neg(), mul(a, det_inv)]]


def qfloat_inverse_2x2_multi(qfloat_M, qfloat_len, qfloat_ints):
    """reference :558-584."""
    [a, b] = qfloat_M[0]
    [c, d] = qfloat_M[1]
    [ad, bc] = QFloat.multi_from_mul([a, b], [d, c], 2 * qfloat_ints + 3, 2 * qfloat_ints)
    det = ad + bc.neg()
    det_inv = det.invert(1, qfloat_len, 0)
    [mula, mulb, mulc, muld] = QFloat.multi_from_mul([a, b, c, d], [det_inv] * 4, qfloat_len, qfloat_ints)
    return [[muld, mulb.neg()], [mulc.neg(), mula]]


def qfloat_matrix_inverse(qfloat_arrays, qfloat_signs, n, qfloat_len, qfloat_ints, qfloat_base, true_division,
                          tensorize=False):
    """reference :672-720 — the circuit body.  Input: (n^2, len) digits (>= 0, MSD first) and (n^2,) signs;
    output: (n^2) x (len + 1) scalars, sign in the last column."""
    assert n * n == len(qfloat_arrays)
    assert qfloat_len == len(qfloat_arrays[0])
    qfloat_M = qfloat_arrays_to_qfloat_matrix(qfloat_arrays, qfloat_signs, qfloat_ints, qfloat_base)
    if n == 2:
        Minv = qfloat_inverse_2x2_multi(qfloat_M, qfloat_len, qfloat_ints) if tensorize \
            else qfloat_inverse_2x2(qfloat_M, qfloat_len, qfloat_ints)
    else:
        P, L, U = qfloat_lu_decomposition(qfloat_M, qfloat_len, qfloat_ints, true_division, tensorize)
        Minv = qfloat_lu_inverse(P, L, U, qfloat_len, qfloat_ints, true_division, tensorize)
    return qfloat_matrix_to_arrays_and_signs(Minv, qfloat_len, qfloat_ints, qfloat_base)

"""ORACLE (test infrastructure) — ctypes loader for oracle/_build/libtfhe_oracle.so
(the exact CPU PBS of oracle/tfhe_oracle.c).  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# BMI_ORACLE_SO selects another build of the same sources (the AddressSanitizer / UBSan one: `make -C oracle asan`, run
# with LD_PRELOAD of libasan; profiles/r02_oracle_asan.txt)
SO = os.environ.get("BMI_ORACLE_SO") or os.path.join(HERE, "_build", "libtfhe_oracle.so")
Q = 0xFFFFFFFF00000001


class Params(C.Structure):
    _fields_ = [("n", C.c_uint32), ("log_N", C.c_uint32), ("k", C.c_uint32), ("bs_levels", C.c_uint32),
                ("bs_base_log", C.c_uint32), ("ks_levels", C.c_uint32), ("ks_base_log", C.c_uint32),
                ("q_bits", C.c_uint32), ("lwe_noise", C.c_double), ("glwe_noise", C.c_double)]

    @property
    def N(self):
        return 1 << self.log_N

    @property
    def big(self):
        return self.k * self.N + 1


GOLD = 0xFFFFFFFF00000001
P49 = 562949952700417
TORUS64 = 65                     # q_bits value selecting q = 2^64 exactly (Concrete's torus)
MODULUS = {64: GOLD, 49: P49, TORUS64: 1 << 64}


def log_q(q_bits):
    """bits of the torus the messages are scaled on: 49 for the 49-bit prime, 64 for Goldilocks and for 2^64"""
    return 49 if q_bits == 49 else 64


def default_params(**kw):
    """North-star set: n=630, N=1024, k=1, l=3 (BASELINE.json); the rest is this build's choice.
    q_bits selects the ciphertext modulus: 64 -> 2^64 - 2^32 + 1, 49 -> 2^49 - 720895."""
    d = dict(n=630, log_N=10, k=1, bs_levels=3, bs_base_log=15, ks_levels=8, ks_base_log=4, q_bits=64,
             lwe_noise=2.0 ** -25, glwe_noise=2.0 ** -44)
    d.update(kw)
    if d["q_bits"] == 49 and "glwe_noise" not in kw:
        d["glwe_noise"] = 2.0 ** -40   # 49-bit modulus: keep the absolute noise above the integer grid
    if d["q_bits"] == TORUS64 and "bs_base_log" not in kw:
        d["bs_base_log"] = 10          # the torus set decomposes in base 2^10 (two-limb key at 48 bits of precision)
    return Params(**d)


def default_bsk_precision(P):
    """bits of precision the library stores a torus bootstrap key at by default (bmi_set_bsk_precision): 48 where the
    decomposition base leaves room for two 24-bit limbs (Bg <= 2^10), else the exact key; 46 (two 23-bit limbs) at N = 2048, 44 (two 22-bit limbs) at N = 4096;
    other moduli: exact"""
    if P.q_bits != TORUS64:
        return 64
    if P.log_N == 11:
        return 46
    if P.log_N == 12:
        return 44
    return 48 if P.bs_base_log <= 10 else 64


def round_key(key, precision):
    """a torus bootstrap key (plain or unrolled) stored at `precision` bits: words rounded half up, as signed integers, to
    multiples of 2^(64 - precision) (ora_round_key); returns a new array"""
    out = u64(key).copy()
    lib().ora_round_key(_p(out), C.c_size_t(out.size), C.c_uint32(int(precision)))
    return out


def set_field(q_bits):
    """selects the oracle's ciphertext modulus (global); returns it"""
    if lib().ora_set_field(C.c_uint32(q_bits)) != 0:
        raise ValueError("unsupported q_bits")
    return MODULUS[q_bits]


def build():
    src = os.path.join(HERE, "tfhe_oracle.c")
    if (not os.path.exists(SO)) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = C.CDLL(SO)
        _lib.ora_ctx_create.restype = C.c_void_p
        _lib.ora_fctx_create.restype = C.c_void_p
        _lib.ora_tctx_create.restype = C.c_void_p
        _lib.ora_modswitch.restype = C.c_uint32
        _lib.ora_num_threads.restype = C.c_int
        _lib.ora_modulus.restype = C.c_uint64
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def u64(a):
    return np.ascontiguousarray(a, dtype=np.uint64)


class Keys:
    def __init__(self, P, sk_small, sk_big, bsk, ksk):
        self.P, self.sk_small, self.sk_big, self.bsk, self.ksk = P, sk_small, sk_big, bsk, ksk


def key_shapes(P):
    rows = (P.k + 1) * P.bs_levels
    return (P.n,), (P.k * P.N,), (P.n, rows, P.k + 1, P.N), (P.k * P.N, P.ks_levels, P.n + 1)


def keygen(P, seed):
    s1, s2, s3, s4 = key_shapes(P)
    sk_small, sk_big = np.zeros(s1, np.uint64), np.zeros(s2, np.uint64)
    bsk, ksk = np.zeros(s3, np.uint64), np.zeros(s4, np.uint64)
    lib().ora_keygen(C.byref(P), C.c_uint64(seed), _p(sk_small), _p(sk_big), _p(bsk), _p(ksk))
    return Keys(P, sk_small, sk_big, bsk, ksk)


def unrolled_key_shape(P):
    return ((P.n + 1) // 2, 3, (P.k + 1) * P.bs_levels, P.k + 1, P.N)


def keygen_bsk_unrolled(P, seed, sk_small, sk_big):
    """unrolled bootstrap key (two LWE coefficients per blind-rotation step): per pair (s, s') the three GGSW
    encryptions of s s', s (1 - s'), (1 - s) s'"""
    bsk3 = np.zeros(unrolled_key_shape(P), np.uint64)
    lib().ora_keygen_bsk_unrolled(C.byref(P), C.c_uint64(seed), _p(u64(sk_small)), _p(u64(sk_big)), _p(bsk3))
    return bsk3


def torus_negacyclic(logN, d, b, schoolbook=False, bound_log=14):
    """d * b mod (X^N + 1, 2^64); d small signed (decomposition digits).  schoolbook=True: wrap-around definition;
    otherwise the Goldilocks half-transform route the torus blind rotation uses."""
    d = np.ascontiguousarray(d, dtype=np.int64)
    b = u64(b)
    c = np.zeros(1 << logN, np.uint64)
    if schoolbook:
        lib().ora_torus_negacyclic_schoolbook(C.c_uint32(logN), _p(d.view(np.uint64)), _p(b), _p(c))
    elif lib().ora_torus_negacyclic_split(C.c_uint32(logN), _p(d), C.c_uint32(bound_log), _p(b), _p(c)) != 0:
        raise ValueError("operand bound too large for the split product")
    return c


def negacyclic(logN, a, b, schoolbook=False):
    a, b = u64(a), u64(b)
    c = np.zeros(1 << logN, np.uint64)
    (lib().ora_negacyclic_schoolbook if schoolbook else lib().ora_negacyclic_ntt)(C.c_uint32(logN), _p(a), _p(b), _p(c))
    return c


def decompose(a, levels, base_log):
    d = np.zeros(levels, np.int64)
    lib().ora_decompose(C.c_uint64(int(a)), C.c_uint32(levels), C.c_uint32(base_log), _p(d))
    return d


def modswitch(a, log2N):
    return int(lib().ora_modswitch(C.c_uint64(int(a)), C.c_uint32(log2N)))


def modulus():
    return int(lib().ora_modulus()) or (1 << 64)      # the library reports 2^64 as 0


def encode(msgs, delta_log):
    """signed integers -> torus values m * 2^delta_log mod q (q = the currently selected field)"""
    q = modulus()
    return np.array([(int(m) << delta_log) % q for m in np.asarray(msgs).reshape(-1)], dtype=np.uint64)


def lwe_encrypt(key, noise, seed, first, torus):
    key, torus = u64(key), u64(torus)
    out = np.zeros((torus.size, key.size + 1), np.uint64)
    lib().ora_lwe_encrypt(_p(key), C.c_uint32(key.size), C.c_double(noise), C.c_uint64(seed), C.c_uint64(first),
                          _p(torus), C.c_uint32(torus.size), _p(out))
    return out


def lwe_phase(key, cts):
    key, cts = u64(key), u64(cts)
    cts = cts.reshape(-1, key.size + 1)
    ph = np.zeros(cts.shape[0], np.uint64)
    lib().ora_lwe_phase(_p(key), C.c_uint32(key.size), _p(cts), C.c_uint32(cts.shape[0]), _p(ph))
    return ph


def decode(phase, delta_log):
    phase = u64(phase)
    m = np.zeros(phase.size, np.int64)
    lib().ora_decode(_p(phase), C.c_uint32(phase.size), C.c_uint32(delta_log), _p(m))
    return m


def make_test_vector(log_N, p, table, out_delta_log):
    table = np.ascontiguousarray(table, dtype=np.int64)
    assert table.size == 1 << p
    tv = np.zeros(1 << log_N, np.uint64)
    lib().ora_make_test_vector(C.c_uint32(log_N), C.c_uint32(p), _p(table), C.c_uint32(out_delta_log), _p(tv))
    return tv


class Ctx:
    """Holds the NTT-domain bootstrap key; runs the PBS stages on the host cores."""

    def __init__(self, P, bsk, ksk):
        self.P = P
        self.bsk, self.ksk = u64(bsk), u64(ksk)  # keep alive (ksk is referenced, not copied)
        self.h = C.c_void_p(lib().ora_ctx_create(C.byref(P), _p(self.bsk), _p(self.ksk)))

    def close(self):
        if self.h:
            lib().ora_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_bsk_unrolled(self, bsk3):
        """attaches an unrolled bootstrap key; `unrolled=True` below then runs the two-coefficients-per-step blind rotation"""
        bsk3 = u64(bsk3)
        assert bsk3.size == int(np.prod(unrolled_key_shape(self.P)))
        lib().ora_ctx_set_bsk_unrolled(self.h, _p(bsk3))

    def keyswitch(self, cts):
        cts = u64(cts).reshape(-1, self.P.big)
        out = np.zeros((cts.shape[0], self.P.n + 1), np.uint64)
        lib().ora_keyswitch_batch(self.h, _p(cts), C.c_uint32(cts.shape[0]), _p(out))
        return out

    def blind_rotate(self, small, tvs, tv_ids, unrolled=False):
        small = u64(small).reshape(-1, self.P.n + 1)
        tvs = u64(tvs).reshape(-1, self.P.N)
        ids = np.ascontiguousarray(tv_ids, dtype=np.uint32)
        out = np.zeros((small.shape[0], self.P.big), np.uint64)
        f = lib().ora_blind_rotate_batch_unrolled if unrolled else lib().ora_blind_rotate_batch
        if f(self.h, _p(small), _p(tvs), _p(ids), C.c_uint32(small.shape[0]), _p(out)) and unrolled:
            raise ValueError("no unrolled bootstrap key attached")
        return out

    def pbs(self, cts, tvs, tv_ids, want_ks=False, unrolled=False):
        cts = u64(cts).reshape(-1, self.P.big)
        tvs = u64(tvs).reshape(-1, self.P.N)
        ids = np.ascontiguousarray(tv_ids, dtype=np.uint32)
        out = np.zeros_like(cts)
        ks = np.zeros((cts.shape[0], self.P.n + 1), np.uint64) if want_ks else None
        f = lib().ora_pbs_batch_unrolled if unrolled else lib().ora_pbs_batch
        if f(self.h, _p(cts), _p(tvs), _p(ids), C.c_uint32(cts.shape[0]), _p(out), _p(ks) if want_ks else None) and unrolled:
            raise ValueError("no unrolled bootstrap key attached")
        return (out, ks) if want_ks else out


class FastCtx:
    """The CPU baseline of bench.py: same PBS, same bits as Ctx.pbs, written for speed (f64 exact arithmetic, vectorisable loops,
    no allocation per call).  49-bit field, and the 2^64 torus at its default set (bootstrap key at 48 bits = two 24-bit limbs,
    base 2^10: exact limb sums mod 2^49 - 720895).  tests/test_oracle_tfhe.py holds it to Ctx.pbs on both."""

    def __init__(self, P, bsk, ksk):
        self.P = P
        bsk, ksk = u64(bsk), u64(ksk)
        self.torus = P.q_bits == TORUS64
        h = (lib().ora_tctx_create if self.torus else lib().ora_fctx_create)(C.byref(P), _p(bsk), _p(ksk))
        if not h:
            raise ValueError("the fast path exists for q_bits = 49 and for the 2^64 torus with a 48-bit key in base <= 2^10 (N = 1024)")
        self.h = C.c_void_p(h)

    def close(self):
        if self.h:
            (lib().ora_tctx_destroy if self.torus else lib().ora_fctx_destroy)(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def pbs(self, cts, tvs, tv_ids, want_ks=False):
        cts = u64(cts).reshape(-1, self.P.big)
        tvs = u64(tvs).reshape(-1, self.P.N)
        ids = np.ascontiguousarray(tv_ids, dtype=np.uint32)
        out = np.zeros_like(cts)
        ks = np.zeros((cts.shape[0], self.P.n + 1), np.uint64) if want_ks else None
        (lib().ora_tfast_pbs_batch if self.torus else lib().ora_fast_pbs_batch)(
            self.h, _p(cts), _p(tvs), _p(ids), C.c_uint32(cts.shape[0]), _p(out), _p(ks) if want_ks else None)
        return (out, ks) if want_ks else out


def lincomb(width, cts, row_ptr, idx, coef, const_body):
    cts = u64(cts).reshape(-1, width)
    row_ptr = np.ascontiguousarray(row_ptr, dtype=np.uint32)
    idx = np.ascontiguousarray(idx, dtype=np.uint32)
    coef = np.ascontiguousarray(coef, dtype=np.int64)
    const_body = u64(const_body)
    out = np.zeros((row_ptr.size - 1, width), np.uint64)
    lib().ora_lincomb(C.c_uint32(width), _p(cts), _p(row_ptr), _p(idx), _p(coef), _p(const_body),
                      C.c_uint32(row_ptr.size - 1), _p(out))
    return out


def num_threads():
    return int(lib().ora_num_threads())


def set_num_threads(n):
    lib().ora_set_num_threads(C.c_int(int(n)))

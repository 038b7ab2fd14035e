"""The exact CPU PBS oracle (oracle/tfhe_oracle.c): self-consistency pins.
Ciphertext-level parity with Concrete is unpinned (Concrete is absent); these pins are
(1) NTT == schoolbook, (2) decrypt(PBS(enc m)) == LUT[m] for all m, (3) noise within budget."""
import numpy as np
import pytest

from oracle import tfhe_oracle as to

Q = to.Q
RNG = np.random.default_rng(42)


def rand_q(n):
    v = RNG.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + RNG.integers(0, 2, n, dtype=np.uint64)
    return np.where(v >= np.uint64(Q), v - np.uint64(Q), v)


@pytest.mark.parametrize("logN", [2, 4, 6, 8])
def test_ntt_matches_schoolbook(logN):
    a, b = rand_q(1 << logN), rand_q(1 << logN)
    assert np.array_equal(to.negacyclic(logN, a, b), to.negacyclic(logN, a, b, schoolbook=True))
    # edge: X^(N-1) * X = -1
    N = 1 << logN
    e1 = np.zeros(N, np.uint64); e1[N - 1] = 1
    e2 = np.zeros(N, np.uint64); e2[1] = 1
    c = to.negacyclic(logN, e1, e2)
    assert int(c[0]) == Q - 1 and not c[1:].any()


def test_fast_mulq_matches_division():
    import ctypes
    lib = to.lib()
    lib.ora_selftest_mulq.restype = ctypes.c_int
    assert lib.ora_selftest_mulq(ctypes.c_uint64(12345), ctypes.c_uint32(2_000_000)) == 1


def test_decompose_recomposes():
    for levels, bl in [(3, 15), (8, 4), (2, 8), (1, 23)]:
        for a in list(rand_q(200)) + [0, 1, Q - 1, Q // 2, Q // 2 + 1, (1 << 63) - 1]:
            d = to.decompose(a, levels, bl)
            assert all(-(1 << (bl - 1)) <= int(x) < (1 << (bl - 1)) for x in d[1:])
            assert -(1 << (bl - 1)) <= int(d[0]) <= (1 << (bl - 1))
            rec = sum(int(d[i]) << (64 - bl * (i + 1)) for i in range(levels))
            c = int(a) if int(a) <= Q // 2 else int(a) - Q
            err = rec - c
            assert abs(err) <= 1 << (63 - levels * bl), (a, d, err)  # exact: the top digit absorbs the carry


def test_modswitch():
    for a in list(rand_q(200)) + [0, Q - 1, Q // 2]:
        for lg in (5, 11, 13):
            assert to.modswitch(a, lg) == ((int(a) * (1 << lg) + Q // 2) // Q) % (1 << lg)


def small_params():
    return to.default_params(n=16, log_N=8, bs_levels=3, bs_base_log=15, ks_levels=8, ks_base_log=4,
                             lwe_noise=2.0 ** -40, glwe_noise=2.0 ** -50)


def test_keygen_structure():
    P = small_params()
    K = to.keygen(P, 7)
    K2 = to.keygen(P, 7)
    assert np.array_equal(K.bsk, K2.bsk) and np.array_equal(K.ksk, K2.ksk)
    assert set(np.unique(K.sk_small)) <= {0, 1} and set(np.unique(K.sk_big)) <= {0, 1}
    assert (K.bsk < np.uint64(Q)).all() and (K.ksk < np.uint64(Q)).all()
    # every KSK row decrypts to sk_big[j] * 2^(64 - 4*(lev+1)) up to noise
    ph = to.lwe_phase(K.sk_small, K.ksk.reshape(-1, P.n + 1)).reshape(P.k * P.N, P.ks_levels)
    for j in range(0, P.k * P.N, 7):
        for lev in range(P.ks_levels):
            want = int(K.sk_big[j]) << (64 - 4 * (lev + 1))
            got = int(ph[j, lev])
            err = (got - want) % Q
            err = err - Q if err > Q // 2 else err
            assert abs(err) < 2 ** 30


@pytest.mark.parametrize("p", [1, 2, 3, 4])
def test_pbs_evaluates_every_lut_entry(p):
    P = small_params()  # N = 256, n = 16: box half-width 8 positions vs mod-switch sigma ~0.9
    K = to.keygen(P, 11 + p)
    ctx = to.Ctx(P, K.bsk, K.ksk)
    M = 1 << p
    msgs = np.arange(-M // 2, M // 2)
    table = np.array([(3 * m * m - 5 * m + 1) % M - M // 2 for m in msgs], dtype=np.int64)
    dl = 63 - p
    tv = to.make_test_vector(P.log_N, p, table, dl)
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 99, 0, to.encode(msgs, dl))
    assert list(to.decode(to.lwe_phase(K.sk_big, ct), dl)) == list(msgs)
    out, ks = ctx.pbs(ct, tv, np.zeros(M, np.uint32), want_ks=True)
    assert list(to.decode(to.lwe_phase(K.sk_small, ks), dl)) == list(msgs)  # keyswitch keeps the message
    assert np.array_equal(ks, ctx.keyswitch(ct))
    assert np.array_equal(out, ctx.blind_rotate(ks, tv, np.zeros(M, np.uint32)))
    assert list(to.decode(to.lwe_phase(K.sk_big, out), dl)) == list(table)
    # output noise well inside half a box
    ph = to.lwe_phase(K.sk_big, out)
    for x, f in zip(ph, table):
        e = (int(x) - (int(f) << dl)) % Q
        e = e - Q if e > Q // 2 else e
        assert abs(e) < 2 ** (dl - 8)


def test_pbs_default_params_identity_and_sign():
    P = to.default_params()
    assert (P.n, P.N, P.k, P.bs_levels) == (630, 1024, 1, 3)  # BASELINE.json north-star set
    K = to.keygen(P, 0x5EED)
    ctx = to.Ctx(P, K.bsk, K.ksk)
    p, dl = 4, 59
    msgs = np.array([-8, -3, 0, 5, 7])
    ident = np.arange(-8, 8, dtype=np.int64)
    neg = np.array([1 if m < 0 else 0 for m in range(-8, 8)], dtype=np.int64)
    tvs = np.stack([to.make_test_vector(10, p, ident, dl), to.make_test_vector(10, p, neg, dl)])
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 5, 0, to.encode(msgs, dl))
    cts = np.concatenate([ct, ct])
    ids = np.array([0] * 5 + [1] * 5, np.uint32)
    out = ctx.pbs(cts, tvs, ids)
    got = to.decode(to.lwe_phase(K.sk_big, out), dl)
    assert list(got[:5]) == list(msgs)
    assert list(got[5:]) == [1, 1, 0, 0, 0]


def test_lincomb():
    P = small_params()
    K = to.keygen(P, 3)
    dl = 59
    msgs = np.array([1, -2, 3, 0])
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 1, 0, to.encode(msgs, dl))
    # row0 = 2*c0 - c1 + 1 ; row1 = c2 + c3 - c0 ; row2 = const -4
    row_ptr = [0, 2, 5, 5]
    idx = [0, 1, 2, 3, 0]
    coef = [2, -1, 1, 1, -1]
    consts = to.encode([1, 0, -4], dl)
    out = to.lincomb(P.big, ct, row_ptr, idx, coef, consts)
    assert list(to.decode(to.lwe_phase(K.sk_big, out), dl)) == [5, 2, -4]


# ----------------------------------------------------------------------------------------------------------
# The second field: q = 2^49 - 720895 (q_bits = 49), the set whose GPU kernels carry exact integers in f64.
@pytest.fixture
def field49():
    q = to.set_field(49)
    yield q
    to.set_field(64)


def rand_q49(n, q):
    return RNG.integers(0, q, n, dtype=np.uint64)


def test_p49_is_an_ntt_prime_and_ntt_matches_schoolbook(field49):
    q = field49
    assert q == 562949952700417 and (q - 1) % (1 << 16) == 0 and pow(5, (q - 1) // 2, q) == q - 1
    for logN in (3, 6, 8):
        a, b = rand_q49(1 << logN, q), rand_q49(1 << logN, q)
        assert np.array_equal(to.negacyclic(logN, a, b), to.negacyclic(logN, a, b, schoolbook=True))


def test_p49_decompose_round_half_up(field49):
    """One decomposition rule on every modulus (oracle/tfhe_oracle.c ora_decompose): round half up to the top
    levels * base_log bits, balanced digits in [-B/2, B/2) from the least significant one, the top digit taking the last carry."""
    q = field49
    for levels, bl in [(3, 15), (8, 4), (2, 8), (2, 15), (1, 23)]:
        shift = 49 - levels * bl
        B = 1 << bl
        for a in list(rand_q49(300, q)) + [0, 1, q - 1, q // 2, q // 2 + 1, (1 << shift) // 2, 3 * (1 << shift) // 2,
                                           (B // 2) << shift, ((B // 2) << shift) - (1 << (shift - 1))]:
            d = [int(x) for x in to.decompose(a, levels, bl)]
            assert all(-B // 2 <= x < B // 2 for x in d[1:]) and -B // 2 <= d[0] <= B // 2
            c = int(a) if int(a) <= q // 2 else int(a) - q
            rec = sum(d[i] << (49 - bl * (i + 1)) for i in range(levels))
            assert rec == ((c + (1 << (shift - 1))) >> shift) << shift          # floor(c / 2^shift + 1/2)
            # digit by digit: r <- floor(r / B + 1/2)
            r = (c + (1 << (shift - 1))) >> shift
            for lev in range(levels - 1, 0, -1):
                rn = (r + B // 2) >> bl
                assert d[lev] == r - (rn << bl)
                r = rn
            assert d[0] == r


@pytest.mark.parametrize("p", [1, 3, 4])
def test_p49_pbs_evaluates_every_lut_entry(field49, p):
    P = to.default_params(n=16, log_N=8, q_bits=49, lwe_noise=2.0 ** -36, glwe_noise=2.0 ** -40)
    K = to.keygen(P, 21 + p)
    assert (K.bsk < np.uint64(field49)).all() and (K.ksk < np.uint64(field49)).all()
    ctx = to.Ctx(P, K.bsk, K.ksk)
    M = 1 << p
    msgs = np.arange(-M // 2, M // 2)
    table = np.array([(5 * m * m + m + 2) % M - M // 2 for m in msgs], dtype=np.int64)
    dl = 48 - p
    tv = to.make_test_vector(P.log_N, p, table, dl)
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 7, 0, to.encode(msgs, dl))
    assert list(to.decode(to.lwe_phase(K.sk_big, ct), dl)) == list(msgs)
    out = ctx.pbs(ct, tv, np.zeros(M, np.uint32))
    assert list(to.decode(to.lwe_phase(K.sk_big, out), dl)) == list(table)


def test_p49_pbs_default_params(field49):
    P = to.default_params(q_bits=49)
    K = to.keygen(P, 0x5EED)
    ctx = to.Ctx(P, K.bsk, K.ksk)
    dl = 44
    msgs = np.array([-8, -3, 0, 5, 7])
    ident = np.arange(-8, 8, dtype=np.int64)
    tv = to.make_test_vector(10, 4, ident, dl)
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 5, 0, to.encode(msgs, dl))
    out = ctx.pbs(ct, tv, np.zeros(5, np.uint32))
    assert list(to.decode(to.lwe_phase(K.sk_big, out), dl)) == list(msgs)
    ph = to.lwe_phase(K.sk_big, out)
    q = field49
    for x, m in zip(ph, msgs):
        e = (int(x) - (int(m) << dl)) % q
        e = e - q if e > q // 2 else e
        assert abs(e) < 2 ** (dl - 6)


def test_pbs_known_answer_digests():
    """tests/golden/pbs_kat.json (tools/gen_pbs_golden.py): SHA-256 of keys, test vector, ciphertexts, keyswitched and
    bootstrapped outputs of a fixed seed / messages / table, for every supported (field, N).  The oracle must still
    produce them (the GPU suite checks the library against the same file)."""
    import hashlib
    import json
    import os
    kat = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pbs_kat.json")))
    h = lambda a: hashlib.sha256(np.ascontiguousarray(a, dtype=np.uint64).tobytes()).hexdigest()  # noqa: E731
    case = next(c for c in kat["cases"] if c["q_bits"] == 49 and c["log_N"] == 10)   # one case keeps the CPU suite short
    to.set_field(49)
    P = to.default_params(q_bits=49, log_N=10)
    K = to.keygen(P, kat["seed"])
    assert (h(K.sk_big), h(K.bsk), h(K.ksk)) == (case["sk_big"], case["bsk"], case["ksk"])
    dl = 49 - 1 - 4
    tv = to.make_test_vector(10, 4, np.array(kat["table"]), dl)
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, kat["seed"], 0, to.encode(kat["msgs"], dl))
    ctx = to.Ctx(P, K.bsk, K.ksk)
    assert h(tv) == case["test_vector"] and h(ct) == case["ciphertexts"]
    assert h(ctx.keyswitch(ct)) == case["keyswitched"]
    assert h(ctx.pbs(ct, tv[None, :], np.zeros(len(kat["msgs"]), np.uint32))) == case["bootstrapped"]
    ctx.close()


# ----------------------------------------------------------------------------------------------------------
# The 2^64 torus (q_bits = 65, Concrete's modulus): wrap-around arithmetic, exact external products through Goldilocks
# transforms of the key's 32-bit halves - pinned to the wrap-around schoolbook product.
@pytest.fixture
def torus():
    to.set_field(to.TORUS64)
    yield 1 << 64
    to.set_field(64)


def rand_u64(n):
    return RNG.integers(0, 2**63, n, dtype=np.uint64) * np.uint64(2) + RNG.integers(0, 2, n, dtype=np.uint64)


@pytest.mark.parametrize("logN", [2, 5, 8, 10])
def test_torus_split_product_matches_wraparound_schoolbook(torus, logN):
    N = 1 << logN
    d = RNG.integers(-(1 << 14), (1 << 14) + 1, N)            # decomposition digits, both extremes reachable
    d[0], d[-1] = -(1 << 14), 1 << 14
    b = rand_u64(N)
    b[:3] = [0, (1 << 64) - 1, 1 << 63]
    assert np.array_equal(to.torus_negacyclic(logN, d, b), to.torus_negacyclic(logN, d, b, schoolbook=True))
    e1 = np.zeros(N, np.int64); e1[N - 1] = 1
    e2 = np.zeros(N, np.uint64); e2[1] = 5
    c = to.torus_negacyclic(logN, e1, e2)
    assert int(c[0]) == (1 << 64) - 5 and not c[1:].any()      # X^(N-1) * 5X = -5
    with pytest.raises(ValueError):
        to.torus_negacyclic(10, d[:1024] if N >= 1024 else np.zeros(1024, np.int64), np.zeros(1024, np.uint64), bound_log=30)


def test_torus_modswitch_decompose_and_wraparound(torus):
    for a in list(rand_u64(200)) + [0, (1 << 64) - 1, 1 << 63, (1 << 63) - 1, 1 << 52, (1 << 52) - 1]:
        for lg in (5, 11):
            assert to.modswitch(a, lg) == ((int(a) * (1 << lg) + (1 << 63)) >> 64) % (1 << lg)
        d = to.decompose(a, 3, 15)
        assert all(-(1 << 14) <= int(x) < (1 << 14) for x in d[1:]) and abs(int(d[0])) <= 1 << 14
        c = int(a) - (1 << 64) if int(a) >> 63 else int(a)
        rec = sum(int(d[i]) << (64 - 15 * (i + 1)) for i in range(3))
        assert abs(rec - c) <= 1 << 18 and rec % (1 << 19) == 0
    assert to.modulus() == 1 << 64 and list(to.encode([-1, 8], 59)) == [(1 << 64) - (1 << 59), 1 << 62]


@pytest.mark.parametrize("p", [1, 4])
def test_torus_pbs_evaluates_every_lut_entry(torus, p):
    P = to.default_params(n=16, log_N=8, q_bits=to.TORUS64, lwe_noise=2.0 ** -40, glwe_noise=2.0 ** -50)
    K = to.keygen(P, 31 + p)
    ctx = to.Ctx(P, K.bsk, K.ksk)
    M = 1 << p
    msgs = np.arange(-M // 2, M // 2)
    table = np.array([(3 * m * m + m + 1) % M - M // 2 for m in msgs], dtype=np.int64)
    dl = 63 - p
    tv = to.make_test_vector(P.log_N, p, table, dl)
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 7, 0, to.encode(msgs, dl))
    assert list(to.decode(to.lwe_phase(K.sk_big, ct), dl)) == list(msgs)
    out = ctx.pbs(ct, tv, np.zeros(M, np.uint32))
    assert list(to.decode(to.lwe_phase(K.sk_big, out), dl)) == list(table)
    ctx.close()


def test_torus_key_stored_at_48_bits_rule_noise_and_pbs(torus):
    """The default torus set keeps its bootstrap key at 48 bits of precision (ora_round_key: words rounded half up, as signed
    integers, to multiples of 2^16).  The rule against its unsigned restatement (what the library computes) on edge words; the
    error it adds (uniform on the grid: variance 2^32 / 12); decrypt(PBS(enc m)) == LUT[m] for every m with the ROUNDED plain
    and unrolled keys at the set's own decomposition (l 3, Bg 2^10)."""
    P = to.default_params(n=15, log_N=8, q_bits=to.TORUS64, lwe_noise=2.0 ** -40)
    assert P.bs_base_log == 10 and to.default_bsk_precision(P) == 48
    assert to.default_bsk_precision(to.default_params(q_bits=to.TORUS64, bs_base_log=15)) == 64
    edge = np.array([0, 1 << 15, (1 << 15) - 1, (3 << 15), 2 ** 64 - 1, 2 ** 64 - (1 << 15), 2 ** 64 - (1 << 15) - 1, 2 ** 63 - 1, 2 ** 63,
                     2 ** 63 - (1 << 15)], dtype=np.uint64)
    words = np.concatenate([edge, RNG.integers(0, 2 ** 63, 5000, dtype=np.uint64) * np.uint64(2) + RNG.integers(0, 2, 5000, dtype=np.uint64)])
    for prec in (48, 42):
        drop = 64 - prec
        want = np.array([(((int(w) + (1 << (drop - 1))) >> drop) << drop) % (1 << 64) for w in words], dtype=np.uint64)
        got = to.round_key(words, prec)
        assert np.array_equal(got, want) and not (got & np.uint64((1 << drop) - 1)).any()
        d = (got - words).astype(np.int64)[len(edge):].astype(np.float64)
        assert np.abs(d).max() <= 2.0 ** (drop - 1) and abs(np.var(d) / (4.0 ** drop / 12) - 1) < 0.1
    assert np.array_equal(to.round_key(words, 64), words)
    K = to.keygen(P, 35)
    bsk = to.round_key(K.bsk, 48)
    ctx = to.Ctx(P, bsk, K.ksk)
    ctx.set_bsk_unrolled(to.round_key(to.keygen_bsk_unrolled(P, 35, K.sk_small, K.sk_big), 48))
    msgs = np.arange(-8, 8)
    table = np.array([(5 * m * m + 3 * m + 2) % 16 - 8 for m in msgs], dtype=np.int64)
    dl = 59
    tv = to.make_test_vector(P.log_N, 4, table, dl)
    ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 7, 0, to.encode(msgs, dl))
    for unrolled in (False, True):
        out = ctx.pbs(ct, tv[None, :], np.zeros(16, np.uint32), unrolled=unrolled)
        assert list(to.decode(to.lwe_phase(K.sk_big, out), dl)) == list(table)
        assert not (out & np.uint64(0xFFFF)).any()        # the accumulator stays on the key's grid
    ctx.close()


# ----------------------------------------------------------------------------------------------------------
# The fast path (bench.py's CPU baseline): bit for bit the generic path.
def test_fast_path_equals_generic_path(field49):
    for kw in (dict(n=24, log_N=10), dict(n=12, log_N=10, ks_levels=5, ks_base_log=6)):
        P = to.default_params(q_bits=49, **kw)
        K = to.keygen(P, 77)
        slow, fast = to.Ctx(P, K.bsk, K.ksk), to.FastCtx(P, K.bsk, K.ksk)
        dl = 44
        tvs = np.stack([to.make_test_vector(10, 4, np.arange(-8, 8), dl), to.make_test_vector(10, 4, RNG.integers(-8, 8, 16), dl)])
        msgs = RNG.integers(-8, 8, 19)                      # two full groups of 8 and a ragged one
        ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 77, 0, to.encode(msgs, dl))
        ct[3, :1024] = rand_q49(1024, field49)              # arbitrary masks, extreme words
        ct[4, :1024] = field49 - 1
        ct[5, :] = 0
        ids = (np.arange(19) % 2).astype(np.uint32)
        a, ka = slow.pbs(ct, tvs, ids, want_ks=True)
        b, kb = fast.pbs(ct, tvs, ids, want_ks=True)
        assert np.array_equal(ka, kb) and np.array_equal(a, b)
        slow.close(); fast.close()
    with pytest.raises(ValueError):
        to.FastCtx(to.default_params(q_bits=64, n=4), np.zeros((4, 6, 2, 1024), np.uint64), np.zeros((1024, 8, 5), np.uint64))


def test_torus_fast_path_equals_generic_path():
    """FastCtx on the 2^64 torus (48-bit key as two 24-bit limbs, exact limb sums mod 2^49 - 720895 in doubles, accumulator as
    word / 2^16 in a double) against the generic path (Goldilocks transforms of the key's 32-bit halves, u64 accumulator) on the
    same rounded key: identical words, keyswitch included; adversarial rows (random masks, extreme words, zeros)."""
    to.set_field(to.TORUS64)
    try:
        for kw in (dict(n=20), dict(n=9, bs_levels=2, ks_levels=5, ks_base_log=6)):
            P = to.default_params(q_bits=to.TORUS64, **kw)
            assert P.bs_base_log == 10 and to.default_bsk_precision(P) == 48
            K = to.keygen(P, 78)
            bsk = to.round_key(K.bsk, 48)
            slow, fast = to.Ctx(P, bsk, K.ksk), to.FastCtx(P, bsk, K.ksk)
            dl = 59
            tvs = np.stack([to.make_test_vector(10, 4, np.arange(-8, 8), dl), to.make_test_vector(10, 4, RNG.integers(-8, 8, 16), dl)])
            msgs = RNG.integers(-8, 8, 11)                      # a full group of 8 and a ragged one
            ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 78, 0, to.encode(msgs, dl))
            ct[3, :1024] = RNG.integers(0, 1 << 63, 1024, dtype=np.uint64) * np.uint64(2) + RNG.integers(0, 2, 1024, dtype=np.uint64)
            ct[4, :1024] = np.uint64(0xFFFFFFFFFFFFFFFF)
            ct[5, :] = 0
            ct[6, :1024] = np.uint64(1 << 63)
            ids = (np.arange(11) % 2).astype(np.uint32)
            a, ka = slow.pbs(ct, tvs, ids, want_ks=True)
            b, kb = fast.pbs(ct, tvs, ids, want_ks=True)
            assert np.array_equal(ka, kb) and np.array_equal(a, b)
            ok = [0, 1, 2, 7, 8, 9, 10]
            tabs = [np.arange(-8, 8), None]
            dec = to.decode(to.lwe_phase(K.sk_big, b[ok]), dl)
            assert [int(d) for d, i in zip(dec, ok) if i % 2 == 0] == [int(tabs[0][msgs[i] + 8]) for i in ok if i % 2 == 0]
            slow.close(); fast.close()
        with pytest.raises(ValueError):   # the exact (unrounded) key has no two-limb form
            P = to.default_params(q_bits=to.TORUS64, n=4)
            to.FastCtx(P, to.keygen(P, 1).bsk, to.keygen(P, 1).ksk)
    finally:
        to.set_field(49)


# ------------------------------------------------------------------ unrolled bootstrap key (two coefficients per step)

@pytest.mark.parametrize("q_bits,n", [(49, 16), (49, 15), (64, 16), (to.TORUS64, 15)])
def test_unrolled_blind_rotation_evaluates_every_lut_entry(q_bits, n):
    """ACC <- ACC + sum_j (X^c_j - 1)(K_j [.] ACC), c = (a + a', a, a'), K = GGSW(s s'), GGSW(s (1 - s')), GGSW((1 - s) s'):
    decrypt(PBS(enc m)) == LUT[m] for every m, even and odd n (an odd n is completed by a zero key bit), every modulus."""
    to.set_field(q_bits)
    try:
        P = to.default_params(n=n, log_N=8, q_bits=q_bits, lwe_noise=2.0 ** -36, glwe_noise=2.0 ** -40)
        K = to.keygen(P, 77 + n)
        ctx = to.Ctx(P, K.bsk, K.ksk)
        with pytest.raises(ValueError):
            ctx.pbs(np.zeros((1, P.big), np.uint64), np.zeros((1, P.N), np.uint64), np.zeros(1, np.uint32), unrolled=True)
        bsk3 = to.keygen_bsk_unrolled(P, 77 + n, K.sk_small, K.sk_big)
        assert bsk3.shape == ((n + 1) // 2, 3, 6, 2, 256)
        ctx.set_bsk_unrolled(bsk3)
        p = 4
        M = 1 << p
        msgs = np.arange(-M // 2, M // 2)
        table = np.array([(3 * m * m + 5 * m + 1) % M - M // 2 for m in msgs], dtype=np.int64)
        dl = to.log_q(q_bits) - 1 - p
        tv = to.make_test_vector(P.log_N, p, table, dl)
        ct = to.lwe_encrypt(K.sk_big, P.glwe_noise, 9, 0, to.encode(msgs, dl))
        out_u, ks = ctx.pbs(ct, tv, np.zeros(M, np.uint32), want_ks=True, unrolled=True)
        out_s = ctx.pbs(ct, tv, np.zeros(M, np.uint32))
        assert list(to.decode(to.lwe_phase(K.sk_big, out_u), dl)) == list(table)
        assert list(to.decode(to.lwe_phase(K.sk_big, out_s), dl)) == list(table)
        assert not np.array_equal(out_u, out_s)                    # different ciphertexts of the same messages
        assert np.array_equal(ctx.blind_rotate(ks, tv, np.zeros(M, np.uint32), unrolled=True), out_u)
        ctx.close()
    finally:
        to.set_field(64)


def test_unrolled_blind_rotation_output_noise_on_the_formula(field49):
    """Unrolling keeps the decomposition term and triples the key-noise term of a pair of steps (three GGSW products,
    each scaled by X^c - 1 of squared norm 2, against two plain products): output variance 3 x the CGGI value when the key
    noise dominates, as it does at the default parameters."""
    P = to.default_params(q_bits=49, n=64)            # 64 coefficients: 1/10 of the work, same per-step noise
    K = to.keygen(P, 0xABCD)
    ctx = to.Ctx(P, K.bsk, K.ksk)
    ctx.set_bsk_unrolled(to.keygen_bsk_unrolled(P, 0xABCD, K.sk_small, K.sk_big))
    dl = 44
    rng = np.random.default_rng(5)
    msgs = rng.integers(-8, 8, 96)
    ident = np.arange(-8, 8, dtype=np.int64)
    tv = to.make_test_vector(10, 4, ident, dl)
    small = to.lwe_encrypt(K.sk_small, 2.0 ** -30, 3, 0, to.encode(msgs, dl))      # straight to the blind rotation
    q = float(field49)
    var = {}
    for unrolled in (False, True):
        out = ctx.blind_rotate(small, tv, np.zeros(msgs.size, np.uint32), unrolled=unrolled)
        ph = to.lwe_phase(K.sk_big, out).astype(np.int64)
        assert list(to.decode(ph.astype(np.uint64), dl)) == list(msgs)
        err = ph - (msgs.astype(np.int64) << dl)
        err = np.where(err > q / 2, err - q, np.where(err < -q / 2, err + q, err)) / q
        var[unrolled] = float(np.mean(err ** 2))
    N, l, Bg = 1024, 3, 2.0 ** 15
    cggi = P.n * (l * 2 * N * (Bg * Bg + 2) / 12 * P.glwe_noise ** 2 + (1 + N / 2) / (12 * Bg ** (2 * l)))
    assert 0.6 < var[False] / cggi < 1.5
    assert 2.0 < var[True] / cggi < 4.2                # 3 x, 96 samples
    ctx.close()

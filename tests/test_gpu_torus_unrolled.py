"""GPU parity tests of the UNROLLED blind rotation on the 2^64 TORUS (Concrete's ciphertext modulus; bmi_set_bsk_unroll(ctx, 2)
on a torus context with the key pinned at 48 bits, k_blind_rotate_lat2u_t64 - the exact-transform predecessor of the unrolled FFT
kernel of tests/test_gpu_torus_unrolled_fft.py): two LWE coefficients per step, exact limb-split products against a bootstrap key
stored at 48 bits of precision.  Bit for bit against oracle/tfhe_oracle.c ora_blind_rotate_extract_unrolled (which rotates in
the coefficient domain and multiplies through Goldilocks transforms of the key's 32-bit halves - a different route to the same
integers) on the same keys; output noise on the formula; the encrypted inverses against the reference's golden digits."""
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EED
QB = 65


def _engine(seed=SEED, **kw):
    from bmi_amd import tfhe
    e = tfhe.Engine(tfhe.default_params(q_bits=QB, **kw))
    e.set_bsk_precision(48)       # the exact-transform unrolled kernel (without this, unrolling picks the 42-bit key and the FFT route:
    e.set_bsk_unroll(2)           # tests/test_gpu_torus_unrolled_fft.py)
    e.keygen(seed)
    return e


def _oracle(eng, bsk3=None):
    from oracle import tfhe_oracle as to
    to.set_field(QB)
    sk_small, sk_big, bsk, ksk = eng.export_keys()
    P = to.default_params(q_bits=QB, n=eng.P.n, bs_levels=eng.P.bs_levels, bs_base_log=eng.P.bs_base_log)
    ctx = to.Ctx(P, bsk, ksk)
    ctx.set_bsk_unrolled(eng.export_bsk_unrolled() if bsk3 is None else bsk3)
    return to, P, ctx, sk_small, sk_big


@pytest.fixture(scope="module")
def eng():
    e = _engine()
    yield e
    e.close()


def test_default_torus_set_and_seeded_keys_match_the_oracle(eng):
    """the torus set: (l, Bg) = (3, 2^10), bootstrap key at 48 bits; the library's keys = the oracle's keys rounded by the
    oracle's own statement of the rule, for the plain and for the unrolled key"""
    from oracle import tfhe_oracle as to
    to.set_field(QB)
    P = to.default_params(q_bits=QB)
    assert (eng.P.n, eng.P.N, eng.P.k, eng.P.bs_levels, eng.P.bs_base_log) == (630, 1024, 1, 3, 10) == (P.n, P.N, P.k, P.bs_levels, P.bs_base_log)
    assert eng.bsk_precision == 48 == to.default_bsk_precision(P)
    K = to.keygen(P, SEED)
    sk_small, sk_big, bsk, ksk = eng.export_keys()
    assert np.array_equal(K.sk_small, sk_small) and np.array_equal(K.ksk, ksk)
    assert np.array_equal(to.round_key(K.bsk, 48), bsk)
    bsk3 = eng.export_bsk_unrolled()
    assert np.array_equal(to.round_key(to.keygen_bsk_unrolled(P, SEED, K.sk_small, K.sk_big), 48), bsk3)
    assert not (bsk3 & np.uint64(0xFFFF)).any() and not (bsk & np.uint64(0xFFFF)).any()


@pytest.mark.parametrize("count", [1, 5, 300, 700])
def test_unrolled_torus_pbs_bit_exact_every_batch_size(eng, count):
    """one kernel for every batch size: the ciphertext bits do not depend on the batch a ciphertext travelled in"""
    to, P, ctx, sk_small, sk_big = _oracle(eng)
    rng = np.random.default_rng(count)
    dl = eng.delta_log()
    tables = [np.arange(-8, 8), rng.integers(-8, 8, 16)]
    lids = [eng.lut_register(t, 4, dl) for t in tables]
    tvs = np.stack([eng.lut_get(l) for l in lids])
    msgs = rng.integers(-8, 8, count)
    sel = rng.integers(0, 2, count).astype(np.uint32)
    ct = eng.encrypt(msgs, dl)
    got = eng.pbs_host(ct, np.array(lids, np.uint32)[sel])
    assert list(eng.decrypt(got, dl)) == [int(tables[s][m + 8]) for s, m in zip(sel, msgs)]
    assert not (got & np.uint64(0xFFFF)).any()      # the accumulator lives on the key's 2^16 grid
    pick = np.arange(count) if count <= 8 else np.unique(np.concatenate([[0, count - 1, 255, 256, 511, 512][:6], rng.integers(0, count, 6)]) % count)
    want = ctx.pbs(ct[pick], tvs, sel[pick], unrolled=True)
    assert np.array_equal(got[pick], want)
    ctx.close()


def test_unrolled_torus_blind_rotation_extreme_inputs(eng):
    """arbitrary small-key words (zeros, maxima, the pair sums that wrap 2N) straight into the blind rotation"""
    to, P, ctx, sk_small, sk_big = _oracle(eng)
    rng = np.random.default_rng(11)
    small = rng.integers(0, 2 ** 63, (10, P.n + 1), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (10, P.n + 1), dtype=np.uint64)
    small[0] = 0                                # every exponent zero: the accumulator is the test polynomial
    small[1] = np.uint64(2 ** 64 - 1)
    small[2] = np.uint64(2 ** 63)               # every a = N: the pair sums wrap to 0
    small[3, ::2] = 0                           # first coefficient of every pair zero
    small[4, 1::2] = 0
    small[5, :-1] = np.uint64(2 ** 53)          # a = 1 everywhere
    lid = eng.lut_register(rng.integers(-8, 8, 16), 4, eng.delta_log())
    ids = np.full(10, lid, np.uint32)
    got = eng.blind_rotate_host(small, ids)
    want = ctx.blind_rotate(small, eng.lut_get(lid)[None, :], np.zeros(10, np.uint32), unrolled=True)
    assert np.array_equal(got, want)
    ctx.close()


@pytest.mark.parametrize("kw", [dict(n=629), dict(n=1024), dict(bs_levels=2), dict(n=1)], ids=["odd_n", "n1024", "l2", "n1"])
def test_unrolled_torus_other_shapes_bit_exact(kw):
    e = _engine(seed=77, **kw)
    try:
        to, P, ctx, sk_small, sk_big = _oracle(e)
        rng = np.random.default_rng(3)
        dl = e.delta_log()
        table = rng.integers(-8, 8, 16)
        lid = e.lut_register(table, 4, dl)
        msgs = rng.integers(-8, 8, 6)
        ct = e.encrypt(msgs, dl)
        got = e.pbs_host(ct, np.full(6, lid, np.uint32))
        if kw.get("n") != 1:      # (one coefficient cannot hold a message's phase; the bits are still compared)
            assert list(e.decrypt(got, dl)) == [int(table[m + 8]) for m in msgs]
        assert np.array_equal(got, ctx.pbs(ct, e.lut_get(lid)[None, :], np.zeros(6, np.uint32), unrolled=True))
        ctx.close()
    finally:
        e.close()


def test_unrolling_on_the_torus_is_refused_where_the_limb_sums_would_not_fit():
    """the three scaled products of an unrolled step must keep every limb sum below p/2: base 2^10 with the 48-bit key only"""
    from bmi_amd import tfhe
    for kw, prec in ((dict(bs_base_log=15), None), (dict(), 64)):
        e = tfhe.Engine(tfhe.default_params(q_bits=QB, **kw))
        try:
            if prec:
                e.set_bsk_precision(prec)
            with pytest.raises(tfhe.BmiError):
                e.set_bsk_unroll(2)
        finally:
            e.close()
    e = tfhe.Engine(tfhe.default_params(q_bits=QB))
    try:
        e.set_bsk_unroll(2)
        assert e.bsk_precision == 42         # the precision nobody chose follows the mode: the unrolled step takes the FFT route
        with pytest.raises(tfhe.BmiError):
            e.set_bsk_precision(64)          # would leave the unrolled mode without a kernel
        e.set_bsk_precision(48)              # the exact-transform unrolled kernel stays selectable
        e.set_bsk_unroll(1)
        assert e.bsk_precision == 48         # (an explicit choice is kept)
    finally:
        e.close()


def test_unrolled_torus_key_under_csprng_and_on_an_evaluation_only_context():
    """production key generation (no seed): plain keys first, the unrolled key derived from the secrets held; a second,
    evaluation-only context imports both evaluation keys (already on the 2^16 grid) and reproduces the same ciphertexts"""
    from bmi_amd import tfhe
    e = tfhe.Engine(tfhe.default_params(q_bits=QB))
    e.keygen()
    e.set_bsk_unroll(2)
    ev = tfhe.Engine(tfhe.default_params(q_bits=QB))
    try:
        to, P, ctx, sk_small, sk_big = _oracle(e)
        rng = np.random.default_rng(8)
        dl = e.delta_log()
        table = rng.integers(-8, 8, 16)
        msgs = rng.integers(-8, 8, 7)
        ct = e.encrypt(msgs, dl)
        lid = e.lut_register(table, 4, dl)
        got = e.pbs_host(ct, np.full(7, lid, np.uint32))
        assert list(e.decrypt(got, dl)) == [int(table[m + 8]) for m in msgs]
        assert np.array_equal(got, ctx.pbs(ct, e.lut_get(lid)[None, :], np.zeros(7, np.uint32), unrolled=True))
        _, _, bsk, ksk = e.export_keys(secret=False)
        ev.import_keys(None, None, bsk, ksk)
        ev.set_bsk_unroll(2)
        with pytest.raises(tfhe.BmiError):      # unrolling selected, no unrolled key yet
            ev.pbs_host(ct, np.full(7, ev.lut_register(table, 4, dl), np.uint32))
        ev.import_bsk_unrolled(e.export_bsk_unrolled())
        assert np.array_equal(ev.pbs_host(ct, np.full(7, ev.lut_register(table, 4, dl), np.uint32)), got)
        ctx.close()
    finally:
        e.close()
        ev.close()


def test_unrolled_torus_output_noise_on_the_formula_and_timing(eng, capsys):
    """4,096 bootstraps: output variance = 3 x the key-noise term (the key noise a bootstrap sees includes the rounding of the
    48-bit key: body + the mask words the GLWE key selects) + the decomposition term of the pairs whose key bits are not both
    zero, doubled by the factor X^c - 1 (at Bg = 2^10 this term is the larger one); prints the latency of the unrolled kernel
    beside the plain torus latency kernel's (same set, same precision)"""
    from bmi_amd import tfhe
    rng = np.random.default_rng(21)
    B = 4096
    dl = eng.delta_log()
    ident = np.arange(-8, 8)
    msgs = rng.integers(-8, 8, B)
    lid = eng.lut_register(ident, 4, dl)
    ct = eng.encrypt(msgs, dl)
    out = eng.pbs_host(ct, np.full(B, lid, np.uint32))
    assert list(eng.decrypt(out, dl)) == list(msgs)
    Q = 1 << 64
    err = np.array([((int(x) - (int(m) << dl)) + Q // 2) % Q - Q // 2 for x, m in zip(eng.phase(out), msgs)], dtype=np.float64) / Q
    P = eng.P
    N, l, Bg = 1024, P.bs_levels, 2.0 ** P.bs_base_log
    sk_small, sk_big = eng.export_keys()[:2]
    hw = int(sk_big.sum())
    sigma2 = P.glwe_noise ** 2 + (1 + hw) * 4.0 ** 16 / 12 / 2.0 ** 128
    key_term = P.n * l * 2 * N * (Bg * Bg + 2) / 12 * sigma2
    # decomposition rounding: a step's rounding error is multiplied by the bit its GGSW encrypts - exactly one of the three keys
    # of a pair encrypts 1 unless both key bits are 0 - then scaled by X^c - 1 (x 2) and carried by the GLWE key's set bits
    live_pairs = int((sk_small[0::2] | np.concatenate([sk_small[1::2], np.zeros(sk_small[0::2].size - sk_small[1::2].size, np.uint64)])).sum())
    dec_term = 2 * live_pairs * (1 + hw) / (12 * Bg ** (2 * l))
    want = 3 * key_term + dec_term
    ratio = float(np.mean(err ** 2)) / want
    plain = tfhe.Engine(tfhe.default_params(q_bits=QB))
    plain.keygen(SEED)
    t = {}
    for name, e in (("unrolled", eng), ("plain latency kernel", plain)):
        l2 = e.lut_register(ident, 4, dl)
        for cnt in (1, 256):
            c = e.encrypt(msgs[:cnt], dl)
            ids = np.full(cnt, l2, np.uint32)
            small = e.keyswitch_host(c)
            e.blind_rotate_host(small, ids)
            t0 = time.perf_counter()
            for _ in range(3):
                e.blind_rotate_host(small, ids)
            t[(name, cnt)] = (time.perf_counter() - t0) / 3 * 1e3
    plain.close()
    with capsys.disabled():
        print(f"\nunrolled torus PBS: output log2 std {0.5 * np.log2(np.mean(err ** 2)):.2f} (3 x key term + rounding term: {0.5 * np.log2(want):.2f}, "
              f"variance ratio {ratio:.3f}, worst 2^{np.log2(np.abs(err).max()):.2f}); blind rotation ms (host-buffer calls, copies included): "
              + ", ".join(f"{k[0]} x{k[1]}: {v:.2f}" for k, v in t.items()))
    assert 0.88 < ratio < 1.12


@pytest.mark.parametrize("tag", ["baseline_n2_len20_ints8", "baseline_n3_len30_ints12", "baseline_n4_len40_ints16", "baseline_n8_len48_ints16",
                                 "overflow_digit_2x2", "overflow_digit_3x3", "uniform_3x3_small_truediv", "uniform_2x2_tensorize"])
def test_encrypted_inverse_on_the_torus_with_the_unrolled_key_matches_reference_golden(tag, capsys):
    """BASELINE configs 2-5, the overflow-digit cases and the true-division / tensorize modes on 2^64-torus ciphertexts with
    EncryptedMatrixInversion(q_bits=65, unroll=True) - the wrapper's unrolled torus engine: 42-bit key, k_blind_rotate_lat2u_t64f: decrypted digits == the reference's plaintext QFloat output
    (tests/golden/inverse.json, generated from the reference)."""
    from bmi_amd.main import EncryptedMatrixInversion
    with open(os.path.join(os.path.dirname(__file__), "golden", "inverse.json")) as f:
        cases = json.load(f)
    c = next(x for x in cases if x["tag"] == tag)
    emi = EncryptedMatrixInversion(c["n"], None, 2, c["len"], c["ints"], c["true_division"], c["tensorize"], unroll=True, q_bits=QB)
    try:
        emi.keygen()                                # CSPRNG keys
        assert emi.engine.q_bits == QB and emi.engine.bsk_precision == 42 and emi.engine.P.glwe_noise == 2.0 ** -44   # the FFT-route unrolled kernel
        M = np.array(c["M"]).reshape(c["n"], c["n"])
        q, s = emi.quantize(M)
        enc = emi.encrypt(q, s)
        emi._executor()
        if c["n"] < 8:
            emi.evaluate(enc)                       # warm-up (skipped for the long 8x8 run)
        t0 = time.time()
        res = emi.evaluate(enc)
        wall = time.time() - t0
        out = emi.decrypt(res)
        assert out.tolist() == c["out"], f"circuit failure probability by noise under these parameters: {emi.error_budget['p_fail']:.1e}"
        with capsys.disabled():
            print(f"\ntorus, unrolled key, {tag}: evaluate {wall:.2f} s, {emi.circuit.summary()['depth']} levels, {wall / emi.circuit.summary()['depth'] * 1e3:.2f} ms per level")
    finally:
        emi.engine.close()

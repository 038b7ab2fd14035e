"""GPU parity tests of the UNROLLED blind rotation on the 2^64 torus through the FLOATING-POINT transform
(k_blind_rotate_lat2u_t64f, csrc/bmi_kernels_t64fu.hip): bmi_set_bsk_precision(ctx, 42) + bmi_set_bsk_unroll(ctx, 2) - two LWE
coefficients per step, the three scaled GGSW products of a step summed per limb as exact integers (below 2^45: the key is stored
at 42 bits = two 21-bit limbs so that the six-times larger sums stay inside the transform's certified range).  Bit for bit against
oracle/tfhe_oracle.c ora_blind_rotate_extract_unrolled (integer arithmetic: coefficient-domain rotation, Goldilocks transforms of
the key's 32-bit halves) on the same rounded keys; the rounding distance of the limb sums; output noise on the formula; BASELINE
configs 2-4 through EncryptedMatrixInversion(q_bits=65, unroll=True) against the reference's golden digits."""
import json
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EED
QB = 65
PREC = 42


def _engine(seed=SEED, **kw):
    from bmi_amd import tfhe
    e = tfhe.Engine(tfhe.default_params(q_bits=QB, **kw))
    e.set_bsk_unroll(2)           # selects the 42-bit key and the FFT route by itself ...
    assert e.bsk_precision == PREC
    e.set_bsk_precision(PREC)     # ... and the explicit choice is accepted in unrolled mode
    e.keygen(seed)
    return e


def _oracle(eng):
    from oracle import tfhe_oracle as to
    to.set_field(QB)
    sk_small, sk_big, bsk, ksk = eng.export_keys()
    P = to.default_params(q_bits=QB, n=eng.P.n, bs_levels=eng.P.bs_levels, bs_base_log=eng.P.bs_base_log)
    ctx = to.Ctx(P, bsk, ksk)
    ctx.set_bsk_unrolled(eng.export_bsk_unrolled())
    return to, P, ctx, sk_small, sk_big


@pytest.fixture(scope="module")
def eng():
    e = _engine()
    yield e
    e.close()


def test_keys_are_the_oracles_rounded_to_42_bits_and_plain_pbs_is_refused(eng):
    from bmi_amd import tfhe
    from oracle import tfhe_oracle as to
    to.set_field(QB)
    P = to.default_params(q_bits=QB)
    assert eng.bsk_precision == PREC
    K = to.keygen(P, SEED)
    _, _, bsk, ksk = eng.export_keys()
    assert np.array_equal(to.round_key(K.bsk, PREC), bsk) and np.array_equal(K.ksk, ksk)
    bsk3 = eng.export_bsk_unrolled()
    assert np.array_equal(to.round_key(to.keygen_bsk_unrolled(P, SEED, K.sk_small, K.sk_big), PREC), bsk3)
    assert not (bsk3 & np.uint64((1 << 22) - 1)).any()
    # the 42-bit key at base 2^10 exists for the unrolled kernel only: a context in plain mode refuses the precision, and one that
    # leaves the unrolled mode with the precision pinned refuses to bootstrap
    e = tfhe.Engine(tfhe.default_params(q_bits=QB))
    try:
        with pytest.raises(tfhe.BmiError):
            e.set_bsk_precision(PREC)
        e.set_bsk_unroll(2)
        e.set_bsk_precision(PREC)
        e.set_bsk_unroll(1)
        e.keygen(SEED)
        lid = e.lut_register(np.arange(-8, 8), 4, e.delta_log())
        with pytest.raises(tfhe.BmiError):
            e.blind_rotate_host(np.zeros((1, e.P.small), np.uint64), np.full(1, lid, np.uint32))
    finally:
        e.close()


@pytest.mark.parametrize("count", [1, 2, 5, 256, 257, 300, 701])
def test_two_ciphertexts_per_workgroup_give_the_same_words(eng, count):
    """the throughput form (k_blind_rotate_tp2u_t64f: two ciphertexts per workgroup sharing every key word in registers; what auto
    runs beyond 256 ciphertexts) against the one-ciphertext form, word for word, for even and odd batches and adversarial rows"""
    rng = np.random.default_rng(1000 + count)
    lid = eng.lut_register(rng.integers(-8, 8, 16), 4, eng.delta_log())
    ids = np.full(count, lid, np.uint32)
    small = rng.integers(0, 1 << 63, (count, eng.P.small), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (count, eng.P.small), dtype=np.uint64)
    small[0] = 0                                   # a ciphertext whose every step is skipped beside one that takes them all
    if count > 2:
        small[2, ::2] = 0
        small[count - 1, 14::16] = 0
        small[count - 1, 15::16] = 0
    try:
        eng.set_kernel_variant(2)
        one = eng.blind_rotate_host(small, ids)
        eng.set_kernel_variant(1)
        two = eng.blind_rotate_host(small, ids)
    finally:
        eng.set_kernel_variant(0)
    assert np.array_equal(one, two)
    assert np.array_equal(eng.blind_rotate_host(small, ids), one)     # auto: whichever it picked


@pytest.mark.parametrize("count", [1, 5, 300, 700])
def test_unrolled_fft_pbs_bit_exact_every_batch_size(eng, count):
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
    assert not (got & np.uint64((1 << 22) - 1)).any()      # the accumulator lives on the key's 2^22 grid
    pick = np.arange(count) if count <= 8 else np.unique(np.concatenate([[0, count - 1, 255, 256, 511, 512][:6], rng.integers(0, count, 6)]) % count)
    assert np.array_equal(got[pick], ctx.pbs(ct[pick], tvs, sel[pick], unrolled=True))
    ctx.close()


def test_unrolled_fft_blind_rotation_extreme_inputs_and_rounding_margin(eng):
    """arbitrary small-key words (zeros, maxima, pair sums that wrap 2N, skipped steps) straight into the blind rotation; and the
    largest distance of a limb sum from the integer it is rounded to over 512 bootstraps of uniformly random words (digits at
    their full range): far below 1/2 (a-priori bound 0.29)"""
    to, P, ctx, sk_small, sk_big = _oracle(eng)
    rng = np.random.default_rng(11)
    small = rng.integers(0, 2 ** 63, (10, P.n + 1), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (10, P.n + 1), dtype=np.uint64)
    small[0] = 0                                # every exponent zero: the accumulator is the test polynomial
    small[1] = np.uint64(2 ** 64 - 1)
    small[2] = np.uint64(2 ** 63)               # every a = N: the pair sums wrap to 0
    small[3, ::2] = 0                           # first coefficient of every pair zero
    small[4, 1::2] = 0
    small[5, :-1] = np.uint64(2 ** 53)          # a = 1 everywhere
    small[6, 14::16] = 0                        # a whole pair zero every eighth step: the f64 accumulator is re-centred every
    small[6, 15::16] = 0                        # eight steps TAKEN
    lid = eng.lut_register(rng.integers(-8, 8, 16), 4, eng.delta_log())
    ids = np.full(10, lid, np.uint32)
    got = eng.blind_rotate_host(small, ids)
    assert np.array_equal(got, ctx.blind_rotate(small, eng.lut_get(lid)[None, :], np.zeros(10, np.uint32), unrolled=True))
    ctx.close()
    count = 512
    rnd = rng.integers(0, 1 << 63, (count, P.n + 1), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (count, P.n + 1), dtype=np.uint64)
    out, dist = eng.fft_margin_host(rnd, np.full(count, lid, np.uint32))
    print(f"\nunrolled FFT kernel: largest distance from an integer before rounding 2^{np.log2(max(dist, 1e-300)):.1f}")
    assert 0.0 < dist < 2.0 ** -7, dist
    assert np.array_equal(out, eng.blind_rotate_host(rnd, np.full(count, lid, np.uint32)))


@pytest.mark.parametrize("kw", [dict(n=629), dict(n=1024), dict(bs_levels=2), dict(n=1)], ids=["odd_n", "n1024", "l2", "n1"])
def test_unrolled_fft_other_shapes_bit_exact(kw):
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
        if kw.get("n") != 1:
            assert list(e.decrypt(got, dl)) == [int(table[m + 8]) for m in msgs]
        assert np.array_equal(got, ctx.pbs(ct, e.lut_get(lid)[None, :], np.zeros(6, np.uint32), unrolled=True))
        ctx.close()
    finally:
        e.close()


def test_unrolled_fft_output_noise_on_the_formula_and_timing(eng, capsys):
    """2,048 bootstraps: the output variance against the unrolled CGGI formula with the 42-bit key's effective noise
    (bmi_amd/error_budget.pbs_output_variance with the exact key weights); latency of 1 and of a full round of 256"""
    import torch
    from bmi_amd import error_budget
    rng = np.random.default_rng(5)
    dl = eng.delta_log()
    table = rng.integers(-8, 8, 16)
    lid = eng.lut_register(table, 4, dl)
    count = 2048
    msgs = rng.integers(-8, 8, count)
    ct = eng.encrypt(msgs, dl)
    out = eng.pbs_host(ct, np.full(count, lid, np.uint32))
    want = table[msgs + 8]
    assert np.array_equal(eng.decrypt(out, dl), want)
    Q = 1 << 64
    err = np.array([((int(x) - (int(m) << dl)) + Q // 2) % Q - Q // 2 for x, m in zip(eng.phase(out), want)], dtype=np.float64) / Q
    sk_small, sk_big = eng.export_keys()[:2]
    pairs = sk_small[0::2].copy()
    pairs[:sk_small[1::2].size] |= sk_small[1::2]
    P = eng.P
    Bg = 2.0 ** P.bs_base_log
    hb = int(sk_big.sum())
    s2 = P.glwe_noise ** 2 + (1 + hb) * 4.0 ** (64 - PREC) / 12 / 2.0 ** 128
    analytic = 3 * P.n * P.bs_levels * 2 * P.N * (Bg * Bg + 2) / 12.0 * s2 + 2 * int(pairs.sum()) * (1 + hb) / (12.0 * Bg ** (2 * P.bs_levels))
    ratio = float(np.var(err)) / analytic
    model = error_budget.pbs_output_variance(P, PREC, unroll=True)
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    small = eng.keyswitch_host(ct[:256])
    d_small = torch.from_numpy(small.view(np.int64)).to(dev)
    d_ids = torch.full((256,), lid, dtype=torch.int32, device=dev)
    d_out = torch.empty((256, P.N + 1), dtype=torch.int64, device=dev)
    t = {}
    for cnt in (1, 256):
        eng.blind_rotate(d_small, d_ids, cnt, d_out, s)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(5):
            eng.blind_rotate(d_small, d_ids, cnt, d_out, s)
        b.record()
        torch.cuda.synchronize()
        t[cnt] = a.elapsed_time(b) / 5
    with capsys.disabled():
        print(f"\nunrolled FFT kernel (42-bit key): output log2 std {0.5 * np.log2(np.var(err)):.2f} (formula {0.5 * np.log2(analytic):.2f}, variance "
              f"ratio {ratio:.3f}; error-budget model {0.5 * np.log2(model):.2f}); blind rotation {t[1]:.2f} ms for 1, {t[256]:.2f} ms for 256")
    assert 0.88 < ratio < 1.12 and abs(0.5 * np.log2(model / analytic)) < 0.2


@pytest.mark.parametrize("tag", ["baseline_n2_len20_ints8", "baseline_n3_len30_ints12", "baseline_n4_len40_ints16"])
def test_encrypted_inverse_with_the_unrolled_fft_kernel_matches_reference_golden(eng, tag, capsys):
    from bmi_amd.main import EncryptedMatrixInversion
    with open(os.path.join(os.path.dirname(__file__), "golden", "inverse.json")) as f:
        c = next(x for x in json.load(f) if x["tag"] == tag)
    emi = EncryptedMatrixInversion(c["n"], None, 2, c["len"], c["ints"], False, False, engine=eng, unroll=True)
    M = np.array(c["M"]).reshape(c["n"], c["n"])
    q, s = emi.quantize(M)
    enc = emi.encrypt(q, s)
    emi._executor()
    emi.evaluate(enc)                           # warm-up
    t0 = time.time()
    res = emi.evaluate(enc)
    wall = time.time() - t0
    out = emi.decrypt(res)
    assert out.tolist() == c["out"], f"circuit failure probability by noise under these parameters: {emi.error_budget['p_fail']:.1e}"
    depth = emi.circuit.summary()["depth"]
    with capsys.disabled():
        print(f"\ntorus, unrolled FFT kernel, {tag}: evaluate {wall:.2f} s, {depth} levels, {wall / depth * 1e3:.2f} ms per level, "
              f"p_fail {emi.error_budget['p_fail']:.1e}")

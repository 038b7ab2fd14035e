"""GPU parity tests of the 2^64 torus at N = 2048 (preset "secure128_torus": n 742, k 1, l 3 x 10 bits, bootstrap key at 46 bits
of precision = two 23-bit limbs; csrc/bmi_kernels_t64w.hip, fft_quarter_f64.hpp).  The specification is the oracle's INTEGER
arithmetic on the same (rounded, exported) key - the generic path of oracle/tfhe_oracle.c, whose torus product goes through
Goldilocks transforms of the key's 32-bit halves, a different route from the GPU's floating-point transform on purpose: every
output word must be identical for every batch shape, and the limb sums must sit far from the half-integers when they are rounded."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EED
QB = 65


@pytest.fixture(scope="module")
def eng():
    from bmi_amd import tfhe
    e = tfhe.Engine(tfhe.preset_params("secure128_torus"))
    e.keygen(SEED + 9)
    yield e
    e.close()


@pytest.fixture(scope="module")
def ora(eng):
    from bmi_amd import tfhe
    from oracle import tfhe_oracle as to
    to.set_field(QB)
    OP = to.Params(**{f: getattr(eng.P, f) for f, _ in tfhe.Params._fields_})
    sk_small, sk_big, bsk, ksk = eng.export_keys()
    ctx = to.Ctx(OP, bsk, ksk)
    yield to, ctx, OP, sk_small, sk_big
    ctx.close()


def test_preset_shape_keys_and_refusals(eng, ora):
    """the preset's numbers; the key generator reproduces the oracle's word for word up to the rounding to 46 bits; what this
    shape refuses (other precisions, unrolling, tables below the key's grid)"""
    from bmi_amd import tfhe
    to, _, OP, sk_small, _ = ora
    P = eng.P
    assert (P.n, P.N, P.k, P.bs_levels, P.bs_base_log, P.q_bits, P.ks_levels, P.ks_base_log) == (742, 2048, 1, 3, 10, QB, 8, 2)
    assert abs(np.log2(P.lwe_noise) + 17.11) < 0.01 and eng.bsk_precision == 46 == to.default_bsk_precision(OP)
    K = to.keygen(OP, SEED + 9)
    _, _, bsk, ksk = eng.export_keys()
    assert np.array_equal(to.round_key(K.bsk, 46), bsk) and np.array_equal(K.ksk, ksk) and np.array_equal(K.sk_small, sk_small)
    assert not np.array_equal(K.bsk, bsk) and np.all(bsk & np.uint64((1 << 18) - 1) == 0)
    e2 = tfhe.Engine(tfhe.preset_params("secure128_torus"))
    try:
        for bits in (64, 48, 42):
            with pytest.raises(tfhe.BmiError):
                e2.set_bsk_precision(bits)
        e2.set_bsk_precision(46)
        with pytest.raises(tfhe.BmiError):
            e2.set_bsk_unroll(2)
    finally:
        e2.close()
    with pytest.raises(tfhe.BmiError):   # accumulators on the rounded key are multiples of 2^18: no table below that scale
        eng.lut_register(np.arange(-8, 8), 4, 17)
    with pytest.raises(tfhe.BmiError):   # N = 2048 on the torus: (l, Bg) = (3 or 2, 2^10) only
        tfhe.Engine(tfhe.preset_params("secure128_torus", bs_base_log=15))


def _batch(eng, count, seed):
    rng = np.random.default_rng(seed)
    tables = [np.arange(-8, 8), rng.integers(-8, 8, 16)]
    ids = np.array([eng.lut_register(t, 4, eng.delta_log()) for t in tables], np.uint32)
    tvs = np.stack([eng.lut_get(i) for i in ids])
    msgs = rng.integers(-8, 8, count)
    sel = rng.integers(0, 2, count).astype(np.uint32)
    small = eng.keyswitch_host(eng.encrypt(msgs, eng.delta_log()))
    # adversarial rows: uniformly random words (not a valid encryption: they drive the digits to their full range), zeros, ones
    small[0] = rng.integers(0, 1 << 63, small.shape[1], dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, small.shape[1], dtype=np.uint64)
    if count > 2:
        small[1] = 0
        small[2] = np.uint64(0xFFFFFFFFFFFFFFFF)
    if count > 4:
        # random words whose every eighth coefficient switches to 0 (a skipped step): the f64 accumulator is re-centred every eight
        # steps TAKEN, whichever steps a ciphertext skips
        small[3] = rng.integers(0, 1 << 63, small.shape[1], dtype=np.uint64) * np.uint64(2)
        small[3, 7::8] = 0
    return tables, ids, tvs, msgs, sel, small


@pytest.mark.parametrize("count", [1, 5, 257, 600])
def test_blind_rotation_bit_exact_every_batch_shape(eng, ora, count):
    to, octx, _, _, sk_big = ora
    tables, ids, tvs, msgs, sel, small = _batch(eng, count, 300 + count)
    got = eng.blind_rotate_host(small, ids[sel])
    rng = np.random.default_rng(count)
    pick = np.arange(count) if count <= 8 else np.unique(np.concatenate([[0, 1, 2, 3, 4, count - 1, 255, 256], rng.integers(0, count, 2)]) % count)
    assert np.array_equal(got[pick], octx.blind_rotate(small[pick], tvs, sel[pick]))
    ok = np.arange(4, count)
    if ok.size:
        dec = to.decode(to.lwe_phase(sk_big, got[ok]), eng.delta_log())
        assert list(dec) == [int(tables[s][m + 8]) for s, m in zip(sel[ok], msgs[ok])]


def test_keyswitch_and_whole_pbs_bit_exact_noise_and_margin(eng, ora):
    """keyswitch and the whole PBS against the oracle; every 4-bit message through a random table; keyswitch noise at its analytic
    value and the look-up margin it leaves; bootstrap output noise on the CGGI formula with the rounded key's effective noise"""
    from test_gpu_parity import cggi_output_variance, effective_params
    to, octx, _, sk_small, sk_big = ora
    P = eng.P
    rng = np.random.default_rng(43)
    dl = eng.delta_log()
    table = rng.integers(-8, 8, 16)
    lid = eng.lut_register(table, 4, dl)
    msgs = np.concatenate([np.arange(-8, 8)] * 64)                    # 1,024 ciphertexts, every message 64 times
    ct = eng.encrypt(msgs, dl)
    small = eng.keyswitch_host(ct)
    assert np.array_equal(small[:24], octx.keyswitch(ct[:24]))
    out = eng.pbs_host(ct, np.full(msgs.size, lid, np.uint32))
    pick = rng.choice(msgs.size, 4, replace=False)
    assert np.array_equal(out[pick], octx.pbs(ct[pick], eng.lut_get(lid)[None, :], np.zeros(4, np.uint32)))
    assert np.array_equal(eng.decrypt(out, dl), table[msgs + 8])
    Q = 1 << 64
    ph = to.lwe_phase(sk_small, small)
    err = np.array([((int(x) - (int(m) << dl)) + Q // 2) % Q - Q // 2 for x, m in zip(ph, msgs)], dtype=np.float64) / Q
    B = 2.0 ** P.ks_base_log
    kN = P.k * P.N
    analytic = kN * P.ks_levels * (B * B + 2) / 12.0 * P.lwe_noise ** 2 + kN / 2.0 / (12.0 * B ** (2 * P.ks_levels))
    ratio = float(np.var(err)) / analytic
    sigma_pos = np.sqrt(np.var(err) * (2 * P.N) ** 2 + (P.n / 2.0 + 1) / 12.0)
    margin = (P.N / 32.0) / sigma_pos       # half a 4-bit box (boxes are N / 2^4 positions wide) in sigmas
    print(f"\nsecure128_torus: keyswitch log2 std {0.5 * np.log2(np.var(err)):.2f} (analytic {0.5 * np.log2(analytic):.2f}, ratio "
          f"{ratio:.3f}); positions sigma {sigma_pos:.2f} of {2 * P.N}; 4-bit look-up margin {margin:.1f} sigma")
    assert 0.75 < ratio < 1.3 and margin > 8.0
    want_m = table[msgs + 8]
    oerr = np.array([((int(x) - (int(m) << dl)) + Q // 2) % Q - Q // 2 for x, m in zip(eng.phase(out), want_m)], dtype=np.float64) / Q
    oratio = float(np.var(oerr)) / cggi_output_variance(effective_params(eng), 64, hw_small=int(sk_small.sum()), hw_big=int(sk_big.sum()))
    print(f"secure128_torus: PBS output log2 std {0.5 * np.log2(np.var(oerr)):.2f} (variance / formula {oratio:.3f})")
    assert 0.85 < oratio < 1.15 and np.var(oerr) * 75 ** 2 < np.var(err) / 4     # x75: the widest linear combination of the circuits
    import time
    ids256 = np.full(256, lid, np.uint32)
    eng.pbs_host(ct[:256], ids256)
    t0 = time.perf_counter(); eng.pbs_host(ct[:256], ids256); t256 = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.pbs_host(ct[:1], ids256[:1]); t1 = time.perf_counter() - t0
    print(f"secure128_torus: 1 PBS {t1 * 1e3:.2f} ms, 256 PBS {t256 * 1e3:.2f} ms (host-buffer calls, copies included)")


def test_rounding_margin_of_the_limb_sums(eng):
    """bmi_fft_margin_host on this shape: over 1,024 bootstraps (512 of them uniformly random words, which drive the digits to
    their full range) the limb sums stay within 2^-9 of the integers they are rounded to - against the 1/2 at which a result
    would change (a-priori bound 0.42: tools/fft_bound.py) - and the words equal the product kernel's"""
    rng = np.random.default_rng(12)
    count = 1024
    lid = eng.lut_register(rng.integers(-8, 8, 16), 4, eng.delta_log())
    small = eng.keyswitch_host(eng.encrypt(rng.integers(-8, 8, count), eng.delta_log()))
    small[:512] = rng.integers(0, 1 << 63, (512, small.shape[1]), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (512, small.shape[1]), dtype=np.uint64)
    ids = np.full(count, lid, np.uint32)
    out, dist = eng.fft_margin_host(small, ids)
    print(f"\nsecure128_torus: largest distance from an integer before rounding 2^{np.log2(max(dist, 1e-300)):.1f}")
    assert 0.0 < dist < 2.0 ** -9, dist
    assert np.array_equal(out, eng.blind_rotate_host(small, ids))


def test_l2_shape_bit_exact():
    """(l, Bg) = (2, 2^10) at N = 2048: the other instantiated shape (16 forward tasks: one per wavefront)"""
    from bmi_amd import tfhe
    from oracle import tfhe_oracle as to
    e = tfhe.Engine(tfhe.preset_params("secure128_torus", bs_levels=2, n=33))
    try:
        e.keygen(SEED)
        to.set_field(QB)
        OP = to.Params(**{f: getattr(e.P, f) for f, _ in tfhe.Params._fields_})
        _, _, bsk, ksk = e.export_keys()
        octx = to.Ctx(OP, bsk, ksk)
        rng = np.random.default_rng(3)
        lid = e.lut_register(rng.integers(-8, 8, 16), 4, e.delta_log())
        small = rng.integers(0, 1 << 63, (9, e.P.small), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (9, e.P.small), dtype=np.uint64)
        got = e.blind_rotate_host(small, np.full(9, lid, np.uint32))
        assert np.array_equal(got, octx.blind_rotate(small, e.lut_get(lid)[None, :], np.zeros(9, np.uint32)))
        octx.close()
    finally:
        e.close()


@pytest.mark.parametrize("count", [257, 300, 601])
def test_two_ciphertexts_per_workgroup_form_same_words(eng, ora, count):
    """k_blind_rotate_w2_t64f (csrc/bmi_kernels_t64w2.hip: what auto dispatch runs beyond 256 ciphertexts; variants 1 / 3 pin it, 2 pins
    the one-ciphertext form): the same words as the one-ciphertext kernel on every batch shape - odd batches (the last workgroup runs
    one ciphertext twice), adversarial rows, skipped steps - and as the oracle"""
    to, octx, _, _, sk_big = ora
    tables, ids, tvs, msgs, sel, small = _batch(eng, count, 900 + count)
    small[5] = small[4]
    small[5, ::2] = 0                      # a pair whose ciphertexts skip different steps
    eng.set_kernel_variant(2)
    one = eng.blind_rotate_host(small, ids[sel])
    eng.set_kernel_variant(0)
    two = eng.blind_rotate_host(small, ids[sel])
    assert np.array_equal(one, two)
    for c in (1, 2, 3, 7):                 # pinned on small and odd batches as well
        eng.set_kernel_variant(3)
        got = eng.blind_rotate_host(small[:c], ids[sel][:c])
        eng.set_kernel_variant(0)
        assert np.array_equal(got, one[:c]), c
    pick = np.array([0, 1, 2, 3, 4, 5, count - 1])
    assert np.array_equal(two[pick], octx.blind_rotate(small[pick], tvs, sel[pick]))

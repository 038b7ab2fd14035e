"""GPU parity tests of the 2^64 torus at N = 4096 (preset "secure128_torus_wide": n 742, k 1, l 3 x 10 bits, bootstrap key at 44
bits of precision = two 22-bit limbs; csrc/bmi_kernels_t64q.hip, fft_eighth_f64.hpp).  The specification is the oracle's INTEGER
arithmetic on the same (rounded, exported) key - the generic path of oracle/tfhe_oracle.c - : every output word must be identical
for every batch shape, and the limb sums must sit far from the half-integers when they are rounded."""
import gzip
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SEED = 0x5EED
QB = 65


@pytest.fixture(scope="module")
def eng():
    from bmi_amd import tfhe
    e = tfhe.Engine(tfhe.preset_params("secure128_torus_wide"))
    e.keygen(SEED + 11)
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
    from bmi_amd import tfhe
    to, _, OP, sk_small, _ = ora
    P = eng.P
    assert (P.n, P.N, P.k, P.bs_levels, P.bs_base_log, P.q_bits, P.ks_levels, P.ks_base_log) == (742, 4096, 1, 3, 10, QB, 16, 1)
    assert abs(np.log2(P.lwe_noise) + 17.11) < 0.01 and eng.bsk_precision == 44 == to.default_bsk_precision(OP)
    K = to.keygen(OP, SEED + 11)
    _, _, bsk, ksk = eng.export_keys()
    assert np.array_equal(to.round_key(K.bsk, 44), bsk) and np.array_equal(K.ksk, ksk) and np.array_equal(K.sk_small, sk_small)
    assert not np.array_equal(K.bsk, bsk) and np.all(bsk & np.uint64((1 << 20) - 1) == 0)
    e2 = tfhe.Engine(tfhe.preset_params("secure128_torus_wide"))
    try:
        for bits in (64, 48, 46, 42):
            with pytest.raises(tfhe.BmiError):
                e2.set_bsk_precision(bits)
        e2.set_bsk_precision(44)
        with pytest.raises(tfhe.BmiError):
            e2.set_bsk_unroll(2)
    finally:
        e2.close()
    with pytest.raises(tfhe.BmiError):   # accumulators on the rounded key are multiples of 2^20: no table below that scale
        eng.lut_register(np.arange(-8, 8), 4, 19)
    with pytest.raises(tfhe.BmiError):   # N = 4096 on the torus: (l, Bg) = (3 or 2, 2^10) only
        tfhe.Engine(tfhe.preset_params("secure128_torus_wide", bs_base_log=15))


def _batch(eng, count, seed, bits=5):
    rng = np.random.default_rng(seed)
    h = 1 << (bits - 1)
    tables = [np.arange(-h, h), rng.integers(-h, h, 2 * h)]
    dl = 64 - 1 - bits
    ids = np.array([eng.lut_register(t, bits, dl) for t in tables], np.uint32)
    tvs = np.stack([eng.lut_get(i) for i in ids])
    msgs = rng.integers(-h, h, count)
    sel = rng.integers(0, 2, count).astype(np.uint32)
    small = eng.keyswitch_host(eng.encrypt(msgs, dl))
    small[0] = rng.integers(0, 1 << 63, small.shape[1], dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, small.shape[1], dtype=np.uint64)
    if count > 2:
        small[1] = 0
        small[2] = np.uint64(0xFFFFFFFFFFFFFFFF)
    if count > 4:
        small[3] = rng.integers(0, 1 << 63, small.shape[1], dtype=np.uint64) * np.uint64(2)
        small[3, 7::8] = 0
    return tables, ids, tvs, msgs, sel, small, dl


@pytest.mark.parametrize("count", [1, 5, 257, 520])
def test_blind_rotation_bit_exact_every_batch_shape(eng, ora, count):
    """5-bit messages (the width this set carries at the secure LWE noise) through identity / random tables; adversarial rows"""
    to, octx, _, _, sk_big = ora
    tables, ids, tvs, msgs, sel, small, dl = _batch(eng, count, 700 + count)
    got = eng.blind_rotate_host(small, ids[sel])
    rng = np.random.default_rng(count)
    pick = np.arange(count) if count <= 8 else np.unique(np.concatenate([[0, 1, 2, 3, 4, count - 1, 255, 256], rng.integers(0, count, 1)]) % count)
    assert np.array_equal(got[pick], octx.blind_rotate(small[pick], tvs, sel[pick]))
    ok = np.arange(4, count)
    if ok.size:
        dec = to.decode(to.lwe_phase(sk_big, got[ok]), dl)
        assert list(dec) == [int(tables[s][m + 16]) for s, m in zip(sel[ok], msgs[ok])]


def test_keyswitch_and_whole_pbs_bit_exact_noise_and_margin(eng, ora):
    """keyswitch (16 levels of 1 bit through the matrix-core kernel) and the whole PBS against the oracle; every 5-bit message through
    a random table; the look-up margin the keyswitch noise leaves - its MEAN SQUARE: the digits' mean of -1/2 makes a constant offset
    per key, which the analytic (B^2 + 2) / 12 counts and a variance over ciphertexts of one key would miss -; bootstrap output noise on
    the CGGI formula with the rounded key's effective noise"""
    from test_gpu_parity import cggi_output_variance, effective_params
    to, octx, _, sk_small, sk_big = ora
    P = eng.P
    rng = np.random.default_rng(47)
    bits, dl = 5, 58
    table = rng.integers(-16, 16, 32)
    lid = eng.lut_register(table, bits, dl)
    msgs = np.concatenate([np.arange(-16, 16)] * 32)                    # 1,024 ciphertexts, every message 32 times
    ct = eng.encrypt(msgs, dl)
    small = eng.keyswitch_host(ct)
    assert np.array_equal(small[:12], octx.keyswitch(ct[:12]))
    out = eng.pbs_host(ct, np.full(msgs.size, lid, np.uint32))
    pick = rng.choice(msgs.size, 3, replace=False)
    assert np.array_equal(out[pick], octx.pbs(ct[pick], eng.lut_get(lid)[None, :], np.zeros(3, np.uint32)))
    assert np.array_equal(eng.decrypt(out, dl), table[msgs + 16])
    Q = 1 << 64
    ph = to.lwe_phase(sk_small, small)
    err = np.array([((int(x) - (int(m) << dl)) + Q // 2) % Q - Q // 2 for x, m in zip(ph, msgs)], dtype=np.float64) / Q
    B = 2.0 ** P.ks_base_log
    kN = P.k * P.N
    analytic = kN * P.ks_levels * (B * B + 2) / 12.0 * P.lwe_noise ** 2 + kN / 2.0 / (12.0 * B ** (2 * P.ks_levels))
    ms = float(np.mean(err ** 2))
    ratio = ms / analytic
    sigma_pos = np.sqrt(ms * (2 * P.N) ** 2 + (P.n / 2.0 + 1) / 12.0)
    sigma_analytic = np.sqrt(analytic * (2 * P.N) ** 2 + (P.n / 2.0 + 1) / 12.0)
    margin = (P.N / 64.0) / sigma_pos       # half a 5-bit box (boxes are N / 2^5 positions wide) in sigmas
    print(f"\nsecure128_torus_wide: keyswitch log2 rms {0.5 * np.log2(ms):.2f} (of which offset {np.mean(err):.2e}; analytic {0.5 * np.log2(analytic):.2f}, "
          f"ratio {ratio:.3f}); positions sigma {sigma_pos:.2f} of {2 * P.N}; 5-bit look-up margin {margin:.1f} sigma (analytic "
          f"{(P.N / 64.0) / sigma_analytic:.1f})")
    assert 0.45 < ratio < 2.0 and margin > 4.5 and (P.N / 64.0) / sigma_analytic > 5.3
    want_m = table[msgs + 16]
    oerr = np.array([((int(x) - (int(m) << dl)) + Q // 2) % Q - Q // 2 for x, m in zip(eng.phase(out), want_m)], dtype=np.float64) / Q
    oratio = float(np.var(oerr)) / cggi_output_variance(effective_params(eng), 64, hw_small=int(sk_small.sum()), hw_big=int(sk_big.sum()))
    print(f"secure128_torus_wide: PBS output log2 std {0.5 * np.log2(np.var(oerr)):.2f} (variance / formula {oratio:.3f})")
    assert 0.85 < oratio < 1.15
    import time
    ids256 = np.full(256, lid, np.uint32)
    eng.pbs_host(ct[:256], ids256)
    t0 = time.perf_counter(); eng.pbs_host(ct[:256], ids256); t256 = time.perf_counter() - t0
    t0 = time.perf_counter(); eng.pbs_host(ct[:1], ids256[:1]); t1 = time.perf_counter() - t0
    print(f"secure128_torus_wide: 1 PBS {t1 * 1e3:.2f} ms, 256 PBS {t256 * 1e3:.2f} ms (host-buffer calls, copies included)")


def test_rounding_margin_of_the_limb_sums(eng):
    """bmi_fft_margin_host on this shape: over 512 bootstraps (256 of them uniformly random words, which drive the digits to their
    full range) the limb sums stay within 2^-8 of the integers they are rounded to - against the 1/2 at which a result would
    change (a-priori bound 0.45: tools/fft_bound.py) - and the words equal the product kernel's"""
    rng = np.random.default_rng(12)
    count = 512
    lid = eng.lut_register(rng.integers(-16, 16, 32), 5, 58)
    small = eng.keyswitch_host(eng.encrypt(rng.integers(-16, 16, count), 58))
    small[:256] = rng.integers(0, 1 << 63, (256, small.shape[1]), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (256, small.shape[1]), dtype=np.uint64)
    ids = np.full(count, lid, np.uint32)
    out, dist = eng.fft_margin_host(small, ids)
    print(f"\nsecure128_torus_wide: largest distance from an integer before rounding 2^{np.log2(max(dist, 1e-300)):.1f}")
    assert 0.0 < dist < 2.0 ** -8, dist
    assert np.array_equal(out, eng.blind_rotate_host(small, ids))


def test_l2_shape_and_six_bit_tables_bit_exact():
    """(l, Bg) = (2, 2^10) at N = 4096, the other instantiated shape, under 6-bit tables (the widest look-up the tracer emits)"""
    from bmi_amd import tfhe
    from oracle import tfhe_oracle as to
    e = tfhe.Engine(tfhe.preset_params("secure128_torus_wide", bs_levels=2, n=35))
    try:
        e.keygen(SEED)
        to.set_field(QB)
        OP = to.Params(**{f: getattr(e.P, f) for f, _ in tfhe.Params._fields_})
        _, _, bsk, ksk = e.export_keys()
        octx = to.Ctx(OP, bsk, ksk)
        rng = np.random.default_rng(3)
        lid = e.lut_register(rng.integers(-32, 32, 64), 6, 57)
        small = rng.integers(0, 1 << 63, (9, e.P.small), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (9, e.P.small), dtype=np.uint64)
        got = e.blind_rotate_host(small, np.full(9, lid, np.uint32))
        assert np.array_equal(got, octx.blind_rotate(small, e.lut_get(lid)[None, :], np.zeros(9, np.uint32)))
        octx.close()
    finally:
        e.close()



def test_reference_five_bit_circuits_at_128_bit_security(eng):
    """What this set is for: the reference's UNMODIFIED qfloat_matrix_inverse (tests/golden/ref_traced_inverse.json.gz: 2x2 as
    written and lazily fused; 5-bit look-ups) and the 5-bit circuits of its own FHE test file (ref_own_fhe_tests.json.gz), on
    Concrete's modulus under the 128-bit-secure LWE pair; the error budget of each is what EncryptedMatrixInversion(p_error=...)
    would hold it to (reference: matrix_inversion/main.py:53-66, tests/test_qfloat_fhe.py:136-335)"""
    from bmi_amd.circuit import Circuit
    from bmi_amd.executor import Executor
    from bmi_amd.program import Program
    dl = eng.delta_log(5)
    with gzip.open(os.path.join(G, "ref_traced_inverse.json.gz"), "rt") as f:
        traced = json.load(f)["cases"]
    for case in (traced[0], traced[2]):
        c = Circuit.from_dict(case["circuit"])
        assert c.msg_bits == 5
        rep = Program.from_circuit(c).failure_probability(eng)
        print(f"\n{case['name']}: {rep['lookups']} look-ups, worst margin {rep['worst_margin_sigma']:.2f} sigma, p_fail {rep['p_fail']:.1e}")
        assert rep["p_fail"] < 1e-5 and 5.2 < rep["worst_margin_sigma"] < 5.5
        ex = Executor(c, eng)
        v = case["vectors"][0]
        assert list(eng.decrypt(ex.run(eng.encrypt(v["inputs"], dl)), dl)) == v["expected"], case["name"]
    with gzip.open(os.path.join(G, "ref_own_fhe_tests.json.gz"), "rt") as f:
        own = [c for c in json.load(f)["cases"] if c["msg_bits"] == 5 and c["function"] != "div_qfloats"]
    assert len(own) >= 2
    for case in own:
        ex = Executor(Circuit.from_dict(case["circuit"]), eng)
        for r in case["runs"]:
            assert list(eng.decrypt(ex.run(eng.encrypt(r["inputs"], dl)), dl)) == r["outputs"], case["function"]

"""GPU parity tests of the 2^64-torus kernels whose exact limb products are carried by a floating-point transform
(bmi_kernels_t64f.hip, fft_wave_f64.hpp; kernel variants 5 = wave pairs, 6 = latency form; what `auto` runs on the torus default
set: bootstrap key at 48 bits of precision, base 2^10).  The specification is the oracle's INTEGER arithmetic
(oracle/tfhe_oracle.c ora_blind_rotate_extract on the same, rounded key): every output word must be identical, for every batch
shape, and the limb sums must sit far from the half-integers when they are rounded - that margin, not the order of the
floating-point operations, is what makes the results exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SEED = 0x5EED
QB = 65


def _engine(seed=SEED, **kw):
    from bmi_amd import tfhe
    e = tfhe.Engine(tfhe.default_params(q_bits=QB, **kw))
    e.keygen(seed)
    return e


def _oracle(eng):
    from oracle import tfhe_oracle as to
    to.set_field(QB)
    sk_small, sk_big, bsk, ksk = eng.export_keys()
    P = to.default_params(q_bits=QB, n=eng.P.n, bs_levels=eng.P.bs_levels, bs_base_log=eng.P.bs_base_log)
    return to, to.Ctx(P, bsk, ksk), sk_small, sk_big


@pytest.fixture(scope="module")
def eng():
    e = _engine()
    yield e
    e.close()


def _batch(eng, octx, count, seed):
    rng = np.random.default_rng(seed)
    tables = [np.arange(-8, 8), rng.integers(-8, 8, 16)]
    ids = np.array([eng.lut_register(t, 4, eng.delta_log()) for t in tables], np.uint32)
    tvs = np.stack([eng.lut_get(i) for i in ids])
    msgs = rng.integers(-8, 8, count)
    sel = rng.integers(0, 2, count).astype(np.uint32)
    small = octx.keyswitch(eng.encrypt(msgs, eng.delta_log())) if count <= 64 else eng.keyswitch_host(eng.encrypt(msgs, eng.delta_log()))
    # adversarial rows: uniformly random words (not a valid encryption), all zeros, all ones
    small[0] = rng.integers(0, 1 << 63, small.shape[1], dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, small.shape[1], dtype=np.uint64)
    if count > 2:
        small[1] = 0
        small[2] = np.uint64(0xFFFFFFFFFFFFFFFF)
    if count > 4:
        # random words whose every eighth coefficient switches to 0 (a skipped step): the latency form re-centres its f64
        # accumulator every eight steps TAKEN, whichever steps a ciphertext skips
        small[3] = rng.integers(0, 1 << 63, small.shape[1], dtype=np.uint64) * np.uint64(2)
        small[3, 7::8] = 0
    return tables, ids, tvs, msgs, sel, small


@pytest.mark.parametrize("count", [1, 3, 4, 5, 257, 600, 1100])
@pytest.mark.parametrize("variant", [5, 6], ids=["wave_pairs", "latency"])
def test_float_transform_kernels_bit_exact_every_batch_shape(eng, variant, count):
    """ragged against the 4 ciphertexts per workgroup of the wave-pair kernel; the oracle checks a sample of the larger batches,
    the exact-transform kernel (variant 1, itself held to the oracle elsewhere) all of them"""
    to, octx, _, sk_big = _oracle(eng)
    tables, ids, tvs, msgs, sel, small = _batch(eng, octx, count, 100 + count)
    if variant == 6 and count > 600:
        pytest.skip("the latency form is a small-batch kernel")
    eng.set_kernel_variant(variant)
    try:
        got = eng.blind_rotate_host(small, ids[sel])
        eng.set_kernel_variant(1)
        ref = eng.blind_rotate_host(small, ids[sel])
    finally:
        eng.set_kernel_variant(0)
    assert np.array_equal(got, ref)
    rng = np.random.default_rng(count)
    pick = np.arange(count) if count <= 8 else np.unique(np.concatenate([[0, 1, 2, 3, 4, count - 1, 255, 256], rng.integers(0, count, 4)]) % count)
    assert np.array_equal(got[pick], octx.blind_rotate(small[pick], tvs, sel[pick]))
    ok = np.arange(4, count)
    if ok.size:
        dec = to.decode(to.lwe_phase(sk_big, got[ok]), eng.delta_log())
        assert list(dec) == [int(tables[s][m + 8]) for s, m in zip(sel[ok], msgs[ok])]
    octx.close()


def test_auto_dispatch_takes_the_float_transform_kernels(eng):
    """variant 0 on the torus default set = variant 6 up to 512 ciphertexts, variant 5 beyond: same words either way (all
    kernels are exact), so the check is on agreement at both sides of the threshold and on the refusals"""
    from bmi_amd import tfhe
    to, octx, _, _ = _oracle(eng)
    for count in (512, 513):
        _, ids, tvs, _, sel, small = _batch(eng, octx, count, 7 + count)
        got = eng.blind_rotate_host(small, ids[sel])
        pick = np.array([0, 1, 2, 3, count - 1])
        assert np.array_equal(got[pick], octx.blind_rotate(small[pick], tvs, sel[pick]))
    octx.close()
    with pytest.raises(tfhe.BmiError):   # accumulators on the rounded key are multiples of 2^16: no table below that scale
        eng.lut_register(np.arange(-8, 8), 4, 15)
    e49 = tfhe.Engine(tfhe.default_params(q_bits=49))
    try:
        for v in (5, 6):
            with pytest.raises(tfhe.BmiError):
                e49.set_kernel_variant(v)
        with pytest.raises(tfhe.BmiError):
            e49.fft_margin_host(np.zeros((1, 631), np.uint64), np.zeros(1, np.uint32))
    finally:
        e49.close()


def test_half_transforms_on_the_gpu_against_the_definition(tmp_path):
    """tests/hip/fft_half_check.hip: the two 256-point half transforms of the latency kernel (register swaps and DPP moves instead
    of LDS exchanges) against the folded transform's definition evaluated in long double, and the inverse halves on a round trip"""
    import os
    import subprocess
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "fft_half_check")
    subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-o", exe,
                    os.path.join(repo, "tests", "hip", "fft_half_check.hip")], check=True, capture_output=True, timeout=600)
    p = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert p.returncode == 0 and p.stdout.startswith("ok"), p.stdout + p.stderr
    fwd, back = (float(x) for x in p.stdout.split()[1:3])
    assert fwd < 1e-9 and back < 1e-10


def test_exact_key_has_no_float_transform_copy():
    """the error bound of the transform is stated for 24-bit limbs against base-2^10 digits: a context with the exact 64-bit key
    (three 22-bit limbs against base 2^15) keeps the exact transform, and pinning variant 5 / 6 there is an error, not a fallback"""
    from bmi_amd import tfhe
    e = tfhe.Engine(tfhe.default_params(q_bits=QB, bs_base_log=15))
    try:
        e.keygen(SEED)
        assert e.bsk_precision == 64
        small = np.zeros((2, e.P.small), np.uint64)
        lid = e.lut_register(np.arange(-8, 8), 4, e.delta_log())
        for v in (5, 6):
            e.set_kernel_variant(v)
            with pytest.raises(tfhe.BmiError):
                e.blind_rotate_host(small, np.full(2, lid, np.uint32))
        e.set_kernel_variant(0)
        e.blind_rotate_host(small, np.full(2, lid, np.uint32))
    finally:
        e.close()


@pytest.mark.parametrize("kw", [dict(), dict(bs_levels=2), dict(n=1024), dict(n=1)], ids=["default", "l2", "n1024", "n1"])
def test_rounding_margin_of_the_limb_sums(kw):
    """bmi_fft_margin_host: over 2,048 bootstraps (2 x 10^9 rounded values at the default set) the limb sums stay within 2^-9 of
    the integers they are rounded to - against the 1/2 at which a result would change - and the words equal the exact-transform
    kernel's.  The first ciphertexts are uniformly random words, which drive the digits to their full range."""
    e = _engine(**kw)
    try:
        rng = np.random.default_rng(11)
        count = 2048
        lid = e.lut_register(rng.integers(-8, 8, 16), 4, e.delta_log())
        small = e.keyswitch_host(e.encrypt(rng.integers(-8, 8, count), e.delta_log()))
        small[:512] = rng.integers(0, 1 << 63, (512, small.shape[1]), dtype=np.uint64) * np.uint64(2) + rng.integers(0, 2, (512, small.shape[1]), dtype=np.uint64)
        ids = np.full(count, lid, np.uint32)
        out, dist = e.fft_margin_host(small, ids)
        print(f"\n{kw or 'default'}: largest distance from an integer before rounding 2^{np.log2(max(dist, 1e-300)):.1f}")
        assert 0.0 < dist < 2.0 ** -9, dist
        e.set_kernel_variant(1)
        assert np.array_equal(out, e.blind_rotate_host(small, ids))
        e.set_kernel_variant(6)   # the latency form at this shape (random words first): same words, and its own rounding distance
        assert np.array_equal(out[:40], e.blind_rotate_host(small[:40], ids[:40]))
        assert np.array_equal(out[600:640], e.blind_rotate_host(small[600:640], ids[600:640]))
        out6, dist6 = e.fft_margin_host(small[:256], ids[:256])
        print(f"{kw or 'default'}: latency form (half transforms), largest distance 2^{np.log2(max(dist6, 1e-300)):.1f}")
        assert 0.0 < dist6 < 2.0 ** -9 and np.array_equal(out6, out[:256])
    finally:
        e.close()

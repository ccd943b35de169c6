"""The C-ABI library builds, loads and exports every symbol include/mzk.h declares; host-side
parameter math works; compute calls fail loudly without a GPU (no CPU fallback).  CPU only."""
import ctypes
import numpy as np
import pytest
import orc


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd.build as b
    b.build()
    import myzkp_amd
    return myzkp_amd


def test_exports_every_declared_symbol(mz):
    assert len(mz.DECLARED_SYMBOLS) >= 20
    missing = [s for s in mz.DECLARED_SYMBOLS if s not in mz.exported_symbols()]
    assert missing == []
    assert mz.lib().mzk_abi_version() == 2


def test_root_of_unity_is_host_math(mz):
    # get_nth_root_of_m128 (fri.rs:423-447) vs the oracle, and the Fr table vs 5^((r-1)/2^28)
    for lg in (0, 1, 3, 8, 20, 24, 119):
        assert mz.root_of_unity(mz.FIELD_M128, lg) == orc.m128_root(lg)
    assert mz.root_of_unity(mz.FIELD_M128, 3) == 131076302407280330469229082343774091404
    for lg in (0, 1, 10, 20, 24, 28):
        assert mz.root_of_unity(mz.FIELD_FR, lg) == orc.fr_root(lg)
    with pytest.raises(mz.MzkError):
        mz.root_of_unity(mz.FIELD_M128, 120)
    with pytest.raises(mz.MzkError):
        mz.root_of_unity(mz.FIELD_FR, 29)


def test_no_cpu_fallback(mz):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    v = np.zeros((4, 4), dtype=np.uint64)
    with pytest.raises(mz.MzkError) as e:
        mz.ntt(mz.FIELD_FR, 1, v)
    assert e.value.code == -8
    with pytest.raises(mz.MzkError) as e:
        mz.msm_g1(np.zeros((1, 4), dtype=np.uint64), np.zeros((1, 8), dtype=np.uint64))
    assert e.value.code == -8


def test_host_parameter_arithmetic_without_a_gpu(mz):
    """the library's host-side field arithmetic (Montgomery on 64-bit limbs) against Python big ints and against its own
    bit-serial reference, for all three fields"""
    import ctypes, random
    import numpy as np
    import orc
    L = mz.lib()
    rnd = random.Random(44)

    def op(fid, o, a, b=0):
        nl = 2 if fid == mz.FIELD_M128 else 4
        aa, bb = orc.to_limbs([a], nl), orc.to_limbs([b], nl)
        out = np.zeros(4, dtype=np.uint64)
        assert L.mzk_host_field_op(fid, o, orc.ptr(aa), orc.ptr(bb), orc.ptr(out)) == 0
        return orc.from_limbs(out[:nl].reshape(1, nl))[0]

    for fid in (mz.FIELD_FR, mz.FIELD_M128, mz.FIELD_FQ):
        p = mz.MODULUS[fid]
        vals = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, 1 << 64, (1 << 64) - 1] + [rnd.randrange(p) for _ in range(300)]
        for i, a in enumerate(vals):
            b = vals[(7 * i + 3) % len(vals)]
            assert op(fid, 0, a, b) == a * b % p
            if i < 30:
                assert op(fid, 3, a, b) == a * b % p
                assert op(fid, 1, a) == (pow(a, -1, p) if a else 0)
                e = rnd.getrandbits(64)
                assert op(fid, 2, a, e) == pow(a, e, p)

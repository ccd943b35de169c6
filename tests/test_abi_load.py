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


def test_shipped_library_reads_no_environment_variable():
    """a caller's environment must not be able to change the code path.  The shipped .so does not import
    getenv at all, and the sources mention it only inside the MZK_TUNING helper of mzk_common.h."""
    import glob, os, subprocess
    import myzkp_amd.build as b
    so = b.build()
    und = subprocess.run(["nm", "-D", "--undefined-only", so], capture_output=True, text=True, check=True).stdout
    assert "getenv" not in und
    hits = []
    for f in glob.glob(os.path.join(os.path.dirname(so), "csrc", "*")):
        for i, line in enumerate(open(f), 1):
            if "getenv" in line and not line.lstrip().startswith("//"):
                hits.append((os.path.basename(f), i))
    assert [h[0] for h in hits] == ["mzk_common.h"], hits


def test_no_hot_kernel_spills_registers():
    """the 1024-lane BN254 NTT pass shipped with vgpr_spill_count 32 and nobody had noticed.  Read the kernel
    metadata out of the shipped .so (tools/kernel_resources.py): no kernel may spill vector registers unless it is listed here
    with its reason; scratch (private segment) is allowed only where a kernel indexes a local array dynamically or calls a
    non-inlined function."""
    import os, sys
    import myzkp_amd.build as b
    so = b.build()
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import kernel_resources
    res = kernel_resources.kernel_resources(so)
    assert len(res) > 100, "kernel metadata not found in the library"
    allowed = {
        "k_fb_table": "one-time fixed-base table of setup_kzg (built once per generator, 32 x 256 entries): launch-bound 128 VGPRs, off the timed paths",
        "k_selftest_field_asm": "device self-test, not a product path",
    }
    spilled = {k: v["vgpr_spill"] for k, v in res.items() if v["vgpr_spill"] > 0}
    bad = {k: n for k, n in spilled.items() if not any(a in k for a in allowed)}
    assert not bad, "kernels spilling VGPRs: %s" % bad
    # the kernels the bench prices, by name: present and spill-free
    for must in kernel_resources.PRICED:
        hits = [k for k in res if must in k]
        assert hits, must
        assert all(res[k]["vgpr_spill"] == 0 for k in hits), must
    # scalar-register spills and scratch in the priced kernels too.  A spilled SGPR lives in a lane of a VGPR
    # (v_writelane / v_readlane: an instruction each way, no memory), so a bounded number is tolerated where it has a reason;
    # scratch (private memory) is tolerated in no priced kernel but the one listed.
    sgpr_ok = {      # substring of the demangled name: (most spilled SGPRs seen + slack, why)
        "k_ntt_strided<mzk::FrParams": (24, "three wave-uniform Shoup twiddles are 2 x 9 scalar registers each, next to 102 usable SGPRs"),
        "k_ntt_last<mzk::FrParams": (24, "same"),
        "k_many_sort1": (64, "per-window histogram bases kept in scalar registers across the walk"),
        "k_open_many": (16, "chunk bookkeeping of the suffix scan"),
        "k_seg_combine_heavy": (8, "skewed inputs only (returns at once when no bucket was deferred, i.e. on every uniform input); two forms in one kernel -- a workgroup per bucket / several per bucket -- rather than a launch more on the path of every MSM"),
        "k_reduce_tail_row": (128, "single-workgroup latency tail: non-inlined one-shot operations, constants of four Horner chains"),
        "k_direct_finish": (96, "one workgroup per polynomial, latency tail: the non-inlined row operations of the tail"),
        "k_window_combine_row": (64, "one wave, latency tail (112 doublings): same non-inlined row operations"),
    }
    row_tail = "non-inlined one-shot group operations pass records through the stack by design (DESIGN 5.3), off every throughput path"
    scratch_ok = {"k_reduce_tail_row": row_tail, "k_direct_finish": row_tail, "k_window_combine_row": row_tail}
    problems = []
    for must in kernel_resources.PRICED:
        for k in (k for k in res if must in k):
            v = res[k]
            cap = max([c for a, (c, _) in sgpr_ok.items() if a in k] or [0])
            if v["sgpr_spill"] > cap:
                problems.append("%s spills %d SGPRs (allowed %d)" % (k, v["sgpr_spill"], cap))
            if v["scratch"] and not any(a in k for a in scratch_ok):
                problems.append("%s uses %d bytes of scratch" % (k, v["scratch"]))
    assert not problems, problems

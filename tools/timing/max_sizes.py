"""Sizes beyond BASELINE's 2^24 up to the limits the ABI states (MSM 2^27 pairs, Fr / M128 transforms of 2^28 points), on one
MI355X, checked by closed forms that need no CPU transform of that size:

  NTT    coset LDE of the all-ones polynomial: out[k] = sum_j a^j w^(jk) = (a^n - 1) / (a w^k - 1)   (ntt.rs:254-269), 64
         sampled k against Python integers; the plain transform of a seeded random vector against the LDE with offset 1
         (the same sums through the other first pass), two outputs against the oracle's Horner evaluation of all n
         coefficients (ntt.rs:40-46), and intt(ntt(x)) == x
  MSM    sum_i s_i [alpha^i]G = [f(alpha)]G  (polynomial.rs:156-165, kzg.rs:57-59): generic layout and SRS window tables

usage: max_sizes.py [ntt LG ...] [ntt_m128 LG ...] [msm LG ...]      e.g.  max_sizes.py ntt 26 28 ntt_m128 30 msm 26 27
The oracle (tests/orc.py) is the checker here, as in tests/."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np, torch
import myzkp_amd as mz
import orc
from orc import FR, M128, P_FR

mz.init(0)
L = mz.lib()
dev = torch.device("cuda", 0)
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
G = (1, 2)


def ok(rc):
    assert rc == 0, L.mzk_last_error().decode()


def dp(t, off=0):
    return ctypes.c_void_p(t.data_ptr() + off)


def vp(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def timed(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def mem():
    free, total = torch.cuda.mem_get_info()
    return "HBM free %.1f of %.1f GiB" % (free / 2**30, total / 2**30)


def elem(t, k, nl):
    return orc.from_limbs(t[k * nl:(k + 1) * nl].cpu().numpy().view(np.uint64).reshape(1, nl))[0]


def ntt_case(fid, lg):
    nl = 4 if fid == FR else 2
    p = orc.MOD[fid]
    n = 1 << lg
    name = "Fr" if fid == FR else "M128"
    w = orc.root_of(fid, lg)
    root, one_l = mz.to_limbs([w], nl), mz.to_limbs([1], nl)
    a = 0x1234567 if fid == FR else 3
    off = mz.to_limbs([a], nl)
    x = torch.empty(n * nl, dtype=torch.int64, device=dev)
    y = torch.empty(n * nl, dtype=torch.int64, device=dev)
    # all-ones coefficients
    x.view(-1, nl).zero_()
    x.view(-1, nl)[:, 0] = 1
    ok(L.mzk_coset_lde_dev(fid, dp(x), ctypes.c_size_t(n), vp(off), vp(root), dp(y), ctypes.c_size_t(n), st))   # plan + tables
    t_lde = timed(lambda: ok(L.mzk_coset_lde_dev(fid, dp(x), ctypes.c_size_t(n), vp(off), vp(root), dp(y), ctypes.c_size_t(n), st)))
    num = (pow(a, n, p) - 1) % p
    rng = np.random.default_rng(lg * 10 + fid)
    ks = [0, 1, n - 1, n // 2, n // 2 + 1] + [int(k) for k in rng.integers(0, n, 59)]
    for k in ks:
        want = num * pow((a * pow(w, k, p) - 1) % p, -1, p) % p
        assert elem(y, k, nl) == want, "LDE closed form differs at k=%d" % k
    # plain transform of a random vector == LDE with offset 1; round trip
    ok(L.mzk_synth_field_dev(fid, ctypes.c_uint64(9000 + lg), ctypes.c_size_t(n), dp(x), st))
    ok(L.mzk_ntt_dev(fid, vp(root), dp(x), dp(y), ctypes.c_size_t(n), 0, st))
    t_ntt = timed(lambda: ok(L.mzk_ntt_dev(fid, vp(root), dp(x), dp(y), ctypes.c_size_t(n), 0, st)))
    z = torch.empty(n * nl, dtype=torch.int64, device=dev)
    ok(L.mzk_coset_lde_dev(fid, dp(x), ctypes.c_size_t(n), vp(one_l), vp(root), dp(z), ctypes.c_size_t(n), st))
    torch.cuda.synchronize()
    assert torch.equal(y, z), "plain transform differs from the LDE with offset 1"
    del z
    direct = ""
    if lg <= 26:
        v = orc.synth_vector(fid, 9000 + lg, n)
        for k in (1, n - 7):
            assert elem(y, k, nl) == orc.poly_eval(fid, v, pow(w, k, p)), "output %d differs from direct evaluation" % k
        del v
        direct = ", 2 outputs == Horner over all coefficients"
    ok(L.mzk_ntt_dev(fid, vp(root), dp(y), dp(y), ctypes.c_size_t(n), 1, st))
    torch.cuda.synchronize()
    assert torch.equal(x, y), "intt(ntt(x)) != x"
    print("NTT %-4s 2^%d: forward %.3f ms (%.2f G elems/s), LDE %.3f ms; 64 closed-form outputs ok, == LDE(offset 1), round trip ok%s; %s"
          % (name, lg, t_ntt, n / t_ntt / 1e6, t_lde, direct, mem()), flush=True)
    del x, y
    torch.cuda.empty_cache()


def msm_case(lg):
    n = 1 << lg
    alpha = orc.from_limbs(orc.synth_vector(FR, 7700 + lg, 1))[0]
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ok(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(7800 + lg), ctypes.c_size_t(n), dp(sc), st))
    pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
    a_l, g_l = mz.to_limbs([alpha], 4), mz.points_to_array([G])
    t_setup = timed(lambda: ok(L.mzk_kzg_setup_g1_dev(vp(a_l), vp(g_l), ctypes.c_size_t(n - 1), dp(pts), st)))
    out = torch.zeros(16, dtype=torch.int64, device=dev)
    ok(L.mzk_msm_g1_bn254_dev(dp(sc), dp(pts), ctypes.c_size_t(n), dp(out), st))
    t_gen = timed(lambda: ok(L.mzk_msm_g1_bn254_dev(dp(sc), dp(pts), ctypes.c_size_t(n), dp(out), st)))
    h = ctypes.c_void_p()
    t_tab = timed(lambda: ok(L.mzk_srs_from_device(dp(pts), ctypes.c_size_t(n), ctypes.byref(h), st)))
    del pts
    torch.cuda.empty_cache()
    ok(L.mzk_kzg_commit_srs_dev(h, dp(sc), ctypes.c_size_t(n), dp(out, 64), ctypes.c_int(0), st))
    t_srs = timed(lambda: ok(L.mzk_kzg_commit_srs_dev(h, dp(sc), ctypes.c_size_t(n), dp(out, 64), ctypes.c_int(0), st)))
    m_now = mem()
    t0 = time.perf_counter()
    s_cpu = orc.synth_vector(FR, 7800 + lg, n)
    assert np.array_equal(s_cpu.view(np.int64).reshape(-1)[-4096:], sc[-4096:].cpu().numpy())
    want = orc.ec_mul(0, G, orc.poly_eval(FR, s_cpu, alpha))
    t_cpu = time.perf_counter() - t0
    del s_cpu
    got_gen = mz.array_to_points(out[:8].cpu().numpy().view(np.uint64))[0]
    got_srs = mz.array_to_points(out[8:].cpu().numpy().view(np.uint64))[0]
    assert got_gen == want, "generic MSM != [f(alpha)]G"
    assert got_srs == want, "SRS-table commit != [f(alpha)]G"
    L.mzk_srs_free(h)
    del sc
    torch.cuda.empty_cache()
    print("MSM 2^%d: setup %.1f ms, generic %.2f ms (%.3g pairs/s), window tables %.1f ms, commit against tables %.2f ms (%.3g pairs/s); "
          "both == [f(alpha)]G (host Horner %.1f s); with tables resident: %s" % (lg, t_setup, t_gen, n / t_gen * 1e3, t_tab, t_srs, n / t_srs * 1e3, t_cpu, m_now), flush=True)


if __name__ == "__main__":
    print(mem(), flush=True)
    mode = None
    for a in sys.argv[1:]:
        if a in ("ntt", "ntt_m128", "msm"):
            mode = a
        elif mode == "ntt":
            ntt_case(FR, int(a))
            ntt_case(M128, int(a))
        elif mode == "ntt_m128":
            ntt_case(M128, int(a))
        elif mode == "msm":
            msm_case(int(a))

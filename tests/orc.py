"""ctypes binding of the CPU oracle (oracle/libmzk_oracle.so).  Test infrastructure only."""
import ctypes, json, os, subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FR, M128, FQ, F631, F17, F31 = 0, 1, 2, 3, 4, 5
LIMBS = {FR: 4, M128: 2, FQ: 4, F631: 1, F17: 1, F31: 1}
P_FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
P_FQ = 21888242871839275222246405745257275088696311157297823662689037894645226208583
P_M128 = 270497897142230380135924736767050121217
MOD = {FR: P_FR, M128: P_M128, FQ: P_FQ, F631: 631, F17: 17, F31: 31}
M128_GEN = 85408008396924667383611388730472331217
FR_OMEGA28 = 19103219067921713944291392827692070036145651957329286315305642004821462161904

_lib = None


def usable_threads(cap=32):
    """Threads worth giving the oracle on this box: the cgroup's CPU quota when there is one (the GPU boxes report 256 CPUs
    but schedule about 16: more threads made every oracle routine SLOWER there, tools/timing/oracle_scaling.py), else the
    affinity mask, capped."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except Exception:
        pass
    return max(1, min(n, cap))


def lib():
    global _lib
    if _lib is None:
        so = os.path.join(ROOT, "oracle", "libmzk_oracle.so")
        src = os.path.join(ROOT, "oracle", "mzk_oracle.c")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL)
        _lib = ctypes.CDLL(so)
        _lib.orc_synth_element.restype = None
        _lib.orc_synth_vector.restype = None
        _lib.orc_synth_g1_points.restype = None
    return _lib


def to_limbs(vals, nl):
    """list of python ints -> (len, nl) uint64 array, little-endian limbs"""
    a = np.zeros((len(vals), nl), dtype=np.uint64)
    for i, v in enumerate(vals):
        v = int(v)
        for j in range(nl):
            a[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return a


def from_limbs(a):
    a = np.asarray(a, dtype=np.uint64)
    a = a.reshape(-1, a.shape[-1])
    return [sum(int(a[i, j]) << (64 * j) for j in range(a.shape[1])) for i in range(a.shape[0])]


def ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def one(fid, v):
    return to_limbs([v], LIMBS[fid])


def field_op(name, fid, *args):
    out = np.zeros((1, LIMBS[fid]), dtype=np.uint64)
    arrs = [one(fid, a) for a in args]
    rc = getattr(lib(), "orc_field_" + name)(fid, *[ptr(a) for a in arrs], ptr(out))
    assert rc == 0
    return from_limbs(out)[0]


def field_pow(fid, a, e):
    out = np.zeros((1, LIMBS[fid]), dtype=np.uint64)
    ne = max(1, (int(e).bit_length() + 63) // 64)
    aa, ee = one(fid, a), to_limbs([e], ne)
    assert lib().orc_field_pow(fid, ptr(aa), ptr(ee), ne, ptr(out)) == 0
    return from_limbs(out)[0]


def m128_root(log2n):
    out = np.zeros((1, 2), dtype=np.uint64)
    rc = lib().orc_m128_nth_root(log2n, ptr(out))
    assert rc == 0, rc
    return from_limbs(out)[0]


def fr_root(log2n):
    return pow(FR_OMEGA28, 1 << (28 - log2n), P_FR)


def root_of(fid, log2n):
    return m128_root(log2n) if fid == M128 else fr_root(log2n)


def _vec_call(fn, fid, root, vals_arr, n, *extra):
    out = np.zeros((max(n, 1), LIMBS[fid]), dtype=np.uint64)
    r = one(fid, root)
    rc = fn(fid, ptr(r), ptr(vals_arr), ptr(out), ctypes.c_size_t(n), *extra)
    return rc, out[:n]


def ntt_ref(fid, root, arr):
    return _vec_call(lib().orc_ntt_ref, fid, root, arr, arr.shape[0])


def intt_ref(fid, root, arr):
    return _vec_call(lib().orc_intt_ref, fid, root, arr, arr.shape[0])


def ntt_fast(fid, root, arr, inverse=False, threads=8):
    return _vec_call(lib().orc_ntt_fast, fid, root, arr, arr.shape[0], int(inverse), threads)


def coset_ref(fid, coef_arr, offset, generator, order):
    out = np.zeros((max(order, 1), LIMBS[fid]), dtype=np.uint64)
    o, g = one(fid, offset), one(fid, generator)
    rc = lib().orc_fast_coset_evaluate_ref(fid, ptr(coef_arr), ctypes.c_size_t(coef_arr.shape[0]), ptr(o), ptr(g), ptr(out), ctypes.c_size_t(order))
    return rc, out[:order]


def fft_multiply_ref(fid, a, b, omega):
    out = np.zeros((max(a.shape[0] + b.shape[0], 1), LIMBS[fid]), dtype=np.uint64)
    ol = ctypes.c_size_t(0)
    w = one(fid, omega)
    rc = lib().orc_fft_multiply_ref(fid, ptr(a), ctypes.c_size_t(a.shape[0]), ptr(b), ctypes.c_size_t(b.shape[0]), ptr(w), ptr(out), ctypes.byref(ol))
    return rc, out[:ol.value]


def fast_multiply_ref(fid, a, b, root, root_order):
    out = np.zeros((max(root_order, a.shape[0] + b.shape[0], 1), LIMBS[fid]), dtype=np.uint64)
    ol = ctypes.c_size_t(0)
    w = one(fid, root)
    rc = lib().orc_fast_multiply_ref(fid, ptr(a), ctypes.c_size_t(a.shape[0]), ptr(b), ctypes.c_size_t(b.shape[0]), ptr(w), ctypes.c_size_t(root_order), ptr(out), ctypes.byref(ol))
    return rc, out[:ol.value]


def poly_eval(fid, coef_arr, x):
    out = np.zeros((1, LIMBS[fid]), dtype=np.uint64)
    xx = one(fid, x)
    assert lib().orc_poly_eval(fid, ptr(coef_arr), ctypes.c_size_t(coef_arr.shape[0]), ptr(xx), ptr(out)) == 0
    return from_limbs(out)[0]


# ---- curve helpers: points are (n, 8) uint64 arrays x||y, all-zero = infinity -----------------
def pts_to_arr(pts, nl=4):
    a = np.zeros((len(pts), 2 * nl), dtype=np.uint64)
    for i, p in enumerate(pts):
        a[i, :nl] = to_limbs([p[0]], nl)[0]
        a[i, nl:] = to_limbs([p[1]], nl)[0]
    return a


def arr_to_pts(a, nl=4):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 2 * nl)
    return [(from_limbs(a[i:i + 1, :nl])[0], from_limbs(a[i:i + 1, nl:])[0]) for i in range(a.shape[0])]


def ec_mul(cid, P, k, nl=4):
    out = np.zeros((1, 2 * nl), dtype=np.uint64)
    nk = max(1, (int(k).bit_length() + 63) // 64)
    pa, ka = pts_to_arr([P], nl), to_limbs([k], nk)
    assert lib().orc_ec_mul(cid, ptr(pa), ptr(ka), nk, ptr(out)) == 0
    return arr_to_pts(out, nl)[0]


def ec_add(cid, P, Q, nl=4):
    out = np.zeros((1, 2 * nl), dtype=np.uint64)
    pa, qa = pts_to_arr([P], nl), pts_to_arr([Q], nl)
    assert lib().orc_ec_add(cid, ptr(pa), ptr(qa), ptr(out)) == 0
    return arr_to_pts(out, nl)[0]


def msm_ref(scal_arr, pts_arr):
    out = np.zeros((1, 8), dtype=np.uint64)
    assert lib().orc_msm_ref(ptr(scal_arr), ptr(pts_arr), ctypes.c_size_t(scal_arr.shape[0]), ptr(out)) == 0
    return arr_to_pts(out)[0]


def msm_fast(scal_arr, pts_arr, threads=8):
    out = np.zeros((1, 8), dtype=np.uint64)
    assert lib().orc_msm_fast(ptr(scal_arr), ptr(pts_arr), ctypes.c_size_t(scal_arr.shape[0]), ptr(out), threads) == 0
    return arr_to_pts(out)[0]


def kzg_setup_ref(alpha, max_d):
    out = np.zeros((max_d + 1, 8), dtype=np.uint64)
    g, a = pts_to_arr([(1, 2)]), one(FR, alpha)
    assert lib().orc_kzg_setup_g1_ref(ptr(g), ptr(a), ctypes.c_size_t(max_d), ptr(out)) == 0
    return out


def kzg_open_ref(coef_arr, u, srs_arr):
    y = np.zeros((1, 4), dtype=np.uint64)
    w = np.zeros((1, 8), dtype=np.uint64)
    uu = one(FR, u)
    assert lib().orc_kzg_open_ref(ptr(coef_arr), ctypes.c_size_t(coef_arr.shape[0]), ptr(uu), ptr(srs_arr), ptr(y), ptr(w)) == 0
    return from_limbs(y)[0], arr_to_pts(w)[0]


def kzg_batch_open_ref(coef_arr, us, srs_arr):
    k = len(us)
    ys = np.zeros((max(k, 1), 4), dtype=np.uint64)
    w = np.zeros((1, 8), dtype=np.uint64)
    ua = to_limbs(list(us), 4)
    assert lib().orc_kzg_batch_open_ref(ptr(coef_arr), ctypes.c_size_t(coef_arr.shape[0]), ptr(ua), ctypes.c_size_t(k), ptr(srs_arr), ptr(ys), ptr(w)) == 0
    return from_limbs(ys[:k]), arr_to_pts(w)[0]


def kzg_degree_bound_ref(coef_arr, srs_arr, d):
    out = np.zeros((1, 8), dtype=np.uint64)
    rc = lib().orc_kzg_prove_degree_bound_ref(ptr(coef_arr), ctypes.c_size_t(coef_arr.shape[0]), ptr(srs_arr), ctypes.c_size_t(srs_arr.shape[0]), ctypes.c_size_t(d), ptr(out))
    return rc, arr_to_pts(out)[0]


def fri_fold_ref(fid, cw_arr, alpha, offset, omega):
    out = np.zeros((max(cw_arr.shape[0] // 2, 1), LIMBS[fid]), dtype=np.uint64)
    a, o, w = one(fid, alpha), one(fid, offset), one(fid, omega)
    assert lib().orc_fri_fold_ref(fid, ptr(cw_arr), ctypes.c_size_t(cw_arr.shape[0]), ptr(a), ptr(o), ptr(w), ptr(out)) == 0
    return out[:cw_arr.shape[0] // 2]


def fixed_base_batch(base, scal_arr, threads=8):
    out = np.zeros((scal_arr.shape[0], 8), dtype=np.uint64)
    b = pts_to_arr([base])
    assert lib().orc_g1_fixed_base_mul_batch(ptr(b), ptr(scal_arr), ctypes.c_size_t(scal_arr.shape[0]), ptr(out), threads) == 0
    return out


def synth_vector(fid, seed, n, threads=8):
    out = np.zeros((n, LIMBS[fid]), dtype=np.uint64)
    lib().orc_synth_vector(fid, ctypes.c_uint64(seed), ctypes.c_size_t(n), ptr(out), threads)
    return out


def synth_points(seed, n, threads=8):
    out = np.zeros((n, 8), dtype=np.uint64)
    lib().orc_synth_g1_points(ctypes.c_uint64(seed), ctypes.c_size_t(n), ptr(out), threads)
    return out


def golden(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def I(x):
    if isinstance(x, list):
        return [I(v) for v in x]
    return int(x)


# ---- Merkle / SHA3 / leaf bytes (oracle/mzk_oracle_merkle.c) ---------------------------------------------
def sha3_256(data):
    out = (ctypes.c_uint8 * 32)()
    lib().orc_sha3_256(bytes(data), ctypes.c_size_t(len(data)), out)
    return bytes(out)


def bincode_field(v, nl):
    a = to_limbs([v], nl)
    out = (ctypes.c_uint8 * (9 + 8 * nl))()
    lib().orc_bincode_field.restype = ctypes.c_size_t
    ln = lib().orc_bincode_field(ptr(a), nl, out)
    return bytes(out[:ln])


def bincode_field_signed(mag, nl, negative):
    a = to_limbs([mag], nl)
    out = (ctypes.c_uint8 * (9 + 8 * nl))()
    lib().orc_bincode_field_signed.restype = ctypes.c_size_t
    ln = lib().orc_bincode_field_signed(ptr(a), nl, int(bool(negative)), out)
    return bytes(out[:ln])


def _leaf_blob(leaves):
    blob = b"".join(leaves)
    off = np.zeros(len(leaves) + 1, dtype=np.uint64)
    if leaves:
        off[1:] = np.cumsum([len(x) for x in leaves], dtype=np.uint64)
    return (ctypes.c_uint8 * max(len(blob), 1)).from_buffer_copy(blob or b"\0"), off


def merkle_commit_ref(leaves):
    buf, off = _leaf_blob(leaves)
    cap = max([32] + [len(x) for x in leaves])
    root = (ctypes.c_uint8 * cap)()
    ln = ctypes.c_size_t()
    assert lib().orc_merkle_commit_ref(buf, ptr(off), ctypes.c_size_t(len(leaves)), root, ctypes.byref(ln)) == 0
    return bytes(root[:ln.value])


def merkle_open_ref(index, leaves):
    buf, off = _leaf_blob(leaves)
    stride = max([32] + [len(x) for x in leaves])
    cap = max(len(leaves).bit_length(), 1)
    path = (ctypes.c_uint8 * (stride * cap))()
    lens = (ctypes.c_uint64 * cap)()
    depth = ctypes.c_size_t()
    rc = lib().orc_merkle_open_ref(buf, ptr(off), ctypes.c_size_t(len(leaves)), ctypes.c_size_t(index), path, lens,
                                   ctypes.c_size_t(stride), ctypes.byref(depth))
    if rc == -2:        # Merkle::open does not terminate for this index (one-leaf slice, merkle.rs:36-45)
        return None
    assert rc == 0
    raw = bytes(path)
    return [raw[k * stride:k * stride + lens[k]] for k in range(depth.value)]


def merkle_verify_ref(root, index, path, leaf):
    stride = max([32] + [len(x) for x in path])
    buf = (ctypes.c_uint8 * (stride * len(path)))()
    lens = (ctypes.c_uint64 * len(path))()
    for k, e in enumerate(path):
        buf[k * stride:k * stride + len(e)] = e
        lens[k] = len(e)
    return bool(lib().orc_merkle_verify_ref(bytes(root), ctypes.c_size_t(len(root)), ctypes.c_size_t(index), buf, lens,
                                            ctypes.c_size_t(stride), ctypes.c_size_t(len(path)), bytes(leaf), ctypes.c_size_t(len(leaf))))


def field_leaves(fid, arr):
    nl = arr.shape[1]
    return [bincode_field(v, nl) for v in from_limbs(arr)]


def field_leaves_fast(fid, arr):
    """same as field_leaves, serialised in C (for 2^20-element codewords)"""
    nl = arr.shape[1]
    n = arr.shape[0]
    blob = (ctypes.c_uint8 * (n * (9 + 8 * nl)))()
    off = np.zeros(n + 1, dtype=np.uint64)
    lib().orc_bincode_field_vector(ptr(arr), nl, ctypes.c_size_t(n), blob, ptr(off))
    raw = bytes(blob)
    return [raw[int(off[i]):int(off[i + 1])] for i in range(n)]


def merkle_commit_field_ref(fid, arr):
    """Merkle::commit(&codeword.map(bincode::serialize)) entirely in the C oracle (no per-leaf Python objects)."""
    nl, n = arr.shape[1], arr.shape[0]
    blob = (ctypes.c_uint8 * (n * (9 + 8 * nl)))()
    off = np.zeros(n + 1, dtype=np.uint64)
    lib().orc_bincode_field_vector(ptr(arr), nl, ctypes.c_size_t(n), blob, ptr(off))
    root = (ctypes.c_uint8 * 48)()
    ln = ctypes.c_size_t()
    assert lib().orc_merkle_commit_ref(blob, ptr(off), ctypes.c_size_t(n), root, ctypes.byref(ln)) == 0
    return bytes(root[:ln.value])


def fast_coset_divide_ref(fid, lhs, rhs, offset, root, root_order):
    nl = LIMBS[fid]
    out = np.zeros((max(lhs.shape[0], 1), nl), dtype=np.uint64)
    ln = ctypes.c_size_t()
    o, r = one(fid, offset), one(fid, root)
    rc = lib().orc_fast_coset_divide_ref(fid, ptr(lhs), ctypes.c_size_t(lhs.shape[0]), ptr(rhs), ctypes.c_size_t(rhs.shape[0]), ptr(o), ptr(r),
                                         ctypes.c_size_t(root_order), ptr(out), ctypes.byref(ln))
    return rc, out[:ln.value]


# ---- subproduct-tree routines (ntt.rs:118-252) ---------------------------------------------------------------
def _np2(x):
    p = 1
    while p < x:
        p <<= 1
    return p


def fast_zerofier_ref(fid, dom, root, root_order):
    nl, n = LIMBS[fid], dom.shape[0]
    out = np.zeros((max(n + 1, 2 * _np2(n + 1)), nl), dtype=np.uint64)
    ln = ctypes.c_size_t()
    r = one(fid, root)
    rc = lib().orc_fast_zerofier_ref(fid, ptr(dom), ctypes.c_size_t(n), ptr(r), ctypes.c_size_t(root_order), ptr(out), ctypes.byref(ln))
    return rc, out[:ln.value]


def fast_evaluate_ref(fid, coef, dom, root, root_order):
    nl, n = LIMBS[fid], dom.shape[0]
    out = np.zeros((max(n, 1), nl), dtype=np.uint64)
    r = one(fid, root)
    rc = lib().orc_fast_evaluate_ref(fid, ptr(coef), ctypes.c_size_t(coef.shape[0]), ptr(dom), ctypes.c_size_t(n), ptr(r), ctypes.c_size_t(root_order), ptr(out))
    return rc, out[:n]


def fast_interpolate_ref(fid, dom, vals, root, root_order):
    nl, n = LIMBS[fid], dom.shape[0]
    out = np.zeros((max(2 * n, 1), nl), dtype=np.uint64)
    ln = ctypes.c_size_t()
    r = one(fid, root)
    rc = lib().orc_fast_interpolate_ref(fid, ptr(dom), ptr(vals), ctypes.c_size_t(n), ptr(r), ctypes.c_size_t(root_order), ptr(out), ctypes.byref(ln))
    return rc, out[:ln.value]


# ---- G2 (Fq2) -----------------------------------------------------------------------------------------------
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))   # bn128.rs:190-206
G2_INF = ((0, 0), (0, 0))


def g2_to_arr(pts):
    a = np.zeros((len(pts), 16), dtype=np.uint64)
    for i, (x, y) in enumerate(pts):
        a[i] = to_limbs([x[0], x[1], y[0], y[1]], 4).reshape(-1)
    return a


def arr_to_g2(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 16)
    out = []
    for row in a:
        v = from_limbs(row.reshape(4, 4))
        out.append(((v[0], v[1]), (v[2], v[3])))
    return out


def g2_add(P, Q):
    out = np.zeros(16, dtype=np.uint64)
    assert lib().orc_g2_add(ptr(g2_to_arr([P])), ptr(g2_to_arr([Q])), ptr(out)) == 0
    return arr_to_g2(out)[0]


def g2_mul(P, k):
    out = np.zeros(16, dtype=np.uint64)
    kk = to_limbs([k], 4)
    assert lib().orc_g2_mul(ptr(g2_to_arr([P])), ptr(kk), 4, ptr(out)) == 0
    return arr_to_g2(out)[0]


def g2_on_curve(P):
    return bool(lib().orc_g2_on_curve(ptr(g2_to_arr([P]))))


def g2_msm_ref(scal_arr, pts_arr):
    out = np.zeros(16, dtype=np.uint64)
    assert lib().orc_g2_msm_ref(ptr(scal_arr), ptr(pts_arr), ctypes.c_size_t(scal_arr.shape[0]), ptr(out)) == 0
    return arr_to_g2(out)[0]


def kzg_setup_g2_ref(alpha, max_d, g2=None):
    out = np.zeros((max_d + 1, 16), dtype=np.uint64)
    a = to_limbs([alpha], 4)
    assert lib().orc_kzg_setup_g2_ref(ptr(g2_to_arr([g2 or G2_GEN])), ptr(a), ctypes.c_size_t(max_d), ptr(out)) == 0
    return out

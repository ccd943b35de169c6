"""ctypes loader + thin wrappers for include/mzk.h.  Arrays are numpy uint64, shape (n, limbs)."""
import ctypes, os, re
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.environ.get("MZK_HIP_LIB") or os.path.join(HERE, "libmzk_hip.so")   # MZK_HIP_LIB: experimental builds (tools/)

FIELD_FR, FIELD_M128, FIELD_FQ = 0, 1, 2
LIMBS = {FIELD_FR: 4, FIELD_M128: 2, FIELD_FQ: 4}
MODULUS = {
    FIELD_FR: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    FIELD_M128: 270497897142230380135924736767050121217,
    FIELD_FQ: 21888242871839275222246405745257275088696311157297823662689037894645226208583,
}
ERRORS = {0: "MZK_OK", -1: "MZK_E_ARG", -2: "MZK_E_NOT_POW2", -3: "MZK_E_ROOT_ORDER", -4: "MZK_E_ROOT_PRIM",
          -5: "MZK_E_LENGTH", -6: "MZK_E_RANGE", -7: "MZK_E_HIP", -8: "MZK_E_NOGPU", -9: "MZK_E_CALLBACK", -10: "MZK_E_IO",
          -11: "MZK_E_BUSY", -12: "MZK_E_NOMEM"}


class MzkError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERRORS.get(code, "?"), code, msg))
        self.code = code
        self.message = msg


_lib = None


def _declared_symbols():
    hdr = os.path.join(ROOT, "include", "mzk.h")
    txt = open(hdr).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(mzk_[a-z0-9_]+)\s*\(", txt)))


DECLARED_SYMBOLS = _declared_symbols()


def lib():
    """Load the HIP library; fail loudly if it has not been built (no silent fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            raise ImportError("myzkp_amd/libmzk_hip.so is missing: run `python -m myzkp_amd.build` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Two HIP runtimes in one process
        # cannot both own the GPU, and whichever is loaded first wins the SONAME: load torch's first (when torch
        # is installed) so that this library binds to the same runtime and device tensors can be shared.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = ctypes.CDLL(SO)
        _lib.mzk_last_error.restype = ctypes.c_char_p
    return _lib


def exported_symbols():
    l = lib()
    return [s for s in DECLARED_SYMBOLS if hasattr(l, s)]


def _check(rc):
    if rc != 0:
        raise MzkError(rc, lib().mzk_last_error().decode())


def init(device=0):
    _check(lib().mzk_init(int(device)))


def set_workspace_budget(nbytes):
    """bytes of scratch a context may keep between calls (0 = no limit); idle buffers above it are released at once"""
    _check(lib().mzk_set_workspace_budget(ctypes.c_size_t(int(nbytes))))


def trim_workspace():
    """release every workspace buffer and cached transform plan of every context; returns the workspace bytes given back"""
    out = ctypes.c_size_t(0)
    _check(lib().mzk_trim_workspace(ctypes.byref(out)))
    return out.value


def workspace_bytes():
    out = ctypes.c_size_t(0)
    _check(lib().mzk_workspace_bytes(ctypes.byref(out)))
    return out.value


def shutdown():
    lib().mzk_shutdown()


def init_devices(ordinals):
    """One process driving several GPUs: context r = ordinals[r] (duplicates allowed).  Context 0 becomes current."""
    arr = (ctypes.c_int * len(ordinals))(*[int(o) for o in ordinals])
    _check(lib().mzk_init_devices(arr, len(ordinals)))


def ctx_count():
    return int(lib().mzk_ctx_count())


def ctx_select(index):
    _check(lib().mzk_ctx_select(int(index)))


def shard_range(n, rank, world):
    lo, hi = ctypes.c_size_t(), ctypes.c_size_t()
    lib().mzk_shard_range(ctypes.c_size_t(n), int(rank), int(world), ctypes.byref(lo), ctypes.byref(hi))
    return lo.value, hi.value


def to_limbs(vals, nl):
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


def points_to_array(pts):
    a = np.zeros((len(pts), 8), dtype=np.uint64)
    for i, p in enumerate(pts):
        a[i, :4] = to_limbs([p[0]], 4)[0]
        a[i, 4:] = to_limbs([p[1]], 4)[0]
    return a


def array_to_points(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 8)
    return [(from_limbs(a[i:i + 1, :4])[0], from_limbs(a[i:i + 1, 4:])[0]) for i in range(a.shape[0])]


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _one(fid, v):
    return to_limbs([v], LIMBS[fid])


def _arr(fid, a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, LIMBS[fid])


def ntt(fid, root, values, inverse=False):
    """ntt::ntt / ntt::intt (algebra/ntt.rs:7-64)."""
    v = _arr(fid, values)
    out = np.empty_like(v)
    r = _one(fid, root)
    _check(lib().mzk_ntt(fid, _p(r), _p(v), _p(out), ctypes.c_size_t(v.shape[0]), int(bool(inverse))))
    return out


def intt(fid, root, values):
    return ntt(fid, root, values, inverse=True)


def merkle_commit_field_batch(fid, codewords):
    """Merkle::commit of every row of `codewords` (batch x n x limbs, n a power of two): list of 32-byte roots."""
    c = np.ascontiguousarray(codewords, dtype=np.uint64)
    batch = c.shape[0]
    if batch == 0:
        return []
    c = c.reshape(batch, -1, LIMBS[fid])
    roots = (ctypes.c_uint8 * (32 * batch))()
    _check(lib().mzk_merkle_commit_field_batch(fid, _p(c), ctypes.c_size_t(c.shape[1]), ctypes.c_size_t(batch), roots))
    raw = bytes(roots)
    return [raw[32 * k: 32 * k + 32] for k in range(batch)]


def coset_lde_batch(fid, coefs, offset, generator, order):
    """ntt::fast_coset_evaluate of every row of `coefs` (batch x n_coef x limbs) onto one coset: mzk_coset_lde_batch."""
    c = np.ascontiguousarray(coefs, dtype=np.uint64)
    batch = c.shape[0]
    c = c.reshape(batch, -1, LIMBS[fid]) if batch else c.reshape(0, 0, LIMBS[fid])
    out = np.empty((batch, order, LIMBS[fid]), dtype=np.uint64)
    _check(lib().mzk_coset_lde_batch(fid, _p(c), ctypes.c_size_t(c.shape[1]), _p(_one(fid, offset)), _p(_one(fid, generator)), _p(out),
                                     ctypes.c_size_t(order), ctypes.c_size_t(batch)))
    return out


def ntt_batch(fid, root, columns, inverse=False):
    """ntt::ntt / ntt::intt of every row of `columns` (batch x n x limbs) in one launch per pass: mzk_ntt_batch."""
    v = np.ascontiguousarray(columns, dtype=np.uint64)
    batch = v.shape[0]
    if batch == 0:
        return v.copy()
    v = v.reshape(batch, -1, LIMBS[fid])
    out = np.empty_like(v)
    r = _one(fid, root)
    _check(lib().mzk_ntt_batch(fid, _p(r), _p(v), _p(out), ctypes.c_size_t(v.shape[1]), ctypes.c_size_t(batch), int(bool(inverse))))
    return out


def coset_lde(fid, coef, offset, generator, order):
    """ntt::fast_coset_evaluate (algebra/ntt.rs:254-269)."""
    c = _arr(fid, coef)
    out = np.empty((order, LIMBS[fid]), dtype=np.uint64)
    o, g = _one(fid, offset), _one(fid, generator)
    _check(lib().mzk_coset_lde(fid, _p(c), ctypes.c_size_t(c.shape[0]), _p(o), _p(g), _p(out), ctypes.c_size_t(order)))
    return out


def poly_scale(fid, coef, ratio, lead=None):
    """Polynomial::scale (algebra/polynomial.rs:167-174), optionally times a leading constant: lead * coef[i] * ratio^i."""
    c = _arr(fid, coef)
    out = np.empty_like(c)
    _check(lib().mzk_poly_scale(fid, _p(c), ctypes.c_size_t(c.shape[0]), _p(_one(fid, ratio)), _p(_one(fid, lead)) if lead is not None else None, _p(out)))
    return out


def fft_multiply(fid, a, b, omega):
    """Polynomial::fft_multiply (algebra/polynomial.rs:242-276)."""
    a, b = _arr(fid, a), _arr(fid, b)
    out = np.zeros((max(a.shape[0] + b.shape[0], 1), LIMBS[fid]), dtype=np.uint64)
    n = ctypes.c_size_t(0)
    w = _one(fid, omega)
    _check(lib().mzk_fft_multiply(fid, _p(a), ctypes.c_size_t(a.shape[0]), _p(b), ctypes.c_size_t(b.shape[0]), _p(w), _p(out), ctypes.byref(n)))
    return out[:n.value]


def fast_multiply(fid, a, b, root, root_order):
    """ntt::fast_multiply (algebra/ntt.rs:66-116)."""
    a, b = _arr(fid, a), _arr(fid, b)
    out = np.zeros((max(root_order, a.shape[0] + b.shape[0], 1), LIMBS[fid]), dtype=np.uint64)
    n = ctypes.c_size_t(0)
    w = _one(fid, root)
    _check(lib().mzk_fast_multiply(fid, _p(a), ctypes.c_size_t(a.shape[0]), _p(b), ctypes.c_size_t(b.shape[0]), _p(w),
                                   ctypes.c_size_t(root_order), _p(out), ctypes.byref(n)))
    return out[:n.value]


def root_of_unity(fid, log2n):
    out = np.zeros((1, LIMBS[fid]), dtype=np.uint64)
    _check(lib().mzk_root_of_unity(fid, int(log2n), _p(out)))
    return from_limbs(out)[0]


def msm_g1(scalars, points):
    """Polynomial::eval_with_powers_on_curve (algebra/polynomial.rs:156-165) / commit_kzg (kzg.rs:57-59)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    if p.shape[0] < s.shape[0]:
        raise MzkError(-5, "index out of bounds: the len is %d but the index is %d" % (p.shape[0], p.shape[0]))
    out = np.zeros((1, 8), dtype=np.uint64)
    _check(lib().mzk_msm_g1_bn254(_p(s), _p(p), ctypes.c_size_t(s.shape[0]), _p(out)))
    return array_to_points(out)[0]


kzg_commit = msm_g1


def msm_g1_multi(scalars, points):
    """The same MSM sharded over every context of init_devices (contiguous shards, gather of 128-byte partials, fold)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    if p.shape[0] < s.shape[0]:
        raise MzkError(-5, "index out of bounds: the len is %d but the index is %d" % (p.shape[0], p.shape[0]))
    out = np.zeros((1, 8), dtype=np.uint64)
    _check(lib().mzk_msm_g1_bn254_multi(_p(s), _p(p), ctypes.c_size_t(s.shape[0]), _p(out)))
    return array_to_points(out)[0]


LAYOUT_CONTIGUOUS, LAYOUT_CYCLIC = 0, 1


def ntt_multi(fid, root, values, inverse=False):
    """ntt::ntt / ntt::intt of ONE vector spread over every context of init_devices (four-step layout, mzk_ntt_multi)."""
    v = _arr(fid, values)
    out = np.empty_like(v)
    _check(lib().mzk_ntt_multi(fid, _p(_one(fid, root)), _p(v), _p(out), ctypes.c_size_t(v.shape[0]), int(bool(inverse))))
    return out


def ntt_multi_dev(fid, root, in_ptrs, out_ptrs, n, inverse=False, layout_in=LAYOUT_CONTIGUOUS, layout_out=LAYOUT_CONTIGUOUS):
    """The same with the parts resident in HBM: in_ptrs[r] / out_ptrs[r] = device pointers (ints) on context r's GPU."""
    W = len(in_ptrs)
    ins = (ctypes.c_void_p * W)(*[ctypes.c_void_p(int(p)) for p in in_ptrs])
    outs = (ctypes.c_void_p * W)(*[ctypes.c_void_p(int(p)) for p in out_ptrs])
    _check(lib().mzk_ntt_multi_dev(fid, _p(_one(fid, root)), ins, outs, ctypes.c_size_t(n), int(bool(inverse)), int(layout_in), int(layout_out)))


class SrsMulti:
    """PublicKeyKZG.powers_1 sharded over the contexts of init_devices (kzg.rs:8-11): from host points, or built on
    the GPUs from (alpha, g1) like setup_kzg (kzg.rs:27-40)."""

    def __init__(self, powers=None, alpha=None, max_d=None, g1=(1, 2), with_tables=1):
        self._h = ctypes.c_void_p()
        if powers is not None:
            p = np.ascontiguousarray(powers, dtype=np.uint64).reshape(-1, 8)
            self.n = p.shape[0]
            _check(lib().mzk_srs_upload_multi(_p(p), ctypes.c_size_t(self.n), ctypes.byref(self._h)))
        else:
            a, g = _one(FIELD_FR, alpha), points_to_array([g1])
            self.n = max_d + 1
            _check(lib().mzk_kzg_setup_srs_multi(_p(a), _p(g), ctypes.c_size_t(max_d), int(with_tables), ctypes.byref(self._h)))
        lib().mzk_srs_multi_shard_lo.restype = ctypes.c_size_t
        self.world = int(lib().mzk_srs_multi_world(self._h))
        self.lo = [int(lib().mzk_srs_multi_shard_lo(self._h, r)) for r in range(self.world + 1)]

    def commit(self, coef):
        c = np.ascontiguousarray(coef, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((1, 8), dtype=np.uint64)
        _check(lib().mzk_kzg_commit_srs_multi(self._h, _p(c), ctypes.c_size_t(c.shape[0]), _p(out)))
        return array_to_points(out)[0]

    def commit_dev(self, shard_ptrs, n):
        """shard_ptrs[r]: device pointer (int) on context r's GPU to coefficients [lo[r], min(lo[r+1], n))."""
        arr = (ctypes.c_void_p * self.world)(*[ctypes.c_void_p(int(x)) for x in shard_ptrs])
        out = np.zeros((1, 8), dtype=np.uint64)
        _check(lib().mzk_kzg_commit_srs_multi_dev(self._h, arr, ctypes.c_size_t(n), _p(out)))
        return array_to_points(out)[0]

    def close(self):
        if self._h:
            lib().mzk_srs_multi_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def kzg_setup_g1(alpha, max_d, g1=(1, 2)):
    """setup_kzg (algebra/kzg.rs:27-40), G1 powers for a caller-supplied trapdoor."""
    out = np.zeros((max_d + 1, 8), dtype=np.uint64)
    a, g = _one(FIELD_FR, alpha), points_to_array([g1])
    _check(lib().mzk_kzg_setup_g1(_p(a), _p(g), ctypes.c_size_t(max_d), _p(out)))
    return out


def kzg_open(coef, u, powers):
    """open_kzg (algebra/kzg.rs:61-72) -> (y, w)."""
    c = np.ascontiguousarray(coef, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(powers, dtype=np.uint64).reshape(-1, 8)
    if c.shape[0] > 1 and p.shape[0] < c.shape[0] - 1:
        raise MzkError(-5, "index out of bounds: the len is %d but the index is %d" % (p.shape[0], p.shape[0]))
    y = np.zeros((1, 4), dtype=np.uint64)
    w = np.zeros((1, 8), dtype=np.uint64)
    uu = _one(FIELD_FR, u)
    _check(lib().mzk_kzg_open(_p(c), ctypes.c_size_t(c.shape[0]), _p(uu), _p(p), _p(y), _p(w)))
    return from_limbs(y)[0], array_to_points(w)[0]


def kzg_batch_open(coef, us, powers):
    """batch_open_kzg (algebra/kzg.rs:74-88) -> (ys, w)."""
    c = np.ascontiguousarray(coef, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(powers, dtype=np.uint64).reshape(-1, 8)
    u = to_limbs(list(us), 4)
    k = u.shape[0]
    nq = max(c.shape[0] - k, 0)
    if p.shape[0] < nq:
        raise MzkError(-5, "index out of bounds: the len is %d but the index is %d" % (p.shape[0], p.shape[0]))
    ys = np.zeros((max(k, 1), 4), dtype=np.uint64)
    w = np.zeros((1, 8), dtype=np.uint64)
    _check(lib().mzk_kzg_batch_open(_p(c), ctypes.c_size_t(c.shape[0]), _p(u), ctypes.c_size_t(k), _p(p), _p(ys), _p(w)))
    return from_limbs(ys[:k]), array_to_points(w)[0]


def kzg_prove_degree_bound(coef, powers, d):
    """prove_degree_bound (algebra/kzg.rs:121-134)."""
    c = np.ascontiguousarray(coef, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(powers, dtype=np.uint64).reshape(-1, 8)
    out = np.zeros((1, 8), dtype=np.uint64)
    _check(lib().mzk_kzg_prove_degree_bound(_p(c), ctypes.c_size_t(c.shape[0]), _p(p), ctypes.c_size_t(p.shape[0]), ctypes.c_size_t(d), _p(out)))
    return array_to_points(out)[0]


def fri_fold(fid, codeword, alpha, offset, omega):
    """FRI split-and-fold (zkstark/fri.rs:182-193)."""
    c = _arr(fid, codeword)
    out = np.zeros((max(c.shape[0] // 2, 1), LIMBS[fid]), dtype=np.uint64)
    a, o, w = _one(fid, alpha), _one(fid, offset), _one(fid, omega)
    _check(lib().mzk_fri_fold(fid, _p(c), ctypes.c_size_t(c.shape[0]), _p(a), _p(o), _p(w), _p(out)))
    return out[:c.shape[0] // 2]


class Srs:
    """Device-resident PublicKeyKZG.powers_1 (algebra/kzg.rs:8-11) for repeated commits."""

    def __init__(self, powers):
        p = np.ascontiguousarray(powers, dtype=np.uint64).reshape(-1, 8)
        self._h = ctypes.c_void_p()
        self.n = p.shape[0]
        _check(lib().mzk_srs_upload(_p(p), ctypes.c_size_t(self.n), ctypes.byref(self._h)))

    def commit(self, coef):
        c = np.ascontiguousarray(coef, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((1, 8), dtype=np.uint64)
        _check(lib().mzk_kzg_commit_srs(self._h, _p(c), ctypes.c_size_t(c.shape[0]), _p(out)))
        return array_to_points(out)[0]

    def commit_batch(self, coefs):
        """commit_kzg of every row of `coefs` (count x n x 4 limbs; all polynomials of one length), one commit in flight
        per context of this GPU (init_devices([d, d, d, d])): mzk_kzg_commit_srs_batch."""
        c = np.ascontiguousarray(coefs, dtype=np.uint64)
        count = c.shape[0]
        if count == 0:
            return []
        c = c.reshape(count, -1, 4)
        out = np.zeros((max(count, 1), 8), dtype=np.uint64)
        _check(lib().mzk_kzg_commit_srs_batch(self._h, _p(c), ctypes.c_size_t(c.shape[1]), ctypes.c_size_t(count), _p(out)))
        return array_to_points(out[:count])

    def commit_many(self, coefs):
        """commit_kzg of every row of `coefs` as ONE grid-batched pass (mzk_kzg_commit_srs_many_dev): the reference's per-row /
        per-chunk loops (das/avail.rs:88-98, das/eigenda.rs:92-101, algebra/gemini.rs:112-114)."""
        import torch
        c = np.ascontiguousarray(coefs, dtype=np.uint64)
        count = c.shape[0]
        if count == 0:
            return []
        c = c.reshape(count, -1, 4)
        st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        d_c = torch.from_numpy(c.view(np.int64).reshape(-1).copy()).cuda() if c.size else torch.zeros(4, dtype=torch.int64, device="cuda")
        d_o = torch.zeros(count * 8, dtype=torch.int64, device="cuda")
        _check(lib().mzk_kzg_commit_srs_many_dev(self._h, ctypes.c_void_p(d_c.data_ptr()), ctypes.c_size_t(c.shape[1]), ctypes.c_size_t(count),
                                                 ctypes.c_void_p(d_o.data_ptr()), st))
        torch.cuda.synchronize()
        return array_to_points(d_o.cpu().numpy().view(np.uint64).reshape(count, 8))

    def build_direct(self, window_bits=0, max_bytes=0):
        """Direct tables for batches of short polynomials (mzk_srs_build_direct); returns the width built."""
        _check(lib().mzk_srs_build_direct(self._h, int(window_bits), ctypes.c_size_t(max_bytes), None))
        return int(lib().mzk_srs_direct_bits(self._h))

    def drop_direct(self):
        lib().mzk_srs_drop_direct(self._h)

    def save(self, path, with_tables=False):
        """Raw little-endian dump of powers_1 (+ optionally the window tables): mzk_srs_save."""
        _check(lib().mzk_srs_save(self._h, os.fsencode(path), int(bool(with_tables))))

    @classmethod
    def load(cls, path, with_tables=1):
        self = cls.__new__(cls)
        self._h = ctypes.c_void_p()
        _check(lib().mzk_srs_load(os.fsencode(path), int(with_tables), ctypes.byref(self._h)))
        lib().mzk_srs_len.restype = ctypes.c_size_t
        self.n = int(lib().mzk_srs_len(self._h))
        return self

    def download(self):
        """powers_1 back as an (n, 8) limb array."""
        out = np.zeros((max(self.n, 1), 8), dtype=np.uint64)
        _check(lib().mzk_srs_download(self._h, _p(out), ctypes.c_size_t(self.n)))
        return out[:self.n]

    def close(self):
        if self._h:
            lib().mzk_srs_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MerkleTree:
    """Merkle::commit / Merkle::open (algebra/merkle.rs:15-46) over a codeword, resident in HBM.

    Leaves are bincode(FiniteFieldElement) of the elements (zkstark/fri.rs:160-166) or arbitrary byte strings."""

    def __init__(self, fid=None, elems=None, leaves=None, negative=None):
        self._h = ctypes.c_void_p()
        self.fid = fid if leaves is None else None
        if negative is not None:       # (magnitude, Sign::Minus flag) pairs: unsanitized elements (field.rs:98-110)
            e = _arr(fid, elems)
            ng = np.ascontiguousarray(negative, dtype=np.uint8)
            self.n = e.shape[0]
            self.stride = 48
            _check(lib().mzk_merkle_build_field_signed(fid, _p(e), _p(ng), ctypes.c_size_t(self.n), ctypes.byref(self._h)))
        elif leaves is not None:
            blob = b"".join(leaves)
            off = np.zeros(len(leaves) + 1, dtype=np.uint64)
            off[1:] = np.cumsum([len(x) for x in leaves], dtype=np.uint64) if leaves else []
            buf = (ctypes.c_uint8 * max(len(blob), 1)).from_buffer_copy(blob or b"\0")
            self.n = len(leaves)
            self.stride = max([32] + [len(x) for x in leaves])
            _check(lib().mzk_merkle_build_bytes(buf, _p(off), ctypes.c_size_t(self.n), ctypes.byref(self._h)))
        else:
            e = _arr(fid, elems)
            self.n = e.shape[0]
            self.stride = 48
            _check(lib().mzk_merkle_build_field(fid, _p(e), ctypes.c_size_t(self.n), ctypes.byref(self._h)))

    def root(self):
        buf = (ctypes.c_uint8 * max(self.stride, 48))()
        ln = ctypes.c_size_t()
        _check(lib().mzk_merkle_root(self._h, buf, ctypes.c_size_t(len(buf)), ctypes.byref(ln)))
        return bytes(buf[:ln.value])

    def open(self, index):
        depth_cap = max(self.n.bit_length(), 1)
        buf = (ctypes.c_uint8 * (self.stride * depth_cap))()
        lens = (ctypes.c_uint64 * depth_cap)()
        depth = ctypes.c_size_t()
        _check(lib().mzk_merkle_open(self._h, ctypes.c_size_t(index), buf, ctypes.c_size_t(self.stride), lens, ctypes.byref(depth)))
        raw = bytes(buf)
        return [raw[k * self.stride:k * self.stride + lens[k]] for k in range(depth.value)]

    def open_many(self, indices):
        """Merkle::open for every index in one gather (mzk_merkle_open_batch): list of paths as `open` returns them."""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        count = idx.shape[0]
        depth_cap = max(self.n.bit_length(), 1)
        buf = (ctypes.c_uint8 * max(self.stride * depth_cap * count, 1))()
        lens = (ctypes.c_uint64 * max(depth_cap * count, 1))()
        depth = ctypes.c_size_t()
        _check(lib().mzk_merkle_open_batch(self._h, _p(idx), ctypes.c_size_t(count), buf, ctypes.c_size_t(self.stride), lens, ctypes.byref(depth)))
        raw, d = bytes(buf), depth.value
        return [[raw[(q * d + k) * self.stride:(q * d + k) * self.stride + lens[q * d + k]] for k in range(d)] for q in range(count)]

    def leaves(self, indices, with_sign=False):
        """The elements the tree was built over at `indices` (mzk_merkle_leaves): (n, limbs) magnitudes[, Sign::Minus flags]."""
        idx = np.ascontiguousarray(indices, dtype=np.uint64)
        nl = LIMBS[self.fid] if getattr(self, "fid", None) is not None else None
        if nl is None:
            raise MzkError(-1, "leaves: field-element trees only")
        out = np.zeros((idx.shape[0], nl), dtype=np.uint64)
        neg = np.zeros(idx.shape[0], dtype=np.uint8)
        _check(lib().mzk_merkle_leaves(self._h, _p(idx), ctypes.c_size_t(idx.shape[0]), _p(out), _p(neg) if with_sign else None))
        return (out, neg) if with_sign else out

    def close(self):
        if self._h:
            lib().mzk_merkle_free(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def merkle_open_multi(trees, index_lists):
    """Openings of several trees in ONE call (mzk_merkle_open_multi): index_lists[t] are the indices to open in trees[t] (a
    MerkleTree or None with an empty list).  Returns, per tree, the list of paths `MerkleTree.open` would return."""
    T = len(trees)
    counts = [len(ix) for ix in index_lists]
    flat = np.ascontiguousarray([i for ix in index_lists for i in ix], dtype=np.uint64)
    handles = (ctypes.c_void_p * max(T, 1))(*[t._h if t is not None else None for t in trees])
    cnt = (ctypes.c_size_t * max(T, 1))(*counts)
    depths = (ctypes.c_size_t * max(T, 1))()
    entries = sum(c * max(t.n.bit_length(), 1) for t, c in zip(trees, counts) if t is not None)
    stride = 48
    buf = (ctypes.c_uint8 * max(stride * entries, 1))()
    lens = (ctypes.c_uint64 * max(entries, 1))()
    _check(lib().mzk_merkle_open_multi(handles, ctypes.c_size_t(T), _p(flat) if flat.size else None, cnt, buf, ctypes.c_size_t(stride), lens, depths))
    raw, out, at = memoryview(buf), [], 0
    for t in range(T):
        d, paths = depths[t], []
        for q in range(counts[t]):
            paths.append([bytes(raw[(at + k) * stride:(at + k) * stride + lens[at + k]]) for k in range(d)])
            at += d
        out.append(paths)
    return out


def merkle_commit_field(fid, elems):
    """Merkle::commit(&codeword.map(bincode::serialize)) (fri.rs:160-166)."""
    e = _arr(fid, elems)
    buf = (ctypes.c_uint8 * 48)()
    ln = ctypes.c_size_t()
    _check(lib().mzk_merkle_commit_field(fid, _p(e), ctypes.c_size_t(e.shape[0]), buf, ctypes.c_size_t(48), ctypes.byref(ln)))
    return bytes(buf[:ln.value])


_FRI_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t,
                           ctypes.POINTER(ctypes.c_uint64))


def fri_commit(fid, codeword, omega, offset, num_rounds, challenge, negative=None, keep_trees=False, codewords=True, device_ptr=None, n=None):
    """FRI::commit (zkstark/fri.rs:144-209), codewords resident in HBM.  challenge(round, last, root_bytes) -> alpha
    (int; ignored when last).  Returns (codewords, roots).  negative: optional Sign::Minus flags of the initial
    codeword, then given as magnitudes (round 0 commits to the unsanitized elements, fri.rs:160-166).
    keep_trees: also return the rounds' Merkle trees (MerkleTree objects on the device, None for a one-element round)
    for the query phase: (codewords, roots, trees).  codewords=False (with keep_trees): nothing but the roots comes back to
    the host -- (None, roots, trees); the query phase reads values and paths from the trees (MerkleTree.leaves,
    merkle_open_multi).  device_ptr / n: the initial codeword is already in HBM (mzk_fri_commit_keep_trees_dev); `codeword`
    is then ignored."""
    nl = LIMBS[fid]
    if device_ptr is None:
        c = _arr(fid, codeword)
        n = c.shape[0]
    elif not keep_trees:
        raise MzkError(-1, "fri_commit: a device-resident codeword needs keep_trees=True")

    failure = []

    def cb(user, rnd, last, root, root_len, alpha_out):
        # An exception must not escape into C (ctypes would print it and carry on with whatever alpha_out holds):
        # keep it, make the C loop stop (non-zero status -> MZK_E_CALLBACK), re-raise below.
        try:
            a = challenge(rnd, bool(last), ctypes.string_at(root, root_len))      # (slicing the pointer builds a list first: ~4 us per round)
            if last:
                return 0
            if a is None:
                raise ValueError("challenge(%d) returned no alpha" % rnd)
            a = int(a)
            for j in range(nl):
                alpha_out[j] = (a >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
            return 0
        except BaseException as ex:      # noqa: BLE001 -- re-raised by fri_commit
            failure.append(ex)
            return 1

    total = sum(n >> r for r in range(num_rounds))
    roots = (ctypes.c_uint8 * (48 * max(num_rounds, 1)))()
    lens = (ctypes.c_uint64 * max(num_rounds, 1))()
    allcw = np.zeros((max(total, 1) if (codewords or not keep_trees) else 1, nl), dtype=np.uint64)
    w, o = _one(fid, omega), _one(fid, offset)
    fn = _FRI_CB(cb)
    handles = (ctypes.c_void_p * max(num_rounds, 1))()
    if keep_trees:
        ng = None if negative is None else np.ascontiguousarray(negative, dtype=np.uint8)
        cw_out = _p(allcw) if codewords else None
        if device_ptr is not None:
            rc = lib().mzk_fri_commit_keep_trees_dev(fid, ctypes.c_void_p(int(device_ptr)), None if ng is None else _p(ng), ctypes.c_size_t(n), _p(w), _p(o),
                                                     int(num_rounds), fn, None, roots, lens, cw_out, handles)
        else:
            rc = lib().mzk_fri_commit_keep_trees(fid, _p(c), None if ng is None else _p(ng), ctypes.c_size_t(n), _p(w), _p(o), int(num_rounds), fn, None,
                                                 roots, lens, cw_out, handles)
    elif negative is not None:
        ng = np.ascontiguousarray(negative, dtype=np.uint8)
        rc = lib().mzk_fri_commit_signed(fid, _p(c), _p(ng), ctypes.c_size_t(n), _p(w), _p(o), int(num_rounds), fn, None, roots, lens, _p(allcw))
    else:
        rc = lib().mzk_fri_commit(fid, _p(c), ctypes.c_size_t(n), _p(w), _p(o), int(num_rounds), fn, None, roots, lens, _p(allcw))
    if failure:
        raise failure[0]
    _check(rc)
    cws, rts, at = [], [], 0
    raw = bytes(roots)
    for r in range(num_rounds):
        if codewords or not keep_trees:
            cws.append(allcw[at:at + (n >> r)].copy())
        rts.append(raw[48 * r:48 * r + lens[r]])
        at += n >> r
    if keep_trees:
        trees = []
        for r in range(num_rounds):
            if not handles[r]:
                trees.append(None)
                continue
            t = MerkleTree.__new__(MerkleTree)
            t._h = ctypes.c_void_p(handles[r])
            t.n = n >> r
            t.stride = 48
            t.fid = fid
            trees.append(t)
        return (cws if codewords else None), rts, trees
    return cws, rts


def fast_coset_divide(fid, lhs, rhs, offset, root, root_order):
    """ntt::fast_coset_divide (algebra/ntt.rs:271-330)."""
    a, b = _arr(fid, lhs), _arr(fid, rhs)
    out = np.zeros((max(a.shape[0], 1), LIMBS[fid]), dtype=np.uint64)
    ln = ctypes.c_size_t()
    o, r = _one(fid, offset), _one(fid, root)
    _check(lib().mzk_fast_coset_divide(fid, _p(a), ctypes.c_size_t(a.shape[0]), _p(b), ctypes.c_size_t(b.shape[0]), _p(o), _p(r),
                                       ctypes.c_size_t(root_order), _p(out), ctypes.byref(ln)))
    return out[:ln.value]


def _np2(x):
    p = 1
    while p < x:
        p <<= 1
    return p


def fast_zerofier(fid, domain, root, root_order):
    """ntt::fast_zerofier (algebra/ntt.rs:118-144)."""
    d = _arr(fid, domain)
    n = d.shape[0]
    out = np.zeros((max(n + 1, _np2(n + 1)), LIMBS[fid]), dtype=np.uint64)
    ln = ctypes.c_size_t()
    r = _one(fid, root)
    _check(lib().mzk_fast_zerofier(fid, _p(d), ctypes.c_size_t(n), _p(r), ctypes.c_size_t(root_order), _p(out), ctypes.byref(ln)))
    return out[:ln.value]


def fast_evaluate(fid, coef, domain, root, root_order):
    """ntt::fast_evaluate (algebra/ntt.rs:146-189)."""
    c, d = _arr(fid, coef), _arr(fid, domain)
    out = np.zeros((max(d.shape[0], 1), LIMBS[fid]), dtype=np.uint64)
    r = _one(fid, root)
    _check(lib().mzk_fast_evaluate(fid, _p(c), ctypes.c_size_t(c.shape[0]), _p(d), ctypes.c_size_t(d.shape[0]), _p(r), ctypes.c_size_t(root_order), _p(out)))
    return out[:d.shape[0]]


def fast_interpolate(fid, domain, values, root, root_order):
    """ntt::fast_interpolate (algebra/ntt.rs:191-252)."""
    d, v = _arr(fid, domain), _arr(fid, values)
    if d.shape[0] != v.shape[0]:
        raise MzkError(-5, "assertion `left == right` failed (domain.len(), values.len())")
    out = np.zeros((max(d.shape[0], 1), LIMBS[fid]), dtype=np.uint64)
    ln = ctypes.c_size_t()
    r = _one(fid, root)
    _check(lib().mzk_fast_interpolate(fid, _p(d), _p(v), ctypes.c_size_t(d.shape[0]), _p(r), ctypes.c_size_t(root_order), _p(out), ctypes.byref(ln)))
    return out[:ln.value]


def fast_interpolate_batch(fid, domain, values, root, root_order):
    """ntt::fast_interpolate of every row of `values` (batch x n x limbs) over one domain: list of coefficient arrays."""
    d = _arr(fid, domain)
    v = np.ascontiguousarray(values, dtype=np.uint64)
    batch = v.shape[0]
    if batch == 0:
        return []
    v = v.reshape(batch, -1, LIMBS[fid])
    if d.shape[0] != v.shape[1]:
        raise MzkError(-5, "assertion `left == right` failed (domain.len(), values.len())")
    n = d.shape[0]
    out = np.zeros((batch, max(n, 1), LIMBS[fid]), dtype=np.uint64)
    lens = (ctypes.c_size_t * batch)()
    r = _one(fid, root)
    _check(lib().mzk_fast_interpolate_batch(fid, _p(d), _p(v), ctypes.c_size_t(n), ctypes.c_size_t(batch), _p(r), ctypes.c_size_t(root_order), _p(out), lens))
    return [out[k, :lens[k]] for k in range(batch)]


def fast_interpolate_batch_dev(fid, domain, d_values_ptr, batch, root, root_order, d_out_ptr, stream=0):
    """mzk_fast_interpolate_batch_dev: values and coefficients in HBM (raw device pointers, batch rows of len(domain) elements each);
    returns the trimmed lengths."""
    d = _arr(fid, domain)
    n = d.shape[0]
    lens = (ctypes.c_size_t * max(batch, 1))()
    r = _one(fid, root)
    _check(lib().mzk_fast_interpolate_batch_dev(fid, _p(d), ctypes.c_void_p(d_values_ptr), ctypes.c_size_t(n), ctypes.c_size_t(batch), _p(r),
                                                ctypes.c_size_t(root_order), ctypes.c_void_p(d_out_ptr), lens, ctypes.c_void_p(stream)))
    return [int(lens[k]) for k in range(batch)]


def g2_points_to_array(pts):
    """[((x0, x1), (y0, y1)), ...] -> (n, 16) limbs; infinity = ((0, 0), (0, 0))."""
    a = np.zeros((len(pts), 16), dtype=np.uint64)
    for i, (x, y) in enumerate(pts):
        a[i] = to_limbs([x[0], x[1], y[0], y[1]], 4).reshape(-1)
    return a


def array_to_g2_points(a):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 16)
    out = []
    for row in a:
        v = from_limbs(row.reshape(4, 4))
        out.append(((v[0], v[1]), (v[2], v[3])))
    return out


def msm_g2(scalars, points):
    """Polynomial::eval_with_powers_on_curve over G2 (kzg.rs:114)."""
    s = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    p = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 16)
    if p.shape[0] < s.shape[0]:
        raise MzkError(-5, "index out of bounds: powers.len() < coef.len() (polynomial.rs:162)")
    out = np.zeros(16, dtype=np.uint64)
    _check(lib().mzk_msm_g2_bn254(_p(s), _p(p), ctypes.c_size_t(s.shape[0]), _p(out)))
    return array_to_g2_points(out)[0]


def kzg_setup_g2(alpha, max_d, g2):
    """powers_2 of setup_kzg_with_full_g2 (kzg.rs:42-55) for a given alpha."""
    a = to_limbs([alpha], 4)
    g = g2_points_to_array([g2])
    out = np.zeros((max_d + 1, 16), dtype=np.uint64)
    _check(lib().mzk_kzg_setup_g2(_p(a), _p(g), ctypes.c_size_t(max_d), _p(out)))
    return out

"""Multi-GPU sharding of the MSM (one process per GPU, torch.distributed; backend "nccl" = RCCL).

sum_i s_i P_i is a sum over independent pairs: rank g of G owns the contiguous slice
[g n/G, (g+1) n/G), reduces it to ONE partial group element on its GPU, and the only exchange is an
all-gather of G fixed-size partial records (RCCL has no elliptic-curve reduction operator, so the
"reduce" is gather + local fold).  The compute callbacks are injected so that the exchange logic can
be exercised on CPU with the gloo backend (tests/test_sharded_gloo.py) where the oracle plays the
local compute."""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced slice of [0, n) for `rank` of `world` (first n % world ranks get one more)."""
    base, extra = divmod(n, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


# A world of ONE rank needs no exchange and normally takes none.  With FORCE_COLLECTIVES set the collectives are issued all the
# same (a gather of one record, an all-to-all of one chunk): that is how a one-GPU box runs the RCCL branch -- process group,
# stream ordering against the C ABI's kernels, int64 device tensors -- before the first multi-GPU job does
# (tests/test_gpu_rccl_world1.py, `bench.py --force-process-group`).
FORCE_COLLECTIVES = False


def all_gather_partials(partial):
    """partial: 1-D tensor (one fixed-size record per rank) -> (world, record) tensor on every rank."""
    if not dist.is_available() or not dist.is_initialized() or (dist.get_world_size() == 1 and not FORCE_COLLECTIVES):
        return partial.reshape(1, -1)
    world = dist.get_world_size()
    flat = partial.contiguous().reshape(-1)
    out = torch.empty(world * flat.numel(), dtype=partial.dtype, device=partial.device)
    dist.all_gather_into_tensor(out, flat)
    return out.reshape(world, flat.numel())


def sharded_msm(local_partial_fn, fold_fn):
    """local_partial_fn() -> this rank's partial record (tensor); fold_fn(records) -> final result."""
    return fold_fn(all_gather_partials(local_partial_fn()))


# ---- open_kzg sharded over the ranks: the quotient's suffix recurrence with ONE 32-byte value per rank exchanged ----------------------
# kzg.rs:61-72: y = f(u), q = (f - y) / (X - u), w = MSM(q, powers).  q_{i-1} = b_i with b_i = c_i + u b_{i+1} (b_n = 0) runs over the
# WHOLE coefficient vector; the rank that holds c[lo, hi) only needs b_hi from the ranks above it:
#   1. value(g) = sum_t c[lo_g + t] u^t        -- local (ops.slice_value); what b_lo(g) would be if nothing followed the slice
#   2. all-gather of the W values (32 bytes each)
#   3. every rank, top slice down: b_lo(g) = value(g) + u^(hi_g - lo_g) * b_hi(g),  b_hi(g - 1) = b_lo(g);  y = b_lo(0) = f(u)
#   4. q[lo_g .. hi_g) = b[lo_g + 1 .. hi_g]   -- local (ops.slice_quotient with carry b_hi(g)); the rank commits it against ITS powers
# and the W witness partials are gathered and folded like every other sharded MSM.  Rounds 2-5 computed the whole quotient on every
# rank (0.55 ms of the 2^22 end-to-end run that did not shrink with the number of GPUs).
def all_gather_values(value_limbs):
    """value_limbs: 1-D int64 tensor (4 limbs) -> (world, 4) on every rank; FORCE_COLLECTIVES as for all_gather_partials."""
    return all_gather_partials(value_limbs)


def open_carries(values, lens, u, modulus):
    """values[g]: slice g as a polynomial at u; lens[g]: its length.  Returns (y, carries): carries[g] = b at the first index behind
    slice g (0 for the top slice), y = f(u)."""
    W = len(values)
    carries, b = [0] * W, 0
    for g in range(W - 1, -1, -1):
        carries[g] = b
        b = (values[g] + pow(u, lens[g], modulus) * b) % modulus
    return b, carries


def sharded_open_quotient(ops, coef_slice, n, u, modulus, rank, world):
    """This rank's slice q[lo, hi) of open_kzg's quotient and y = f(u), from its slice c[lo, hi) of the n coefficients.
    ops.slice_value(coef_slice, u) -> int;  ops.gather_values(int) -> list of the W ranks' ints;
    ops.slice_quotient(coef_slice, u, carry) -> the slice of q (same length as the slice; the top rank's last element is b_n = 0)."""
    if world == 1 and not FORCE_COLLECTIVES and hasattr(ops, "whole"):
        return ops.whole(coef_slice, u)            # one rank holds everything: the one-piece recurrence (one pass, no exchange)
    lens = [shard_range(n, g, world)[1] - shard_range(n, g, world)[0] for g in range(world)]
    values = ops.gather_values(ops.slice_value(coef_slice, u))
    y, carries = open_carries(values, lens, u, modulus)
    return y, ops.slice_quotient(coef_slice, u, carries[rank])


class DeviceOpenOps:
    """The local passes of sharded_open_quotient on the GPU (C ABI on torch's current stream) and the gather over the process group."""

    def __init__(self, stream=None, out=None):
        """stream: a raw HIP stream the local passes run on (default: torch's current stream).  The gather is torch's, on torch's
        stream, so a pass on another stream is waited for before its value is handed on.  out: the caller's buffer for the slice of q
        (int64, as long as the coefficient slice; a fresh tensor per call otherwise)."""
        import ctypes
        import myzkp_amd as mz
        self.ct, self.mz, self.L, self.stream, self.out = ctypes, mz, mz.lib(), stream, out

    def _q(self, coef):
        return self.out if self.out is not None and self.out.numel() == coef.numel() else torch.empty_like(coef)

    def whole(self, coef, u):
        """world 1: y and q in one pass (mzk_kzg_open_quotient_dev); q has the slice's length, its last element b_n = 0"""
        import numpy as np
        ul = self.mz.to_limbs([u], 4)
        q = self._q(coef)
        y = torch.zeros(4, dtype=torch.int64, device=coef.device)
        n = coef.numel() // 4
        q[-4:] = 0
        self._ok(self.L.mzk_kzg_open_quotient_dev(self.ct.c_void_p(coef.data_ptr()), self.ct.c_size_t(n), ul.ctypes.data_as(self.ct.c_void_p),
                                                  self.ct.c_void_p(y.data_ptr()), self.ct.c_void_p(q.data_ptr()), self._st()))
        self._join()
        return self.mz.from_limbs(y.cpu().numpy().view(np.uint64).reshape(1, 4))[0], q

    def _st(self):
        return self.ct.c_void_p(self.stream if self.stream is not None else torch.cuda.current_stream().cuda_stream)

    def _join(self):
        if self.stream is not None:
            torch.cuda.ExternalStream(self.stream).synchronize()

    def _ok(self, rc):
        if rc != 0:
            raise self.mz.MzkError(rc, self.L.mzk_last_error().decode())

    def slice_value(self, coef, u):
        ul = self.mz.to_limbs([u], 4)
        out = torch.zeros(4, dtype=torch.int64, device=coef.device)
        self._ok(self.L.mzk_kzg_open_slice_value_dev(self.ct.c_void_p(coef.data_ptr()), self.ct.c_size_t(coef.numel() // 4), ul.ctypes.data_as(self.ct.c_void_p),
                                                     self.ct.c_void_p(out.data_ptr()), self._st()))
        self._join()
        return out                       # stays on the device: the gather takes it from there

    def gather_values(self, value):
        import numpy as np
        recs = all_gather_values(value)
        return self.mz.from_limbs(recs.cpu().numpy().view(np.uint64).reshape(-1, 4))

    def slice_quotient(self, coef, u, carry):
        ul, cl = self.mz.to_limbs([u], 4), self.mz.to_limbs([carry], 4)
        q = self._q(coef)
        self._ok(self.L.mzk_kzg_open_slice_quotient_dev(self.ct.c_void_p(coef.data_ptr()), self.ct.c_size_t(coef.numel() // 4), ul.ctypes.data_as(self.ct.c_void_p),
                                                        cl.ctypes.data_as(self.ct.c_void_p), self.ct.c_void_p(q.data_ptr()), self._st()))
        return q


# ---- one transform sharded over the ranks (the four-step layout with an all-to-all) ---------------------------
# n = W m points over W ranks (W a power of two, W^2 <= n), natural order in and out as in ntt.rs:7-64.  Two layouts of a
# distributed vector:   "contiguous": rank r holds x[r m .. (r+1) m)          "cyclic": rank r holds x[r], x[r + W], ...
# Write j = a m + t (a = rank of a contiguous input) and k = W k2 + k1:
#     X[W k2 + k1] = sum_t  w_m^(t k2) * w^(t k1) * [ sum_a x[a m + t] w_W^(a k1) ]          w_m = w^W,  w_W = w^m
# "dif" (contiguous in -> cyclic out, two exchanges):
#     all-to-all (rank s gets the slice t in [s m/W, (s+1) m/W) of every rank) -> m/W transforms of W points across the
#     ranks' values (mzk_ntt_columns_dev: one trip over the data) -> all-to-all (rank k1 gets y[k1][all t]) -> the m-point transform of y[k1][t] * (w^k1)^t, which IS
#     fast_coset_evaluate with offset w^k1 and generator w^W (ntt.rs:254-269: the twiddle costs no pass of its own).
# "dit" (cyclic in -> contiguous out, two exchanges), from j = j2 W + j1, k = k1 m + k2:
#     X[k1 m + k2] = sum_j1 w_W^(j1 k1) * [ w^(j1 k2) * sum_j2 x[j2 W + j1] w_m^(j2 k2) ]
#     local m-point transform -> Polynomial::scale by w^j1 (polynomial.rs:167-174) -> all-to-all -> W-point transforms
#     -> all-to-all.
# Contiguous in AND out costs a third all-to-all (dif + redistribution of the cyclic result); a forward transform followed
# by an inverse one (polynomial products, quotients) should keep the cyclic layout in between: 2 + 2 exchanges, not 3 + 3.
# Every exchange moves (W-1)/W of the rank's n/W elements, each to a different peer: on xGMI all seven links of a GPU carry
# one seventh of it at the same time.  The inverse transform uses w^-1 in the same schedule, with W^-1 and m^-1 supplied by
# the local inverse transforms (mzk_ntt_batch_dev / mzk_ntt_dev, inverse = 1).
# The local compute is injected (`ops`), like the MSM's: DeviceOps below is the GPU path (C ABI calls on HBM-resident
# buffers + torch for the two transposes and the collective); the gloo test runs the same schedule with the oracle.

def _ilog2(n):
    l = n.bit_length() - 1
    if n <= 0 or (1 << l) != n:
        raise ValueError("cannot compute ntt of non-power-of-two sequence")
    return l


def ntt_sharded_steps(modulus, log2n, world, root, inverse, layout_in, layout_out):
    """The schedule as a list of steps: None = all-to-all of W equal chunks, else fn(ops, rank, buf) -> buf."""
    n, W = 1 << log2n, world
    _ilog2(W)
    if W * W > n:
        raise ValueError("sharded transform needs world^2 <= n")
    if (layout_in, layout_out) not in (("contiguous", "contiguous"), ("contiguous", "cyclic"), ("cyclic", "contiguous")):
        raise ValueError("layouts: contiguous->contiguous, contiguous->cyclic or cyclic->contiguous")
    m = n // W
    p = modulus
    w = pow(root, p - 2, p) if inverse else root           # the root the sums run over
    root_W, root_m = pow(root, m, p), pow(root, W, p)      # forward roots of the W- and m-point transforms (as the callers pass them)
    if W == 1:
        local = lambda ops, r, b: ops.ntt(b, m, root_m, inverse)
        return [None, local] if FORCE_COLLECTIVES else [local]        # (forced: the exchange of one chunk with oneself, then the transform)

    def across(ops, r, b):            # [a][t'] -> m/W transforms over a -> [k1][t']
        fused = getattr(ops, "ntt_columns", None)
        out = fused(b, W, m // W, root_W, inverse) if fused else None       # one trip over the data where the backend has it
        if out is not None:
            return out
        b = ops.transpose(b, W, m // W)
        b = ops.ntt_rows(b, m // W, W, root_W, inverse)
        return ops.transpose(b, m // W, W)

    def local_dif(ops, r, b):
        if not inverse:
            return ops.lde(b, m, pow(w, r, p), root_m)
        return ops.ntt(ops.scale(b, m, pow(w, r, p)), m, root_m, True)

    def local_dit(ops, r, b):
        return ops.scale(ops.ntt(b, m, root_m, inverse), m, pow(w, r, p))

    def interleave(ops, r, b):        # [k1][k2'] -> k2' W + k1
        return ops.transpose(b, W, m // W)

    if layout_in == "cyclic":
        return [local_dit, None, across, None]
    steps = [None, across, None, local_dif]
    if layout_out == "contiguous":
        steps += [None, interleave]
    return steps


class DeviceOps:
    """Local steps on the GPU: flat int64 device tensors of (elements x limbs), C ABI calls on torch's current stream."""

    def __init__(self, fid):
        import ctypes
        import myzkp_amd as mz
        self.ct, self.mz, self.fid, self.nl, self.L = ctypes, mz, fid, mz.LIMBS[fid], mz.lib()

    def _st(self):
        return self.ct.c_void_p(torch.cuda.current_stream().cuda_stream)

    def _one(self, v):
        a = self.mz.to_limbs([v], self.nl)
        return a, a.ctypes.data_as(self.ct.c_void_p)

    def _ok(self, rc):
        if rc != 0:
            raise self.mz.MzkError(rc, self.L.mzk_last_error().decode())

    def ntt(self, b, m, root, inverse):        # out of place: the caller's part is never overwritten
        keep, r = self._one(root)
        out = torch.empty_like(b)
        self._ok(self.L.mzk_ntt_dev(self.fid, r, self.ct.c_void_p(b.data_ptr()), self.ct.c_void_p(out.data_ptr()), self.ct.c_size_t(m),
                                    int(bool(inverse)), self._st()))
        return out

    def ntt_rows(self, b, rows, n, root, inverse):
        keep, r = self._one(root)
        p = self.ct.c_void_p(b.data_ptr())
        self._ok(self.L.mzk_ntt_batch_dev(self.fid, r, p, p, self.ct.c_size_t(n), self.ct.c_size_t(rows), int(bool(inverse)), self._st()))
        return b

    def ntt_columns(self, b, W, cols, root, inverse):
        if W > 16:
            return None                      # the schedule falls back to transpose + row transforms + transpose
        keep, r = self._one(root)
        out = torch.empty_like(b)
        self._ok(self.L.mzk_ntt_columns_dev(self.fid, r, self.ct.c_void_p(b.data_ptr()), self.ct.c_void_p(out.data_ptr()), self.ct.c_size_t(W),
                                            self.ct.c_size_t(cols), int(bool(inverse)), self._st()))
        return out

    def lde(self, b, m, offset, generator):
        k1, o = self._one(offset)
        k2, g = self._one(generator)
        out = torch.empty_like(b)
        self._ok(self.L.mzk_coset_lde_dev(self.fid, self.ct.c_void_p(b.data_ptr()), self.ct.c_size_t(m), o, g, self.ct.c_void_p(out.data_ptr()),
                                          self.ct.c_size_t(m), self._st()))
        return out

    def scale(self, b, m, ratio):
        keep, r = self._one(ratio)
        p = self.ct.c_void_p(b.data_ptr())
        self._ok(self.L.mzk_poly_scale_dev(self.fid, p, self.ct.c_size_t(m), r, None, p, self._st()))
        return b

    def transpose(self, b, rows, cols):
        return b.view(rows, cols, self.nl).transpose(0, 1).contiguous().view(-1)

    def chunks(self, b, W):
        return list(b.chunk(W))

    def cat(self, parts):
        return torch.cat(parts)

    def all_to_all(self, b, group=None):
        out = torch.empty_like(b)
        dist.all_to_all_single(out, b.contiguous(), group=group)
        return out


def ntt_sharded(x_local, modulus, log2n, root, ops, inverse=False, layout_in="contiguous", layout_out="contiguous", group=None):
    """This rank's part of ONE n-point transform (ntt.rs:7-64) whose vector is spread over the ranks of `group`."""
    live = dist.is_available() and dist.is_initialized()
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if live else (0, 1)
    buf = x_local
    for step in ntt_sharded_steps(modulus, log2n, world, root, inverse, layout_in, layout_out):
        if step is None:
            buf = ops.all_to_all(buf, group) if live else buf
        else:
            buf = step(ops, rank, buf)
    return buf


def ntt_sharded_simulate(xs, modulus, log2n, root, ops, inverse=False, layout_in="contiguous", layout_out="contiguous"):
    """The same schedule for ALL ranks inside one process (xs[r] = rank r's part): the exchanges become slicing.  Tests, and a
    one-GPU rehearsal of the multi-GPU path."""
    W = len(xs)
    bufs = list(xs)
    for step in ntt_sharded_steps(modulus, log2n, W, root, inverse, layout_in, layout_out):
        if step is None:
            parts = [ops.chunks(b, W) for b in bufs]
            bufs = [ops.cat([parts[src][dst] for src in range(W)]) for dst in range(W)]
        else:
            bufs = [step(ops, r, bufs[r]) for r in range(W)]
    return bufs

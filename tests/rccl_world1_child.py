"""Child process of tests/test_gpu_rccl_world1.py: a process group of ONE rank over the `nccl` backend (= RCCL on ROCm)
on cuda:0, with the world-1 shortcuts of myzkp_amd/sharded.py switched off, so that every collective the multi-GPU path
issues really goes through RCCL on a one-GPU box:

  * dist.all_gather_into_tensor of the 16 x int64 XYZZ partial that mzk_msm_g1_bn254_partial_dev left in HBM, then
    mzk_g1_fold_partials_dev on the gathered tensor (contiguous shards -> one 128-byte partial per rank ->
    gather -> fold); the point must equal the oracle's Pippenger on the same streams (polynomial.rs:156-165);
  * the same for the KZG commit against an SRS handle (mzk_kzg_commit_srs_dev, partial output);
  * DeviceOps.all_to_all (dist.all_to_all_single on int64 device tensors) of a 2^16-element part, against itself, and the
    forced one-rank schedule of ntt_sharded against the plain transform;
  * the ordering between torch's NCCL stream and the stream handed to the C ABI: the partial is produced, gathered and folded
    back to back with NO host synchronisation in between, 20 times, on fresh scalars each time.

Prints one JSON line; exits non-zero on any failure.  The parent never touches HIP."""
import ctypes, json, os, socket, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in os.environ:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    import myzkp_amd as mz
    from myzkp_amd import sharded
    import orc

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", world_size=1, rank=0, device_id=dev)
    sharded.FORCE_COLLECTIVES = True
    mz.init(0)
    L = mz.lib()
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rec = {"backend": dist.get_backend(), "world_size": dist.get_world_size()}

    def check(rc):
        if rc != 0:
            raise RuntimeError(L.mzk_last_error().decode())

    def dptr(t):
        return ctypes.c_void_p(t.data_ptr())

    # ---- MSM: partial -> all_gather_into_tensor -> fold, against the oracle
    n = 1 << 14
    sc = torch.empty(n * 4, dtype=torch.int64, device=dev)
    pt = torch.empty(n * 8, dtype=torch.int64, device=dev)
    check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(611), ctypes.c_size_t(n), dptr(sc), st))
    check(L.mzk_synth_g1_points_dev(ctypes.c_uint64(612), ctypes.c_size_t(n), dptr(pt), st))
    partial = torch.zeros(16, dtype=torch.int64, device=dev)
    out = torch.zeros(8, dtype=torch.int64, device=dev)
    check(L.mzk_msm_g1_bn254_partial_dev(dptr(sc), dptr(pt), ctypes.c_size_t(n), dptr(partial), st))
    recs = sharded.all_gather_partials(partial)
    assert recs.shape == (1, 16) and recs.data_ptr() != partial.data_ptr(), "the gather did not run"
    check(L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(1), dptr(out), st))
    torch.cuda.synchronize()
    s_h = orc.synth_vector(orc.FR, 611, n)
    p_h = orc.synth_points(612, n)
    want = orc.msm_fast(s_h, p_h)
    got = mz.array_to_points(out.cpu().numpy().view(np.uint64))[0]
    rec["msm_gather_fold_equals_oracle"] = bool(got == want)

    # ---- the KZG commit against a handle, same exchange
    h = ctypes.c_void_p()
    check(L.mzk_srs_from_device(dptr(pt), ctypes.c_size_t(n), ctypes.byref(h), st))
    check(L.mzk_kzg_commit_srs_dev(h, dptr(sc), ctypes.c_size_t(n), dptr(partial), ctypes.c_int(1), st))
    recs = sharded.all_gather_partials(partial)
    check(L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(1), dptr(out), st))
    torch.cuda.synchronize()
    rec["commit_gather_fold_equals_oracle"] = bool(mz.array_to_points(out.cpu().numpy().view(np.uint64))[0] == want)

    # ---- stream ordering: produce / gather / fold back to back without a host synchronisation, fresh scalars every time;
    # every result against the same chain with a synchronize after every call
    outs = torch.zeros(20, 8, dtype=torch.int64, device=dev)
    scs = torch.empty(20, n * 4, dtype=torch.int64, device=dev)
    for k in range(20):
        check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(700 + k), ctypes.c_size_t(n), dptr(scs[k]), st))
    for k in range(20):
        check(L.mzk_kzg_commit_srs_dev(h, dptr(scs[k]), ctypes.c_size_t(n), dptr(partial), ctypes.c_int(1), st))
        recs = sharded.all_gather_partials(partial)
        check(L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(1), dptr(outs[k]), st))
    torch.cuda.synchronize()
    ok = True
    one = torch.zeros(8, dtype=torch.int64, device=dev)
    for k in range(20):
        check(L.mzk_kzg_commit_srs_dev(h, dptr(scs[k]), ctypes.c_size_t(n), dptr(one), ctypes.c_int(0), st))
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(one, outs[k]))
    rec["unsynchronised_chain_equals_synchronised"] = ok
    L.mzk_srs_free(h)

    # ---- all_to_all_single on an int64 device part, and the forced one-rank schedule of the sharded transform
    lg = 16
    ops = sharded.DeviceOps(mz.FIELD_FR)
    x = torch.empty((1 << lg) * 4, dtype=torch.int64, device=dev)
    check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(613), ctypes.c_size_t(1 << lg), dptr(x), st))
    y = ops.all_to_all(x)
    torch.cuda.synchronize()
    rec["all_to_all_part_equals_itself"] = bool(y.data_ptr() != x.data_ptr() and torch.equal(x, y))
    w = mz.root_of_unity(mz.FIELD_FR, lg)
    got_t = sharded.ntt_sharded(x, mz.MODULUS[mz.FIELD_FR], lg, w, ops)
    want_t = torch.empty_like(x)
    rt = mz.to_limbs([w], 4)
    check(L.mzk_ntt_dev(mz.FIELD_FR, rt.ctypes.data_as(ctypes.c_void_p), dptr(x), dptr(want_t), ctypes.c_size_t(1 << lg), 0, st))
    torch.cuda.synchronize()
    rec["forced_sharded_transform_equals_plain"] = bool(torch.equal(got_t, want_t))
    rc_o, ref = orc.ntt_fast(orc.FR, w, x.cpu().numpy().view(np.uint64).reshape(-1, 4))
    rec["forced_sharded_transform_equals_oracle"] = bool(rc_o == 0 and np.array_equal(ref, got_t.cpu().numpy().view(np.uint64).reshape(-1, 4)))

    maps = open("/proc/self/maps").read()
    rec["librccl_mapped"] = "librccl" in maps
    rec["libmzk_hip_mapped"] = "libmzk_hip" in maps
    dist.barrier()
    dist.destroy_process_group()
    good = all(v is True for k, v in rec.items() if k not in ("backend", "world_size")) and rec["backend"] == "nccl"
    print(json.dumps(rec), flush=True)
    return 0 if good else 1


if __name__ == "__main__":
    sys.exit(main())

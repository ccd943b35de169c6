#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MyZKP MSM / NTT prover path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 20] [--skip-cpu]

One JSON line on stdout (rank 0).  A "step" is one pass of the hot path over one batch of synthetic
input already resident in HBM:
  * primary (`value`): one BN254 G1 MSM of 2^log2n (scalar, point) pairs PER GPU (BASELINE.json
    configs[2]; weak scaling: the N-GPU job is one MSM of N * 2^log2n pairs, contiguous shards, one
    all-gather of N 128-byte partials over RCCL, local fold -- configs[3] shape);
  * `ntt`: one forward radix-2 NTT of 2^log2n BN254-Fr elements per GPU (configs[1]; the transform
    does not shard without an all-to-all, so N GPUs run N independent transforms -- "replicas").
Both are checked bit-for-bit against the CPU oracle before timing.  `roofline` prices the dominant
kernel against HBM bandwidth as BASELINE.md section 4 defines it (MSM: 96 B per pair, NTT: 2*32 B per
element); `alu` adds the integer-multiply roofline that actually binds.
"""
import argparse, ctypes, json, os, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4D595A4B50  # "MYZKP"
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MAD_PEAK_PER_S = 256 * 64 * 2.4e9   # v_mad_u64_u32 is half rate: 64 lanes/clk/CU (profiles/r01_ubench_instr_rates.txt)


METRIC = "G1 MSM pairs/sec + NTT elems/sec at 2^20 and 2^24; bit-exact vs CPU"


def fail_line(n_gpus, msg, **more):
    """One JSON line that says why there is no measurement (never a silent 1-rank run), then a non-zero exit."""
    print(json.dumps({"metric": METRIC, "value": None, "unit": "pairs/s", "n_gpus": n_gpus, "error": msg, **more}), flush=True)
    return 2


def recorded_traffic(kernel_prefix, section="KZG commit 2^20"):
    """FETCH_SIZE + WRITE_SIZE per launch (bytes) of a kernel from the newest profiles/r*_hbm_traffic_pmc.txt that names it in
    the given section (`== KZG commit 2^20 ...`, `== generic MSM 2^20 ...`; written by tools/timing/pmc_summary.py under
    rocprofv3 --pmc; the bench itself never runs under the profiler).  None if no such record exists."""
    import glob, re
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.txt"))):
        inside = False
        for line in open(path):
            if line.startswith("== "):
                inside = section in line
                continue
            if not inside:
                continue
            m = re.match(r"(.*?)\s+launches=.*FETCH_SIZE avg=\s*([0-9.]+) KiB\s+WRITE_SIZE avg=\s*([0-9.]+) KiB", line)
            if m and kernel_prefix in m.group(1):
                best = {"bytes": int((float(m.group(2)) + float(m.group(3))) * 1024),
                        "source": "profiles/%s (recorded rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, separate runs, raw counters; not collected by this run).  Calibration as the microarchitecture guide asks (profiles/r02o_hbm_traffic_pmc.txt, known byte counts): requests of 128 contiguous bytes are tallied at 64 (BN254 NTT passes: 32 MB of data + 32 MB of twiddles read -> FETCH_SIZE 33.0 MB, the guide's x2), reads in 64-byte runs are counted 1:1 (M128 strided pass: 16 + 16 MB read -> 32.9 MB); k_seg_accumulate gathers 64-byte table rows, so its raw figure stands" % os.path.basename(path)}
                break       # first match per file = the KZG-commit section
    return best


def recorded_valu_instructions(kernel_prefix, section="KZG commit 2^20"):
    """Wave-level VALU instructions per launch (SQ_INSTS_VALU, mean) of a kernel from the newest profiles/r*_sq_counters.txt that names it in
    the given section (written by tools/timing/pmc_sq_summary.py under rocprofv3 --pmc, a run of its own).  None if no such record exists."""
    import glob, re
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_sq_counters.txt"))):
        inside, in_kernel = False, False
        for line in open(path):
            if line.startswith("== "):
                inside, in_kernel = section in line, False
                continue
            if not inside:
                continue
            if not line.startswith(" "):
                in_kernel = kernel_prefix in line
                continue
            m = re.match(r"\s+SQ_INSTS_VALU\s+mean\s+([0-9.]+)", line)
            if m and in_kernel:
                best = {"instructions": float(m.group(1)), "source": "profiles/%s" % os.path.basename(path)}
                in_kernel = False
    return best


def recorded_ntt_traffic(field_tag):
    """Bytes per 2^20-point transform from the newest profiles/r*_hbm_traffic_pmc.txt that has a section for this field
    (`== NTT <field_tag> 2^20`: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/timing/ntt_only.py): the sum over
    the transform's kernels, raw and corrected.  Correction as calibrated on known byte counts (profiles/r02o_hbm_traffic_pmc.txt,
    the microarchitecture guide's rule): FETCH_SIZE tallies a 128-byte request at 64, so streams read in runs of >= 128 bytes
    count double -- every BN254 pass, and the contiguous rows of the M128 last pass; the M128 strided pass reads 64-byte runs
    (4 columns x 16 bytes), counted 1:1.  WRITE_SIZE is exact.  None if no such record exists."""
    import glob, re
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_hbm_traffic_pmc.txt"))):
        inside, raw, corr, kernels = False, 0.0, 0.0, []
        for line in open(path):
            if line.startswith("== "):
                inside = ("NTT %s 2^20" % field_tag) in line
                continue
            m = re.match(r"(.*?)\s+launches=.*FETCH_SIZE avg=\s*([0-9.]+) KiB\s+WRITE_SIZE avg=\s*([0-9.]+) KiB", line)
            if inside and m and "k_ntt_" in m.group(1):
                fe, wr = float(m.group(2)) * 1024, float(m.group(3)) * 1024
                double = not (field_tag == "M128" and "k_ntt_strided" in m.group(1))
                raw += fe + wr
                corr += (2 * fe if double else fe) + wr
                kernels.append(m.group(1).strip()[:40])
        if kernels:
            best = {"bytes": int(corr), "raw_bytes": int(raw), "kernels": kernels,
                    "source": "profiles/%s (recorded rocprofv3 --pmc passes of tools/timing/ntt_only.py; FETCH_SIZE doubled for streams read in runs of >= 128 bytes, see recorded_ntt_traffic; not collected by this run)" % os.path.basename(path)}
    return best


def visible_gpus_without_hip(kfd_root="/sys/class/kfd/kfd/topology/nodes", environ=None):
    """GPUs this process could use, counted WITHOUT touching the HIP runtime (the launcher parent must stay GPU-free by
    construction): KFD topology nodes with simd_count > 0 (CPU nodes have 0), then the usual filters -- ROCR_VISIBLE_DEVICES
    restricts the physical devices, HIP_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES the ones of those HIP exposes (a comma list of
    ordinals or UUIDs; an empty string or -1 hides all; entries after the first invalid ordinal are dropped, as the runtime does).
    Returns None when the topology cannot be read (then the ranks' own check inside rank 0 decides)."""
    environ = os.environ if environ is None else environ
    try:
        nodes = sorted(os.listdir(kfd_root), key=lambda x: int(x) if x.isdigit() else 1 << 30)
    except OSError:
        return None
    count = 0
    for nd in nodes:
        try:
            props = dict(l.split(None, 1) for l in open(os.path.join(kfd_root, nd, "properties")) if " " in l.strip())
            if int(props.get("simd_count", "0").strip()) > 0:
                count += 1
        except (OSError, ValueError):
            continue
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        if var not in environ:
            continue
        kept = 0
        for tok in environ[var].split(","):
            tok = tok.strip()
            if tok.isdigit() and int(tok) < count:
                kept += 1
            elif tok.upper().startswith("GPU-"):          # a UUID: cannot be resolved without the runtime -- count it
                kept += 1
            else:
                break
        count = min(count, kept)
    return count


def launch_ranks(n_gpus, argv):
    """`python bench.py --gpus N` with no WORLD_SIZE in the environment: start the N ranks (one process per GPU, RCCL) as a
    child `python -m torch.distributed.run` and hand on rank 0's JSON line and the exit code.  This process never touches the
    HIP runtime -- the GPUs are counted from the KFD topology in sysfs (visible_gpus_without_hip); torch's own count is the
    cross-check INSIDE the ranks -- so the children are ordinary child processes, not an exec from a GPU process."""
    import socket, subprocess
    shared = os.environ.get("MZK_BENCH_SHARED_GPU_TEST") == "1"
    visible = visible_gpus_without_hip()
    # None = the KFD topology is not readable here (a container without /sys/class/kfd): do not guess -- start the ranks, whose own
    # check (torch's device count inside the rank processes) refuses with a JSON line if there are fewer GPUs than ranks
    if visible is not None and visible < n_gpus and not shared:
        return fail_line(n_gpus, "--gpus %d asked for but only %d GPU(s) visible; refusing to run fewer ranks than asked" % (n_gpus, visible),
                         visible_devices=visible)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MZK_BENCH_LAUNCHED_BY"] = "bench.py launch_ranks (python -m torch.distributed.run child)"
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_gpus)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    got_line = False
    for line in proc.stdout:
        st = line.strip()
        if st.startswith("{") and '"metric"' in st:
            got_line = True
        sys.stdout.write(line)
        sys.stdout.flush()
    rc = proc.wait()
    if not got_line:
        fail_line(n_gpus, "the %d-rank job exited with code %d without printing a result line" % (n_gpus, rc), launcher_cmd=" ".join(cmd))
        return rc or 2
    return rc


def main():
    T_START = time.perf_counter()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--log2n", type=int, default=20)
    ap.add_argument("--skip-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--settle-s", type=float, default=0.5, help="untimed settle time per leg before the W warm-up steps")
    ap.add_argument("--e2e-log2n", type=int, default=22, help="degree of the end-to-end KZG run (0 = skip)")
    ap.add_argument("--strong-log2n", type=int, default=24,
                    help="total size of the fixed-size MSM split over all ranks (BASELINE configs[3]; 0 = skip)")
    ap.add_argument("--strong-ntt-log2n", type=int, default=24, help="size of the ONE transform sharded over all ranks (0 = skip)")
    ap.add_argument("--extra-sizes", type=str, default="24", help="comma list of extra log2 sizes timed once each (rank 0 view)")
    ap.add_argument("--no-two-in-flight", action="store_true",
                    help="skip the two-commits-in-flight leg (profiler runs: overlapped kernels would distort the per-kernel averages)")
    ap.add_argument("--inproc-devices", type=str, default="",
                    help="comma list of device ordinals: additionally run the fixed-size MSM (configs[3]) from THIS one process over "
                         "those GPUs through the C ABI's mzk_*_multi entry points (no torch.distributed); single-process runs only")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher BEFORE anything touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import numpy as np
    import torch
    import torch.distributed as dist
    import myzkp_amd as mz
    from myzkp_amd import sharded

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N>1 code path on a ONE-GPU box (the pool's boxes have one): all ranks share cuda:0 and the
    # exchange goes over gloo through host memory.  Timings of such a run mean nothing and the line says so.
    shared_gpu_test = world > 1 and os.environ.get("MZK_BENCH_SHARED_GPU_TEST") == "1"
    if shared_gpu_test:
        local_rank = 0
    if world != args.gpus:
        if rank == 0:
            fail_line(args.gpus, "WORLD_SIZE=%d but --gpus %d: launch `python bench.py --gpus N` (it starts the ranks itself) or "
                                 "torch.distributed.run --nproc-per-node N bench.py --gpus N" % (world, args.gpus))
        sys.exit(2)
    if world > 1 and not shared_gpu_test and torch.cuda.device_count() < world:
        if rank == 0:
            fail_line(args.gpus, "%d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()), visible_devices=torch.cuda.device_count())
        sys.exit(2)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(local_rank)
        if shared_gpu_test:
            dist.init_process_group("gloo")
            _real_gather = sharded.all_gather_partials
            sharded.all_gather_partials = lambda t: _real_gather(t.cpu()).to(t.device)
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    import faulthandler
    if os.environ.get("MZK_BENCH_WATCHDOG_S"):      # dump every thread's stack if the run is still going after that long
        faulthandler.dump_traceback_later(float(os.environ["MZK_BENCH_WATCHDOG_S"]), repeat=True, file=sys.stderr)

    def progress(msg):
        if os.environ.get("MZK_BENCH_VERBOSE") == "1":
            print("[bench rank %d +%.1fs] %s" % (rank, time.perf_counter() - T_START, msg), file=sys.stderr, flush=True)

    dev = torch.device("cuda", local_rank)
    mz.init_devices([local_rank] * 4)     # context 0: every leg; contexts 1..3 (same GPU): further commits in flight, see below
    L = mz.lib()
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

    def check(rc):
        if rc != 0:
            raise RuntimeError(L.mzk_last_error().decode())

    def dptr(t):
        return ctypes.c_void_p(t.data_ptr())

    def barrier_sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device="cpu" if shared_gpu_test else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    n = 1 << args.log2n
    # ------------------------------------------------------------------ synthetic inputs, resident in HBM
    # global problem = world * n pairs; rank g owns indices [g n, (g+1) n) of the global streams
    def synth_shard(nn, r):
        sc = torch.empty(nn * 4, dtype=torch.int64, device=dev)
        pt = torch.empty(nn * 8, dtype=torch.int64, device=dev)
        # distinct streams per rank: seed offset keeps shards independent and reproducible
        check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 1000003 * r), ctypes.c_size_t(nn), dptr(sc), stream))
        check(L.mzk_synth_g1_points_dev(ctypes.c_uint64(SEED + 7 + 1000003 * r), ctypes.c_size_t(nn), dptr(pt), stream))
        return sc, pt

    progress("process group up; generating inputs")
    scalars, points = synth_shard(n, rank)
    ntt_in = torch.empty(n * 4, dtype=torch.int64, device=dev)
    ntt_out = torch.empty(n * 4, dtype=torch.int64, device=dev)
    check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 99 + rank), ctypes.c_size_t(n), dptr(ntt_in), stream))
    root = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, args.log2n)], 4)
    partial = torch.zeros(16, dtype=torch.int64, device=dev)
    result = torch.zeros(8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()

    def msm_step():
        if world == 1:   # nothing to exchange: the kernel chain ends in the affine point
            check(L.mzk_msm_g1_bn254_dev(dptr(scalars), dptr(points), ctypes.c_size_t(n), dptr(result), stream))
            return
        check(L.mzk_msm_g1_bn254_partial_dev(dptr(scalars), dptr(points), ctypes.c_size_t(n), dptr(partial), stream))
        recs = sharded.all_gather_partials(partial)
        check(L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(recs.shape[0]), dptr(result), stream))

    # device-resident SRS built from the same points (tables T[w][i] = 2^(16 w) P_i; one-time, untimed)
    progress("building SRS handle")
    class _Handle:
        _h = None
    srs = _Handle()
    srs_build_ms = None
    for attempt in range(2):            # the first build also grows the workspace (hipMalloc): time the second
        hh = ctypes.c_void_p()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(L.mzk_srs_from_device(dptr(points), ctypes.c_size_t(n), ctypes.byref(hh), stream))
        torch.cuda.synchronize()
        srs_build_ms = (time.perf_counter() - t0) * 1e3
        if attempt == 0:
            L.mzk_srs_free(hh)
    srs._h = hh
    srs_window_bits = int(L.mzk_srs_window_bits(srs._h))       # the library's default width for an SRS of this size (msm_srs_window_bits)
    srs_table_windows = 254 // srs_window_bits + 1
    progress("SRS handle built")
    # The HIP runtime stalls once for 35-45 ms a few thousand dispatches into a process (measured: one stall in 120 000
    # launches, at dispatch ~3500; tools note in DESIGN.md section 8) -- get past it before anything is timed.
    tiny = torch.empty(64 * 4, dtype=torch.int64, device=dev)
    for _ in range(6000):
        L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(1), ctypes.c_size_t(64), dptr(tiny), stream)
    torch.cuda.synchronize()
    del tiny
    result_srs = torch.zeros(8, dtype=torch.int64, device=dev)

    def srs_step():
        if world == 1:
            check(L.mzk_kzg_commit_srs_dev(srs._h, dptr(scalars), ctypes.c_size_t(n), dptr(result_srs), ctypes.c_int(0), stream))
            return
        check(L.mzk_kzg_commit_srs_dev(srs._h, dptr(scalars), ctypes.c_size_t(n), dptr(partial), ctypes.c_int(1), stream))
        recs = sharded.all_gather_partials(partial)
        check(L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(recs.shape[0]), dptr(result_srs), stream))

    # M128 (the STARK field): forward NTT of 2^log2n elements, and the low-degree extension
    # fast_coset_evaluate of a degree-2^(log2n-2) polynomial onto a 2^log2n coset (blow-up 4)
    m_in = torch.empty(n * 2, dtype=torch.int64, device=dev)
    m_out = torch.empty(n * 2, dtype=torch.int64, device=dev)
    check(L.mzk_synth_field_dev(mz.FIELD_M128, ctypes.c_uint64(SEED + 199 + rank), ctypes.c_size_t(n), dptr(m_in), stream))
    m_root = mz.to_limbs([mz.root_of_unity(mz.FIELD_M128, args.log2n)], 2)
    m_off = mz.to_limbs([85408008396924667383611388730472331217], 2)   # fast_stark.rs:573-616: offset = generator

    def ntt_m128_step():
        check(L.mzk_ntt_dev(mz.FIELD_M128, m_root.ctypes.data_as(ctypes.c_void_p), dptr(m_in), dptr(m_out), ctypes.c_size_t(n), 0, stream))

    def lde_m128_step():
        check(L.mzk_coset_lde_dev(mz.FIELD_M128, dptr(m_in), ctypes.c_size_t(n // 4), m_off.ctypes.data_as(ctypes.c_void_p),
                                  m_root.ctypes.data_as(ctypes.c_void_p), dptr(m_out), ctypes.c_size_t(n), stream))

    def merkle_m128_step():
        check(L.mzk_merkle_commit_field_dev(mz.FIELD_M128, dptr(m_out), ctypes.c_size_t(n), mk_root, ctypes.c_size_t(48), ctypes.byref(mk_len), stream))

    mk_root, mk_len = (ctypes.c_uint8 * 48)(), ctypes.c_size_t()

    def ntt_step():
        check(L.mzk_ntt_dev(mz.FIELD_FR, root.ctypes.data_as(ctypes.c_void_p), dptr(ntt_in), dptr(ntt_out), ctypes.c_size_t(n), 0, stream))

    # ------------------------------------------------------------------ parity before timing (bit-exact vs CPU oracle)
    # Every rank checks ITS OWN shard on its share of the host cores (GPU shard result == oracle Pippenger on the
    # same synthetic streams), then rank 0 checks that the folded N-GPU result == the CPU sum of the N shard points.
    parity = {}
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc
    threads = max(1, orc.usable_threads(64) // world)     # the GPU boxes report 256 CPUs and schedule ~16: more threads are slower
    s_cpu = orc.synth_vector(orc.FR, SEED + 1000003 * rank, n, threads)
    p_cpu = orc.synth_points(SEED + 7 + 1000003 * rank, n, threads)
    ok_inputs = bool(np.array_equal(s_cpu.view(np.int64).reshape(-1), scalars.cpu().numpy()) and
                     np.array_equal(p_cpu.view(np.int64).reshape(-1), points.cpu().numpy()))
    assert ok_inputs, "GPU synthetic inputs != oracle streams"
    shard_want = orc.msm_fast(s_cpu, p_cpu, threads)
    shard_out = torch.zeros(8, dtype=torch.int64, device=dev)
    check(L.mzk_msm_g1_bn254_dev(dptr(scalars), dptr(points), ctypes.c_size_t(n), dptr(shard_out), stream))
    torch.cuda.synchronize()
    shard_got = mz.array_to_points(shard_out.cpu().numpy().view(np.uint64))[0]
    assert shard_got == shard_want, "rank %d: MSM shard mismatch vs CPU oracle" % rank
    check(L.mzk_kzg_commit_srs_dev(srs._h, dptr(scalars), ctypes.c_size_t(n), dptr(shard_out), ctypes.c_int(0), stream))
    torch.cuda.synchronize()
    assert mz.array_to_points(shard_out.cpu().numpy().view(np.uint64))[0] == shard_want, "rank %d: SRS commit mismatch" % rank
    progress("shard parity ok; exchanging shard points")
    shard_pts = sharded.all_gather_partials(shard_out)          # (world, 8) affine shard results
    progress("exchange done")
    msm_step(); srs_step()
    torch.cuda.synchronize()
    if rank == 0:
        want = (0, 0)
        for r in range(world):
            want = orc.ec_add(0, want, mz.array_to_points(shard_pts[r].cpu().numpy().view(np.uint64))[0])
        got = mz.array_to_points(result.cpu().numpy().view(np.uint64))[0]
        got2 = mz.array_to_points(result_srs.cpu().numpy().view(np.uint64))[0]
        parity["msm_bit_exact_vs_cpu"] = bool(got == want)
        parity["kzg_commit_srs_bit_exact_vs_cpu"] = bool(got2 == want)
        assert got == want and got2 == want, "folded N-GPU MSM mismatch vs CPU"
    progress("folded MSM parity ok")
    ntt_step()
    torch.cuda.synchronize()
    v_cpu = orc.synth_vector(orc.FR, SEED + 99 + rank, n, threads)
    rc, want_ntt = orc.ntt_fast(orc.FR, mz.from_limbs(root)[0], v_cpu, threads=threads)
    ok = rc == 0 and np.array_equal(want_ntt.view(np.int64).reshape(-1), ntt_out.cpu().numpy())
    assert ok, "rank %d: NTT mismatch vs CPU oracle" % rank
    parity["ntt_bit_exact_vs_cpu"] = bool(ok)
    lde_m128_step()
    torch.cuda.synchronize()
    c_cpu = orc.synth_vector(orc.M128, SEED + 199 + rank, n // 4, threads)
    off = 85408008396924667383611388730472331217
    acc, scaled = 1, []
    for x in orc.from_limbs(c_cpu[:4096]):
        scaled.append(x * acc % orc.P_M128); acc = acc * off % orc.P_M128
    # full-size check: scale on the CPU with a running power, then the oracle's iterative NTT
    sc_all = np.zeros((n, 2), dtype=np.uint64)
    vals = orc.from_limbs(c_cpu)
    acc = 1
    for i, x in enumerate(vals):
        vals[i] = x * acc % orc.P_M128
        acc = acc * off % orc.P_M128
    sc_all[: n // 4] = orc.to_limbs(vals, 2)
    rc, want_lde = orc.ntt_fast(orc.M128, mz.from_limbs(m_root)[0], sc_all, threads=threads)
    ok_lde = rc == 0 and np.array_equal(want_lde.view(np.int64).reshape(-1), m_out.cpu().numpy())
    assert ok_lde, "rank %d: M128 coset LDE mismatch vs CPU oracle" % rank
    parity["m128_coset_lde_bit_exact_vs_cpu"] = bool(ok_lde)
    # Merkle root of that LDE codeword (fri.rs:160-166) vs the oracle's literal recursion over the same leaves
    mroot, mlen = (ctypes.c_uint8 * 48)(), ctypes.c_size_t()
    check(L.mzk_merkle_commit_field_dev(mz.FIELD_M128, dptr(m_out), ctypes.c_size_t(n), mroot, ctypes.c_size_t(48), ctypes.byref(mlen), stream))
    ok_merkle = bytes(mroot[:mlen.value]) == orc.merkle_commit_field_ref(orc.M128, want_lde)
    assert ok_merkle, "rank %d: Merkle root mismatch vs CPU oracle" % rank
    parity["m128_codeword_merkle_root_bit_exact_vs_cpu"] = bool(ok_merkle)
    del s_cpu, p_cpu, v_cpu, want_ntt, c_cpu, sc_all, want_lde, vals

    progress("parity done; timing")
    # ------------------------------------------------------------------ timed regions
    N_PHASES = 13
    PH_ACC, PH_NTT_TOTAL, PH_MERKLE = 2, 12, 10

    def timed(step, K, W, price_mask=0, priced_in_timed_region=True):
        """The contract's timed region: W warm-up steps, then EXACTLY K steps between barrier + synchronize; returns its wall time
        (max over ranks) and the kernel-phase averages of two further, untimed passes.
        priced_in_timed_region (the headline and the generic MSM): the one priced kernel (price_mask) carries its HIP-event pair INSIDE the
        timed region, as the roofline contract asks (0.1 % of a 1.4-ms step).  False (the short sub-legs: transforms, LDE, Merkle): the
        timed region carries NO event -- the pair's two marker packets are ~5-8 us, 7-14 % of such a step -- so `ms_per_step` and
        `value` ARE the contract's timed region, and the priced kernel group is measured in a pass of its own
        (`*_in_priced_pass`, step time `ms_per_step_with_event_pair`)."""
        # settle: workspace growth / plan building, and the DVFS ramp -- on MI355X the same kernel runs ~25 %
        # slower during the first few hundred ms after idle (tools/microbench/mulv.hip: 136 -> 174 G mul/s).  Never
        # counted; the W warm-up steps follow.
        t_settle = time.perf_counter()
        # The exit decision is COLLECTIVE: a step may contain an all-gather, so every rank must run the same
        # number of settle steps (a per-rank clock desynchronises the ranks and deadlocks the next collective).
        while True:
            step()
            torch.cuda.synchronize()
            if max_over_ranks(1.0 if time.perf_counter() - t_settle > args.settle_s else 0.0) > 0.0:
                break
        for _ in range(W):
            step()

        def read_phases():
            out = {}
            for ph in range(N_PHASES):
                ms, cnt = ctypes.c_double(0), ctypes.c_uint64(0)
                check(L.mzk_prof_read(ph, ctypes.byref(ms), ctypes.byref(cnt)))
                if cnt.value:
                    out[L.mzk_prof_name(ph).decode()] = {"avg_ms": ms.value / cnt.value, "launches": cnt.value}
            return out

        # timed region: K steps; only the priced kernel carries a HIP-event pair (on the launch stream) -- an event pair
        # costs a few microseconds of stream time, ten of them per step distorted the 0.15 ms NTT step by ~10 %
        L.mzk_prof_reset()
        L.mzk_prof_select(ctypes.c_uint32(price_mask if priced_in_timed_region else 0))
        L.mzk_prof_enable(1 if priced_in_timed_region else 0)
        barrier_sync()
        t0 = time.perf_counter()
        for _ in range(K):
            step()
        barrier_sync()
        dt = time.perf_counter() - t0
        L.mzk_prof_enable(0)
        priced = read_phases()
        priced_tag, priced_step_ms = "_in_timed_region", None
        if not priced_in_timed_region:
            # the priced kernel group, in K further steps of its own (one event pair per step on the launch stream)
            L.mzk_prof_reset()
            L.mzk_prof_select(ctypes.c_uint32(price_mask))
            L.mzk_prof_enable(1)
            barrier_sync()
            tp = time.perf_counter()
            for _ in range(K):
                step()
            barrier_sync()
            priced_step_ms = max_over_ranks(time.perf_counter() - tp) / K * 1e3
            L.mzk_prof_enable(0)
            priced = read_phases()
            priced_tag = "_in_priced_pass"
        # second pass of K identical steps, NOT timed: event pairs around every kernel group for the phase breakdown
        L.mzk_prof_reset()
        L.mzk_prof_select(ctypes.c_uint32(0xffffffff))
        L.mzk_prof_enable(1)
        for _ in range(K):
            step()
        torch.cuda.synchronize()
        L.mzk_prof_enable(0)
        phases = read_phases()
        L.mzk_prof_reset()
        for k, v in priced.items():
            phases[k + priced_tag] = v
        if priced_in_timed_region:
            # a further pass, K steps with NO event at all: what the event pair of the timed region costs the step.  Reported beside
            # ms_per_step (`ms_per_step_without_event_pair`), never instead of it.
            barrier_sync()
            t1 = time.perf_counter()
            for _ in range(K):
                step()
            barrier_sync()
            phases["_ms_per_step_without_events"] = max_over_ranks(time.perf_counter() - t1) / K * 1e3
        else:
            phases["_ms_per_step_with_event_pair"] = priced_step_ms
        return max_over_ranks(dt), phases

    def run_in_flight(nctx, handle=None, what="KZG commit"):
        """`nctx` commits in flight through ONE C-ABI call per batch: mzk_kzg_commit_srs_batch_dev spreads the polynomials of
        a batch over the contexts of this GPU (the process made four in mz.init_devices above; max_in_flight = nctx).
        handle: the SRS handle to commit against (default: the one with window tables)."""
        if world != 1 or args.no_two_in_flight:
            return None
        hsrs = srs._h if handle is None else handle
        try:
            batch = 32          # polynomials per call: the pipeline drains at the end of every call, so short batches overlap less
            coefs = torch.empty(batch * n * 4, dtype=torch.int64, device=dev)
            for k in range(batch):
                seed_k = SEED if k == 0 else SEED + 4242 + k
                check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(seed_k), ctypes.c_size_t(n), ctypes.c_void_p(coefs.data_ptr() + k * n * 32), stream))
            outs = torch.zeros(8 * batch, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            lanes = ctypes.c_int(nctx)

            def one_batch():
                check(L.mzk_kzg_commit_srs_batch_dev(hsrs, dptr(coefs), ctypes.c_size_t(n), ctypes.c_size_t(batch), dptr(outs), lanes, stream))
            for _ in range(4):
                one_batch()
            torch.cuda.synchronize()
            reps = max(4, K // 2)
            t0 = time.perf_counter()
            for _ in range(reps):
                one_batch()
            torch.cuda.synchronize()
            dtp = (time.perf_counter() - t0) / (reps * batch)
            check(L.mzk_kzg_commit_srs_dev(srs._h, dptr(scalars), ctypes.c_size_t(n), dptr(result_srs), ctypes.c_int(0), stream))
            torch.cuda.synchronize()
            same = bool(torch.equal(outs[:8], result_srs))       # polynomial 0 of the batch is `scalars`: same point as the single call
            return {"metric": "%s pairs/s, batches of %d polynomials through mzk_kzg_commit_srs_batch_dev with %d commits in flight "
                              "(%d contexts on one GPU, shared SRS handle)" % (what, batch, nctx, nctx),
                    "value": n / dtp, "unit": "pairs/s", "ms_per_commit": dtp * 1e3, "commits_timed": reps * batch,
                    "same_point_as_single_call": same}
        except Exception as ex:
            return {"error": str(ex)[:300]}

    L.mzk_prof_name.restype = ctypes.c_char_p
    K, W = args.steps, args.warmup
    msm_dt, msm_ph = timed(msm_step, K, W, 1 << PH_ACC)
    srs_dt, srs_ph = timed(srs_step, K, W, 1 << PH_ACC)
    ntt_dt, ntt_ph = timed(ntt_step, K, W, 1 << PH_NTT_TOTAL, priced_in_timed_region=False)
    nttm_dt, nttm_ph = timed(ntt_m128_step, K, W, 1 << PH_NTT_TOTAL, priced_in_timed_region=False)
    lde_dt, lde_ph = timed(lde_m128_step, K, W, 1 << PH_NTT_TOTAL, priced_in_timed_region=False)
    mk_dt, mk_ph = timed(merkle_m128_step, K, W, 1 << PH_MERKLE, priced_in_timed_region=False)

    # Two commits in flight: a prover commits to many polynomials against one SRS, and the last ~0.3 ms of a commit
    # (bucket reduction, inversion) run on a nearly idle GPU.  Two / four contexts on the SAME device (own stream + workspace
    # each, one shared SRS handle) take turns, so that tail and the memory-bound sort overlap the other commits' accumulate.
    # Reported beside `value`, which stays one commit at a time.
    pipelined = run_in_flight(2)
    pipelined4 = run_in_flight(4)
    # the same against a handle WITHOUT window tables (prepared points only: the GLV / Horner layout of the generic MSM, no
    # precomputation beyond the Montgomery conversion): what several commits in flight are worth where 40 % of one MSM is
    # sort, reduction tails and the window Horner
    generic4 = None
    if world == 1 and not args.no_two_in_flight:
        hplain = ctypes.c_void_p()
        try:
            check(L.mzk_srs_from_device_ex(dptr(points), ctypes.c_size_t(n), ctypes.c_int(0), ctypes.byref(hplain), stream))
            generic4 = {"one_at_a_time": run_in_flight(1, hplain, "G1 MSM against prepared points (no window tables)"),
                        "four_in_flight": run_in_flight(4, hplain, "G1 MSM against prepared points (no window tables)")}
        except Exception as ex:
            generic4 = {"error": str(ex)[:300]}
        finally:
            if hplain:
                L.mzk_srs_free(hplain)
            torch.cuda.empty_cache()

    def run_other_width(c):
        """The same commit against a handle with c-bit windows (mzk_srs_from_device_ex).  `value` uses the library's default
        width for the size (17 bits from 2^19 points: profiles/r03b_window_sweep.txt); BASELINE configs[2] names 16-bit
        windows, so that width is always reported beside it as its own leg."""
        if world != 1 or args.no_two_in_flight:
            return None
        hw = ctypes.c_void_p()
        try:
            check(L.mzk_srs_from_device_ex(dptr(points), ctypes.c_size_t(n), ctypes.c_int(c), ctypes.byref(hw), stream))
            outw = torch.zeros(8, dtype=torch.int64, device=dev)

            def stepw():
                check(L.mzk_kzg_commit_srs_dev(hw, dptr(scalars), ctypes.c_size_t(n), dptr(outw), ctypes.c_int(0), stream))
            for _ in range(max(W, 2)):
                stepw()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                stepw()
            torch.cuda.synchronize()
            dtw = (time.perf_counter() - t0) / K
            return {"window_bits": c, "tables": 254 // c + 1, "ms_per_step": dtw * 1e3, "value": n / dtw, "unit": "pairs/s",
                    "same_point_as_default_width": bool(torch.equal(outw, result_srs))}
        except Exception as ex:
            return {"error": str(ex)[:300]}
        finally:
            if hw:
                L.mzk_srs_free(hw)
            torch.cuda.empty_cache()
    width16 = run_other_width(16) if srs_window_bits != 16 else None

    def run_small_batch():
        """The reference's actual call pattern: hundreds of SHORT polynomials committed against one pk in a loop (das/avail.rs:88-98
        per row, das/eigenda.rs:92-101 per chunk, algebra/gemini.rs:112-114).  One call for the batch (mzk_kzg_commit_srs_many_dev:
        the whole batch as one bucket problem; with mzk_srs_build_direct no buckets at all) against the same batch one commit at a
        time; first and last point of every batch checked against the oracle's Pippenger, all of them against the single calls."""
        if world != 1 or args.no_two_in_flight:
            return None
        res = {}
        L.mzk_srs_table_bytes.restype = ctypes.c_size_t
        for lg, count in ((10, 256), (12, 64), (14, 16)):
            nn = 1 << lg
            with_direct = lg <= 12          # direct tables of 2^14 powers would be 14 GiB at 10 bits: the bucket pass only
            key = "%d_x_2^%d" % (count, lg)
            hs = ctypes.c_void_p()
            try:
                pt = torch.empty(nn * 8, dtype=torch.int64, device=dev)
                cf = torch.empty(count * nn * 4, dtype=torch.int64, device=dev)
                check(L.mzk_synth_g1_points_dev(ctypes.c_uint64(SEED + 31), ctypes.c_size_t(nn), dptr(pt), stream))
                check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 32), ctypes.c_size_t(count * nn), dptr(cf), stream))
                check(L.mzk_srs_from_device(dptr(pt), ctypes.c_size_t(nn), ctypes.byref(hs), stream))
                o_many = torch.zeros(count * 8, dtype=torch.int64, device=dev)
                o_one = torch.zeros(count * 8, dtype=torch.int64, device=dev)

                def many():
                    check(L.mzk_kzg_commit_srs_many_dev(hs, dptr(cf), ctypes.c_size_t(nn), ctypes.c_size_t(count), dptr(o_many), stream))

                def loop():
                    for k in range(count):
                        check(L.mzk_kzg_commit_srs_dev(hs, ctypes.c_void_p(cf.data_ptr() + k * nn * 32), ctypes.c_size_t(nn),
                                                       ctypes.c_void_p(o_one.data_ptr() + k * 64), ctypes.c_int(0), stream))

                def clock(fn, reps):
                    for _ in range(3 if reps > 2 else 1):
                        fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        fn()
                    torch.cuda.synchronize()
                    return (time.perf_counter() - t0) / reps * 1e3
                e = {"polynomials": count, "coefficients_each": nn, "window_bits": int(L.mzk_srs_window_bits(hs))}
                e["one_at_a_time_ms"] = clock(loop, 2)
                e["one_call_ms"] = clock(many, max(K, 30))
                e["same_points_as_single_calls"] = bool(torch.equal(o_many, o_one))
                pts_h = pt.cpu().numpy().view(np.uint64).reshape(nn, 8)
                cf_h = cf.cpu().numpy().view(np.uint64).reshape(count, nn, 4)
                got = mz.array_to_points(o_many.cpu().numpy().view(np.uint64).reshape(count, 8))
                e["first_and_last_equal_oracle"] = bool(got[0] == orc.msm_fast(cf_h[0], pts_h) and got[-1] == orc.msm_fast(cf_h[-1], pts_h))
                e["table_bytes_after_the_pass"] = int(L.mzk_srs_table_bytes(hs))     # from 2^13 coefficients on: + the 12-bit tables the pass builds once
                e["us_per_commit"] = e["one_call_ms"] / count * 1e3
                e["speedup_over_one_at_a_time"] = e["one_at_a_time_ms"] / e["one_call_ms"]
                if lg == 10:
                    # the same batch on 31-byte coefficients (the DAS callers chunk their data into 31-byte field elements,
                    # das/avail.rs:88-98): the top window of every scalar is empty but for the carry of the signed digits, which
                    # all lands in one bucket per polynomial (mzk_msm.hip bucket_end / HEAVY_SLOTS)
                    cf31 = cf.clone()
                    cf31.view(-1, 4)[:, 3] &= (1 << 56) - 1
                    o31 = torch.zeros(count * 8, dtype=torch.int64, device=dev)

                    def many31():
                        check(L.mzk_kzg_commit_srs_many_dev(hs, dptr(cf31), ctypes.c_size_t(nn), ctypes.c_size_t(count), dptr(o31), stream))
                    d31 = {"one_call_ms": clock(many31, max(K, 30))}
                    c31_h = cf31.cpu().numpy().view(np.uint64).reshape(count, nn, 4)
                    g31 = mz.array_to_points(o31.cpu().numpy().view(np.uint64).reshape(count, 8))
                    d31["first_and_last_equal_oracle"] = bool(g31[0] == orc.msm_fast(c31_h[0], pts_h) and g31[-1] == orc.msm_fast(c31_h[-1], pts_h))
                    d31["us_per_commit"] = d31["one_call_ms"] / count * 1e3
                    e["coefficients_of_31_bytes"] = d31
                    del cf31, o31
                if not with_direct:
                    res[key] = e
                    continue
                b0 = L.mzk_srs_table_bytes(hs)            # the window tables alone (a direct build replaces the previous direct tables)
                for bits in (10, 12):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    check(L.mzk_srs_build_direct(hs, ctypes.c_int(bits), ctypes.c_size_t(16 << 30), stream))
                    build_ms = (time.perf_counter() - t0) * 1e3
                    o_many.zero_()
                    d = {"table_bytes": int(L.mzk_srs_table_bytes(hs) - b0), "table_build_ms": build_ms, "one_call_ms": clock(many, max(K, 30))}
                    d["same_points_as_single_calls"] = bool(torch.equal(o_many, o_one))
                    d["us_per_commit"] = d["one_call_ms"] / count * 1e3
                    e["direct_tables_%d_bit" % bits] = d
                e["us_per_commit"] = e["one_call_ms"] / count * 1e3
                e["speedup_over_one_at_a_time"] = e["one_at_a_time_ms"] / e["one_call_ms"]
                # open_kzg per polynomial at its own point (das/avail.rs:132 opens per cell): one call (over the 12-bit direct tables
                # built above) against one mzk_kzg_open_srs_dev per polynomial on the same handle
                us_h = orc.synth_vector(orc.FR, SEED + 33, count)
                ys_m = torch.zeros(count * 4, dtype=torch.int64, device=dev); ws_m = torch.zeros(count * 8, dtype=torch.int64, device=dev)
                ys_1 = torch.zeros(count * 4, dtype=torch.int64, device=dev); ws_1 = torch.zeros(count * 8, dtype=torch.int64, device=dev)

                def open_many():
                    check(L.mzk_kzg_open_srs_many_dev(hs, dptr(cf), ctypes.c_size_t(nn), ctypes.c_size_t(count), us_h.ctypes.data_as(ctypes.c_void_p), dptr(ys_m), dptr(ws_m), stream))

                def open_loop():
                    for k in range(count):
                        check(L.mzk_kzg_open_srs_dev(hs, ctypes.c_void_p(cf.data_ptr() + k * nn * 32), ctypes.c_size_t(nn), us_h[k].ctypes.data_as(ctypes.c_void_p),
                                                     ctypes.c_void_p(ys_1.data_ptr() + k * 32), ctypes.c_void_p(ws_1.data_ptr() + k * 64), stream))
                o = {"one_at_a_time_ms": clock(open_loop, 1), "one_call_ms": clock(open_many, max(K, 30))}
                o["same_values_and_witnesses_as_single_calls"] = bool(torch.equal(ys_m, ys_1) and torch.equal(ws_m, ws_1))
                y0 = orc.from_limbs(ys_m[:4].cpu().numpy().view(np.uint64).reshape(1, 4))[0]
                o["first_value_equals_oracle_horner"] = bool(y0 == orc.poly_eval(orc.FR, cf_h[0], orc.from_limbs(us_h[:1])[0]))
                o["us_per_opening"] = o["one_call_ms"] / count * 1e3
                e["openings_over_direct_tables_12_bit"] = o
                res[key] = e
            except Exception as ex:
                res[key] = {"error": str(ex)[:300]}
            finally:
                if hs:
                    L.mzk_srs_free(hs)
                torch.cuda.empty_cache()
        res["metric"] = ("ms per batch of KZG commitments of short polynomials against one SRS: mzk_kzg_commit_srs_many_dev (one call, default narrow "
                         "window tables; direct_tables_*: after mzk_srs_build_direct) vs one mzk_kzg_commit_srs_dev per polynomial")
        return res
    small_batch = run_small_batch()

    def run_ntt_batched():
        """Many transforms per call (mzk_ntt_batch_dev): a prover interpolates / extends every column of a trace, and a batch
        gives the kernels several rounds of workgroups per CU, i.e. loads and stores under other tiles' butterflies."""
        if world != 1 or args.no_two_in_flight:
            return None
        res = {}
        try:
            for lgb, batch in ((args.log2n, 16), (16, 64), (12, 64)):
                if lgb > args.log2n:
                    continue
                nb = 1 << lgb
                vb = torch.empty(batch * nb * 4, dtype=torch.int64, device=dev)
                for k in range(batch):
                    check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 7000 + k), ctypes.c_size_t(nb), ctypes.c_void_p(vb.data_ptr() + k * nb * 32), stream))
                ob = torch.empty_like(vb)
                rb = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, lgb)], 4)

                def stepb():
                    check(L.mzk_ntt_batch_dev(mz.FIELD_FR, rb.ctypes.data_as(ctypes.c_void_p), dptr(vb), dptr(ob), ctypes.c_size_t(nb), ctypes.c_size_t(batch), 0, stream))
                for _ in range(3):
                    stepb()
                torch.cuda.synchronize()
                reps = max(K, 10)
                t0 = time.perf_counter()
                for _ in range(reps):
                    stepb()
                torch.cuda.synchronize()
                dtb = (time.perf_counter() - t0) / reps
                # row 0 of the batch against the single-transform entry point
                one = torch.empty(nb * 4, dtype=torch.int64, device=dev)
                check(L.mzk_ntt_dev(mz.FIELD_FR, rb.ctypes.data_as(ctypes.c_void_p), dptr(vb), dptr(one), ctypes.c_size_t(nb), 0, stream))
                torch.cuda.synchronize()
                res["%d x 2^%d" % (batch, lgb)] = {"ms_per_call": dtb * 1e3, "ms_per_transform": dtb * 1e3 / batch, "value": batch * nb / dtb, "unit": "elems/s",
                                                  "row_0_equals_single_transform": bool(torch.equal(one, ob[:nb * 4]))}
                del vb, ob, one
                torch.cuda.empty_cache()
            # the STARK side: low-degree extension (blow-up 4) of 64 trace columns of 2^14 coefficients, M128
            nc, order, batch = 1 << 14, 1 << 16, 64
            cb = torch.empty(batch * nc * 2, dtype=torch.int64, device=dev)
            for k in range(batch):
                check(L.mzk_synth_field_dev(mz.FIELD_M128, ctypes.c_uint64(SEED + 8000 + k), ctypes.c_size_t(nc), ctypes.c_void_p(cb.data_ptr() + k * nc * 16), stream))
            ob = torch.empty(batch * order * 2, dtype=torch.int64, device=dev)
            gen_l = mz.to_limbs([mz.root_of_unity(mz.FIELD_M128, 16)], 2)
            off_l = mz.to_limbs([orc.M128_GEN], 2)

            def stepl():
                check(L.mzk_coset_lde_batch_dev(mz.FIELD_M128, dptr(cb), ctypes.c_size_t(nc), off_l.ctypes.data_as(ctypes.c_void_p), gen_l.ctypes.data_as(ctypes.c_void_p),
                                                dptr(ob), ctypes.c_size_t(order), ctypes.c_size_t(batch), stream))
            for _ in range(3):
                stepl()
            torch.cuda.synchronize()
            reps = max(K, 10)
            t0 = time.perf_counter()
            for _ in range(reps):
                stepl()
            torch.cuda.synchronize()
            dtl = (time.perf_counter() - t0) / reps
            one = torch.empty(order * 2, dtype=torch.int64, device=dev)
            check(L.mzk_coset_lde_dev(mz.FIELD_M128, dptr(cb), ctypes.c_size_t(nc), off_l.ctypes.data_as(ctypes.c_void_p), gen_l.ctypes.data_as(ctypes.c_void_p),
                                      dptr(one), ctypes.c_size_t(order), stream))
            torch.cuda.synchronize()
            res["coset_lde_m128 64 x (2^14 -> 2^16)"] = {"ms_per_call": dtl * 1e3, "ms_per_column": dtl * 1e3 / batch, "value": batch * order / dtl, "unit": "out elems/s",
                                                        "row_0_equals_single_call": bool(torch.equal(one, ob[:order * 2]))}
        except Exception as ex:
            res["error"] = str(ex)[:300]
        return res
    ntt_batched = run_ntt_batched()

    def run_pcie_inclusive():
        """The host-buffer entry points a MyZKP caller binds first (Vec<u64> limbs in, point / vector
        out), timed with the transfers inside: pageable host memory, plain hipMemcpyAsync (already 56 GB/s either way on this runtime;
        the points of the generic MSM travel on a side stream under the digit sort -- profiles/r03h_pcie_probe.txt).
        Reported beside `value`, never as `value` (inputs of `value` are resident in HBM)."""
        if world != 1 or args.no_two_in_flight:
            return None
        res = {}
        try:
            hs, hp = scalars.cpu().numpy().view(np.uint64).reshape(-1, 4).copy(), points.cpu().numpy().view(np.uint64).reshape(-1, 8).copy()
            hv = ntt_in.cpu().numpy().view(np.uint64).reshape(-1, 4).copy()
            wr = mz.root_of_unity(mz.FIELD_FR, args.log2n)
            hcommit_out = np.zeros((1, 8), dtype=np.uint64)

            def commit_host():                    # the bench's device-resident handle, host coefficients (no wrapper object that could free it)
                check(L.mzk_kzg_commit_srs(srs._h, hs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), hcommit_out.ctypes.data_as(ctypes.c_void_p)))
            hout = np.zeros_like(hv)             # the caller's output vector, allocated once (a fresh one per call costs page faults)
            wl = mz.to_limbs([wr], 4)

            def ntt_host():
                check(L.mzk_ntt(mz.FIELD_FR, wl.ctypes.data_as(ctypes.c_void_p), hv.ctypes.data_as(ctypes.c_void_p), hout.ctypes.data_as(ctypes.c_void_p),
                                ctypes.c_size_t(n), 0))
            legs = (("msm_g1_bn254_host_buffers", lambda: mz.msm_g1(hs, hp), n, "pairs/s", 96 * n),
                    ("kzg_commit_srs_host_scalars", commit_host, n, "pairs/s", 32 * n),
                    ("ntt_host_buffers", ntt_host, n, "elems/s", 64 * n))
            for name, fn, units, unit, nbytes in legs:
                fn(); fn()
                reps = 5
                t0 = time.perf_counter()
                for _ in range(reps):
                    r = fn()
                dtp = (time.perf_counter() - t0) / reps
                res[name] = {"ms_per_call": dtp * 1e3, "value": units / dtp, "unit": unit, "bytes_over_pcie": nbytes}
            res["msm_matches_resident_result"] = bool(mz.msm_g1(hs, hp) == mz.array_to_points(result.cpu().numpy().view(np.uint64))[0]) if world == 1 else None
        except Exception as ex:
            res["error"] = str(ex)[:300]
        return res
    pcie_inclusive = run_pcie_inclusive()

    def run_stark_commit_pipeline(lgt=14, regs=16):
        """The commit side of the reference's STARK prover on M128, stage by stage (fast_stark.rs:209-337): interpolate the trace
        registers over the omicron domain (fast_stark.rs:209-229 -> ntt.rs:225-252), fast_coset_evaluate every register onto the FRI
        domain (:231, blow-up 4), Merkle::commit every codeword (:231-243), FRI::commit one codeword with the rounds' trees kept for
        the query phase (fri.rs:144-209).  One batched call per stage, codewords resident in HBM from the extension on.  Before
        timing: register 0's polynomial against the oracle's subproduct-tree interpolation, its codeword against the oracle's
        coset evaluation, its Merkle root and the first FRI fold against the oracle's."""
        if world != 1 or args.no_two_in_flight:
            return None
        import hashlib
        try:
            fid = mz.FIELD_M128
            p128 = mz.MODULUS[fid]
            cycles = (1 << lgt) - 3                     # a trace that does not fill its power-of-two domain
            lg_fri = lgt + 2
            n_fri = 1 << lg_fri
            omicron, omega = mz.root_of_unity(fid, lgt), mz.root_of_unity(fid, lg_fri)
            gen = orc.M128_GEN
            dom, acc = [], 1
            for _ in range(cycles):
                dom.append(acc); acc = acc * omicron % p128
            dom = mz.to_limbs(dom, 2)
            trace = np.stack([orc.synth_vector(orc.M128, 100 + r, cycles) for r in range(regs)])
            rounds = lg_fri - 4

            def challenge(rnd, last, root):
                return None if last else int.from_bytes(hashlib.sha3_256(root + bytes([rnd])).digest(), "little") % p128

            def stage_interpolate():
                return mz.fast_interpolate_batch(fid, dom, trace, omicron, 1 << lgt)
            polys = stage_interpolate()
            coefs = np.zeros((regs, 1 << lgt, 2), dtype=np.uint64)
            for k, c in enumerate(polys):
                coefs[k, :len(c)] = c
            # the same stage as the prover would run it: the trace goes up once (that copy is inside the stage's time), the coefficients stay
            # in HBM (rows of `cycles` elements, zeros behind the trimmed length) and feed the extension directly
            trace_flat = np.ascontiguousarray(trace).view(np.int64).reshape(-1)
            d_coefs = torch.zeros(regs * cycles * 2, dtype=torch.int64, device=dev)
            lens_dev = []

            def stage_interpolate_hbm():
                d_trace = torch.from_numpy(trace_flat).to(dev)
                lens_dev[:] = mz.fast_interpolate_batch_dev(fid, dom, d_trace.data_ptr(), regs, omicron, 1 << lgt, d_coefs.data_ptr(), torch.cuda.current_stream().cuda_stream)
            stage_interpolate_hbm()
            rows = d_coefs.cpu().numpy().view(np.uint64).reshape(regs, cycles, 2)
            ok_hbm = all(lens_dev[k] == len(polys[k]) and np.array_equal(rows[k, :lens_dev[k]], np.asarray(polys[k])) and not rows[k, lens_dev[k]:].any() for k in range(regs))
            d_cw = torch.empty(regs * n_fri * 2, dtype=torch.int64, device=dev)
            off_l, gen_l = mz.to_limbs([gen], 2), mz.to_limbs([omega], 2)
            roots = (ctypes.c_uint8 * (32 * regs))()

            def stage_lde():
                check(L.mzk_coset_lde_batch_dev(fid, dptr(d_coefs), ctypes.c_size_t(cycles), off_l.ctypes.data_as(ctypes.c_void_p), gen_l.ctypes.data_as(ctypes.c_void_p),
                                                dptr(d_cw), ctypes.c_size_t(n_fri), ctypes.c_size_t(regs), stream))
                torch.cuda.synchronize()

            def stage_merkle():
                check(L.mzk_merkle_commit_field_batch_dev(fid, dptr(d_cw), ctypes.c_size_t(n_fri), ctypes.c_size_t(regs), roots, stream))

            def stage_fri():
                _, froots, trees = mz.fri_commit(fid, None, omega, gen, rounds, challenge, keep_trees=True, codewords=False, device_ptr=d_cw.data_ptr(), n=n_fri)
                return froots, trees
            stage_lde(); stage_merkle()
            froots, trees = stage_fri()
            # parity, before any timing
            cw0 = d_cw[: n_fri * 2].cpu().numpy().view(np.uint64).reshape(n_fri, 2)
            rc_i, want_poly = orc.fast_interpolate_ref(orc.M128, dom, np.ascontiguousarray(trace[0]), omicron, 1 << lgt)
            ok_interp = bool(rc_i == 0 and np.array_equal(np.asarray(polys[0]), want_poly))
            rc_l, want_cw = orc.coset_ref(orc.M128, coefs[0], gen, omega, n_fri)
            ok_lde = bool(rc_l == 0 and np.array_equal(cw0, want_cw))
            ok_root = bytes(roots[0:32]) == orc.merkle_commit_field_ref(orc.M128, cw0) == bytes(froots[0])
            a0 = challenge(0, False, bytes(froots[0]))
            folded = orc.fri_fold_ref(orc.M128, np.ascontiguousarray(cw0), a0, gen, omega)
            ok_fold = bool(bytes(froots[1]) == orc.merkle_commit_field_ref(orc.M128, folded))
            for t in trees:
                if t is not None:
                    t.close()
            if not (ok_interp and ok_hbm and ok_lde and ok_root and ok_fold):
                return {"error": "parity: interpolate %s (HBM form %s), lde %s, merkle root %s, first fold %s" % (ok_interp, ok_hbm, ok_lde, ok_root, ok_fold)}
            best = {}
            for rep in range(4):
                for name, fn in (("interpolate_host", stage_interpolate), ("interpolate", stage_interpolate_hbm), ("coset_lde", stage_lde), ("merkle_commit", stage_merkle), ("fri_commit", stage_fri)):
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    r = fn()
                    torch.cuda.synchronize()
                    dtp = (time.perf_counter() - t0) * 1e3
                    if name == "fri_commit":
                        for t in r[1]:
                            if t is not None:
                                t.close()
                    if rep:
                        best[name] = min(best.get(name, dtp), dtp)
            return {"metric": "STARK commit side on M128, ms per stage (one batched call each; fast_stark.rs:209-337, fri.rs:144-209)",
                    "registers": regs, "trace_cycles": cycles, "fri_domain": n_fri, "fri_rounds": rounds,
                    "stages_ms": {"interpolate_%d_registers_trace_uploaded_coefficients_in_hbm" % regs: best["interpolate"], "coset_lde_batch_dev": best["coset_lde"],
                                  "merkle_commit_batch_dev": best["merkle_commit"], "fri_commit_keep_trees_dev": best["fri_commit"]},
                    "total_ms": best["interpolate"] + best["coset_lde"] + best["merkle_commit"] + best["fri_commit"],
                    "interpolate_%d_registers_host_buffers_ms" % regs: best["interpolate_host"],
                    "fri_us_per_round": best["fri_commit"] / rounds * 1e3,
                    "parity": {"interpolate_vs_oracle": ok_interp, "interpolate_hbm_form_equals_host_form": ok_hbm, "coset_lde_vs_oracle": ok_lde, "merkle_root_vs_oracle": ok_root,
                               "first_fri_fold_vs_oracle": ok_fold},
                    "note": "best of three repetitions per stage, each bracketed by a device synchronize; the trace is uploaded inside the interpolation stage "
                            "(mzk_fast_interpolate_batch_dev; the host-buffer form of rounds 4-5, coefficients back over PCIe, is timed beside it), everything after "
                            "it stays in HBM; the challenge callback hashes on the host as the reference's transcript does"}
        except Exception as ex:
            return {"error": str(ex)[:300]}
    stark_pipeline = run_stark_commit_pipeline()
    progress("timed legs done")
    # Every leg's `ms_per_step` and `value` are the contract's timed region (timed(): W warm-up steps, K steps between barrier +
    # synchronize).  For the short sub-legs that region carries no event (priced_in_timed_region=False); the step time of their
    # priced pass is kept as ms_per_step_with_event_pair.  The generic MSM, like the headline, carries the pair of its accumulate
    # kernel inside the region (0.1 % of the step) and reports the event-free step beside it.
    msm_ms, ntt_ms, nttm_ms, lde_ms, mk_ms = (d / K * 1e3 for d in (msm_dt, ntt_dt, nttm_dt, lde_dt, mk_dt))
    msm_ms_plain = msm_ph.pop("_ms_per_step_without_events", None)
    ntt_ms_ev, nttm_ms_ev, lde_ms_ev, mk_ms_ev = (ph.pop("_ms_per_step_with_event_pair", None) for ph in (ntt_ph, nttm_ph, lde_ph, mk_ph))
    msm_rate = world * n / (msm_ms * 1e-3)
    ntt_rate = world * n / (ntt_ms * 1e-3)

    # achievable HBM copy bandwidth on THIS box (SURVEY 8d asks for it next to the nominal 8 TB/s)
    # measured with the library's own 16-byte-per-lane grid-stride copy (mzk_selftest_copy_dev; read + write counted); torch's
    # Tensor.copy_ -- the figure of rounds 2-4 -- is kept beside it
    cp_a = torch.empty(1 << 28, dtype=torch.int32, device=dev)   # 1 GiB
    cp_b = torch.empty_like(cp_a)
    def copy_rate(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return 5 * 2 * cp_a.numel() * 4 / (time.perf_counter() - t0) / 1e9
    copy_gbps_torch = copy_rate(lambda: cp_b.copy_(cp_a))
    copy_gbps = copy_rate(lambda: check(L.mzk_selftest_copy_dev(dptr(cp_a), dptr(cp_b), ctypes.c_size_t(cp_a.numel() * 4), stream)))
    copy_same = bool(torch.equal(cp_a[:1 << 20], cp_b[:1 << 20]) and torch.equal(cp_a[-(1 << 20):], cp_b[-(1 << 20):]))
    del cp_a, cp_b
    torch.cuda.empty_cache()

    def hbm_roofline(alg_bytes, ms):
        ach = alg_bytes / (ms * 1e-3) / 1e9
        return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
                "frac_of_measured_copy_rate": ach / copy_gbps, "measured_copy_GBps": copy_gbps, "traffic": None}

    acc_ms = msm_ph.get("msm_bucket_accumulate_in_timed_region", {}).get("avg_ms", float("nan"))
    roof = hbm_roofline(96.0 * n, acc_ms)
    roof["kernel"] = "k_seg_accumulate"
    roof["algorithmic_bytes_per_launch"] = 96 * n
    if args.log2n == 20:
        # the generic layout gathers 2 x 8 window entries per pair from the prepared points and their endomorphism images
        # (64-byte rows, random: raw counters, see recorded_traffic)
        tr = recorded_traffic("k_seg_accumulate", section="generic MSM 2^20")
        if tr is not None:
            roof["traffic"] = tr["bytes"]
            roof["traffic_source"] = tr["source"]
    ntt_total_ms = ntt_ph.get("ntt_whole_transform_in_priced_pass", {}).get("avg_ms", float("nan"))
    npass = sum(1 for k in ntt_ph if k.startswith("ntt_pass") and not k.endswith("_in_priced_pass"))
    ntt_roof = hbm_roofline(64.0 * n, ntt_total_ms)
    ntt_roof["kernel"] = "k_ntt_strided + k_ntt_last (whole transform: %d passes, one event pair around them)" % npass
    ntt_roof["passes"] = npass
    ntt_roof["algorithmic_bytes_per_launch"] = 64 * n
    ntt_roof["frac_of_wall_clock"] = 64.0 * n / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS
    if args.log2n == 20:
        tr = recorded_ntt_traffic("Fr")
        if tr is not None:
            ntt_roof.update({"traffic": tr["bytes"], "traffic_raw_counters": tr["raw_bytes"], "traffic_source": tr["source"]})
    nttm_total_ms = nttm_ph.get("ntt_whole_transform_in_priced_pass", {}).get("avg_ms", float("nan"))
    nttm_roof = dict(hbm_roofline(32.0 * n, nttm_total_ms), kernel="k_ntt_strided + k_ntt_last (whole transform)", algorithmic_bytes_per_launch=32 * n)
    if args.log2n == 20:
        tr = recorded_ntt_traffic("M128")
        if tr is not None:
            nttm_roof.update({"traffic": tr["bytes"], "traffic_raw_counters": tr["raw_bytes"], "traffic_source": tr["source"]})
    # integer-multiply roofline (the binding one): v_mad_u64_u32 per Montgomery product = 171
    msm_mads = n * 16 * (8 * 171 + 2 * 135)             # generic layout: 2 x 8 GLV windows; madd = 8M + 2S per (pair, window)
    srs_mads = n * srs_table_windows * (8 * 171 + 2 * 135)
    ntt_mads = (n // 2) * args.log2n * 171
    # M128: 31 v_mad_i64_i32 per product since round 5 (5 x 5 limbs + 5 for the sparse modulus 1 + 407 * 2^119 + 1 constant; no v_mul_lo).
    # Round 4's product took 35 + 5 v_mul_lo = 40 half-rate multiplies: the fraction by THAT count is kept beside it, for comparison only.
    nttm_mads = (n // 2) * args.log2n * 31
    nttm_mads_r04 = (n // 2) * args.log2n * 40
    alu = {"unit": "v_mad_u64_u32/s", "peak": MAD_PEAK_PER_S,
           "msm_accumulate_frac": msm_mads / (acc_ms * 1e-3) / MAD_PEAK_PER_S if acc_ms == acc_ms else None,
           "ntt_frac": ntt_mads / (ntt_total_ms * 1e-3) / MAD_PEAK_PER_S if ntt_total_ms else None,
           "ntt_m128_frac": nttm_mads / (nttm_total_ms * 1e-3) / MAD_PEAK_PER_S if nttm_total_ms == nttm_total_ms and nttm_total_ms else None,
           "ntt_frac_of_step": ntt_mads / (ntt_ms * 1e-3) / MAD_PEAK_PER_S,
           "ntt_m128_frac_of_step": nttm_mads / (nttm_ms * 1e-3) / MAD_PEAK_PER_S,
           "ntt_m128_frac_by_round4_count_of_40_multiplies_per_product": nttm_mads_r04 / (nttm_total_ms * 1e-3) / MAD_PEAK_PER_S if nttm_total_ms == nttm_total_ms and nttm_total_ms else None,
           "ntt_m128_note": "(n/2) log2(n) products x 31 half-rate multiply-adds each (round 4: 40); the M128 transform is bound by neither roofline: what its "
                            "two passes wait for is the global loads / stores of the one tile each CU holds (DESIGN.md section 4)"}

    srs_ms = srs_dt / K * 1e3
    srs_rate = world * n / (srs_dt / K)
    srs_acc_ms = srs_ph.get("msm_bucket_accumulate_in_timed_region", {}).get("avg_ms", float("nan"))
    srs_roof = hbm_roofline(96.0 * n, srs_acc_ms)
    srs_roof["kernel"] = "k_seg_accumulate"
    srs_roof["measured"] = "HIP-event pair around the kernel on its launch stream, inside the timed region (the only instrumented kernel there)"
    if args.log2n == 20:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/gpu_jobs/r03c.sh) of k_seg_accumulate at 2^20 pairs with
        # the default window width, raw counters (64-byte random gathers: the gfx950 x2 FETCH_SIZE correction for wide coalesced
        # streams is not applied; with it the figure doubles).  The fixed-base method reads each of the table points of a pair
        # once (15 x 64 B = 0.96 GiB); the 128-byte fetch granule doubles that.  Served by L2 / Infinity Cache.  A recorded
        # profile of the same kernel at the same size -- not collected by this run.
        tr = recorded_traffic("k_seg_accumulate")
        if tr is not None:
            srs_roof["traffic"] = tr["bytes"]
            srs_roof["traffic_source"] = tr["source"]
    srs_roof["algorithmic_bytes_per_launch"] = 96 * n
    alu["kzg_commit_accumulate_frac"] = srs_mads / (srs_acc_ms * 1e-3) / MAD_PEAK_PER_S if srs_acc_ms == srs_acc_ms else None
    # Among multiply-adds EVERY VALU instruction costs a multiply-add's issue slot (tools/microbench/op_rates.hip, profiles/round5_op_issue_rates.txt:
    # 1.75 ns per wave-instruction and SIMD, whatever the mix), so the kernel's issue-side roofline is its executed VALU instructions (recorded
    # SQ_INSTS_VALU of the same kernel at the same size) x that slot / its SIMDs
    vi = recorded_valu_instructions("k_seg_accumulate") if args.log2n == 20 else None
    if vi is not None and srs_acc_ms == srs_acc_ms:
        simds = torch.cuda.get_device_properties(dev).multi_processor_count * 4
        alu["kzg_commit_accumulate_valu_issue"] = {"frac": vi["instructions"] / simds * 1.75e-9 / (srs_acc_ms * 1e-3), "valu_instructions_per_launch": vi["instructions"],
                                                   "slot_ns": 1.75, "source": vi["source"] + " (recorded counters) and profiles/round5_op_issue_rates.txt (the slot)"}
    out = {
        "metric": METRIC,
        "value": srs_rate, "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": srs_ms,
        "process_group": {"world_size": dist.get_world_size() if world > 1 else 1, "backend": dist.get_backend() if world > 1 else None,
                          "launched_by": os.environ.get("MZK_BENCH_LAUNCHED_BY", "external launcher" if world > 1 else "single process")},
        **({"REHEARSAL_NOT_A_MEASUREMENT": "ranks share one GPU, exchange over gloo via host (MZK_BENCH_SHARED_GPU_TEST=1)"} if shared_gpu_test else {}),
        # `value` is measured at the library's default window width for this SRS size (17 bits at 2^20 since round 3; BASELINE
        # configs[2] names 16 bits: the kzg_commit_16_bit_windows leg) -- stated at top level so that a change of the default shows
        "value_window_bits": srs_window_bits,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, 254-bit Montgomery)",
        "data": "synthetic",
        "config": {"workload": "KZG commit = BN254 G1 Pippenger MSM of 2^%d (scalar, point) pairs per GPU against a device-resident SRS "
                               "(commit_kzg, kzg.rs:57-59; %d-bit signed windows, the library's default at this size -- the 16-bit windows "
                               "BASELINE configs[2] names are the kzg_commit_16_bit_windows leg; the SRS handle holds %d window tables "
                               "2^(%d w)*P_i built once at upload like an FFT plan, so all windows share one bucket set); N GPUs = one MSM of "
                               "N*2^%d pairs (BASELINE configs[2]/[3])" % (args.log2n, srs_window_bits, srs_table_windows, srs_window_bits, args.log2n),
                   "pairs_per_gpu": n, "seed": SEED, "window_bits": srs_window_bits, "sharding": "contiguous shards + all_gather of 128 B partials (RCCL) + local fold"},
        "roofline": srs_roof,
        # BASELINE configs[2] read literally -- an MSM on ARBITRARY points, nothing precomputed per point set -- is this rate;
        # `value` is the KZG-commit special case (fixed SRS, window tables built once: srs_precompute)
        "msm_pairs_per_s_arbitrary_points": msm_rate,
        "ms_per_step_without_event_pair": srs_ph.pop("_ms_per_step_without_events", None),
        "phases": srs_ph,
        # what the headline rests on: `value` commits against window tables built ONCE per SRS (like an FFT plan);
        # `msm_generic` below is the same MSM with no per-point-set precomputation at all
        "srs_precompute": {"table_build_ms": srs_build_ms, "table_bytes": srs_table_windows * n * 64, "tables": srs_table_windows,
                           "generic_no_precompute_pairs_per_s": msm_rate,
                           "break_even_commits": (srs_build_ms / (msm_ms - srs_ms)) if msm_ms > srs_ms else None,
                           "note": "table build is outside the timed region; one KZG setup is followed by many commits/opens against the same powers_1 (kzg.rs:57-72)"},
        "kzg_commit_two_in_flight": pipelined,
        "kzg_commit_four_in_flight": pipelined4,
        "kzg_commit_16_bit_windows": width16 if width16 is not None else ({"note": "16 bits is the default width at this size: see `value`"} if srs_window_bits == 16 else None),
        "kzg_commit_small_batch": small_batch,
        "stark_commit_pipeline": stark_pipeline,
        "msm_no_tables_in_flight": generic4,
        "ntt_batched": ntt_batched,
        "pcie_inclusive": pcie_inclusive,
        "msm_generic": {"metric": "G1 MSM pairs/sec, arbitrary points every call (no per-point-set precomputation; 16 bucket sets + window Horner)",
                        "value": msm_rate, "unit": "pairs/s", "ms_per_step": msm_ms, "ms_per_step_without_event_pair": msm_ms_plain, "phases": msm_ph, "roofline": roof},
        "ntt": {"metric": "NTT elems/sec", "value": ntt_rate, "unit": "elems/s", "ms_per_step": ntt_ms, "ms_per_step_with_event_pair": ntt_ms_ev, "field": "BN254 Fr",
                "log2n": args.log2n, "multi_gpu": "replicas (one independent transform per GPU); ONE transform sharded over the ranks is the strong_scaling_ntt leg", "roofline": ntt_roof,
                "phases": ntt_ph},
        "ntt_m128": {"metric": "NTT elems/sec", "value": world * n / (nttm_ms * 1e-3), "unit": "elems/s", "ms_per_step": nttm_ms, "ms_per_step_with_event_pair": nttm_ms_ev,
                     "field": "M128 = 1 + 407*2^119 (fri.rs:408)", "log2n": args.log2n, "phases": nttm_ph,
                     "roofline": nttm_roof},
        "coset_lde_m128": {"metric": "LDE output elems/sec (fast_coset_evaluate, ntt.rs:254-269; blow-up 4)", "value": world * n / (lde_ms * 1e-3), "unit": "elems/s",
                           "ms_per_step": lde_ms, "ms_per_step_with_event_pair": lde_ms_ev, "n_coef": n // 4, "order": n, "phases": lde_ph},
        "merkle_m128": {"metric": "Merkle::commit of a codeword, SHA3-256 hashes/sec (merkle.rs:15-25 over bincode leaves, fri.rs:160-166; "
                                  "root copied to the host every step as FRI::commit needs it for the transcript)",
                        "value": world * (n - 1) / (mk_ms * 1e-3), "unit": "hashes/s", "ms_per_step": mk_ms, "ms_per_step_with_event_pair": mk_ms_ev, "leaves": n, "phases": mk_ph},
        "alu_roofline": alu,
        "hbm_copy_GBps_measured": copy_gbps,
        "hbm_copy": {"GBps_library_copy_kernel": copy_gbps, "GBps_torch_copy_": copy_gbps_torch, "bytes": 2 << 30, "copied_correctly": copy_same,
                     "note": "read + write of 1 GiB each way, five repetitions; every frac_of_measured_copy_rate in this line is against the library kernel's figure"},
        "parity": parity,
    }

    progress("line assembled; extra sizes")
    # ------------------------------------------------------------------ extra sizes (single GPU view, rank 0 only, once each)
    extras = {}
    if rank == 0:
        for lg in [int(x) for x in args.extra_sizes.split(",") if x]:
            if lg == args.log2n:
                continue
            try:
                nn = 1 << lg
                sc, pt = synth_shard(nn, 0)
                vin = torch.empty(nn * 4, dtype=torch.int64, device=dev)
                vout = torch.empty(nn * 4, dtype=torch.int64, device=dev)
                check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 99), ctypes.c_size_t(nn), dptr(vin), stream))
                rt = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, lg)], 4)
                res = torch.zeros(8, dtype=torch.int64, device=dev)
                def m():
                    check(L.mzk_msm_g1_bn254_dev(dptr(sc), dptr(pt), ctypes.c_size_t(nn), dptr(res), stream))
                def t():
                    check(L.mzk_ntt_dev(mz.FIELD_FR, rt.ctypes.data_as(ctypes.c_void_p), dptr(vin), dptr(vout), ctypes.c_size_t(nn), 0, stream))
                e = {}
                hx = ctypes.c_void_p()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                check(L.mzk_srs_from_device(dptr(pt), ctypes.c_size_t(nn), ctypes.byref(hx), stream))
                torch.cuda.synchronize()
                e["srs_table_build_ms"] = (time.perf_counter() - t0) * 1e3
                def c():
                    check(L.mzk_kzg_commit_srs_dev(hx, dptr(sc), ctypes.c_size_t(nn), dptr(res), ctypes.c_int(0), stream))
                for name, fn, reps in (("kzg_commit_srs", c, 3), ("msm_generic", m, 2), ("ntt", t, 3)):
                    fn(); torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(reps):
                        fn()
                    torch.cuda.synchronize()
                    dt = (time.perf_counter() - t0) / reps
                    e[name] = {"ms": dt * 1e3, "rate": nn / dt, "hbm_frac": (64.0 if name == "ntt" else 96.0) * nn / dt / 1e9 / HBM_PEAK_GBPS}
                # round trip property at this size: intt(ntt(x)) == x
                check(L.mzk_ntt_dev(mz.FIELD_FR, rt.ctypes.data_as(ctypes.c_void_p), dptr(vout), dptr(vout), ctypes.c_size_t(nn), 1, stream))
                torch.cuda.synchronize()
                e["ntt_roundtrip_ok"] = bool(torch.equal(vin, vout))
                extras["2^%d" % lg] = e
                L.mzk_srs_free(hx)
                del sc, pt, vin, vout
                torch.cuda.empty_cache()
            except Exception as ex:  # an extra must never sink the headline line
                extras["2^%d" % lg] = {"error": str(ex)[:200]}
    out["extra_sizes_1gpu"] = extras

    # ------------------------------------------------------------------ BASELINE configs[4]: end-to-end KZG at degree 2^e2e, 1 vs N GPUs
    # evaluations -> iNTT -> setup(alpha) -> commit -> open(u).  Strong scaling over the N ranks of this job: the
    # (cheap) iNTT and the synthetic-division quotient are replicated, rank g builds SRS powers [lo, hi) and runs the
    # MSMs of its slice of the coefficient / quotient vectors, partials are all-gathered (2 x N x 128 B) and folded.
    if args.e2e_log2n > 0:
        lg = args.e2e_log2n
        nn = 1 << lg
        lo, hi = sharded.shard_range(nn, rank, world)
        stages, err = {}, None
        rec_c = torch.zeros(16, dtype=torch.int64, device=dev)
        rec_w = torch.zeros(16, dtype=torch.int64, device=dev)
        hh = ctypes.c_void_p()

        class StageFailed(Exception):
            pass

        try:
            ev = torch.empty(nn * 4, dtype=torch.int64, device=dev)
            cf = torch.empty(nn * 4, dtype=torch.int64, device=dev)
            qq = torch.zeros(nn * 4, dtype=torch.int64, device=dev)
            sp = torch.empty((hi - lo) * 8, dtype=torch.int64, device=dev)
            yv = torch.zeros(4, dtype=torch.int64, device=dev)
            check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 555), ctypes.c_size_t(nn), dptr(ev), stream))
            rt = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, lg)], 4)
            alpha = orc.from_limbs(orc.synth_vector(orc.FR, SEED + 556, 1))[0]
            uu = orc.from_limbs(orc.synth_vector(orc.FR, SEED + 557, 1))[0]
            a_l, u_l, g_l = mz.to_limbs([alpha], 4), mz.to_limbs([uu], 4), mz.points_to_array([(1, 2)])

            def stage(name, fn, record):
                # every rank runs the same barriers and the same collective failure check per stage: a rank that
                # raised must not leave its peers waiting in the next stage's barrier
                barrier_sync()
                t0 = time.perf_counter()
                local_err = None
                try:
                    fn()
                    torch.cuda.synchronize()
                except Exception as ex:
                    local_err = str(ex)[:300]
                if record and local_err is None:
                    stages[name] = (time.perf_counter() - t0) * 1e3
                if max_over_ranks(0.0 if local_err is None else 1.0) > 0.0:
                    raise StageFailed(local_err or "a peer rank failed in stage " + name)

            def off(t, elems, limbs):
                return ctypes.c_void_p(t.data_ptr() + elems * limbs * 8)

        except Exception as ex:
            err = str(ex)[:300]
        # allocation / input failures are decided collectively BEFORE any rank enters the stage barriers
        if max_over_ranks(0.0 if err is None else 1.0) > 0.0:
            err = err or "a peer rank failed while allocating"
        try:
            if err is not None:
                raise StageFailed(err)
            for record in (False, True):     # first pass builds plans / workspaces
                if hh:
                    L.mzk_srs_free(hh); hh = ctypes.c_void_p()
                stage("intt", lambda: check(L.mzk_ntt_dev(mz.FIELD_FR, rt.ctypes.data_as(ctypes.c_void_p), dptr(ev), dptr(cf), ctypes.c_size_t(nn), 1, stream)), record)
                stage("setup_srs_powers", lambda: check(L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p),
                                                                                   ctypes.c_size_t(lo), ctypes.c_size_t(hi - lo), dptr(sp), stream)), record)
                # one commit + one open per SRS: keep plain prepared points (no window tables: 173 ms at 2^22 only pays
                # off after ~20 commits)
                stage("srs_prepare", lambda: check(L.mzk_srs_from_device_ex(dptr(sp), ctypes.c_size_t(hi - lo), ctypes.c_int(0), ctypes.byref(hh), stream)), record)
                stage("commit_local", lambda: check(L.mzk_kzg_commit_srs_dev(hh, off(cf, lo, 4), ctypes.c_size_t(hi - lo), dptr(rec_c), ctypes.c_int(1), stream)), record)
                def open_local():
                    check(L.mzk_kzg_open_quotient_dev(dptr(cf), ctypes.c_size_t(nn), u_l.ctypes.data_as(ctypes.c_void_p), dptr(yv), dptr(qq), stream))
                    qhi = min(hi, nn - 1)
                    check(L.mzk_kzg_commit_srs_dev(hh, off(qq, lo, 4), ctypes.c_size_t(max(qhi - lo, 0)), dptr(rec_w), ctypes.c_int(1), stream))
                stage("open_local", open_local, record)
            # the two MSMs of a proof are independent: commit on context 0, open (quotient + its MSM) on context 1 of the same
            # GPU at the same time -- the sort and the latency-bound tails of one run under the accumulation of the other
            rec_c2 = torch.zeros(16, dtype=torch.int64, device=dev)
            rec_w2 = torch.zeros(16, dtype=torch.int64, device=dev)
            L.mzk_ctx_stream.restype = ctypes.c_void_p
            s1 = ctypes.c_void_p(L.mzk_ctx_stream(1))

            def commit_and_open():
                mz.ctx_select(1)
                try:
                    check(L.mzk_kzg_open_quotient_dev(dptr(cf), ctypes.c_size_t(nn), u_l.ctypes.data_as(ctypes.c_void_p), dptr(yv), dptr(qq), s1))
                    qhi = min(hi, nn - 1)
                    check(L.mzk_kzg_commit_srs_dev(hh, off(qq, lo, 4), ctypes.c_size_t(max(qhi - lo, 0)), dptr(rec_w2), ctypes.c_int(1), s1))
                finally:
                    mz.ctx_select(0)
                check(L.mzk_kzg_commit_srs_dev(hh, off(cf, lo, 4), ctypes.c_size_t(hi - lo), dptr(rec_c2), ctypes.c_int(1), stream))
            for record in (False, True):
                stage("commit_and_open_overlapped", commit_and_open, record)
            overlapped_ms = stages.pop("commit_and_open_overlapped")
            # partial records are XYZZ (projective: the entry order inside a bucket comes from atomics, so the representation of
            # the same point differs from run to run): compare the canonical affine points
            aff = torch.zeros(4 * 8, dtype=torch.int64, device=dev)
            for k, r in enumerate((rec_c, rec_c2, rec_w, rec_w2)):
                check(L.mzk_g1_fold_partials_dev(dptr(r), ctypes.c_int(1), ctypes.c_void_p(aff.data_ptr() + 64 * k), stream))
            torch.cuda.synchronize()
            overlapped_same = bool(torch.equal(aff[0:8], aff[8:16]) and torch.equal(aff[16:24], aff[24:32]))
        except Exception as ex:
            err = str(ex)[:300]
        # every rank reaches this point; only fold if all local stages succeeded everywhere
        all_ok = max_over_ranks(0.0 if err is None else 1.0) == 0.0
        e2e = {"log2_degree": lg, "n_gpus": world, "srs_points_this_rank": hi - lo, "srs_points_total": nn,
               "what": "evaluations -> iNTT -> setup(alpha) -> commit -> open(u), device-resident, MSMs and SRS sharded over the ranks (BASELINE configs[4])"}
        if all_ok:
            barrier_sync()
            t0 = time.perf_counter()
            fin = torch.zeros(16, dtype=torch.int64, device=dev)
            rc_all = sharded.all_gather_partials(rec_c)
            rw_all = sharded.all_gather_partials(rec_w)
            check(L.mzk_g1_fold_partials_dev(dptr(rc_all), ctypes.c_int(rc_all.shape[0]), dptr(fin), stream))
            check(L.mzk_g1_fold_partials_dev(dptr(rw_all), ctypes.c_int(rw_all.shape[0]), ctypes.c_void_p(fin.data_ptr() + 64), stream))
            torch.cuda.synchronize()
            stages["gather_and_fold"] = (time.perf_counter() - t0) * 1e3
            stages = {k: max_over_ranks(v) for k, v in stages.items()}
            max_ov = max_over_ranks(overlapped_ms)
            if rank == 0:
                oc = fin.cpu().numpy().view(np.uint64)
                cf_cpu = cf.cpu().numpy().view(np.uint64).reshape(-1, 4)
                fa = orc.poly_eval(orc.FR, cf_cpu, alpha)
                yy = mz.from_limbs(yv.cpu().numpy().view(np.uint64).reshape(1, 4))[0]
                qa = (fa - yy) * pow(alpha - uu, -1, orc.P_FR) % orc.P_FR
                okc = mz.array_to_points(oc[:8])[0] == orc.ec_mul(0, (1, 2), fa)
                oky = yy == orc.poly_eval(orc.FR, cf_cpu, uu)
                okw = mz.array_to_points(oc[8:16])[0] == orc.ec_mul(0, (1, 2), qa)
                e2e.update({"stages_ms": stages, "total_ms": sum(stages.values()), "trapdoor_identities_hold": bool(okc and oky and okw),
                            "commit_and_open_overlapped_ms": max_ov, "overlapped_results_identical": overlapped_same,
                            "total_with_overlap_ms": sum(v for k, v in stages.items() if k not in ("commit_local", "open_local")) + max_ov})
        else:
            e2e["error"] = err or "a rank failed"
        out["e2e_kzg"] = e2e
        if hh:
            L.mzk_srs_free(hh)
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ fixed-size MSM over all ranks (BASELINE configs[3])
    # ONE KZG commit of 2^strong_log2n pairs: rank g builds SRS powers [lo_g, hi_g) on its GPU (tables included),
    # commits its slice, the 128-byte partials are all-gathered and folded.  Strong-scaling view next to the weak-
    # scaling headline: total pairs fixed, time should fall with N.  Verified by the trapdoor identity on rank 0.
    progress("fixed-size (strong scaling) leg")
    if args.strong_log2n > 0:
        tot = 1 << args.strong_log2n
        lo, hi = sharded.shard_range(tot, rank, world)
        m = hi - lo
        alpha = orc.from_limbs(orc.synth_vector(orc.FR, SEED + 901, 1))[0]
        a_l, g_l = mz.to_limbs([alpha], 4), mz.points_to_array([(1, 2)])
        err, hs, step_fn = None, ctypes.c_void_p(), None
        try:
            sc2 = torch.empty(m * 4, dtype=torch.int64, device=dev)
            check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 5000 + 1000003 * rank), ctypes.c_size_t(m), dptr(sc2), stream))
            sp2 = torch.empty(m * 8, dtype=torch.int64, device=dev)
            check(L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(lo),
                                               ctypes.c_size_t(m), dptr(sp2), stream))
            check(L.mzk_srs_from_device(dptr(sp2), ctypes.c_size_t(m), ctypes.byref(hs), stream))
            torch.cuda.synchronize()
            del sp2
            part2 = torch.zeros(16, dtype=torch.int64, device=dev)
            res2 = torch.zeros(8, dtype=torch.int64, device=dev)

            def step_fn():
                check(L.mzk_kzg_commit_srs_dev(hs, dptr(sc2), ctypes.c_size_t(m), dptr(part2), ctypes.c_int(1), stream))
                recs = sharded.all_gather_partials(part2)
                check(L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(recs.shape[0]), dptr(res2), stream))
        except Exception as ex:
            err = str(ex)[:300]
        strong = {"total_pairs": tot, "n_gpus": world, "pairs_per_gpu": m,
                  "what": "one KZG commit of 2^%d pairs, SRS and scalars sharded contiguously over the ranks, all_gather of 128-byte "
                          "partials + fold (BASELINE configs[3]); time should fall with the number of GPUs" % args.strong_log2n}
        if max_over_ranks(0.0 if err is None else 1.0) == 0.0:
            Ks = max(3, min(K, 5))
            sdt, _ = timed(step_fn, Ks, 1)
            strong.update({"ms_per_step": sdt / Ks * 1e3, "value": tot / (sdt / Ks), "unit": "pairs/s"})
            # trapdoor identity: the commitment must be [f(alpha)] G for f = the concatenation of all ranks' scalars
            got2 = mz.array_to_points(res2.cpu().numpy().view(np.uint64))[0]
            if rank == 0:
                cores = orc.usable_threads(64)
                fa, apow = 0, 1
                for r in range(world):
                    rlo, rhi = sharded.shard_range(tot, r, world)
                    sr = orc.synth_vector(orc.FR, SEED + 5000 + 1000003 * r, rhi - rlo, cores)
                    fa = (fa + orc.poly_eval(orc.FR, sr, alpha) * pow(alpha, rlo, orc.P_FR)) % orc.P_FR
                strong["trapdoor_identity_holds"] = bool(got2 == orc.ec_mul(0, (1, 2), fa))
        else:
            strong["error"] = err or "a rank failed"
        out["strong_scaling_msm"] = strong
        if hs:
            L.mzk_srs_free(hs)
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ ONE transform sharded over all ranks (SURVEY 8e four-step layout)
    # A 2^strong_ntt_log2n-point Fr transform whose vector is spread over the ranks in contiguous slices: all-to-all, W-point
    # transforms across the ranks, all-to-all, local n/W-point coset transform (the twiddle rides in the LDE's offset), and a
    # third all-to-all when the result has to be contiguous again (myzkp_amd/sharded.py).  Every rank checks its part against
    # the single-GPU transform of the whole vector, which it computes itself.  At N = 1 this is the plain transform.
    progress("sharded transform leg")
    if args.strong_ntt_log2n > 0 and world * world <= (1 << args.strong_ntt_log2n):
        lgt = args.strong_ntt_log2n
        tot = 1 << lgt
        mt = tot // world
        wt = mz.root_of_unity(mz.FIELD_FR, lgt)
        sn = {"log2n": lgt, "n_gpus": world, "field": "BN254 Fr", "points_per_gpu": mt,
              "what": "one 2^%d-point transform (ntt.rs:7-64), vector sharded over the ranks; all_to_all_single (RCCL) exchanges of "
                      "(N-1)/N of each rank's n/N elements; time should fall with the number of GPUs" % lgt}
        err, fns = None, {}
        try:
            ops = sharded.DeviceOps(mz.FIELD_FR)
            PFR = mz.MODULUS[mz.FIELD_FR]
            if shared_gpu_test:
                _a2a = ops.all_to_all
                ops.all_to_all = lambda b, group=None: _a2a(b.cpu(), group).to(dev)
            full = torch.empty(tot * 4, dtype=torch.int64, device=dev)
            for r in range(world):
                check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 7000 + 1000003 * r), ctypes.c_size_t(mt), ctypes.c_void_p(full.data_ptr() + r * mt * 32), stream))
            xloc = full.view(world, -1)[rank].clone()
            want = torch.empty_like(full)
            rt24 = mz.to_limbs([wt], 4)
            check(L.mzk_ntt_dev(mz.FIELD_FR, rt24.ctypes.data_as(ctypes.c_void_p), dptr(full), dptr(want), ctypes.c_size_t(tot), 0, stream))
            torch.cuda.synchronize()
            want_contig = want.view(world, -1)[rank].clone()
            want_cyclic = want.view(-1, 4)[rank::world].contiguous().view(-1)
            del full, want
            torch.cuda.empty_cache()
            res = {}
            fns["contiguous_to_contiguous"] = lambda: res.__setitem__("cc", sharded.ntt_sharded(xloc, PFR, lgt, wt, ops, False, "contiguous", "contiguous"))
            fns["contiguous_to_cyclic"] = lambda: res.__setitem__("cy", sharded.ntt_sharded(xloc, PFR, lgt, wt, ops, False, "contiguous", "cyclic"))
            fns["inverse_cyclic_to_contiguous"] = lambda: res.__setitem__("inv", sharded.ntt_sharded(want_cyclic, PFR, lgt, wt, ops, True, "cyclic", "contiguous"))
        except Exception as ex:
            err = str(ex)[:300]
        if max_over_ranks(0.0 if err is None else 1.0) == 0.0:
            Kn = max(3, min(K, 5))
            for name, fn in fns.items():
                dtn, _ = timed(fn, Kn, 1)
                sn[name + "_ms"] = dtn / Kn * 1e3
            torch.cuda.synchronize()
            okn = bool(torch.equal(res["cc"], want_contig) and torch.equal(res["cy"], want_cyclic) and torch.equal(res["inv"], xloc))
            sn["every_part_equals_single_gpu_transform"] = max_over_ranks(0.0 if okn else 1.0) == 0.0
            sn.update({"ms_per_step": sn["contiguous_to_contiguous_ms"], "value": tot / (sn["contiguous_to_contiguous_ms"] * 1e-3), "unit": "elems/s",
                       "exchanges": {"contiguous_to_contiguous": 3 if world > 1 else 0, "contiguous_to_cyclic": 2 if world > 1 else 0,
                                     "inverse_cyclic_to_contiguous": 2 if world > 1 else 0},
                       "bytes_sent_per_rank_per_exchange": (world - 1) * (mt // world) * 32})
        else:
            sn["error"] = err or "a rank failed"
        out["strong_scaling_ntt"] = sn
        torch.cuda.empty_cache()

    # ------------------------------------------------------------------ one process, several GPUs, C ABI only (optional)
    if args.inproc_devices and world == 1 and args.strong_log2n > 0:
        ords = [int(x) for x in args.inproc_devices.split(",") if x != ""]
        tot = 1 << args.strong_log2n
        rec = {"devices": ords, "total_pairs": tot,
               "what": "mzk_init_devices + mzk_kzg_setup_srs_multi + mzk_kzg_commit_srs_multi_dev: contiguous shards, every GPU builds its own "
                       "SRS slice and commits it, 128-byte partials gathered through pinned host memory, fold on context 0"}
        try:
            if srs._h:
                L.mzk_srs_free(srs._h); srs._h = None
            torch.cuda.empty_cache()
            mz.init_devices(ords)
            alpha = orc.from_limbs(orc.synth_vector(orc.FR, SEED + 901, 1))[0]
            t0 = time.perf_counter()
            hm = mz.SrsMulti(alpha=alpha, max_d=tot - 1)
            rec["srs_setup_and_tables_ms"] = (time.perf_counter() - t0) * 1e3
            shards, fa = [], 0
            for r, o in enumerate(ords):
                lo_r, hi_r = hm.lo[r], hm.lo[r + 1]
                t = torch.empty((hi_r - lo_r) * 4, dtype=torch.int64, device=torch.device("cuda", o))
                mz.ctx_select(r)
                check(L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 7000 + r), ctypes.c_size_t(hi_r - lo_r), dptr(t), None))
                shards.append(t)
                sr = orc.synth_vector(orc.FR, SEED + 7000 + r, hi_r - lo_r, orc.usable_threads(64))
                fa = (fa + orc.poly_eval(orc.FR, sr, alpha) * pow(alpha, lo_r, orc.P_FR)) % orc.P_FR
            mz.ctx_select(0)
            for o in set(ords):
                torch.cuda.synchronize(o)
            ptrs = [t.data_ptr() for t in shards]
            got = hm.commit_dev(ptrs, tot)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                got = hm.commit_dev(ptrs, tot)
            dt = (time.perf_counter() - t0) / reps
            rec.update({"ms_per_commit": dt * 1e3, "value": tot / dt, "unit": "pairs/s",
                        "trapdoor_identity_holds": bool(got == orc.ec_mul(0, (1, 2), fa))})
            hm.close()
        except Exception as ex:
            rec["error"] = str(ex)[:300]
        finally:
            try:
                mz.init_devices([local_rank] * 4)
            except Exception:
                pass
        out["strong_scaling_msm_single_process"] = rec

    # ------------------------------------------------------------------ CPU baseline (rank 0, N = 1 only, bounded sample)
    if rank == 0 and world == 1 and not args.skip_cpu:
        cores = orc.usable_threads(64)       # threads the oracle's multi-threaded legs run on (cgroup quota / affinity, capped)
        sample = 1 << 11
        s_cpu = orc.synth_vector(orc.FR, SEED, sample, cores)
        p_cpu = orc.synth_points(SEED + 7, sample, cores)
        t0 = time.perf_counter()
        orc.msm_ref(s_cpu, p_cpu)
        dt = time.perf_counter() - t0
        cb = {"value": sample / dt, "unit": "pairs/s", "cores": 1, "kind": "port",
              "sample": "first 2^11 pairs of the same stream through oracle orc_msm_ref (literal restatement of "
                        "polynomial.rs:156-165: affine double-and-add, one inversion per group op); MSM cost is linear in n"}
        # `cpu_fast`: the oracle's own multi-threaded code -- plain Jacobian-coordinate Pippenger with unsigned windows, a plain
        # iterative radix-2 NTT on 4x64-bit CIOS Montgomery limbs, OpenMP over the host cores -- written to be read against the
        # reference, NOT tuned (no signed digits, no endomorphism, no batched affine additions, no assembly): a production CPU
        # library is one to two orders of magnitude faster per core.  Inputs are generated BEFORE the clock starts.
        nn = min(n, 1 << 20)
        s2 = orc.synth_vector(orc.FR, SEED, nn, cores)
        p2 = orc.synth_points(SEED + 7, nn, cores)
        t0 = time.perf_counter()
        orc.msm_fast(s2, p2, cores)
        dt2 = time.perf_counter() - t0
        cb["cpu_fast"] = {"value": nn / dt2, "unit": "pairs/s", "cores": cores, "value_per_core": nn / dt2 / cores,
                          "what": "oracle Pippenger (orc_msm_fast: windows x slices over the cores), 2^%d pairs, inputs generated before the clock; "
                                  "the oracle's un-tuned code, not a CPU library -- do not quote a speed-up from it" % (nn.bit_length() - 1)}
        lgs = 14
        v = orc.synth_vector(orc.FR, SEED + 99, 1 << lgs, cores)
        t0 = time.perf_counter()
        orc.ntt_ref(orc.FR, orc.fr_root(lgs), v)
        dt3 = time.perf_counter() - t0
        vv = orc.synth_vector(orc.FR, SEED + 99, n, cores)
        t0 = time.perf_counter()
        orc.ntt_fast(orc.FR, orc.fr_root(args.log2n), vv, threads=cores)
        dt4 = time.perf_counter() - t0
        cb["ntt"] = {"value": (1 << lgs) / dt3, "unit": "elems/s", "cores": 1, "kind": "port",
                     "sample": "2^14-point oracle orc_ntt_ref (literal ntt.rs:7-48, one pow per output per level; O(n log^2 n), so larger n is slower per element)",
                     "cpu_fast": {"value": n / dt4, "unit": "elems/s", "cores": cores, "value_per_core": n / dt4 / cores,
                                  "what": "oracle iterative radix-2 NTT (orc_ntt_fast), 2^%d points, input generated before the clock; un-tuned "
                                          "checker code (its stages parallelise poorly beyond a few cores)" % args.log2n}}
        cb["host"] = {"os_cpu_count": os.cpu_count(), "threads_used_by_cpu_fast": cores}
        out["cpu_baseline"] = cb
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

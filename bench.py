#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MyZKP MSM / NTT prover path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--log2n 20] [--skip-cpu]

ONE JSON line on stdout (rank 0), under 8 KB: the contract's keys, `roofline`, `cpu_baseline`, the four numbers of
BASELINE.json's metric as top-level scalars (MSM pairs/s and NTT elems/s at 2^20 and 2^24), the GPU's clock and power cap,
and one time per sub-leg (`legs_ms`).  Everything else every leg measures -- phases, parity flags, notes -- goes to
`bench_detail.json` beside this file (`--detail-file`; `detail_file` in the line names it).

A "step" is one pass of the hot path over one batch of synthetic input already resident in HBM:
  * primary (`value`): one KZG commit = BN254 G1 MSM of 2^log2n (scalar, point) pairs PER GPU against a device-resident SRS
    (BASELINE.json configs[2]; weak scaling: the N-GPU job is one MSM of N * 2^log2n pairs, contiguous shards, one
    all-gather of N 128-byte partials over RCCL, local fold -- configs[3] shape);
  * `ntt`: one forward radix-2 NTT of 2^log2n BN254-Fr elements per GPU (configs[1]; N GPUs run N independent transforms;
    ONE transform sharded over the ranks is the strong_scaling_ntt leg).
Both are checked bit-for-bit against the CPU oracle before timing.  `roofline` prices the dominant kernel against HBM
bandwidth as BASELINE.md section 4 defines it (MSM: 96 B per pair, NTT: 2*32 B per element); `alu_roofline` adds the
integer-multiply roofline that actually binds.  One function per leg; main() only orders them.
"""
import argparse, ctypes, glob, json, os, re, sys, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4D595A4B50  # "MYZKP"
HBM_PEAK_GBPS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
MAD_PEAK_PER_S = 256 * 64 * 2.4e9   # v_mad_u64_u32 is half rate: 64 lanes/clk/CU (profiles/r01_ubench_instr_rates.txt)
METRIC = "G1 MSM pairs/sec + NTT elems/sec at 2^20 and 2^24; bit-exact vs CPU"
M128_GEN = 85408008396924667383611388730472331217   # fri.rs:436-438; fast_stark.rs:573-616: offset = generator
N_PHASES = 13
PH_ACC, PH_NTT_TOTAL, PH_MERKLE = 2, 12, 10
LINE_BUDGET_BYTES = 8192


def fail_line(n_gpus, msg, **more):
    """One JSON line that says why there is no measurement (never a silent 1-rank run), then a non-zero exit."""
    print(json.dumps({"metric": METRIC, "value": None, "unit": "pairs/s", "n_gpus": n_gpus, "error": msg, **more}), flush=True)
    return 2


# ---------------------------------------------------------------------------------------------------- recorded counter profiles
def current_fingerprints():
    """Fingerprints of the kernel sources this tree would be built from (tools/source_fingerprint.py)."""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import source_fingerprint as sf
        return {k: sf.fingerprint(k) for k in sf.FAMILIES}, sf.parse_header
    except Exception:
        return {}, (lambda path: {})


def traffic_is_stale(path, family):
    """True when the recorded profile was taken from other kernel sources than this tree's (or does not say: records older
    than round 6 carry no fingerprint).  File dates do not survive a checkout, so the summaries carry a source fingerprint."""
    cur, parse = current_fingerprints()
    rec = parse(path)
    return not (family in cur and rec.get(family) == cur[family])


def recorded_traffic(kernel_prefix, section="KZG commit 2^20"):
    """FETCH_SIZE + WRITE_SIZE per launch (bytes) of a kernel from the newest profiles/r*_hbm_traffic_pmc.txt that names it in
    the given section (`== KZG commit 2^20 ...`, `== generic MSM 2^20 ...`; written by tools/timing/pmc_summary.py under
    rocprofv3 --pmc; the bench itself never runs under the profiler).  None if no such record exists."""
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
                best = {"bytes": int((float(m.group(2)) + float(m.group(3))) * 1024), "stale": traffic_is_stale(path, "msm"),
                        "source": "profiles/%s (recorded rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, separate runs, raw counters; not collected by this run; "
                                  "64-byte table-row gathers are counted 1:1, DESIGN.md section 8)" % os.path.basename(path)}
                break       # first match per file = the KZG-commit section
    return best


def recorded_valu_instructions(kernel_prefix, section="KZG commit 2^20"):
    """Wave-level VALU instructions per launch (SQ_INSTS_VALU, mean) of a kernel from the newest profiles/r*_sq_counters.txt that names it in
    the given section (written by tools/timing/pmc_sq_summary.py under rocprofv3 --pmc, a run of its own).  None if no such record exists."""
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
                best = {"instructions": float(m.group(1)), "source": "profiles/%s" % os.path.basename(path), "stale": traffic_is_stale(path, "msm")}
                in_kernel = False
    return best


def recorded_ntt_traffic(field_tag):
    """Bytes per 2^20-point transform from the newest profiles/r*_hbm_traffic_pmc.txt that has a section for this field
    (`== NTT <field_tag> 2^20`: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/timing/ntt_only.py): the sum over
    the transform's kernels, raw and corrected.  Correction as calibrated on known byte counts (profiles/r02o_hbm_traffic_pmc.txt,
    the microarchitecture guide's rule): FETCH_SIZE tallies a 128-byte request at 64, so streams read in runs of >= 128 bytes
    count double -- every BN254 pass, and the contiguous rows of the M128 last pass; the M128 strided pass reads 64-byte runs
    (4 columns x 16 bytes), counted 1:1.  WRITE_SIZE is exact.  None if no such record exists."""
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
            best = {"bytes": int(corr), "raw_bytes": int(raw), "kernels": kernels, "stale": traffic_is_stale(path, "ntt"),
                    "source": "profiles/%s (recorded rocprofv3 --pmc passes of tools/timing/ntt_only.py; FETCH_SIZE doubled for streams read in runs of >= 128 bytes, "
                              "see recorded_ntt_traffic; not collected by this run)" % os.path.basename(path)}
    return best


# ---------------------------------------------------------------------------------------------------- the GPU's clock, without HIP
def _read(path):
    try:
        with open(path) as f:
            return f.read()
    except OSError:
        return None


def _sclk_levels(text):
    """pp_dpm_sclk -> (list of MHz levels, the level marked `*`)."""
    levels, cur = [], None
    for line in (text or "").splitlines():
        m = re.match(r"\s*\S+:\s*([0-9]+)\s*Mhz\s*(\*)?", line, re.I)
        if m:
            levels.append(int(m.group(1)))
            if m.group(2):
                cur = int(m.group(1))
    return levels, cur


def gpu_sysfs_snapshot(drm_root="/sys/class/drm"):
    """Clock range and power cap of every amdgpu card, from sysfs -- no HIP call, safe in the launcher parent and before the
    runtime starts: {card: {pci, sclk_mhz_max, sclk_mhz_now, power_cap_w}}.  A slow box shows up here (a lower cap or top level)."""
    cards = {}
    for dev_dir in sorted(glob.glob(os.path.join(drm_root, "card[0-9]*", "device"))):
        txt = _read(os.path.join(dev_dir, "pp_dpm_sclk"))
        if txt is None:
            continue
        levels, cur = _sclk_levels(txt)
        cap = None
        for p in glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*", "power1_cap")):
            v = _read(p)
            if v and v.strip().isdigit():
                cap = int(v) / 1e6
        try:
            pci = os.path.basename(os.path.realpath(dev_dir))
        except OSError:
            pci = None
        cards[os.path.basename(os.path.dirname(dev_dir))] = {"pci": pci, "sclk_mhz_max": max(levels) if levels else None, "sclk_mhz_now": cur, "power_cap_w": cap}
    return cards


def gpu_card_of(pci, snapshot):
    for card, rec in snapshot.items():
        if pci and rec.get("pci") and rec["pci"].lower() == pci.lower():
            return card
    return None


def sclk_now_mhz(card, drm_root="/sys/class/drm"):
    """Instantaneous shader clock of one card: hwmon freq1_input (Hz) if present, else the `*` level of pp_dpm_sclk."""
    dev_dir = os.path.join(drm_root, card, "device")
    for p in glob.glob(os.path.join(dev_dir, "hwmon", "hwmon*", "freq1_input")):
        v = _read(p)
        if v and v.strip().isdigit():
            return int(v) / 1e6
    return _sclk_levels(_read(os.path.join(dev_dir, "pp_dpm_sclk")))[1]


# ---------------------------------------------------------------------------------------------------- the launcher
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


def parse_args(argv=None):
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
                    help="skip the side legs (commits in flight, batches, PCIe-inclusive calls, STARK pipeline): profiler runs -- overlapped "
                         "kernels would distort the per-kernel averages")
    ap.add_argument("--sharded-legs-only", action="store_true",
                    help="only the legs with an exchange step (headline, arbitrary-point MSM, e2e KZG, fixed-size MSM, sharded transform)")
    ap.add_argument("--force-process-group", action="store_true",
                    help="with --gpus 1: form a one-rank `nccl` process group anyway and issue every collective of the sharded legs (a gather of "
                         "one record, an all-to-all of one chunk), so that a one-GPU box runs the RCCL branch")
    ap.add_argument("--detail-file", type=str, default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the bulky per-leg records go (the printed line stays under 8 KB); '' = nowhere")
    ap.add_argument("--inproc-devices", type=str, default="",
                    help="comma list of device ordinals: additionally run the fixed-size MSM (configs[3]) from THIS one process over "
                         "those GPUs through the C ABI's mzk_*_multi entry points (no torch.distributed); single-process runs only")
    return ap.parse_args(argv)


# ---------------------------------------------------------------------------------------------------- set-up
class Bench:
    """Everything the legs share: the process group, the library, the synthetic inputs resident in HBM, the SRS handle."""


def setup(args, t_start):
    import numpy as np
    import torch
    import torch.distributed as dist
    import myzkp_amd as mz
    from myzkp_amd import sharded
    B = Bench()
    B.args, B.np, B.torch, B.dist, B.mz, B.sharded, B.t_start = args, np, torch, dist, mz, sharded, t_start
    B.world = world = int(os.environ.get("WORLD_SIZE", "1"))
    B.rank = rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # Rehearsal of the N>1 code path on a ONE-GPU box (the pool's boxes have one): all ranks share cuda:0 and the
    # exchange goes over gloo through host memory.  Timings of such a run mean nothing and the line says so.
    B.shared_gpu_test = shared = world > 1 and os.environ.get("MZK_BENCH_SHARED_GPU_TEST") == "1"
    if shared:
        local_rank = 0
    B.local_rank = local_rank
    if world != args.gpus:
        if rank == 0:
            fail_line(args.gpus, "WORLD_SIZE=%d but --gpus %d: launch `python bench.py --gpus N` (it starts the ranks itself) or "
                                 "torch.distributed.run --nproc-per-node N bench.py --gpus N" % (world, args.gpus))
        sys.exit(2)
    if world > 1 and not shared and torch.cuda.device_count() < world:
        if rank == 0:
            fail_line(args.gpus, "%d ranks but only %d GPU(s) visible" % (world, torch.cuda.device_count()), visible_devices=torch.cuda.device_count())
        sys.exit(2)
    B.forced = forced = bool(args.force_process_group) and world == 1
    B.exchanging = world > 1 or forced            # the steps end in partial -> gather -> fold, not in the affine point directly
    if world > 1 or forced:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if forced and "MASTER_PORT" not in os.environ:
            import socket
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        torch.cuda.set_device(local_rank)
        if shared:
            dist.init_process_group("gloo")
            _real_gather = sharded.all_gather_partials
            sharded.all_gather_partials = lambda t: _real_gather(t.cpu()).to(t.device)
        elif forced:
            dist.init_process_group("nccl", world_size=1, rank=0, device_id=torch.device("cuda", local_rank))
            sharded.FORCE_COLLECTIVES = True
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(0)
    import faulthandler
    if os.environ.get("MZK_BENCH_WATCHDOG_S"):      # dump every thread's stack if the run is still going after that long
        faulthandler.dump_traceback_later(float(os.environ["MZK_BENCH_WATCHDOG_S"]), repeat=True, file=sys.stderr)

    B.dev = dev = torch.device("cuda", local_rank)
    mz.init_devices([local_rank] * 4)     # context 0: every leg; contexts 1..3 (same GPU): further commits in flight
    B.L = L = mz.lib()
    L.mzk_prof_name.restype = ctypes.c_char_p
    B.stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    B.side_legs = world == 1 and not args.no_two_in_flight and not args.sharded_legs_only and not forced
    B.all_legs = not args.sharded_legs_only
    B.K, B.W = args.steps, args.warmup
    B.n = 1 << args.log2n
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orc                     # the CPU oracle: parity asserts before timing, trapdoor expectations, the cpu_baseline leg -- never the thing measured
    B.orc = orc
    return B


def progress(B, msg):
    if os.environ.get("MZK_BENCH_VERBOSE") == "1":
        print("[bench rank %d +%.1fs] %s" % (B.rank, time.perf_counter() - B.t_start, msg), file=sys.stderr, flush=True)


def check(B, rc):
    if rc != 0:
        raise RuntimeError(B.L.mzk_last_error().decode())


def dptr(t):
    return ctypes.c_void_p(t.data_ptr())


def barrier_sync(B):
    B.torch.cuda.synchronize()
    if B.dist.is_initialized():
        B.dist.barrier()
    B.torch.cuda.synchronize()


def max_over_ranks(B, x):
    if B.world == 1:
        return x
    t = B.torch.tensor([x], dtype=B.torch.float64, device="cpu" if B.shared_gpu_test else B.dev)
    B.dist.all_reduce(t, op=B.dist.ReduceOp.MAX)
    return float(t.item())


def synth_shard(B, nn, r):
    """Global problem = world * n pairs; rank g owns indices [g n, (g+1) n) of the global streams (distinct seeds per rank)."""
    torch, L, mz = B.torch, B.L, B.mz
    sc = torch.empty(nn * 4, dtype=torch.int64, device=B.dev)
    pt = torch.empty(nn * 8, dtype=torch.int64, device=B.dev)
    check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 1000003 * r), ctypes.c_size_t(nn), dptr(sc), B.stream))
    check(B, L.mzk_synth_g1_points_dev(ctypes.c_uint64(SEED + 7 + 1000003 * r), ctypes.c_size_t(nn), dptr(pt), B.stream))
    return sc, pt


def make_inputs(B):
    """Synthetic inputs resident in HBM, the device-resident SRS handle (window tables built once, untimed), and the step
    functions of the timed legs."""
    torch, L, mz, n, dev, stream, args = B.torch, B.L, B.mz, B.n, B.dev, B.stream, B.args
    progress(B, "process group up; generating inputs")
    B.scalars, B.points = synth_shard(B, n, B.rank)
    B.ntt_in = torch.empty(n * 4, dtype=torch.int64, device=dev)
    B.ntt_out = torch.empty(n * 4, dtype=torch.int64, device=dev)
    check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 99 + B.rank), ctypes.c_size_t(n), dptr(B.ntt_in), stream))
    B.root = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, args.log2n)], 4)
    B.partial = torch.zeros(16, dtype=torch.int64, device=dev)
    B.result = torch.zeros(8, dtype=torch.int64, device=dev)
    B.result_srs = torch.zeros(8, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    progress(B, "building SRS handle")
    B.srs_h, B.srs_build_ms = None, None
    for attempt in range(2):            # the first build also grows the workspace (hipMalloc): time the second
        hh = ctypes.c_void_p()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        check(B, L.mzk_srs_from_device(dptr(B.points), ctypes.c_size_t(n), ctypes.byref(hh), stream))
        torch.cuda.synchronize()
        B.srs_build_ms = (time.perf_counter() - t0) * 1e3
        if attempt == 0:
            L.mzk_srs_free(hh)
    B.srs_h = hh
    B.srs_window_bits = int(L.mzk_srs_window_bits(B.srs_h))       # the library's default width for an SRS of this size
    B.srs_table_windows = 254 // B.srs_window_bits + 1
    progress(B, "SRS handle built")
    # The HIP runtime stalls once for 35-45 ms a few thousand dispatches into a process (measured: one stall in 120 000
    # launches, at dispatch ~3500) -- get past it before anything is timed.
    tiny = torch.empty(64 * 4, dtype=torch.int64, device=dev)
    for _ in range(6000):
        L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(1), ctypes.c_size_t(64), dptr(tiny), stream)
    torch.cuda.synchronize()
    del tiny
    # M128 (the STARK field): forward NTT of 2^log2n elements, and the low-degree extension fast_coset_evaluate of a
    # degree-2^(log2n-2) polynomial onto a 2^log2n coset (blow-up 4)
    B.m_in = torch.empty(n * 2, dtype=torch.int64, device=dev)
    B.m_out = torch.empty(n * 2, dtype=torch.int64, device=dev)
    check(B, L.mzk_synth_field_dev(mz.FIELD_M128, ctypes.c_uint64(SEED + 199 + B.rank), ctypes.c_size_t(n), dptr(B.m_in), stream))
    B.m_root = mz.to_limbs([mz.root_of_unity(mz.FIELD_M128, args.log2n)], 2)
    B.m_off = mz.to_limbs([M128_GEN], 2)
    B.mk_root, B.mk_len = (ctypes.c_uint8 * 48)(), ctypes.c_size_t()


# ---- the steps of the timed legs
def msm_step(B):
    L, n = B.L, B.n
    if not B.exchanging:   # nothing to exchange: the kernel chain ends in the affine point
        check(B, L.mzk_msm_g1_bn254_dev(dptr(B.scalars), dptr(B.points), ctypes.c_size_t(n), dptr(B.result), B.stream))
        return
    check(B, L.mzk_msm_g1_bn254_partial_dev(dptr(B.scalars), dptr(B.points), ctypes.c_size_t(n), dptr(B.partial), B.stream))
    recs = B.sharded.all_gather_partials(B.partial)
    check(B, L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(recs.shape[0]), dptr(B.result), B.stream))


def srs_step(B):
    L, n = B.L, B.n
    if not B.exchanging:
        check(B, L.mzk_kzg_commit_srs_dev(B.srs_h, dptr(B.scalars), ctypes.c_size_t(n), dptr(B.result_srs), ctypes.c_int(0), B.stream))
        return
    check(B, L.mzk_kzg_commit_srs_dev(B.srs_h, dptr(B.scalars), ctypes.c_size_t(n), dptr(B.partial), ctypes.c_int(1), B.stream))
    recs = B.sharded.all_gather_partials(B.partial)
    check(B, L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(recs.shape[0]), dptr(B.result_srs), B.stream))


def ntt_step(B):
    check(B, B.L.mzk_ntt_dev(B.mz.FIELD_FR, B.root.ctypes.data_as(ctypes.c_void_p), dptr(B.ntt_in), dptr(B.ntt_out), ctypes.c_size_t(B.n), 0, B.stream))


def ntt_m128_step(B):
    check(B, B.L.mzk_ntt_dev(B.mz.FIELD_M128, B.m_root.ctypes.data_as(ctypes.c_void_p), dptr(B.m_in), dptr(B.m_out), ctypes.c_size_t(B.n), 0, B.stream))


def lde_m128_step(B):
    check(B, B.L.mzk_coset_lde_dev(B.mz.FIELD_M128, dptr(B.m_in), ctypes.c_size_t(B.n // 4), B.m_off.ctypes.data_as(ctypes.c_void_p),
                                   B.m_root.ctypes.data_as(ctypes.c_void_p), dptr(B.m_out), ctypes.c_size_t(B.n), B.stream))


def merkle_m128_step(B):
    check(B, B.L.mzk_merkle_commit_field_dev(B.mz.FIELD_M128, dptr(B.m_out), ctypes.c_size_t(B.n), B.mk_root, ctypes.c_size_t(48), ctypes.byref(B.mk_len), B.stream))


# ---------------------------------------------------------------------------------------------------- parity before timing
def leg_parity(B):
    """Bit-exact vs the CPU oracle before anything is timed.  Every rank checks ITS OWN shard on its share of the host cores
    (GPU shard result == oracle Pippenger on the same synthetic streams), then rank 0 checks that the folded N-GPU result ==
    the CPU sum of the N shard points; then the transforms, the M128 extension and the Merkle root of its codeword."""
    torch, L, mz, orc, np, n, dev, stream, rank, world = B.torch, B.L, B.mz, B.orc, B.np, B.n, B.dev, B.stream, B.rank, B.world
    parity = {}
    threads = max(1, orc.usable_threads(64) // world)     # the GPU boxes report 256 CPUs and schedule ~16: more threads are slower
    s_cpu = orc.synth_vector(orc.FR, SEED + 1000003 * rank, n, threads)
    p_cpu = orc.synth_points(SEED + 7 + 1000003 * rank, n, threads)
    ok_inputs = bool(np.array_equal(s_cpu.view(np.int64).reshape(-1), B.scalars.cpu().numpy()) and
                     np.array_equal(p_cpu.view(np.int64).reshape(-1), B.points.cpu().numpy()))
    assert ok_inputs, "GPU synthetic inputs != oracle streams"
    shard_want = orc.msm_fast(s_cpu, p_cpu, threads)
    shard_out = torch.zeros(8, dtype=torch.int64, device=dev)
    check(B, L.mzk_msm_g1_bn254_dev(dptr(B.scalars), dptr(B.points), ctypes.c_size_t(n), dptr(shard_out), stream))
    torch.cuda.synchronize()
    shard_got = mz.array_to_points(shard_out.cpu().numpy().view(np.uint64))[0]
    assert shard_got == shard_want, "rank %d: MSM shard mismatch vs CPU oracle" % rank
    check(B, L.mzk_kzg_commit_srs_dev(B.srs_h, dptr(B.scalars), ctypes.c_size_t(n), dptr(shard_out), ctypes.c_int(0), stream))
    torch.cuda.synchronize()
    assert mz.array_to_points(shard_out.cpu().numpy().view(np.uint64))[0] == shard_want, "rank %d: SRS commit mismatch" % rank
    progress(B, "shard parity ok; exchanging shard points")
    shard_pts = B.sharded.all_gather_partials(shard_out)          # (world, 8) affine shard results
    progress(B, "exchange done")
    msm_step(B); srs_step(B)
    torch.cuda.synchronize()
    if rank == 0:
        want = (0, 0)
        for r in range(world):
            want = orc.ec_add(0, want, mz.array_to_points(shard_pts[r].cpu().numpy().view(np.uint64))[0])
        got = mz.array_to_points(B.result.cpu().numpy().view(np.uint64))[0]
        got2 = mz.array_to_points(B.result_srs.cpu().numpy().view(np.uint64))[0]
        parity["msm_bit_exact_vs_cpu"] = bool(got == want)
        parity["kzg_commit_srs_bit_exact_vs_cpu"] = bool(got2 == want)
        assert got == want and got2 == want, "folded N-GPU MSM mismatch vs CPU"
    progress(B, "folded MSM parity ok")
    del s_cpu, p_cpu
    if not B.all_legs:
        return parity
    ntt_step(B)
    torch.cuda.synchronize()
    v_cpu = orc.synth_vector(orc.FR, SEED + 99 + rank, n, threads)
    rc, want_ntt = orc.ntt_fast(orc.FR, mz.from_limbs(B.root)[0], v_cpu, threads=threads)
    ok = rc == 0 and np.array_equal(want_ntt.view(np.int64).reshape(-1), B.ntt_out.cpu().numpy())
    assert ok, "rank %d: NTT mismatch vs CPU oracle" % rank
    parity["ntt_bit_exact_vs_cpu"] = bool(ok)
    lde_m128_step(B)
    torch.cuda.synchronize()
    c_cpu = orc.synth_vector(orc.M128, SEED + 199 + rank, n // 4, threads)
    # full-size check: scale on the CPU with a running power, then the oracle's iterative NTT
    sc_all = np.zeros((n, 2), dtype=np.uint64)
    vals = orc.from_limbs(c_cpu)
    acc = 1
    for i, x in enumerate(vals):
        vals[i] = x * acc % orc.P_M128
        acc = acc * M128_GEN % orc.P_M128
    sc_all[: n // 4] = orc.to_limbs(vals, 2)
    rc, want_lde = orc.ntt_fast(orc.M128, mz.from_limbs(B.m_root)[0], sc_all, threads=threads)
    ok_lde = rc == 0 and np.array_equal(want_lde.view(np.int64).reshape(-1), B.m_out.cpu().numpy())
    assert ok_lde, "rank %d: M128 coset LDE mismatch vs CPU oracle" % rank
    parity["m128_coset_lde_bit_exact_vs_cpu"] = bool(ok_lde)
    # Merkle root of that LDE codeword (fri.rs:160-166) vs the oracle's literal recursion over the same leaves
    mroot, mlen = (ctypes.c_uint8 * 48)(), ctypes.c_size_t()
    check(B, L.mzk_merkle_commit_field_dev(mz.FIELD_M128, dptr(B.m_out), ctypes.c_size_t(n), mroot, ctypes.c_size_t(48), ctypes.byref(mlen), stream))
    ok_merkle = bytes(mroot[:mlen.value]) == orc.merkle_commit_field_ref(orc.M128, want_lde)
    assert ok_merkle, "rank %d: Merkle root mismatch vs CPU oracle" % rank
    parity["m128_codeword_merkle_root_bit_exact_vs_cpu"] = bool(ok_merkle)
    return parity


# ---------------------------------------------------------------------------------------------------- the contract's timed region
def read_phases(B):
    out = {}
    for ph in range(N_PHASES):
        ms, cnt = ctypes.c_double(0), ctypes.c_uint64(0)
        check(B, B.L.mzk_prof_read(ph, ctypes.byref(ms), ctypes.byref(cnt)))
        if cnt.value:
            out[B.L.mzk_prof_name(ph).decode()] = {"avg_ms": ms.value / cnt.value, "launches": cnt.value}
    return out


def timed(B, step, K, W, price_mask=0, priced_in_timed_region=True):
    """The contract's timed region: W warm-up steps, then EXACTLY K steps between barrier + synchronize; returns its wall time
    (max over ranks) and the kernel-phase averages of two further, untimed passes.
    priced_in_timed_region (the headline and the generic MSM): the one priced kernel (price_mask) carries its HIP-event pair INSIDE the
    timed region, as the roofline contract asks (0.1 % of a 1.4-ms step).  False (the short sub-legs: transforms, LDE, Merkle): the
    timed region carries NO event -- the pair's two marker packets are ~5-8 us, 7-14 % of such a step -- so `ms_per_step` and
    `value` ARE the contract's timed region, and the priced kernel group is measured in a pass of its own
    (`*_in_priced_pass`, step time `ms_per_step_with_event_pair`)."""
    L, torch = B.L, B.torch
    # settle: workspace growth / plan building, and the DVFS ramp -- on MI355X the same kernel runs ~25 % slower during the first
    # few hundred ms after idle (tools/microbench/mulv.hip: 136 -> 174 G mul/s).  Never counted; the W warm-up steps follow.
    t_settle = time.perf_counter()
    # The exit decision is COLLECTIVE: a step may contain an all-gather, so every rank must run the same
    # number of settle steps (a per-rank clock desynchronises the ranks and deadlocks the next collective).
    while True:
        step()
        torch.cuda.synchronize()
        if max_over_ranks(B, 1.0 if time.perf_counter() - t_settle > B.args.settle_s else 0.0) > 0.0:
            break
    for _ in range(W):
        step()
    # timed region: K steps; only the priced kernel carries a HIP-event pair (on the launch stream) -- an event pair
    # costs a few microseconds of stream time, ten of them per step distorted the 0.15 ms NTT step by ~10 %
    L.mzk_prof_reset()
    L.mzk_prof_select(ctypes.c_uint32(price_mask if priced_in_timed_region else 0))
    L.mzk_prof_enable(1 if priced_in_timed_region else 0)
    barrier_sync(B)
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    barrier_sync(B)
    dt = time.perf_counter() - t0
    L.mzk_prof_enable(0)
    priced = read_phases(B)
    priced_tag, priced_step_ms = "_in_timed_region", None
    if not priced_in_timed_region:
        # the priced kernel group, in K further steps of its own (one event pair per step on the launch stream)
        L.mzk_prof_reset()
        L.mzk_prof_select(ctypes.c_uint32(price_mask))
        L.mzk_prof_enable(1)
        barrier_sync(B)
        tp = time.perf_counter()
        for _ in range(K):
            step()
        barrier_sync(B)
        priced_step_ms = max_over_ranks(B, time.perf_counter() - tp) / K * 1e3
        L.mzk_prof_enable(0)
        priced = read_phases(B)
        priced_tag = "_in_priced_pass"
    # second pass of K identical steps, NOT timed: event pairs around every kernel group for the phase breakdown
    L.mzk_prof_reset()
    L.mzk_prof_select(ctypes.c_uint32(0xffffffff))
    L.mzk_prof_enable(1)
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    L.mzk_prof_enable(0)
    phases = read_phases(B)
    L.mzk_prof_reset()
    for k, v in priced.items():
        phases[k + priced_tag] = v
    if priced_in_timed_region:
        # a further pass, K steps with NO event at all: what the event pair of the timed region costs the step.  Reported beside
        # ms_per_step (`ms_per_step_without_event_pair`), never instead of it.
        barrier_sync(B)
        t1 = time.perf_counter()
        for _ in range(K):
            step()
        barrier_sync(B)
        phases["_ms_per_step_without_events"] = max_over_ranks(B, time.perf_counter() - t1) / K * 1e3
    else:
        phases["_ms_per_step_with_event_pair"] = priced_step_ms
    return max_over_ranks(B, dt), phases


def leg_clock_under_load(B, step, card):
    """The shader clock while the headline step runs: a further, UNTIMED pass of the same step (160 steps on every rank), sysfs sampled
    from the host between enqueues (reading sysfs is no HIP call).  None when the card's sysfs node is not visible from here."""
    samples = []
    for _ in range(40):                  # a FIXED number of steps: the step may hold a collective, every rank must run as many as the others
        for _ in range(4):
            step()
        v = sclk_now_mhz(card) if card is not None else None
        if v:
            samples.append(v)
        B.torch.cuda.synchronize()
    if not samples:
        return None
    samples.sort()
    return {"median_mhz": samples[len(samples) // 2], "min_mhz": samples[0], "max_mhz": samples[-1], "samples": len(samples)}


# ---------------------------------------------------------------------------------------------------- side legs (N = 1)
def leg_in_flight(B, nctx, handle=None, what="KZG commit"):
    """`nctx` commits in flight through ONE C-ABI call per batch: mzk_kzg_commit_srs_batch_dev spreads the polynomials of
    a batch over the contexts of this GPU (the process made four in setup; max_in_flight = nctx).  A prover commits to many
    polynomials against one SRS, and the last ~0.2 ms of a commit (bucket reduction, inversion) run on a nearly idle GPU: with
    two / four contexts on the SAME device that tail and the memory-bound sort overlap the other commits' accumulate.
    handle: the SRS handle to commit against (default: the one with window tables)."""
    if not B.side_legs:
        return None
    torch, L, mz, n, dev, stream, K = B.torch, B.L, B.mz, B.n, B.dev, B.stream, B.K
    hsrs = B.srs_h if handle is None else handle
    try:
        batch = 32          # polynomials per call: the pipeline drains at the end of every call, so short batches overlap less
        coefs = torch.empty(batch * n * 4, dtype=torch.int64, device=dev)
        for k in range(batch):
            seed_k = SEED if k == 0 else SEED + 4242 + k
            check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(seed_k), ctypes.c_size_t(n), ctypes.c_void_p(coefs.data_ptr() + k * n * 32), stream))
        outs = torch.zeros(8 * batch, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        lanes = ctypes.c_int(nctx)

        def one_batch():
            check(B, L.mzk_kzg_commit_srs_batch_dev(hsrs, dptr(coefs), ctypes.c_size_t(n), ctypes.c_size_t(batch), dptr(outs), lanes, stream))
        for _ in range(4):
            one_batch()
        torch.cuda.synchronize()
        reps = max(4, K // 2)
        t0 = time.perf_counter()
        for _ in range(reps):
            one_batch()
        torch.cuda.synchronize()
        dtp = (time.perf_counter() - t0) / (reps * batch)
        check(B, L.mzk_kzg_commit_srs_dev(B.srs_h, dptr(B.scalars), ctypes.c_size_t(n), dptr(B.result_srs), ctypes.c_int(0), stream))
        torch.cuda.synchronize()
        same = bool(torch.equal(outs[:8], B.result_srs))       # polynomial 0 of the batch is `scalars`: same point as the single call
        return {"metric": "%s pairs/s, batches of %d polynomials through mzk_kzg_commit_srs_batch_dev with %d commits in flight "
                          "(%d contexts on one GPU, shared SRS handle)" % (what, batch, nctx, nctx),
                "value": n / dtp, "unit": "pairs/s", "ms_per_commit": dtp * 1e3, "commits_timed": reps * batch,
                "same_point_as_single_call": same}
    except Exception as ex:
        return {"error": str(ex)[:300]}


def leg_no_tables_in_flight(B):
    """The same against a handle WITHOUT window tables (prepared points only: the GLV / Horner layout of the generic MSM, no
    precomputation beyond the Montgomery conversion): what several commits in flight are worth where 40 % of one MSM is
    sort, reduction tails and the window Horner."""
    if not B.side_legs:
        return None
    L = B.L
    hplain = ctypes.c_void_p()
    try:
        check(B, L.mzk_srs_from_device_ex(dptr(B.points), ctypes.c_size_t(B.n), ctypes.c_int(0), ctypes.byref(hplain), B.stream))
        return {"one_at_a_time": leg_in_flight(B, 1, hplain, "G1 MSM against prepared points (no window tables)"),
                "four_in_flight": leg_in_flight(B, 4, hplain, "G1 MSM against prepared points (no window tables)")}
    except Exception as ex:
        return {"error": str(ex)[:300]}
    finally:
        if hplain:
            L.mzk_srs_free(hplain)
        B.torch.cuda.empty_cache()


def leg_other_width(B, c):
    """The same commit against a handle with c-bit windows (mzk_srs_from_device_ex).  `value` uses the library's default
    width for the size (17 bits from 2^19 points: profiles/r03b_window_sweep.txt); BASELINE configs[2] names 16-bit
    windows, so that width is always reported beside it as its own leg."""
    if not B.side_legs:
        return None
    torch, L, n, K, W = B.torch, B.L, B.n, B.K, B.W
    hw = ctypes.c_void_p()
    try:
        check(B, L.mzk_srs_from_device_ex(dptr(B.points), ctypes.c_size_t(n), ctypes.c_int(c), ctypes.byref(hw), B.stream))
        outw = torch.zeros(8, dtype=torch.int64, device=B.dev)

        def stepw():
            check(B, L.mzk_kzg_commit_srs_dev(hw, dptr(B.scalars), ctypes.c_size_t(n), dptr(outw), ctypes.c_int(0), B.stream))
        for _ in range(max(W, 2)):
            stepw()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            stepw()
        torch.cuda.synchronize()
        dtw = (time.perf_counter() - t0) / K
        return {"window_bits": c, "tables": 254 // c + 1, "ms_per_step": dtw * 1e3, "value": n / dtw, "unit": "pairs/s",
                "same_point_as_default_width": bool(torch.equal(outw, B.result_srs))}
    except Exception as ex:
        return {"error": str(ex)[:300]}
    finally:
        if hw:
            L.mzk_srs_free(hw)
        torch.cuda.empty_cache()


def _clock(B, fn, reps):
    for _ in range(3 if reps > 2 else 1):
        fn()
    B.torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    B.torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def leg_small_batch(B):
    """The reference's actual call pattern: hundreds of SHORT polynomials committed against one pk in a loop (das/avail.rs:88-98
    per row, das/eigenda.rs:92-101 per chunk, algebra/gemini.rs:112-114).  One call for the batch (mzk_kzg_commit_srs_many_dev:
    the whole batch as one bucket problem; with mzk_srs_build_direct no buckets at all) against the same batch one commit at a
    time; first and last point of every batch checked against the oracle's Pippenger, all of them against the single calls."""
    if not B.side_legs:
        return None
    torch, L, mz, orc, np, dev, stream, K = B.torch, B.L, B.mz, B.orc, B.np, B.dev, B.stream, B.K
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
            check(B, L.mzk_synth_g1_points_dev(ctypes.c_uint64(SEED + 31), ctypes.c_size_t(nn), dptr(pt), stream))
            check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 32), ctypes.c_size_t(count * nn), dptr(cf), stream))
            check(B, L.mzk_srs_from_device(dptr(pt), ctypes.c_size_t(nn), ctypes.byref(hs), stream))
            o_many = torch.zeros(count * 8, dtype=torch.int64, device=dev)
            o_one = torch.zeros(count * 8, dtype=torch.int64, device=dev)

            def many():
                check(B, L.mzk_kzg_commit_srs_many_dev(hs, dptr(cf), ctypes.c_size_t(nn), ctypes.c_size_t(count), dptr(o_many), stream))

            def loop():
                for k in range(count):
                    check(B, L.mzk_kzg_commit_srs_dev(hs, ctypes.c_void_p(cf.data_ptr() + k * nn * 32), ctypes.c_size_t(nn),
                                                      ctypes.c_void_p(o_one.data_ptr() + k * 64), ctypes.c_int(0), stream))
            e = {"polynomials": count, "coefficients_each": nn, "window_bits": int(L.mzk_srs_window_bits(hs))}
            e["one_at_a_time_ms"] = _clock(B, loop, 2)
            e["one_call_ms"] = _clock(B, many, max(K, 30))
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
                    check(B, L.mzk_kzg_commit_srs_many_dev(hs, dptr(cf31), ctypes.c_size_t(nn), ctypes.c_size_t(count), dptr(o31), stream))
                d31 = {"one_call_ms": _clock(B, many31, max(K, 30))}
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
                check(B, L.mzk_srs_build_direct(hs, ctypes.c_int(bits), ctypes.c_size_t(16 << 30), stream))
                build_ms = (time.perf_counter() - t0) * 1e3
                o_many.zero_()
                d = {"table_bytes": int(L.mzk_srs_table_bytes(hs) - b0), "table_build_ms": build_ms, "one_call_ms": _clock(B, many, max(K, 30))}
                d["same_points_as_single_calls"] = bool(torch.equal(o_many, o_one))
                d["us_per_commit"] = d["one_call_ms"] / count * 1e3
                e["direct_tables_%d_bit" % bits] = d
            # open_kzg per polynomial at its own point (das/avail.rs:132 opens per cell): one call (over the 12-bit direct tables
            # built above) against one mzk_kzg_open_srs_dev per polynomial on the same handle
            us_h = orc.synth_vector(orc.FR, SEED + 33, count)
            ys_m = torch.zeros(count * 4, dtype=torch.int64, device=dev); ws_m = torch.zeros(count * 8, dtype=torch.int64, device=dev)
            ys_1 = torch.zeros(count * 4, dtype=torch.int64, device=dev); ws_1 = torch.zeros(count * 8, dtype=torch.int64, device=dev)

            def open_many():
                check(B, L.mzk_kzg_open_srs_many_dev(hs, dptr(cf), ctypes.c_size_t(nn), ctypes.c_size_t(count), us_h.ctypes.data_as(ctypes.c_void_p), dptr(ys_m), dptr(ws_m), stream))

            def open_loop():
                for k in range(count):
                    check(B, L.mzk_kzg_open_srs_dev(hs, ctypes.c_void_p(cf.data_ptr() + k * nn * 32), ctypes.c_size_t(nn), us_h[k].ctypes.data_as(ctypes.c_void_p),
                                                    ctypes.c_void_p(ys_1.data_ptr() + k * 32), ctypes.c_void_p(ws_1.data_ptr() + k * 64), stream))
            o = {"one_at_a_time_ms": _clock(B, open_loop, 1), "one_call_ms": _clock(B, open_many, max(K, 30))}
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


def leg_ntt_batched(B):
    """Many transforms per call (mzk_ntt_batch_dev): a prover interpolates / extends every column of a trace, and a batch
    gives the kernels several rounds of workgroups per CU, i.e. loads and stores under other tiles' butterflies."""
    if not B.side_legs:
        return None
    torch, L, mz, dev, stream, K, args = B.torch, B.L, B.mz, B.dev, B.stream, B.K, B.args
    res = {}
    try:
        for lgb, batch in ((args.log2n, 16), (16, 64), (12, 64)):
            if lgb > args.log2n:
                continue
            nb = 1 << lgb
            vb = torch.empty(batch * nb * 4, dtype=torch.int64, device=dev)
            for k in range(batch):
                check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 7000 + k), ctypes.c_size_t(nb), ctypes.c_void_p(vb.data_ptr() + k * nb * 32), stream))
            ob = torch.empty_like(vb)
            rb = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, lgb)], 4)

            def stepb():
                check(B, L.mzk_ntt_batch_dev(mz.FIELD_FR, rb.ctypes.data_as(ctypes.c_void_p), dptr(vb), dptr(ob), ctypes.c_size_t(nb), ctypes.c_size_t(batch), 0, stream))
            dtb = _clock(B, stepb, max(K, 10)) * 1e-3
            # row 0 of the batch against the single-transform entry point
            one = torch.empty(nb * 4, dtype=torch.int64, device=dev)
            check(B, L.mzk_ntt_dev(mz.FIELD_FR, rb.ctypes.data_as(ctypes.c_void_p), dptr(vb), dptr(one), ctypes.c_size_t(nb), 0, stream))
            torch.cuda.synchronize()
            res["%d x 2^%d" % (batch, lgb)] = {"ms_per_call": dtb * 1e3, "ms_per_transform": dtb * 1e3 / batch, "value": batch * nb / dtb, "unit": "elems/s",
                                              "row_0_equals_single_transform": bool(torch.equal(one, ob[:nb * 4]))}
            del vb, ob, one
            torch.cuda.empty_cache()
        # the STARK side: low-degree extension (blow-up 4) of 64 trace columns of 2^14 coefficients, M128
        nc, order, batch = 1 << 14, 1 << 16, 64
        cb = torch.empty(batch * nc * 2, dtype=torch.int64, device=dev)
        for k in range(batch):
            check(B, L.mzk_synth_field_dev(mz.FIELD_M128, ctypes.c_uint64(SEED + 8000 + k), ctypes.c_size_t(nc), ctypes.c_void_p(cb.data_ptr() + k * nc * 16), stream))
        ob = torch.empty(batch * order * 2, dtype=torch.int64, device=dev)
        gen_l = mz.to_limbs([mz.root_of_unity(mz.FIELD_M128, 16)], 2)
        off_l = mz.to_limbs([M128_GEN], 2)

        def stepl():
            check(B, L.mzk_coset_lde_batch_dev(mz.FIELD_M128, dptr(cb), ctypes.c_size_t(nc), off_l.ctypes.data_as(ctypes.c_void_p), gen_l.ctypes.data_as(ctypes.c_void_p),
                                               dptr(ob), ctypes.c_size_t(order), ctypes.c_size_t(batch), stream))
        dtl = _clock(B, stepl, max(K, 10)) * 1e-3
        one = torch.empty(order * 2, dtype=torch.int64, device=dev)
        check(B, L.mzk_coset_lde_dev(mz.FIELD_M128, dptr(cb), ctypes.c_size_t(nc), off_l.ctypes.data_as(ctypes.c_void_p), gen_l.ctypes.data_as(ctypes.c_void_p),
                                     dptr(one), ctypes.c_size_t(order), stream))
        torch.cuda.synchronize()
        res["coset_lde_m128 64 x (2^14 -> 2^16)"] = {"ms_per_call": dtl * 1e3, "ms_per_column": dtl * 1e3 / batch, "value": batch * order / dtl, "unit": "out elems/s",
                                                    "row_0_equals_single_call": bool(torch.equal(one, ob[:order * 2]))}
    except Exception as ex:
        res["error"] = str(ex)[:300]
    return res


def leg_pcie_inclusive(B):
    """The host-buffer entry points a MyZKP caller binds first (Vec<u64> limbs in, point / vector
    out), timed with the transfers inside: pageable host memory.  Reported beside `value`, never as `value` (inputs of `value`
    are resident in HBM)."""
    if not B.side_legs:
        return None
    L, mz, np, n, args = B.L, B.mz, B.np, B.n, B.args
    res = {}
    try:
        hs, hp = B.scalars.cpu().numpy().view(np.uint64).reshape(-1, 4).copy(), B.points.cpu().numpy().view(np.uint64).reshape(-1, 8).copy()
        hv = B.ntt_in.cpu().numpy().view(np.uint64).reshape(-1, 4).copy()
        wr = mz.root_of_unity(mz.FIELD_FR, args.log2n)
        hcommit_out = np.zeros((1, 8), dtype=np.uint64)

        def commit_host():                    # the bench's device-resident handle, host coefficients (no wrapper object that could free it)
            check(B, L.mzk_kzg_commit_srs(B.srs_h, hs.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n), hcommit_out.ctypes.data_as(ctypes.c_void_p)))
        hout = np.zeros_like(hv)             # the caller's output vector, allocated once (a fresh one per call costs page faults)
        wl = mz.to_limbs([wr], 4)

        def ntt_host():
            check(B, L.mzk_ntt(mz.FIELD_FR, wl.ctypes.data_as(ctypes.c_void_p), hv.ctypes.data_as(ctypes.c_void_p), hout.ctypes.data_as(ctypes.c_void_p),
                               ctypes.c_size_t(n), 0))
        legs = (("msm_g1_bn254_host_buffers", lambda: mz.msm_g1(hs, hp), n, "pairs/s", 96 * n),
                ("kzg_commit_srs_host_scalars", commit_host, n, "pairs/s", 32 * n),
                ("ntt_host_buffers", ntt_host, n, "elems/s", 64 * n))
        for name, fn, units, unit, nbytes in legs:
            fn(); fn()
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            dtp = (time.perf_counter() - t0) / reps
            res[name] = {"ms_per_call": dtp * 1e3, "value": units / dtp, "unit": unit, "bytes_over_pcie": nbytes}
        res["msm_matches_resident_result"] = bool(mz.msm_g1(hs, hp) == mz.array_to_points(B.result.cpu().numpy().view(np.uint64))[0])
        res["commit_matches_resident_result"] = bool(mz.array_to_points(hcommit_out)[0] == mz.array_to_points(B.result_srs.cpu().numpy().view(np.uint64))[0])
    except Exception as ex:
        res["error"] = str(ex)[:300]
    return res


def leg_stark_commit_pipeline(B, lgt=14, regs=16):
    """The commit side of the reference's STARK prover on M128, stage by stage (fast_stark.rs:209-337): interpolate the trace
    registers over the omicron domain (fast_stark.rs:209-229 -> ntt.rs:225-252), fast_coset_evaluate every register onto the FRI
    domain (:231, blow-up 4), Merkle::commit every codeword (:231-243), FRI::commit one codeword with the rounds' trees kept for
    the query phase (fri.rs:144-209).  One batched call per stage, codewords resident in HBM from the extension on.  Before
    timing: register 0's polynomial against the oracle's subproduct-tree interpolation, its codeword against the oracle's
    coset evaluation, its Merkle root and the first FRI fold against the oracle's."""
    if not B.side_legs:
        return None
    import hashlib
    torch, L, mz, orc, np, dev, stream = B.torch, B.L, B.mz, B.orc, B.np, B.dev, B.stream
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
            check(B, L.mzk_coset_lde_batch_dev(fid, dptr(d_coefs), ctypes.c_size_t(cycles), off_l.ctypes.data_as(ctypes.c_void_p), gen_l.ctypes.data_as(ctypes.c_void_p),
                                               dptr(d_cw), ctypes.c_size_t(n_fri), ctypes.c_size_t(regs), stream))
            torch.cuda.synchronize()

        def stage_merkle():
            check(B, L.mzk_merkle_commit_field_batch_dev(fid, dptr(d_cw), ctypes.c_size_t(n_fri), ctypes.c_size_t(regs), roots, stream))

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
        # the same 16 registers over a domain that is NOT a subgroup prefix (arbitrary points): the subproduct-tree path of the same call,
        # timed beside the trace domain's inverse-transform path
        dom_arb = orc.synth_vector(orc.M128, 4242, cycles)

        def stage_interpolate_arbitrary():
            d_trace = torch.from_numpy(trace_flat).to(dev)
            mz.fast_interpolate_batch_dev(fid, dom_arb, d_trace.data_ptr(), regs, omicron, 1 << lgt, d_coefs_arb.data_ptr(), torch.cuda.current_stream().cuda_stream)
        d_coefs_arb = torch.zeros(regs * cycles * 2, dtype=torch.int64, device=dev)
        stage_interpolate_arbitrary()
        best = {}
        for rep in range(4):
            for name, fn in (("interpolate_host", stage_interpolate), ("interpolate", stage_interpolate_hbm), ("interpolate_arbitrary", stage_interpolate_arbitrary),
                             ("coset_lde", stage_lde), ("merkle_commit", stage_merkle), ("fri_commit", stage_fri)):
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
                "interpolate_%d_registers_arbitrary_domain_ms" % regs: best["interpolate_arbitrary"],
                "fri_us_per_round": best["fri_commit"] / rounds * 1e3,
                "parity": {"interpolate_vs_oracle": ok_interp, "interpolate_hbm_form_equals_host_form": ok_hbm, "coset_lde_vs_oracle": ok_lde, "merkle_root_vs_oracle": ok_root,
                           "first_fri_fold_vs_oracle": ok_fold},
                "note": "best of three repetitions per stage, each bracketed by a device synchronize; the trace is uploaded inside the interpolation stage "
                        "(mzk_fast_interpolate_batch_dev; the host-buffer form, coefficients back over PCIe, is timed beside it), everything after "
                        "it stays in HBM; the challenge callback hashes on the host as the reference's transcript does.  The trace domain omicron^i is a "
                        "prefix of a power-of-two subgroup: its interpolation is one inverse transform plus three dot products per register; the same "
                        "call on arbitrary points (subproduct tree) is the _arbitrary_domain_ figure"}
    except Exception as ex:
        return {"error": str(ex)[:300]}


def leg_copy_rate(B):
    """Achievable HBM copy bandwidth on THIS box (SURVEY 8d asks for it next to the nominal 8 TB/s): the library's own
    16-byte-per-lane copy kernel (mzk_selftest_copy_dev; read + write counted), torch's Tensor.copy_ beside it."""
    torch, L = B.torch, B.L
    cp_a = torch.empty(1 << 28, dtype=torch.int32, device=B.dev)   # 1 GiB
    cp_b = torch.empty_like(cp_a)

    def copy_rate(fn):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        return 5 * 2 * cp_a.numel() * 4 / (time.perf_counter() - t0) / 1e9
    gbps_torch = copy_rate(lambda: cp_b.copy_(cp_a))
    gbps = copy_rate(lambda: check(B, L.mzk_selftest_copy_dev(dptr(cp_a), dptr(cp_b), ctypes.c_size_t(cp_a.numel() * 4), B.stream)))
    same = bool(torch.equal(cp_a[:1 << 20], cp_b[:1 << 20]) and torch.equal(cp_a[-(1 << 20):], cp_b[-(1 << 20):]))
    del cp_a, cp_b
    torch.cuda.empty_cache()
    return {"GBps_library_copy_kernel": gbps, "GBps_torch_copy_": gbps_torch, "bytes": 2 << 30, "copied_correctly": same,
            "note": "read + write of 1 GiB each way, five repetitions; every frac_of_measured_copy_rate is against the library kernel's figure"}


# ---------------------------------------------------------------------------------------------------- other sizes, sharded legs
def leg_extra_sizes(B, card=None):
    """Extra sizes (single GPU view, rank 0 only, a few repetitions each): the 2^24 half of BASELINE.json's metric.  The shader clock is
    sampled while each leg's kernels run: under the same power cap the 2^24 accumulate -- 13 GiB of tables gathered from HBM -- runs
    ~12 % below the clock of the 2^20 one (profiles/round6_accumulate_2p20_vs_2p24_counters.txt), which is most of its per-addition gap."""
    torch, L, mz, args, dev, stream = B.torch, B.L, B.mz, B.args, B.dev, B.stream
    extras = {}
    if B.rank != 0 or not B.all_legs:
        return extras
    for lg in [int(x) for x in args.extra_sizes.split(",") if x]:
        if lg == args.log2n:
            continue
        try:
            nn = 1 << lg
            sc, pt = synth_shard(B, nn, 0)
            vin = torch.empty(nn * 4, dtype=torch.int64, device=dev)
            vout = torch.empty(nn * 4, dtype=torch.int64, device=dev)
            check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 99), ctypes.c_size_t(nn), dptr(vin), stream))
            rt = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, lg)], 4)
            res = torch.zeros(8, dtype=torch.int64, device=dev)

            def m():
                check(B, L.mzk_msm_g1_bn254_dev(dptr(sc), dptr(pt), ctypes.c_size_t(nn), dptr(res), stream))

            def t():
                check(B, L.mzk_ntt_dev(mz.FIELD_FR, rt.ctypes.data_as(ctypes.c_void_p), dptr(vin), dptr(vout), ctypes.c_size_t(nn), 0, stream))
            e = {}
            hx = ctypes.c_void_p()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            check(B, L.mzk_srs_from_device(dptr(pt), ctypes.c_size_t(nn), ctypes.byref(hx), stream))
            torch.cuda.synchronize()
            e["srs_table_build_ms"] = (time.perf_counter() - t0) * 1e3
            e["srs_window_bits"] = int(L.mzk_srs_window_bits(hx))

            def c():
                check(B, L.mzk_kzg_commit_srs_dev(hx, dptr(sc), ctypes.c_size_t(nn), dptr(res), ctypes.c_int(0), stream))
            for name, fn, reps in (("kzg_commit_srs", c, 3), ("msm_generic", m, 2), ("ntt", t, 3)):
                fn(); torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(reps):
                    fn()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / reps
                e[name] = {"ms": dt * 1e3, "rate": nn / dt, "hbm_frac": (64.0 if name == "ntt" else 96.0) * nn / dt / 1e9 / HBM_PEAK_GBPS}
                if card is not None and dt > 2e-3:          # an untimed further call, the clock read while its kernels run
                    fn()
                    time.sleep(dt * 0.6)
                    e[name]["sclk_mhz_mid_call"] = sclk_now_mhz(card)
                    torch.cuda.synchronize()
            # round trip property at this size: intt(ntt(x)) == x
            check(B, L.mzk_ntt_dev(mz.FIELD_FR, rt.ctypes.data_as(ctypes.c_void_p), dptr(vout), dptr(vout), ctypes.c_size_t(nn), 1, stream))
            torch.cuda.synchronize()
            e["ntt_roundtrip_ok"] = bool(torch.equal(vin, vout))
            extras["2^%d" % lg] = e
            L.mzk_srs_free(hx)
            del sc, pt, vin, vout
            torch.cuda.empty_cache()
        except Exception as ex:  # an extra must never sink the headline line
            extras["2^%d" % lg] = {"error": str(ex)[:200]}
    return extras


def leg_e2e_kzg(B):
    """BASELINE configs[4]: end-to-end KZG at degree 2^e2e, 1 vs N GPUs.  evaluations -> iNTT -> setup(alpha) -> commit ->
    open(u).  Strong scaling over the N ranks of this job: the (cheap) iNTT is replicated, rank g builds SRS powers [lo, hi),
    commits its slice of the coefficients, computes ITS slice of the quotient (one 32-byte value per rank exchanged:
    sharded.sharded_open_quotient) and commits that; the partials are all-gathered (2 x N x 128 B) and folded."""
    torch, L, mz, orc, np, args, dev, stream, rank, world, sharded = B.torch, B.L, B.mz, B.orc, B.np, B.args, B.dev, B.stream, B.rank, B.world, B.sharded
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
        sp = torch.empty((hi - lo) * 8, dtype=torch.int64, device=dev)
        check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 555), ctypes.c_size_t(nn), dptr(ev), stream))
        rt = mz.to_limbs([mz.root_of_unity(mz.FIELD_FR, lg)], 4)
        alpha = orc.from_limbs(orc.synth_vector(orc.FR, SEED + 556, 1))[0]
        uu = orc.from_limbs(orc.synth_vector(orc.FR, SEED + 557, 1))[0]
        a_l, g_l = mz.to_limbs([alpha], 4), mz.points_to_array([(1, 2)])
        q_buf, q_buf1 = torch.zeros((hi - lo) * 4, dtype=torch.int64, device=dev), torch.zeros((hi - lo) * 4, dtype=torch.int64, device=dev)
        open_ops, y_host, keep_q = sharded.DeviceOpenOps(out=q_buf), [None], [None]

        def stage(name, fn, record):
            # every rank runs the same barriers and the same collective failure check per stage: a rank that
            # raised must not leave its peers waiting in the next stage's barrier
            barrier_sync(B)
            t0 = time.perf_counter()
            local_err = None
            try:
                fn()
                torch.cuda.synchronize()
            except Exception as ex:
                local_err = str(ex)[:300]
            if record and local_err is None:
                stages[name] = min(stages.get(name, float("inf")), (time.perf_counter() - t0) * 1e3)     # best of the recorded passes
            if max_over_ranks(B, 0.0 if local_err is None else 1.0) > 0.0:
                raise StageFailed(local_err or "a peer rank failed in stage " + name)

        def off(t, elems, limbs):
            return ctypes.c_void_p(t.data_ptr() + elems * limbs * 8)

    except Exception as ex:
        err = str(ex)[:300]
    # allocation / input failures are decided collectively BEFORE any rank enters the stage barriers
    if max_over_ranks(B, 0.0 if err is None else 1.0) > 0.0:
        err = err or "a peer rank failed while allocating"
    overlapped_ms, overlapped_same = None, None
    try:
        if err is not None:
            raise StageFailed(err)
        for record in (False, True, True):     # first pass builds plans / workspaces; two recorded passes, the better time of each stage
            # (one pass alone showed a 2.4-ms hiccup in one stage on one box of round 6: profiles/round6_bench_default_after_prefix_interpolation.json)
            if hh:
                L.mzk_srs_free(hh); hh = ctypes.c_void_p()
            stage("intt", lambda: check(B, L.mzk_ntt_dev(mz.FIELD_FR, rt.ctypes.data_as(ctypes.c_void_p), dptr(ev), dptr(cf), ctypes.c_size_t(nn), 1, stream)), record)
            stage("setup_srs_powers", lambda: check(B, L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p),
                                                                                      ctypes.c_size_t(lo), ctypes.c_size_t(hi - lo), dptr(sp), stream)), record)
            # one commit + one open per SRS: keep plain prepared points (no window tables: 173 ms at 2^22 only pays
            # off after ~20 commits)
            stage("srs_prepare", lambda: check(B, L.mzk_srs_from_device_ex(dptr(sp), ctypes.c_size_t(hi - lo), ctypes.c_int(0), ctypes.byref(hh), stream)), record)
            stage("commit_local", lambda: check(B, L.mzk_kzg_commit_srs_dev(hh, off(cf, lo, 4), ctypes.c_size_t(hi - lo), dptr(rec_c), ctypes.c_int(1), stream)), record)

            def open_local():
                # the quotient sharded like everything else (sharded.sharded_open_quotient): this rank's slice from ITS coefficients, one
                # 32-byte value per rank gathered; y = f(u) comes out of the same exchange
                y_host[0], qs = sharded.sharded_open_quotient(open_ops, cf[lo * 4:hi * 4], nn, uu, orc.P_FR, rank, world)
                check(B, L.mzk_kzg_commit_srs_dev(hh, dptr(qs), ctypes.c_size_t(hi - lo), dptr(rec_w), ctypes.c_int(1), stream))
            stage("open_local", open_local, record)
        # the two MSMs of a proof are independent: commit on context 0, open (quotient + its MSM) on context 1 of the same
        # GPU at the same time -- the sort and the latency-bound tails of one run under the accumulation of the other
        rec_c2 = torch.zeros(16, dtype=torch.int64, device=dev)
        rec_w2 = torch.zeros(16, dtype=torch.int64, device=dev)
        L.mzk_ctx_stream.restype = ctypes.c_void_p
        s1 = ctypes.c_void_p(L.mzk_ctx_stream(1))

        open_ops1 = sharded.DeviceOpenOps(stream=s1.value, out=q_buf1)

        def commit_and_open():
            mz.ctx_select(1)
            try:
                _, qs1 = sharded.sharded_open_quotient(open_ops1, cf[lo * 4:hi * 4], nn, uu, orc.P_FR, rank, world)
                keep_q[0] = qs1          # (alive until the stage's synchronize: the MSM reads it on s1)
                check(B, L.mzk_kzg_commit_srs_dev(hh, dptr(qs1), ctypes.c_size_t(hi - lo), dptr(rec_w2), ctypes.c_int(1), s1))
            finally:
                mz.ctx_select(0)
            check(B, L.mzk_kzg_commit_srs_dev(hh, off(cf, lo, 4), ctypes.c_size_t(hi - lo), dptr(rec_c2), ctypes.c_int(1), stream))
        for record in (False, True, True):
            stage("commit_and_open_overlapped", commit_and_open, record)
        overlapped_ms = stages.pop("commit_and_open_overlapped")
        # partial records are XYZZ (projective: the entry order inside a bucket comes from atomics, so the representation of
        # the same point differs from run to run): compare the canonical affine points
        aff = torch.zeros(4 * 8, dtype=torch.int64, device=dev)
        for k, r in enumerate((rec_c, rec_c2, rec_w, rec_w2)):
            check(B, L.mzk_g1_fold_partials_dev(dptr(r), ctypes.c_int(1), ctypes.c_void_p(aff.data_ptr() + 64 * k), stream))
        torch.cuda.synchronize()
        overlapped_same = bool(torch.equal(aff[0:8], aff[8:16]) and torch.equal(aff[16:24], aff[24:32]))
    except Exception as ex:
        err = str(ex)[:300]
    # every rank reaches this point; only fold if all local stages succeeded everywhere
    all_ok = max_over_ranks(B, 0.0 if err is None else 1.0) == 0.0
    e2e = {"log2_degree": lg, "n_gpus": world, "srs_points_this_rank": hi - lo, "srs_points_total": nn,
           "what": "evaluations -> iNTT -> setup(alpha) -> commit -> open(u), device-resident, MSMs and SRS sharded over the ranks (BASELINE configs[4]); "
                   "every stage bracketed by a barrier and a device synchronize, one warm-up pass, the better of two recorded passes per stage"}
    if all_ok:
        barrier_sync(B)
        t0 = time.perf_counter()
        fin = torch.zeros(16, dtype=torch.int64, device=dev)
        rc_all = sharded.all_gather_partials(rec_c)
        rw_all = sharded.all_gather_partials(rec_w)
        check(B, L.mzk_g1_fold_partials_dev(dptr(rc_all), ctypes.c_int(rc_all.shape[0]), dptr(fin), stream))
        check(B, L.mzk_g1_fold_partials_dev(dptr(rw_all), ctypes.c_int(rw_all.shape[0]), ctypes.c_void_p(fin.data_ptr() + 64), stream))
        torch.cuda.synchronize()
        stages["gather_and_fold"] = (time.perf_counter() - t0) * 1e3
        stages = {k: max_over_ranks(B, v) for k, v in stages.items()}
        max_ov = max_over_ranks(B, overlapped_ms)
        if rank == 0:
            oc = fin.cpu().numpy().view(np.uint64)
            cf_cpu = cf.cpu().numpy().view(np.uint64).reshape(-1, 4)
            fa = orc.poly_eval(orc.FR, cf_cpu, alpha)
            yy = y_host[0]
            qa = (fa - yy) * pow(alpha - uu, -1, orc.P_FR) % orc.P_FR
            okc = mz.array_to_points(oc[:8])[0] == orc.ec_mul(0, (1, 2), fa)
            oky = yy == orc.poly_eval(orc.FR, cf_cpu, uu)
            okw = mz.array_to_points(oc[8:16])[0] == orc.ec_mul(0, (1, 2), qa)
            e2e.update({"stages_ms": stages, "total_ms": sum(stages.values()), "trapdoor_identities_hold": bool(okc and oky and okw),
                        "commit_and_open_overlapped_ms": max_ov, "overlapped_results_identical": overlapped_same,
                        "total_with_overlap_ms": sum(v for k, v in stages.items() if k not in ("commit_local", "open_local")) + max_ov})
    else:
        e2e["error"] = err or "a rank failed"
    if hh:
        L.mzk_srs_free(hh)
    torch.cuda.empty_cache()
    return e2e


def leg_strong_msm(B):
    """Fixed-size MSM over all ranks (BASELINE configs[3]).  ONE KZG commit of 2^strong_log2n pairs: rank g builds SRS powers
    [lo_g, hi_g) on its GPU (tables included), commits its slice, the 128-byte partials are all-gathered and folded.  Strong-
    scaling view next to the weak-scaling headline.  Verified by the trapdoor identity on rank 0."""
    torch, L, mz, orc, np, args, dev, stream, rank, world, sharded, K = B.torch, B.L, B.mz, B.orc, B.np, B.args, B.dev, B.stream, B.rank, B.world, B.sharded, B.K
    tot = 1 << args.strong_log2n
    lo, hi = sharded.shard_range(tot, rank, world)
    m = hi - lo
    alpha = orc.from_limbs(orc.synth_vector(orc.FR, SEED + 901, 1))[0]
    a_l, g_l = mz.to_limbs([alpha], 4), mz.points_to_array([(1, 2)])
    err, hs, step_fn, res2 = None, ctypes.c_void_p(), None, None
    try:
        sc2 = torch.empty(m * 4, dtype=torch.int64, device=dev)
        check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 5000 + 1000003 * rank), ctypes.c_size_t(m), dptr(sc2), stream))
        sp2 = torch.empty(m * 8, dtype=torch.int64, device=dev)
        check(B, L.mzk_kzg_setup_g1_range_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(lo),
                                              ctypes.c_size_t(m), dptr(sp2), stream))
        check(B, L.mzk_srs_from_device(dptr(sp2), ctypes.c_size_t(m), ctypes.byref(hs), stream))
        torch.cuda.synchronize()
        del sp2
        part2 = torch.zeros(16, dtype=torch.int64, device=dev)
        res2 = torch.zeros(8, dtype=torch.int64, device=dev)

        def step_fn():
            check(B, L.mzk_kzg_commit_srs_dev(hs, dptr(sc2), ctypes.c_size_t(m), dptr(part2), ctypes.c_int(1), stream))
            recs = sharded.all_gather_partials(part2)
            check(B, L.mzk_g1_fold_partials_dev(dptr(recs), ctypes.c_int(recs.shape[0]), dptr(res2), stream))
    except Exception as ex:
        err = str(ex)[:300]
    strong = {"total_pairs": tot, "n_gpus": world, "pairs_per_gpu": m,
              "what": "one KZG commit of 2^%d pairs, SRS and scalars sharded contiguously over the ranks, all_gather of 128-byte "
                      "partials + fold (BASELINE configs[3]); time should fall with the number of GPUs" % args.strong_log2n}
    if max_over_ranks(B, 0.0 if err is None else 1.0) == 0.0:
        Ks = max(3, min(K, 5))
        sdt, _ = timed(B, step_fn, Ks, 1)
        strong.update({"ms_per_step": sdt / Ks * 1e3, "value": tot / (sdt / Ks), "unit": "pairs/s"})
        # trapdoor identity: the commitment must be [f(alpha)] G for f = the concatenation of all ranks' scalars
        got2 = mz.array_to_points(res2.cpu().numpy().view(np.uint64))[0]
        if rank == 0:
            cores = orc.usable_threads(64)
            fa = 0
            for r in range(world):
                rlo, rhi = sharded.shard_range(tot, r, world)
                sr = orc.synth_vector(orc.FR, SEED + 5000 + 1000003 * r, rhi - rlo, cores)
                fa = (fa + orc.poly_eval(orc.FR, sr, alpha) * pow(alpha, rlo, orc.P_FR)) % orc.P_FR
            strong["trapdoor_identity_holds"] = bool(got2 == orc.ec_mul(0, (1, 2), fa))
    else:
        strong["error"] = err or "a rank failed"
    if hs:
        L.mzk_srs_free(hs)
    torch.cuda.empty_cache()
    return strong


def leg_strong_ntt(B):
    """ONE transform sharded over all ranks (SURVEY 8e four-step layout).  A 2^strong_ntt_log2n-point Fr transform whose vector
    is spread over the ranks in contiguous slices: all-to-all, W-point transforms across the ranks, all-to-all, local n/W-point
    coset transform (the twiddle rides in the LDE's offset), and a third all-to-all when the result has to be contiguous again
    (myzkp_amd/sharded.py).  Every rank checks its part against the single-GPU transform of the whole vector, which it computes
    itself.  At N = 1 this is the plain transform (with --force-process-group: behind one all-to-all of one chunk)."""
    torch, L, mz, args, dev, stream, rank, world, sharded, K = B.torch, B.L, B.mz, B.args, B.dev, B.stream, B.rank, B.world, B.sharded, B.K
    lgt = args.strong_ntt_log2n
    tot = 1 << lgt
    mt = tot // world
    wt = mz.root_of_unity(mz.FIELD_FR, lgt)
    sn = {"log2n": lgt, "n_gpus": world, "field": "BN254 Fr", "points_per_gpu": mt,
          "what": "one 2^%d-point transform (ntt.rs:7-64), vector sharded over the ranks; all_to_all_single (RCCL) exchanges of "
                  "(N-1)/N of each rank's n/N elements; time should fall with the number of GPUs" % lgt}
    err, fns, res = None, {}, {}
    try:
        ops = sharded.DeviceOps(mz.FIELD_FR)
        PFR = mz.MODULUS[mz.FIELD_FR]
        if B.shared_gpu_test:
            _a2a = ops.all_to_all
            ops.all_to_all = lambda b, group=None: _a2a(b.cpu(), group).to(dev)
        full = torch.empty(tot * 4, dtype=torch.int64, device=dev)
        for r in range(world):
            check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 7000 + 1000003 * r), ctypes.c_size_t(mt), ctypes.c_void_p(full.data_ptr() + r * mt * 32), stream))
        xloc = full.view(world, -1)[rank].clone()
        want = torch.empty_like(full)
        rt24 = mz.to_limbs([wt], 4)
        check(B, L.mzk_ntt_dev(mz.FIELD_FR, rt24.ctypes.data_as(ctypes.c_void_p), dptr(full), dptr(want), ctypes.c_size_t(tot), 0, stream))
        torch.cuda.synchronize()
        want_contig = want.view(world, -1)[rank].clone()
        want_cyclic = want.view(-1, 4)[rank::world].contiguous().view(-1)
        del full, want
        torch.cuda.empty_cache()
        fns["contiguous_to_contiguous"] = lambda: res.__setitem__("cc", sharded.ntt_sharded(xloc, PFR, lgt, wt, ops, False, "contiguous", "contiguous"))
        fns["contiguous_to_cyclic"] = lambda: res.__setitem__("cy", sharded.ntt_sharded(xloc, PFR, lgt, wt, ops, False, "contiguous", "cyclic"))
        fns["inverse_cyclic_to_contiguous"] = lambda: res.__setitem__("inv", sharded.ntt_sharded(want_cyclic, PFR, lgt, wt, ops, True, "cyclic", "contiguous"))
    except Exception as ex:
        err = str(ex)[:300]
    if max_over_ranks(B, 0.0 if err is None else 1.0) == 0.0:
        Kn = max(3, min(K, 5))
        for name, fn in fns.items():
            dtn, _ = timed(B, fn, Kn, 1)
            sn[name + "_ms"] = dtn / Kn * 1e3
        torch.cuda.synchronize()
        okn = bool(torch.equal(res["cc"], want_contig) and torch.equal(res["cy"], want_cyclic) and torch.equal(res["inv"], xloc))
        sn["every_part_equals_single_gpu_transform"] = max_over_ranks(B, 0.0 if okn else 1.0) == 0.0
        ex1 = 1 if B.forced else 0          # the forced one-rank schedule: one all-to-all of one chunk in front of the transform
        sn.update({"ms_per_step": sn["contiguous_to_contiguous_ms"], "value": tot / (sn["contiguous_to_contiguous_ms"] * 1e-3), "unit": "elems/s",
                   "exchanges": {"contiguous_to_contiguous": 3 if world > 1 else ex1, "contiguous_to_cyclic": 2 if world > 1 else ex1,
                                 "inverse_cyclic_to_contiguous": 2 if world > 1 else ex1},
                   "bytes_sent_per_rank_per_exchange": (world - 1) * (mt // world) * 32})
    else:
        sn["error"] = err or "a rank failed"
    torch.cuda.empty_cache()
    return sn


def leg_inproc_devices(B):
    """One process, several GPUs, C ABI only (optional): mzk_init_devices + mzk_kzg_setup_srs_multi + mzk_kzg_commit_srs_multi_dev."""
    torch, L, mz, orc, args = B.torch, B.L, B.mz, B.orc, B.args
    ords = [int(x) for x in args.inproc_devices.split(",") if x != ""]
    tot = 1 << args.strong_log2n
    rec = {"devices": ords, "total_pairs": tot,
           "what": "mzk_init_devices + mzk_kzg_setup_srs_multi + mzk_kzg_commit_srs_multi_dev: contiguous shards, every GPU builds its own "
                   "SRS slice and commits it, 128-byte partials gathered through pinned host memory, fold on context 0"}
    try:
        if B.srs_h:
            L.mzk_srs_free(B.srs_h); B.srs_h = None
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
            check(B, L.mzk_synth_field_dev(mz.FIELD_FR, ctypes.c_uint64(SEED + 7000 + r), ctypes.c_size_t(hi_r - lo_r), dptr(t), None))
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
            mz.init_devices([B.local_rank] * 4)
        except Exception:
            pass
    return rec


def leg_cpu_baseline(B):
    """CPU baseline (rank 0, N = 1 only, bounded sample): the oracle's literal restatement (`port`) on one core, and its
    multi-threaded Pippenger / iterative NTT (`cpu_fast`) beside it.  A reported baseline, not the target."""
    orc, n, args = B.orc, B.n, B.args
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
    return cb


# ---------------------------------------------------------------------------------------------------- rooflines, the line, the detail
def hbm_roofline(alg_bytes, ms, copy_gbps):
    ach = alg_bytes / (ms * 1e-3) / 1e9
    return {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS,
            "frac_of_measured_copy_rate": (ach / copy_gbps) if copy_gbps else None, "measured_copy_GBps": copy_gbps, "traffic": None}


def _attach_traffic(roof, tr):
    if tr is not None:
        roof["traffic"] = tr["bytes"]
        roof["traffic_source"] = tr["source"]
        roof["traffic_stale"] = bool(tr["stale"])
        if "raw_bytes" in tr:
            roof["traffic_raw_counters"] = tr["raw_bytes"]


def _nan(x):
    return x != x


def build_rooflines(B, T, copy_gbps):
    """`roofline` of the headline (k_seg_accumulate against HBM: 96 B per pair) and of the sub-legs, plus the integer-multiply
    roofline that actually binds.  T: the timed legs' (wall seconds for K steps, phases)."""
    n, args, K, world = B.n, B.args, B.K, B.world
    lg20 = args.log2n == 20
    R = {}
    srs_dt, srs_ph = T["srs"]
    srs_acc_ms = srs_ph.get("msm_bucket_accumulate_in_timed_region", {}).get("avg_ms", float("nan"))
    roof = hbm_roofline(96.0 * n, srs_acc_ms, copy_gbps)
    roof.update({"kernel": "k_seg_accumulate", "avg_launch_ms": srs_acc_ms, "algorithmic_bytes_per_launch": 96 * n,
                 "measured": "HIP-event pair around the kernel on its launch stream, inside the timed region (the only instrumented kernel there)"})
    if lg20:
        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of k_seg_accumulate at 2^20 pairs with the default window width,
        # raw counters.  The fixed-base method reads each of the table points of a pair once (15 x 64 B = 0.96 GiB), served by L2 /
        # Infinity Cache.  A recorded profile of the same kernel at the same size -- not collected by this run (traffic_stale says
        # whether the kernel sources have changed since).
        _attach_traffic(roof, recorded_traffic("k_seg_accumulate"))
    R["srs"] = roof
    msm_dt, msm_ph = T["msm"]
    acc_ms = msm_ph.get("msm_bucket_accumulate_in_timed_region", {}).get("avg_ms", float("nan"))
    g = hbm_roofline(96.0 * n, acc_ms, copy_gbps)
    g.update({"kernel": "k_seg_accumulate", "avg_launch_ms": acc_ms, "algorithmic_bytes_per_launch": 96 * n})
    if lg20:
        _attach_traffic(g, recorded_traffic("k_seg_accumulate", section="generic MSM 2^20"))
    R["msm"] = g
    # integer-multiply roofline (the binding one): v_mad_u64_u32 per Montgomery product = 171
    gen_madds_per_pair = B.generic_windows_per_pair
    msm_mads = n * gen_madds_per_pair * (8 * 171 + 2 * 135)      # madd = 8M + 2S per (pair, window)
    srs_mads = n * B.srs_table_windows * (8 * 171 + 2 * 135)
    alu = {"unit": "v_mad_u64_u32/s", "peak": MAD_PEAK_PER_S,
           "msm_accumulate_frac": None if _nan(acc_ms) else msm_mads / (acc_ms * 1e-3) / MAD_PEAK_PER_S,
           "kzg_commit_accumulate_frac": None if _nan(srs_acc_ms) else srs_mads / (srs_acc_ms * 1e-3) / MAD_PEAK_PER_S}
    # Among multiply-adds EVERY VALU instruction costs a multiply-add's issue slot (tools/microbench/op_rates.hip, profiles/round5_op_issue_rates.txt:
    # 1.75 ns per wave-instruction and SIMD, whatever the mix), so the kernel's issue-side roofline is its executed VALU instructions (recorded
    # SQ_INSTS_VALU of the same kernel at the same size) x that slot / its SIMDs
    vi = recorded_valu_instructions("k_seg_accumulate") if lg20 else None
    if vi is not None and not _nan(srs_acc_ms):
        simds = B.torch.cuda.get_device_properties(B.dev).multi_processor_count * 4
        alu["kzg_commit_accumulate_valu_issue"] = {"frac": vi["instructions"] / simds * 1.75e-9 / (srs_acc_ms * 1e-3), "valu_instructions_per_launch": vi["instructions"],
                                                   "slot_ns": 1.75, "record_stale": bool(vi["stale"]),
                                                   "source": vi["source"] + " (recorded counters) and profiles/round5_op_issue_rates.txt (the slot)"}
    if "ntt" in T:
        ntt_dt, ntt_ph = T["ntt"]
        ntt_ms = ntt_dt / K * 1e3
        ntt_total_ms = ntt_ph.get("ntt_whole_transform_in_priced_pass", {}).get("avg_ms", float("nan"))
        npass = sum(1 for k in ntt_ph if k.startswith("ntt_pass") and not k.endswith("_in_priced_pass"))
        r = hbm_roofline(64.0 * n, ntt_total_ms, copy_gbps)
        r.update({"kernel": "k_ntt_strided + k_ntt_last (whole transform: %d passes, one event pair around them)" % npass, "passes": npass, "avg_launch_ms": ntt_total_ms,
                  "algorithmic_bytes_per_launch": 64 * n, "frac_of_wall_clock": 64.0 * n / (ntt_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS})
        if lg20:
            _attach_traffic(r, recorded_ntt_traffic("Fr"))
        R["ntt"] = r
        ntt_mads = (n // 2) * args.log2n * 171
        alu["ntt_frac"] = None if _nan(ntt_total_ms) or not ntt_total_ms else ntt_mads / (ntt_total_ms * 1e-3) / MAD_PEAK_PER_S
        alu["ntt_frac_of_step"] = ntt_mads / (ntt_ms * 1e-3) / MAD_PEAK_PER_S
    if "nttm" in T:
        nttm_dt, nttm_ph = T["nttm"]
        nttm_ms = nttm_dt / K * 1e3
        nttm_total_ms = nttm_ph.get("ntt_whole_transform_in_priced_pass", {}).get("avg_ms", float("nan"))
        r = dict(hbm_roofline(32.0 * n, nttm_total_ms, copy_gbps), kernel="k_ntt_strided + k_ntt_last (whole transform)", algorithmic_bytes_per_launch=32 * n,
                 avg_launch_ms=nttm_total_ms)
        if lg20:
            _attach_traffic(r, recorded_ntt_traffic("M128"))
        R["nttm"] = r
        # M128: 31 v_mad_i64_i32 per product since round 5 (5 x 5 limbs + 5 for the sparse modulus 1 + 407 * 2^119 + 1 constant; no v_mul_lo)
        nttm_mads = (n // 2) * args.log2n * 31
        alu["ntt_m128_frac"] = None if _nan(nttm_total_ms) or not nttm_total_ms else nttm_mads / (nttm_total_ms * 1e-3) / MAD_PEAK_PER_S
        alu["ntt_m128_frac_of_step"] = nttm_mads / (nttm_ms * 1e-3) / MAD_PEAK_PER_S
        alu["ntt_m128_note"] = ("(n/2) log2(n) products x 31 half-rate multiply-adds each; the M128 transform is bound by neither roofline: what its two "
                                "passes wait for is the global loads / stores of the one tile each CU holds (DESIGN.md section 4)")
    R["alu"] = alu
    return R


def compact_line(line):
    """Keep the printed line under LINE_BUDGET_BYTES: drop the longest optional strings first (the detail file has them all)."""
    droppable = [("roofline", "traffic_source"), ("roofline", "measured"), ("cpu_baseline", "sample_note"), ("config", "note"), ("gpu", "cards")]
    for path in droppable:
        if len(json.dumps(line)) <= LINE_BUDGET_BYTES:
            break
        d = line
        for k in path[:-1]:
            d = d.get(k, {}) if isinstance(d, dict) else {}
        if isinstance(d, dict):
            d.pop(path[-1], None)
    return line


def main():
    T_START = time.perf_counter()
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` by itself: become the launcher BEFORE anything touches the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))
    sysfs = gpu_sysfs_snapshot()           # clock range and power cap of the cards, read before the HIP runtime starts
    B = setup(args, T_START)
    torch, L, mz, n, K, W, world, rank = B.torch, B.L, B.mz, B.n, B.K, B.W, B.world, B.rank
    make_inputs(B)
    props = torch.cuda.get_device_properties(B.dev)
    try:
        pci = "%04x:%02x:%02x.0" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    except AttributeError:
        pci = None
    card = gpu_card_of(pci, sysfs)
    parity = leg_parity(B)
    progress(B, "parity done; timing")

    # ---- timed legs (the contract's region each).  The generic MSM, like the headline, carries the event pair of its accumulate
    # kernel inside the region (0.1 % of the step); the short sub-legs carry none and price their kernels in a pass of their own.
    T = {}
    T["msm"] = timed(B, lambda: msm_step(B), K, W, 1 << PH_ACC)
    T["srs"] = timed(B, lambda: srs_step(B), K, W, 1 << PH_ACC)
    clock_under_load = leg_clock_under_load(B, lambda: srs_step(B), card)       # (every rank: the step holds the gather)
    if B.all_legs:
        T["ntt"] = timed(B, lambda: ntt_step(B), K, W, 1 << PH_NTT_TOTAL, priced_in_timed_region=False)
        T["nttm"] = timed(B, lambda: ntt_m128_step(B), K, W, 1 << PH_NTT_TOTAL, priced_in_timed_region=False)
        T["lde"] = timed(B, lambda: lde_m128_step(B), K, W, 1 << PH_NTT_TOTAL, priced_in_timed_region=False)
        T["mk"] = timed(B, lambda: merkle_m128_step(B), K, W, 1 << PH_MERKLE, priced_in_timed_region=False)
    # the generic layout's additions per pair: 2 GLV halves x windows (the library reports the width it chose for this size)
    try:
        L.mzk_msm_generic_window_bits.restype = ctypes.c_int
        gbits = int(L.mzk_msm_generic_window_bits(ctypes.c_size_t(n)))
    except AttributeError:
        gbits = 16
    B.generic_window_bits = gbits
    B.generic_windows_per_pair = 2 * (126 // gbits + 1)

    side = {}
    side["kzg_commit_two_in_flight"] = leg_in_flight(B, 2)
    side["kzg_commit_four_in_flight"] = leg_in_flight(B, 4)
    side["msm_no_tables_in_flight"] = leg_no_tables_in_flight(B)
    w16 = leg_other_width(B, 16) if B.srs_window_bits != 16 else None
    side["kzg_commit_16_bit_windows"] = w16 if w16 is not None else ({"note": "16 bits is the default width at this size: see `value`"} if B.srs_window_bits == 16 else None)
    side["kzg_commit_small_batch"] = leg_small_batch(B)
    side["ntt_batched"] = leg_ntt_batched(B)
    side["pcie_inclusive"] = leg_pcie_inclusive(B)
    side["stark_commit_pipeline"] = leg_stark_commit_pipeline(B)
    progress(B, "timed legs done")
    hbm_copy = leg_copy_rate(B) if B.all_legs else None
    copy_gbps = hbm_copy["GBps_library_copy_kernel"] if hbm_copy else None
    R = build_rooflines(B, T, copy_gbps)

    # ---- the headline's numbers
    srs_dt, srs_ph = T["srs"]
    msm_dt, msm_ph = T["msm"]
    srs_ms, msm_ms = srs_dt / K * 1e3, msm_dt / K * 1e3
    srs_rate, msm_rate = world * n / (srs_dt / K), world * n / (msm_dt / K)
    srs_acc_ms = R["srs"]["avg_launch_ms"]
    madds = n * B.srs_table_windows
    clk_max = sysfs.get(card, {}).get("sclk_mhz_max") if card else (max([c["sclk_mhz_max"] or 0 for c in sysfs.values()]) or None if sysfs else None)
    cap_w = sysfs.get(card, {}).get("power_cap_w") if card else (min([c["power_cap_w"] for c in sysfs.values() if c["power_cap_w"]] or [0]) or None if sysfs else None)
    clk_rep = (clock_under_load or {}).get("median_mhz") or clk_max
    simds = props.multi_processor_count * 4
    gpu = {"name": props.name, "compute_units": props.multi_processor_count, "pci": pci, "card": card, "clock_mhz_max": clk_max, "power_cap_w": cap_w,
           "sclk_mhz_under_headline_load": clock_under_load,
           "cards": {k: {"sclk_mhz_max": v["sclk_mhz_max"], "power_cap_w": v["power_cap_w"]} for k, v in sysfs.items()} if card is None else None,
           "how": "sysfs (pp_dpm_sclk, hwmon power1_cap) read before the HIP runtime starts; the clock under load from hwmon freq1_input during a further, "
                  "untimed pass of the headline step"}
    lg = args.log2n
    out = {
        "metric": METRIC,
        "value": srs_rate, "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": srs_ms,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32x9 (29-bit limbs, 254-bit Montgomery)", "data": "synthetic",
        # `value` is measured at the library's default window width for this SRS size (17 bits at 2^20 since round 3; BASELINE
        # configs[2] names 16 bits: legs_ms.kzg_commit_16_bit_windows) -- stated at top level so that a change of the default shows
        "value_window_bits": B.srs_window_bits,
        "config": {"workload": "KZG commit = BN254 G1 Pippenger MSM of 2^%d (scalar, point) pairs per GPU against a device-resident SRS (commit_kzg, kzg.rs:57-59); "
                               "%d-bit signed windows (library default at this size), %d window tables built once at upload; N GPUs = one MSM of N*2^%d pairs "
                               "(BASELINE configs[2]/[3])" % (lg, B.srs_window_bits, B.srs_table_windows, lg),
                   "pairs_per_gpu": n, "seed": SEED, "window_bits": B.srs_window_bits, "sharding": "contiguous shards + all_gather of 128 B partials (RCCL) + local fold"},
        "roofline": R["srs"],
        "gpu_clock_mhz_max": clk_max, "power_cap_w": cap_w, "gpu_clock_mhz_under_load": (clock_under_load or {}).get("median_mhz"),
        # two boxes compared kernel for kernel: chip-wide time per mixed addition of the priced kernel, and the same in SIMD cycles per
        # wave-level addition (64 lanes) at the clock this line reports: t x f x SIMDs / (additions / 64)
        "accumulate_ns_per_madd": None if _nan(srs_acc_ms) else srs_acc_ms * 1e6 / madds,
        "accumulate_cycles_per_madd_at_reported_clock": None if _nan(srs_acc_ms) or not clk_rep else srs_acc_ms * 1e-3 * clk_rep * 1e6 * simds / (madds / 64.0),
        "process_group": {"world_size": B.dist.get_world_size() if B.dist.is_initialized() else 1, "backend": B.dist.get_backend() if B.dist.is_initialized() else None,
                          "forced_collectives": bool(B.forced),
                          "launched_by": os.environ.get("MZK_BENCH_LAUNCHED_BY", "external launcher" if world > 1 else "single process")},
        **({"REHEARSAL_NOT_A_MEASUREMENT": "ranks share one GPU, exchange over gloo via host (MZK_BENCH_SHARED_GPU_TEST=1)"} if B.shared_gpu_test else {}),
        "parity": parity,
    }
    # ---- BASELINE.json's metric as top-level scalars: MSM pairs/s and NTT elems/s at 2^20 and 2^24 (the literal "G1 MSM" on
    # arbitrary points beside the KZG-commit special case `value` measures)
    out["msm_2p%d_pairs_per_s" % lg], out["msm_2p%d_ms" % lg] = srs_rate, srs_ms
    out["msm_2p%d_arbitrary_points_pairs_per_s" % lg], out["msm_2p%d_arbitrary_points_ms" % lg] = msm_rate, msm_ms
    legs_ms = {"kzg_commit": srs_ms, "kzg_commit_without_event_pair": srs_ph.get("_ms_per_step_without_events"), "msm_arbitrary_points": msm_ms,
               "kzg_commit_accumulate_kernel": srs_acc_ms, "msm_arbitrary_points_accumulate_kernel": R["msm"]["avg_launch_ms"]}
    for ph_key, name in (("msm_digit_sort", "kzg_commit_digit_sort"), ("msm_segment_combine", "kzg_commit_segment_combine"), ("msm_bucket_reduce", "kzg_commit_bucket_reduce")):
        if ph_key in srs_ph:
            legs_ms[name] = srs_ph[ph_key]["avg_ms"]
    detail = {"line": None, "phases": {"kzg_commit": srs_ph, "msm_arbitrary_points": msm_ph}, "rooflines": R, "gpu": gpu, "hbm_copy": hbm_copy}
    if "ntt" in T:
        ntt_ms, nttm_ms, lde_ms, mk_ms = (T[k][0] / K * 1e3 for k in ("ntt", "nttm", "lde", "mk"))
        out["ntt_2p%d_elems_per_s" % lg], out["ntt_2p%d_ms" % lg] = world * n / (ntt_ms * 1e-3), ntt_ms
        out["ntt_m128_2p%d_elems_per_s" % lg], out["ntt_m128_2p%d_ms" % lg] = world * n / (nttm_ms * 1e-3), nttm_ms
        out["ntt_roofline"] = {k: R["ntt"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "avg_launch_ms", "passes")}
        out["ntt_m128_roofline"] = {k: R["nttm"].get(k) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_stale", "avg_launch_ms")}
        legs_ms.update({"ntt_fr": ntt_ms, "ntt_m128": nttm_ms, "coset_lde_m128_blowup_4": lde_ms, "merkle_commit_m128_codeword": mk_ms})
        for key, name, unit_n in (("ntt", "ntt", n), ("nttm", "ntt_m128", n), ("lde", "coset_lde_m128", n), ("mk", "merkle_m128", n - 1)):
            ph = T[key][1]
            detail[name] = {"ms_per_step": T[key][0] / K * 1e3, "value": world * unit_n / (T[key][0] / K), "ms_per_step_with_event_pair": ph.get("_ms_per_step_with_event_pair"),
                            "phases": ph, "multi_gpu": "replicas (one independent call per GPU)"}
    out["alu_roofline"] = {k: v for k, v in R["alu"].items() if not isinstance(v, (dict, str)) or k == "unit"}
    out["hbm_copy_GBps_measured"] = copy_gbps

    progress(B, "line assembled; extra sizes")
    extras = leg_extra_sizes(B, card)
    detail["extra_sizes_1gpu"] = extras
    for tag, e in extras.items():
        if "error" in e:
            continue
        lgx = tag[2:]
        out["msm_2p%s_pairs_per_s" % lgx], out["msm_2p%s_ms" % lgx] = e["kzg_commit_srs"]["rate"], e["kzg_commit_srs"]["ms"]
        out["msm_2p%s_arbitrary_points_pairs_per_s" % lgx], out["msm_2p%s_arbitrary_points_ms" % lgx] = e["msm_generic"]["rate"], e["msm_generic"]["ms"]
        out["ntt_2p%s_elems_per_s" % lgx], out["ntt_2p%s_ms" % lgx] = e["ntt"]["rate"], e["ntt"]["ms"]
        out["hbm_frac_2p%s" % lgx] = {"msm": e["kzg_commit_srs"]["hbm_frac"], "msm_arbitrary_points": e["msm_generic"]["hbm_frac"], "ntt": e["ntt"]["hbm_frac"]}
        if "sclk_mhz_mid_call" in e["kzg_commit_srs"]:
            out["gpu_clock_mhz_under_load_2p%s_msm" % lgx] = e["kzg_commit_srs"]["sclk_mhz_mid_call"]

    if args.e2e_log2n > 0:
        e2e = leg_e2e_kzg(B)
        out["e2e_kzg"] = {k: e2e.get(k) for k in ("log2_degree", "n_gpus", "total_ms", "total_with_overlap_ms", "trapdoor_identities_hold", "error") if k in e2e}
        detail["e2e_kzg"] = e2e
    progress(B, "fixed-size (strong scaling) leg")
    if args.strong_log2n > 0:
        sm = leg_strong_msm(B)
        out["strong_scaling_msm"] = {k: sm.get(k) for k in ("total_pairs", "n_gpus", "pairs_per_gpu", "ms_per_step", "value", "unit", "trapdoor_identity_holds", "error") if k in sm}
        detail["strong_scaling_msm"] = sm
    progress(B, "sharded transform leg")
    if args.strong_ntt_log2n > 0 and world * world <= (1 << args.strong_ntt_log2n):
        sn = leg_strong_ntt(B)
        out["strong_scaling_ntt"] = {k: sn.get(k) for k in ("log2n", "n_gpus", "ms_per_step", "value", "unit", "exchanges", "bytes_sent_per_rank_per_exchange",
                                                            "every_part_equals_single_gpu_transform", "error") if k in sn}
        detail["strong_scaling_ntt"] = sn
    if args.inproc_devices and world == 1 and args.strong_log2n > 0:
        detail["strong_scaling_msm_single_process"] = leg_inproc_devices(B)

    # ---- one time per side leg in the line, the records themselves in the detail file
    def pick(d, *path):
        for k in path:
            d = d.get(k) if isinstance(d, dict) else None
        return d
    s = side
    legs_ms.update({
        "kzg_commit_two_in_flight_per_commit": pick(s, "kzg_commit_two_in_flight", "ms_per_commit"),
        "kzg_commit_four_in_flight_per_commit": pick(s, "kzg_commit_four_in_flight", "ms_per_commit"),
        "kzg_commit_16_bit_windows": pick(s, "kzg_commit_16_bit_windows", "ms_per_step"),
        "many_commit_256_x_2p10": pick(s, "kzg_commit_small_batch", "256_x_2^10", "one_call_ms"),
        "many_commit_256_x_2p10_31_byte": pick(s, "kzg_commit_small_batch", "256_x_2^10", "coefficients_of_31_bytes", "one_call_ms"),
        "many_commit_256_x_2p10_direct_10_bit": pick(s, "kzg_commit_small_batch", "256_x_2^10", "direct_tables_10_bit", "one_call_ms"),
        "many_commit_64_x_2p12": pick(s, "kzg_commit_small_batch", "64_x_2^12", "one_call_ms"),
        "many_commit_16_x_2p14": pick(s, "kzg_commit_small_batch", "16_x_2^14", "one_call_ms"),
        "pcie_msm_host_buffers": pick(s, "pcie_inclusive", "msm_g1_bn254_host_buffers", "ms_per_call"),
        "pcie_kzg_commit_host_scalars": pick(s, "pcie_inclusive", "kzg_commit_srs_host_scalars", "ms_per_call"),
        "pcie_ntt_host_buffers": pick(s, "pcie_inclusive", "ntt_host_buffers", "ms_per_call"),
        "stark_commit_pipeline_total": pick(s, "stark_commit_pipeline", "total_ms"),
        "stark_interpolate_16_registers": pick(s, "stark_commit_pipeline", "stages_ms", "interpolate_16_registers_trace_uploaded_coefficients_in_hbm"),
        "stark_interpolate_16_registers_arbitrary_domain": pick(s, "stark_commit_pipeline", "interpolate_16_registers_arbitrary_domain_ms"),
        "stark_fri_us_per_round": pick(s, "stark_commit_pipeline", "fri_us_per_round"),
    })
    out["legs_ms"] = {k: (round(v, 5) if isinstance(v, float) else v) for k, v in legs_ms.items() if v is not None}
    side_errors = {k: v["error"] for k, v in s.items() if isinstance(v, dict) and "error" in v}
    if side_errors:
        out["side_leg_errors"] = side_errors
    detail.update(side)
    detail["srs_precompute"] = {"table_build_ms": B.srs_build_ms, "table_bytes": B.srs_table_windows * n * 64, "tables": B.srs_table_windows,
                                "generic_no_precompute_pairs_per_s": msm_rate, "generic_window_bits": B.generic_window_bits,
                                "break_even_commits": (B.srs_build_ms / (msm_ms - srs_ms)) if msm_ms > srs_ms else None,
                                "note": "table build is outside the timed region; one KZG setup is followed by many commits/opens against the same powers_1 (kzg.rs:57-72)"}

    # ---- CPU baseline (rank 0, N = 1 only, bounded sample)
    if rank == 0 and world == 1 and not args.skip_cpu:
        cb = leg_cpu_baseline(B)
        detail["cpu_baseline"] = cb
        out["cpu_baseline"] = {"value": cb["value"], "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
                               "sample": "first 2^11 pairs of the same stream, oracle orc_msm_ref (literal polynomial.rs:156-165, one core); linear in n",
                               "cpu_fast_pairs_per_s": cb["cpu_fast"]["value"], "cpu_fast_cores": cb["cpu_fast"]["cores"],
                               "ntt_port_elems_per_s": cb["ntt"]["value"], "ntt_cpu_fast_elems_per_s": cb["ntt"]["cpu_fast"]["value"],
                               "sample_note": "cpu_fast = the oracle's un-tuned multi-threaded Pippenger / iterative NTT at full size; not a CPU library -- do not quote a speed-up from it"}
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        out["gpu"] = {k: v for k, v in gpu.items() if k not in ("how",) and v is not None}
        out["detail_file"] = os.path.basename(args.detail_file) if args.detail_file else None
        out = compact_line(out)
        detail["line"] = out
        if args.detail_file:
            try:
                with open(args.detail_file, "w") as f:
                    json.dump(detail, f, indent=1, default=str)
            except OSError as ex:
                out["detail_file"] = "not written: %s" % str(ex)[:80]
        print(json.dumps(out))
    if B.dist.is_initialized():
        B.dist.barrier()
        B.dist.destroy_process_group()


if __name__ == "__main__":
    main()

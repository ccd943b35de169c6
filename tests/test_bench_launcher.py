"""bench.py's own launcher: `python bench.py --gpus N` must never silently run fewer ranks than asked.
On a box with fewer than N visible GPUs (this CPU container has none) it prints ONE JSON line with an `error` field and exits
non-zero, before importing anything that initialises a GPU.  CPU only."""
import json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_more_gpus_than_visible_fails_loudly_with_a_json_line():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("two GPUs visible: the launcher would really start two ranks")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MZK_BENCH_SHARED_GPU_TEST")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env, capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] is None and "error" in rec and rec["metric"].startswith("G1 MSM pairs/sec")


def test_world_size_that_contradicts_gpus_flag_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    rec = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert rec["n_gpus"] == 4 and "WORLD_SIZE=1" in rec["error"]


def _fake_topology(tmp_path, simd_counts):
    for i, sc in enumerate(simd_counts):
        d = tmp_path / str(i)
        d.mkdir()
        (d / "properties").write_text("cpu_cores_count %d\nsimd_count %d\nmem_banks_count 1\n" % (0 if sc else 64, sc))
    return str(tmp_path)


def test_gpus_are_counted_from_sysfs_without_touching_hip(tmp_path):
    """VERDICT r03 #8a: the launcher parent counts devices from the KFD topology (simd_count > 0), honouring the visibility
    variables, and never imports torch or calls HIP for it."""
    sys.path.insert(0, ROOT)
    import bench
    root = _fake_topology(tmp_path, [0, 0, 1024, 1024, 1024, 1024])           # two CPU nodes, four GPUs
    assert bench.visible_gpus_without_hip(root, {}) == 4
    assert bench.visible_gpus_without_hip(root, {"HIP_VISIBLE_DEVICES": "0,2"}) == 2
    assert bench.visible_gpus_without_hip(root, {"ROCR_VISIBLE_DEVICES": "1,2,3", "HIP_VISIBLE_DEVICES": "0,1"}) == 2
    assert bench.visible_gpus_without_hip(root, {"HIP_VISIBLE_DEVICES": ""}) == 0
    assert bench.visible_gpus_without_hip(root, {"HIP_VISIBLE_DEVICES": "-1"}) == 0
    assert bench.visible_gpus_without_hip(root, {"CUDA_VISIBLE_DEVICES": "0,9,1"}) == 1      # cut at the first invalid ordinal
    assert bench.visible_gpus_without_hip(str(tmp_path / "missing"), {}) is None
    src = open(os.path.join(ROOT, "bench.py")).read()
    body = src[src.index("def launch_ranks"):src.index("def parse_args")]
    assert "import torch" not in body and "device_count" not in body


def _newest(pattern):
    import glob
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return recs[-1] if recs else None


def _line(path):
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def test_multi_rank_line_describes_its_process_group_and_exchanges():
    """VERDICT r03 #8b: what the first real SCALE run must carry, checked on the committed one-GPU rehearsals (world 2, 4, 8):
    world_size == N, the nccl backend named for a real run, exchanges > 0, and the sharded transform's bytes per rank and
    exchange = (N - 1) / N * n / N * 32.  From round 6 on the printed line is the compact one and the per-stage records sit in
    the detail file beside it (`*_detail.json`)."""
    import glob
    rounds = sorted({os.path.basename(f)[:8] for f in glob.glob(os.path.join(ROOT, "profiles", "r0*final_rehearsal_world*_shared_gpu.json"))})
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", rounds[-1] + "_rehearsal_world*_shared_gpu.json")))       # the newest round's set
    assert len(recs) >= 3, "world-2/4/8 rehearsals missing from profiles/"

    for path in recs:
        rec = _line(path)
        det_path = path.replace(".json", "_detail.json")
        det = json.load(open(det_path)) if os.path.exists(det_path) else rec      # rounds 3-5: everything was in the line
        N = rec["n_gpus"]
        assert rec["process_group"]["world_size"] == N and N in (2, 4, 8)
        assert rec["process_group"]["backend"] in ("nccl", "gloo")            # gloo only under the tagged one-GPU rehearsal
        if rec["process_group"]["backend"] == "gloo":
            assert "REHEARSAL_NOT_A_MEASUREMENT" in rec
        sn = rec["strong_scaling_ntt"]
        assert all(v > 0 for v in sn["exchanges"].values())
        n = 1 << sn["log2n"]
        assert sn["bytes_sent_per_rank_per_exchange"] == (N - 1) * (n // N // N) * 32
        sm = rec["strong_scaling_msm"]
        assert sm["n_gpus"] == N and sm["pairs_per_gpu"] * N == sm["total_pairs"] and sm["trapdoor_identity_holds"] is True
        assert sn["every_part_equals_single_gpu_transform"] is True
        # the end-to-end leg as well -- the SRS is sharded (each rank builds and commits against ITS powers only), the
        # partials of commit and open are gathered and folded, and the trapdoor identities hold on the folded points
        e2e = det["e2e_kzg"]
        assert e2e["n_gpus"] == N and "error" not in e2e
        assert {"intt", "setup_srs_powers", "srs_prepare", "commit_local", "open_local", "gather_and_fold"} <= set(e2e["stages_ms"])
        assert e2e["trapdoor_identities_hold"] is True and e2e["overlapped_results_identical"] is True
        if "srs_points_this_rank" in e2e:              # (records from round 5 on)
            assert e2e["srs_points_total"] == 1 << e2e["log2_degree"] and e2e["srs_points_this_rank"] * N == e2e["srs_points_total"]


def test_default_line_carries_the_metric_scalars_the_clock_and_fits_its_budget():
    """the driver keeps only the top level of the printed line, so BASELINE.json's four numbers (MSM pairs/s and NTT
    elems/s at 2^20 and 2^24), the GPU's clock and power cap and the per-addition figures that let two boxes be compared must be
    top-level scalars of a line under 8 KB; everything bulky lives in the detail file the line names."""
    path = _newest("r06*_bench_default.json")
    assert path, "no round-6 bench record in profiles/"
    raw = [l for l in open(path) if l.startswith("{")][-1]
    assert len(raw) <= 8192, len(raw)
    rec = json.loads(raw)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline"):
        assert key in rec, key
    for key in ("msm_2p20_pairs_per_s", "msm_2p20_ms", "msm_2p24_pairs_per_s", "msm_2p24_ms", "ntt_2p20_elems_per_s", "ntt_2p20_ms", "ntt_2p24_elems_per_s",
                "ntt_2p24_ms", "msm_2p20_arbitrary_points_pairs_per_s", "msm_2p24_arbitrary_points_pairs_per_s", "gpu_clock_mhz_max", "power_cap_w",
                "accumulate_ns_per_madd", "accumulate_cycles_per_madd_at_reported_clock"):
        assert isinstance(rec[key], (int, float)) and rec[key] > 0, key
    assert rec["value"] == rec["msm_2p20_pairs_per_s"] and rec["config"]["workload"].startswith("KZG commit")
    roof = rec["roofline"]
    assert roof["bound"] == "hbm" and roof["unit"] == "GB/s" and abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    assert isinstance(roof["traffic_stale"], bool)                         # a recorded profile: says whether the kernel has changed since
    cb = rec["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and cb["sample"]
    assert rec["detail_file"] and os.path.exists(path.replace(".json", "_detail.json"))
    det = json.load(open(path.replace(".json", "_detail.json")))
    assert det["line"]["value"] == rec["value"] and "phases" in det and "kzg_commit_small_batch" in det


def test_forced_one_rank_record_went_through_nccl():
    """the sharded legs of bench.py through a one-rank `nccl` group on a one-GPU box (`--force-process-group`)."""
    path = _newest("r06*_bench_forced_nccl_world1.json")
    assert path, "no forced one-rank nccl record in profiles/"
    rec = _line(path)
    pg = rec["process_group"]
    assert pg["backend"] == "nccl" and pg["world_size"] == 1 and pg["forced_collectives"] is True
    assert rec["strong_scaling_msm"]["trapdoor_identity_holds"] is True and rec["e2e_kzg"]["trapdoor_identities_hold"] is True
    assert rec["strong_scaling_ntt"]["every_part_equals_single_gpu_transform"] is True and rec["strong_scaling_ntt"]["exchanges"]["contiguous_to_contiguous"] == 1


def test_clock_and_power_cap_come_from_sysfs_without_hip(tmp_path):
    """The line's gpu_clock_mhz_max / power_cap_w are read from sysfs before the HIP runtime starts (pp_dpm_sclk levels, hwmon
    power1_cap in microwatts), the clock under load from hwmon freq1_input."""
    sys.path.insert(0, ROOT)
    import bench
    pci = tmp_path / "devices" / "0000:26:00.0"
    (pci / "hwmon" / "hwmon7").mkdir(parents=True)
    (pci / "pp_dpm_sclk").write_text("0: 500Mhz\n1: 2259Mhz *\n2: 2400Mhz\n")
    (pci / "hwmon" / "hwmon7" / "power1_cap").write_text("1400000000\n")
    (pci / "hwmon" / "hwmon7" / "freq1_input").write_text("2259000000\n")
    drm = tmp_path / "drm"
    (drm / "card24").mkdir(parents=True)
    os.symlink(str(pci), str(drm / "card24" / "device"))
    (drm / "card24-DP-1").mkdir()                                           # connector nodes are not cards
    (drm / "renderD128").mkdir()
    snap = bench.gpu_sysfs_snapshot(str(drm))
    assert snap == {"card24": {"pci": "0000:26:00.0", "sclk_mhz_max": 2400, "sclk_mhz_now": 2259, "power_cap_w": 1400.0}}
    assert bench.gpu_card_of("0000:26:00.0", snap) == "card24" and bench.gpu_card_of("0000:99:00.0", snap) is None
    assert bench.sclk_now_mhz("card24", str(drm)) == 2259.0
    assert bench._sclk_levels("S: 94Mhz *\n0: 500Mhz\n1: 2400Mhz\n") == ([94, 500, 2400], 94)
    assert bench.gpu_sysfs_snapshot(str(tmp_path / "missing")) == {}


def test_recorded_traffic_is_flagged_stale_when_the_kernel_sources_changed(tmp_path):
    """`roofline.traffic` quotes a recorded rocprofv3 --pmc pass; the record carries a fingerprint of the kernel sources it was taken
    from (tools/source_fingerprint.py) and bench.py says `traffic_stale` when this tree's sources differ -- or when the record is
    older than the fingerprints (rounds 2-5)."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench, source_fingerprint as sf
    fresh, old, other = tmp_path / "a.txt", tmp_path / "b.txt", tmp_path / "c.txt"
    fresh.write_text(sf.header_line() + "\n== KZG commit 2^20\n")
    old.write_text("== KZG commit 2^20\n")
    other.write_text("# source_fingerprint msm=0123456789abcdef ntt=%s\n" % sf.fingerprint("ntt"))
    assert bench.traffic_is_stale(str(fresh), "msm") is False and bench.traffic_is_stale(str(fresh), "ntt") is False
    assert bench.traffic_is_stale(str(old), "msm") is True
    assert bench.traffic_is_stale(str(other), "msm") is True and bench.traffic_is_stale(str(other), "ntt") is False


def test_no_collective_step_runs_on_one_rank_only():
    """Round 6's first rehearsals hung for their whole timeout: the clock-under-load leg ran the headline step -- which holds an
    all-gather when N > 1 -- under `if rank == 0`.  Static guard: inside bench.py no call of a function that may issue a collective
    sits under a condition on `rank` (an `if` statement or a conditional expression)."""
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    collective = {"srs_step", "msm_step", "timed", "leg_e2e_kzg", "leg_strong_msm", "leg_strong_ntt", "leg_clock_under_load", "barrier_sync", "max_over_ranks",
                  "all_gather_partials", "all_gather_values", "sharded_open_quotient", "ntt_sharded", "sharded_msm"}

    def mentions_rank(node):
        return any(isinstance(n, ast.Name) and n.id == "rank" or isinstance(n, ast.Attribute) and n.attr == "rank" for n in ast.walk(node))

    def calls(node):
        out = []
        for n in ast.walk(node):
            if isinstance(n, ast.Call):
                f = n.func
                name = f.id if isinstance(f, ast.Name) else (f.attr if isinstance(f, ast.Attribute) else None)
                if name in collective:
                    out.append((name, n.lineno))
        return out

    bad = []
    for node in ast.walk(tree):
        if isinstance(node, ast.If) and mentions_rank(node.test):
            for part in node.body + node.orelse:
                bad += calls(part)
        if isinstance(node, ast.IfExp) and mentions_rank(node.test):
            bad += calls(node.body) + calls(node.orelse)
    assert not bad, "collective calls under a condition on the rank: %s" % bad

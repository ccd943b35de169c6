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
    body = src[src.index("def launch_ranks"):src.index("def main")]
    assert "import torch" not in body and "device_count" not in body


def test_multi_rank_line_describes_its_process_group_and_exchanges():
    """VERDICT r03 #8b: what the first real SCALE run must carry, checked on the committed one-GPU rehearsals (world 2, 4, 8):
    world_size == N, the nccl backend named for a real run, exchanges > 0, and the sharded transform's bytes per rank and
    exchange = (N - 1) / N * n / N * 32."""
    import glob
    rounds = sorted({os.path.basename(f)[:8] for f in glob.glob(os.path.join(ROOT, "profiles", "r0*final_rehearsal_world*_shared_gpu.json"))})
    recs = sorted(glob.glob(os.path.join(ROOT, "profiles", rounds[-1] + "_rehearsal_world*_shared_gpu.json")))       # the newest round's set
    assert len(recs) >= 3, "world-2/4/8 rehearsals missing from profiles/"

    for path in recs:
        rec = json.loads([l for l in open(path) if l.startswith("{")][-1])
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
        e2e = rec["e2e_kzg"]
        assert e2e["n_gpus"] == N and "error" not in e2e
        assert {"intt", "setup_srs_powers", "srs_prepare", "commit_local", "open_local", "gather_and_fold"} <= set(e2e["stages_ms"])
        assert e2e["trapdoor_identities_hold"] is True and e2e["overlapped_results_identical"] is True
        if "srs_points_this_rank" in e2e:              # (records from round 5 on)
            assert e2e["srs_points_total"] == 1 << e2e["log2_degree"] and e2e["srs_points_this_rank"] * N == e2e["srs_points_total"]

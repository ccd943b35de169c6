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

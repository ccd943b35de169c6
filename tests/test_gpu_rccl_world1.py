"""The RCCL branch of the multi-GPU path, run on ONE GPU before any multi-GPU job does (SURVEY 8e;
north_star "final RCCL reduce over xGMI").  Every committed rehearsal and every CPU test went through gloo, and at world 1
myzkp_amd/sharded.py returns before its collectives, so `init_process_group("nccl")`, all_gather_into_tensor /
all_to_all_single on int64 device tensors and the ordering between torch's NCCL stream and the stream the C ABI launches on
had never executed.  The child (tests/rccl_world1_child.py) forms a one-rank `nccl` group, switches the shortcuts off and
checks the gathered + folded MSM against the oracle; this parent only starts it and reads its line -- it never touches HIP
for this test, and the child is an ordinary child process."""
import json, os, subprocess, sys
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(cmd, timeout):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MZK_BENCH_SHARED_GPU_TEST")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


def test_one_rank_nccl_group_runs_every_collective_of_the_sharded_path():
    r = _run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_child.py")], 900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, "child failed (rc %d):\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    rec = json.loads(lines[-1])
    assert rec["backend"] == "nccl" and rec["world_size"] == 1
    assert rec["librccl_mapped"] is True and rec["libmzk_hip_mapped"] is True
    for key in ("msm_gather_fold_equals_oracle", "commit_gather_fold_equals_oracle", "unsynchronised_chain_equals_synchronised",
                "all_to_all_part_equals_itself", "forced_sharded_transform_equals_plain", "forced_sharded_transform_equals_oracle"):
        assert rec[key] is True, (key, rec)


def test_bench_sharded_legs_through_the_nccl_branch_on_one_gpu():
    """`bench.py --gpus 1 --force-process-group`: the weak-scaling headline (partial -> gather -> fold every step), the
    fixed-size MSM, the end-to-end KZG and the sharded transform all run their exchange steps over a one-rank nccl group."""
    r = _run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-process-group", "--steps", "3", "--warmup", "1", "--log2n", "16",
              "--e2e-log2n", "16", "--strong-log2n", "16", "--strong-ntt-log2n", "16", "--extra-sizes", "", "--skip-cpu", "--sharded-legs-only"], 1200)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode == 0 and lines, "bench failed (rc %d):\n%s\n%s" % (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    rec = json.loads(lines[-1])
    pg = rec["process_group"]
    assert pg["backend"] == "nccl" and pg["world_size"] == 1 and pg["forced_collectives"] is True
    assert rec["parity"]["msm_bit_exact_vs_cpu"] is True and rec["parity"]["kzg_commit_srs_bit_exact_vs_cpu"] is True
    assert rec["strong_scaling_msm"]["trapdoor_identity_holds"] is True
    assert rec["e2e_kzg"]["trapdoor_identities_hold"] is True
    assert rec["strong_scaling_ntt"]["every_part_equals_single_gpu_transform"] is True
    assert rec["strong_scaling_ntt"]["exchanges"]["contiguous_to_contiguous"] >= 1

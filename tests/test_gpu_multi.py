"""Multi-GPU MSM / KZG commit INSIDE the C ABI (mzk_init_devices + the *_multi entry points; SURVEY 8e, BASELINE
configs[3]) -- one process, W contexts.  Contexts may share a device ordinal, so the whole path (W streams, W
workspaces and table caches, pinned gather of the 128-byte partials, fold on context 0) runs on a one-GPU box; the
last case uses every visible device."""
import numpy as np
import pytest
import orc
from orc import FR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd as m
    yield m
    m.init(0)          # leave the process in the single-context state the other modules expect


def _inputs(n, seed):
    s = orc.synth_vector(FR, seed, n)
    p = orc.synth_points(seed + 1, n)
    if n > 20:
        p[17] = 0
        s[3] = 0
    return s, p


@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_msm_multi_equals_oracle(mz, world):
    mz.init_devices([0] * world)
    assert mz.ctx_count() == world
    for n in (0, 1, world - 1, 5000):
        s, p = _inputs(n, 700 + world)
        assert mz.msm_g1_multi(s, p) == orc.msm_fast(s, p), (world, n)
    # context 0 still serves the single-device entry points, and so does any other context once selected
    s, p = _inputs(300, 77)
    want = orc.msm_fast(s, p)
    assert mz.msm_g1(s, p) == want
    mz.ctx_select(world - 1)
    assert mz.msm_g1(s, p) == want
    mz.ctx_select(0)


def test_shard_range_matches_the_python_sharding(mz):
    from myzkp_amd import sharded
    for n in (0, 1, 7, 1000, (1 << 24) + 5):
        for world in (1, 2, 3, 8):
            assert [mz.shard_range(n, r, world) for r in range(world)] == [sharded.shard_range(n, r, world) for r in range(world)]


@pytest.mark.parametrize("world", [2, 5])
def test_srs_multi_commit_host_and_device_coefficients(mz, world):
    import torch
    mz.init_devices([0] * world)
    n = (1 << 15) + 11                   # world 2: shards above the window-table threshold (2^14), world 5: below it
    s, p = _inputs(n, 800 + world)
    h = mz.SrsMulti(powers=p)
    assert h.world == world and h.lo[0] == 0 and h.lo[-1] == n
    assert h.commit(s) == orc.msm_fast(s, p)
    m = n - 4097                         # shorter polynomial: the last shards see fewer (or no) coefficients
    assert h.commit(s[:m]) == orc.msm_fast(s[:m], p[:m])
    assert h.commit(s[:5]) == orc.msm_fast(s[:5], p[:5])
    with pytest.raises(mz.MzkError) as e:
        h.commit(np.concatenate([s, s[:1]]))
    assert e.value.code == -5            # powers[i] out of bounds, polynomial.rs:162
    # device-resident shards (all contexts share cuda:0 here)
    dev = torch.device("cuda", 0)
    shards = [torch.from_numpy(s[h.lo[r]:h.lo[r + 1]].view(np.int64).reshape(-1).copy()).to(dev) for r in range(world)]
    torch.cuda.synchronize()
    assert h.commit_dev([t.data_ptr() for t in shards], n) == orc.msm_fast(s, p)
    h.close()


def test_setup_srs_multi_then_commit_is_the_trapdoor_identity(mz):
    world = 3
    mz.init_devices([0] * world)
    n = (1 << 16) + 3
    alpha = orc.from_limbs(orc.synth_vector(FR, 901, 1))[0]
    s = orc.synth_vector(FR, 902, n)
    want = orc.ec_mul(0, (1, 2), orc.poly_eval(FR, s, alpha))
    for with_tables in (0, 1):
        h = mz.SrsMulti(alpha=alpha, max_d=n - 1, with_tables=with_tables)
        assert h.commit(s) == want, with_tables
        h.close()


def test_every_visible_device(mz):
    import torch
    count = torch.cuda.device_count()
    mz.init_devices(list(range(count)))
    s, p = _inputs(20000, 990)
    assert mz.msm_g1_multi(s, p) == orc.msm_fast(s, p)
    alpha = 0x1234567
    n = 1 << 14
    h = mz.SrsMulti(alpha=alpha, max_d=n - 1)
    c = orc.synth_vector(FR, 991, n)
    assert h.commit(c) == orc.ec_mul(0, (1, 2), orc.poly_eval(FR, c, alpha))
    h.close()


def test_srs_handle_is_shared_by_contexts_on_the_same_device(mz):
    """A handle is plain device memory: a second context on the SAME GPU (own stream + workspace) may commit against it --
    that is how two commits are kept in flight (bench.py `kzg_commit_two_in_flight`).  Only a context on another
    device is refused (MZK_E_ARG; needs two GPUs to provoke)."""
    import torch
    mz.init_devices([0, 0])
    for n in (64, (1 << 14) + 5):
        s, p = _inputs(n, 55 + n)
        want = orc.msm_fast(s, p)
        h = mz.Srs(p)                        # built on context 0
        mz.ctx_select(1)
        assert h.commit(s) == want
        mz.ctx_select(0)
        assert h.commit(s) == want
        if torch.cuda.device_count() > 1:
            mz.init_devices([0, 1])
            mz.ctx_select(1)
            with pytest.raises(mz.MzkError) as e:
                h.commit(s)
            assert e.value.code == -1
            mz.init_devices([0, 0])
        h.close()

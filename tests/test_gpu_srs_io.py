"""SRS persistence (VERDICT r01 missing #1): mzk_srs_save / mzk_srs_load / mzk_srs_download.
The dump holds PublicKeyKZG.powers_1 exactly as it crosses the ABI, so a file written here is readable by anything
that knows `x || y` little-endian limbs; commits against a re-loaded handle must equal the original bit for bit."""
import os, struct
import numpy as np
import pytest
import orc
from orc import FR

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def mz():
    import myzkp_amd
    myzkp_amd.init(0)
    return myzkp_amd


@pytest.mark.parametrize("n", [0, 1, 300, 3000, 5000, (1 << 14) + 7])      # 8-, 10- and 16-bit window tables
@pytest.mark.parametrize("with_tables", [False, True])
def test_save_load_round_trip(mz, tmp_path, n, with_tables):
    p = orc.synth_points(1200 + n, n)
    if n > 10:
        p[5] = 0                                                     # infinity survives the dump
    s = orc.synth_vector(FR, 1201 + n, n)
    want = orc.msm_fast(s, p) if n else (0, 0)
    h = mz.Srs(p)
    path = str(tmp_path / "srs.bin")
    h.save(path, with_tables=with_tables)
    assert np.array_equal(h.download(), p)
    h.close()
    raw = open(path, "rb").read()
    assert raw[:8] == b"MZKSRS\0\0" and struct.unpack_from("<Q", raw, 16)[0] == n
    assert raw[64:64 + 64 * n] == p.tobytes()                        # the ABI wire format, verbatim
    c = 8 if n <= 1024 else (10 if n <= (1 << 14) else 16)             # msm_srs_window_bits (mzk_common.h)
    has_tables = with_tables and n > 0
    assert len(raw) == 64 + 64 * n * ((254 // c + 1) if has_tables else 1)
    for mode in (1, 0, 13):                                          # stored tables / plain points / rebuilt at another width
        g = mz.Srs.load(path, with_tables=mode)
        assert g.n == n
        assert g.commit(s) == want, mode
        assert np.array_equal(g.download(), p)
        g.close()


def test_corrupted_truncated_and_foreign_files_are_rejected(mz, tmp_path):
    n = 500
    p = orc.synth_points(77, n)
    h = mz.Srs(p)
    path = str(tmp_path / "srs.bin")
    h.save(path)
    h.close()
    raw = bytearray(open(path, "rb").read())
    bad = str(tmp_path / "bad.bin")
    flipped = bytearray(raw); flipped[64 + 1000] ^= 1
    for blob in (flipped, raw[:-64], b"not an srs dump at all" * 10, raw[:40]):
        open(bad, "wb").write(bytes(blob))
        with pytest.raises(mz.MzkError) as e:
            mz.Srs.load(bad)
        assert e.value.code == -10
    with pytest.raises(mz.MzkError) as e:
        mz.Srs.load(str(tmp_path / "missing.bin"))
    assert e.value.code == -10


def test_flipped_byte_in_the_table_section_is_rejected_and_format_1_files_rebuild(mz, tmp_path):
    """the points checksum alone let a corrupted table row through (every later commit a wrong point, MZK_OK).
    Format 2 carries a second checksum over the table section; a format-1 file's tables are never trusted."""
    n = 700
    p = orc.synth_points(91, n)
    s = orc.synth_vector(FR, 92, n)
    want = orc.msm_fast(s, p)
    h = mz.Srs(p)
    path = str(tmp_path / "srs.bin")
    h.save(path, with_tables=True)
    h.close()
    raw = bytearray(open(path, "rb").read())
    assert struct.unpack_from("<I", raw, 8)[0] == 2 and len(raw) > 64 + 64 * n * 2
    bad = str(tmp_path / "bad.bin")
    for at in (64 + 64 * n + 5, 64 + 64 * n * 7 + 33, len(raw) - 1):            # first table row, a middle row, the last byte
        blob = bytearray(raw); blob[at] ^= 0x10
        open(bad, "wb").write(bytes(blob))
        with pytest.raises(mz.MzkError) as e:
            mz.Srs.load(bad, with_tables=1)
        assert e.value.code == -10, at
        g = mz.Srs.load(bad, with_tables=0)          # the points section is intact: a load that ignores the tables succeeds
        assert g.commit(s) == want
        g.close()
    # the same bytes relabelled as a format-1 dump (no table checksum) with a corrupted table row: tables are rebuilt, result right
    blob = bytearray(raw); blob[64 + 64 * n * 3 + 9] ^= 0xff
    struct.pack_into("<I", blob, 8, 1)
    open(bad, "wb").write(bytes(blob))
    g = mz.Srs.load(bad, with_tables=1)
    assert g.commit(s) == want
    g.close()


def test_setup_on_device_then_dump_equals_oracle_setup(mz, tmp_path):
    """setup_kzg (kzg.rs:27-40) on the GPU -> handle -> file: the bytes are the oracle's powers_1."""
    import ctypes, torch
    L = mz.lib()
    dev = torch.device("cuda", 0)
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, alpha = 257, 0xabcdef123456789
    pts = torch.empty(n * 8, dtype=torch.int64, device=dev)
    a_l, g_l = mz.to_limbs([alpha], 4), mz.points_to_array([(1, 2)])
    assert L.mzk_kzg_setup_g1_dev(a_l.ctypes.data_as(ctypes.c_void_p), g_l.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(n - 1), ctypes.c_void_p(pts.data_ptr()), st) == 0
    hh = ctypes.c_void_p()
    assert L.mzk_srs_from_device(ctypes.c_void_p(pts.data_ptr()), ctypes.c_size_t(n), ctypes.byref(hh), st) == 0
    path = str(tmp_path / "setup.bin")
    assert L.mzk_srs_save(hh, os.fsencode(path), 0) == 0
    L.mzk_srs_free(hh)
    assert open(path, "rb").read()[64:] == orc.kzg_setup_ref(alpha, n - 1).tobytes()


def test_dumps_and_loads_under_a_table_budget(mz, tmp_path):
    """A handle degraded by mzk_set_table_budget dumps its points only (its every-2nd-table layout is rebuilt on load); a dump WITH
    full tables loaded under a budget they do not fit degrades instead of failing -- commitments bit-identical all the way."""
    import ctypes
    L = mz.lib()
    L.mzk_srs_table_bytes.restype = ctypes.c_size_t
    n = (1 << 15) + 5
    p = orc.synth_points(77, n)
    s = orc.synth_vector(FR, 78, n)
    want = orc.msm_fast(s, p)
    full_path, deg_path = str(tmp_path / "full.bin"), str(tmp_path / "degraded.bin")
    h = mz.Srs(p)
    assert L.mzk_srs_bucket_sets(h._h) == 1
    h.save(full_path, with_tables=True)
    h.close()
    try:
        L.mzk_set_table_budget(ctypes.c_size_t(9 * n * 64))              # 16 tables do not fit, 8 do
        h = mz.Srs(p)
        assert L.mzk_srs_bucket_sets(h._h) == 2 and h.commit(s) == want
        h.save(deg_path, with_tables=True)
        assert os.path.getsize(deg_path) == 64 + 64 * n                  # points only
        h.close()
        g = mz.Srs.load(full_path, with_tables=1)                        # stored tables: 16 rows, over the budget
        assert L.mzk_srs_bucket_sets(g._h) == 2 and L.mzk_srs_table_bytes(g._h) <= 9 * n * 64
        assert g.commit(s) == want and np.array_equal(g.download(), p)
        g.close()
        L.mzk_set_table_budget(ctypes.c_size_t(0))
        g = mz.Srs.load(deg_path, with_tables=1)                         # no budget any more: full tables, built from the points
        assert L.mzk_srs_bucket_sets(g._h) == 1 and g.commit(s) == want
        g.close()
        g = mz.Srs.load(full_path, with_tables=1)                        # and the stored tables are used as they are
        assert L.mzk_srs_bucket_sets(g._h) == 1 and g.commit(s) == want
        g.close()
    finally:
        L.mzk_set_table_budget(ctypes.c_size_t(0))

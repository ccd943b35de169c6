"""Cost of one mzk_merkle_open_batch call at the C boundary (no Python unpacking): 96 openings of a 2^lg-leaf M128 tree."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import numpy as np
import myzkp_amd as mz, orc
mz.init(0); L = mz.lib()
for lg in (8, 14, 18):
    x = orc.synth_vector(orc.M128, 3, 1 << lg)
    t = mz.MerkleTree(orc.M128, x)
    idx = np.random.default_rng(1).integers(0, 1 << lg, 96).astype(np.uint64)
    buf = (ctypes.c_uint8 * (48 * lg * 96))(); lens = (ctypes.c_uint64 * (lg * 96))(); depth = ctypes.c_size_t()
    def call():
        assert L.mzk_merkle_open_batch(t._h, idx.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(96), buf, ctypes.c_size_t(48), lens, ctypes.byref(depth)) == 0
    for _ in range(20): call()
    t0 = time.perf_counter()
    for _ in range(200): call()
    c_us = (time.perf_counter() - t0) / 200 * 1e6
    t0 = time.perf_counter()
    for _ in range(50): t.open_many(idx)
    py_us = (time.perf_counter() - t0) / 50 * 1e6
    print("2^%d leaves, 96 openings: %.1f us per C call, %.1f us through the Python wrapper" % (lg, c_us, py_us), flush=True)

# the query phase of one FRI proof: 10 rounds, trees of 2^14 .. 2^5 leaves, 3 x 32 openings each -- round by round vs one call
trees, lists = [], []
for lg in range(14, 4, -1):
    trees.append(mz.MerkleTree(orc.M128, orc.synth_vector(orc.M128, lg, 1 << lg)))
    lists.append(np.random.default_rng(lg).integers(0, 1 << lg, 96).astype(np.uint64))
T = len(trees)
flat = np.concatenate(lists)
handles = (ctypes.c_void_p * T)(*[t._h for t in trees]); cnt = (ctypes.c_size_t * T)(*[96] * T); depths = (ctypes.c_size_t * T)()
entries = sum(96 * lg for lg in range(14, 4, -1))
buf = (ctypes.c_uint8 * (48 * entries))(); lens = (ctypes.c_uint64 * entries)(); depth = ctypes.c_size_t()
def by_round():
    at = 0
    for t, ix, lg in zip(trees, lists, range(14, 4, -1)):
        assert L.mzk_merkle_open_batch(t._h, ix.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(96), ctypes.byref(buf, 48 * at), ctypes.c_size_t(48),
                                       ctypes.byref(lens, 8 * at), ctypes.byref(depth)) == 0
        at += 96 * lg
def multi():
    assert L.mzk_merkle_open_multi(handles, ctypes.c_size_t(T), flat.ctypes.data_as(ctypes.c_void_p), cnt, buf, ctypes.c_size_t(48), lens, depths) == 0
for fn, name in ((by_round, "10 x mzk_merkle_open_batch"), (multi, "1 x mzk_merkle_open_multi")):
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(200): fn()
    print("query phase, 10 rounds x 96 openings: %-28s %.1f us" % (name, (time.perf_counter() - t0) / 200 * 1e6), flush=True)

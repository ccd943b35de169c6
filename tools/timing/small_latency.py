import sys, time; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, orc, myzkp_amd as mz
mz.init(0)
def t(label, f, reps=20):
    f(); f()
    t0 = time.perf_counter()
    for _ in range(reps): f()
    print("%-44s %8.3f ms" % (label, (time.perf_counter() - t0) / reps * 1e3), flush=True)
for n in (16, 1024, 1 << 14):
    s = orc.synth_vector(orc.FR, 1, n); p = orc.synth_points(2, n)
    t("msm_g1 (host buffers) n=%d" % n, lambda: mz.msm_g1(s, p))
    srs = mz.kzg_setup_g1(12345, n - 1)
    t("kzg_setup_g1 n=%d" % n, lambda: mz.kzg_setup_g1(12345, n - 1), 5)
    t("kzg_commit n=%d" % n, lambda: mz.kzg_commit(s, srs))
    t("kzg_open n=%d" % n, lambda: mz.kzg_open(s, 777, srs))
    h = mz.Srs(srs)
    t("Srs.commit n=%d" % n, lambda: h.commit(s))
    h.close()
    lg = n.bit_length() - 1
    for fid, name in ((orc.FR, "Fr"), (orc.M128, "M128")):
        v = orc.synth_vector(fid, 3, n); root = orc.root_of(fid, lg)
        t("ntt %s n=%d" % (name, n), lambda: mz.ntt(fid, root, v))
        t("intt %s n=%d" % (name, n), lambda: mz.intt(fid, root, v))
    v = orc.synth_vector(orc.M128, 3, n)
    t("merkle_commit_field M128 n=%d" % n, lambda: mz.merkle_commit_field(orc.M128, v))
    t("fri_fold M128 n=%d" % n, lambda: mz.fri_fold(orc.M128, v, 5, orc.M128_GEN, orc.root_of(orc.M128, lg)))

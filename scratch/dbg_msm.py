import sys; sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, orc, myzkp_amd as mz
mz.init(0)
G = (1, 2)
for k in [1, 2, 3, 255, 256, 257, 65535, 65536, 1 << 20, (1 << 200) + 12345, orc.P_FR - 1]:
    s = orc.to_limbs([k], 4); p = orc.pts_to_arr([G])
    got = mz.msm_g1(s, p); want = orc.ec_mul(0, G, k)
    print(hex(k)[:20], "OK" if tuple(got) == tuple(want) else "BAD got=%s" % (str(got)[:60]))

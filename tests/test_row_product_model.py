"""The integer model of the row-cooperative Montgomery product (tools/row_product_model.py mirrors myzkp_amd/csrc/mzk_row.h lane
by lane): column accumulators stay below 2^64, the ripple-free normalisation keeps the value, m = -C / p (mod R), the carry of
the low half is exact from one limb, the fused pair of the last product level fits, and the slack K p constants never borrow --
on random operands and on the extreme limb patterns the kernels allow.  CPU only."""
import os, sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import row_product_model as m


def test_row_product_model_bounds_and_values():
    assert m.self_test(rounds=120, seed=2026)


def test_generated_slack_constants_match_the_model():
    import re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "myzkp_amd", "csrc", "mzk_constants.h")).read()
    body = hdr[hdr.index("struct FqRowParams"):]
    rows = re.findall(r"\{((?:0x[0-9a-f]{8}u(?:, )?){16})\}", body[body.index("KPS[17][16]"):])
    assert len(rows) == 17
    for K in (3, 5, 7, 8, 10):
        got = [int(x[:-1], 16) for x in rows[K].split(", ")]
        assert got[:9] == m.kps(K, m.P_FQ) and got[9:] == [0] * 7
    np_ = [int(x[:-1], 16) for x in re.search(r"NPRIME\[9\] = \{([^}]*)\}", body).group(1).split(", ")]
    assert m.value(np_) == (-pow(m.P_FQ, -1, m.Rr)) % m.Rr
    pinv = int(re.search(r"PINV29 = (0x[0-9a-f]+)u", body).group(1), 16)
    assert (pinv * m.P_FQ) % (1 << 29) == 1

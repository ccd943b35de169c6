#!/usr/bin/env python3
"""Generate tests/golden/*.json -- golden input/output vectors for the MSM / NTT hot path.

The reference is Rust and cannot be run in this environment (no rustc/cargo), so these vectors come
from an INDEPENDENT Python big-integer transcription of the reference algorithms (written against the
Rust sources, cited below; paths relative to myzkp/src/modules/).  Python's unbounded ints stand in
for num-bigint: every modular operation is `op` then `% p`, exactly as field.rs:157-183 does.

The only data read from /root/reference is the Rescue-Prime parameter set (numeric literals of
zkstark/rescueprime.rs:30-399) whose two hashes (rescueprime.rs:606-620) are the reference's only
large-number known-answer vectors for M128 arithmetic; that extraction runs only when
/root/reference exists, and its output (numbers, not source) is committed.

Run:  python3 tests/golden/make_golden.py      (deterministic; rewrites the JSON files in place)
"""
import hashlib, json, os, random, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))

# --- moduli: field.rs:428-431, curve/bn128.rs:19-22, zkstark/fri.rs:408 -------------------------
FR = 21888242871839275222246405745257275088548364400416034343698204186575808495617
FQ = 21888242871839275222246405745257275088696311157297823662689037894645226208583
M128 = 270497897142230380135924736767050121217
M128_GEN = 85408008396924667383611388730472331217  # fri.rs:436-438, order 2^119
FR_OMEGA28 = pow(5, (FR - 1) >> 28, FR)              # Fr has no root helper upstream

def m128_root(n):  # get_nth_root_of_m128, fri.rs:423-447
    assert n <= 1 << 119 and n & (n - 1) == 0
    root, order = M128_GEN, 1 << 119
    while order != n:
        root = root * root % M128
        order >>= 1
    return root

def fr_root(n):
    assert n <= 1 << 28 and n & (n - 1) == 0
    return pow(FR_OMEGA28, (1 << 28) // n, FR)

# --- ntt.rs:7-48 / :50-64 ----------------------------------------------------------------------
def ntt(root, values, p):
    n = len(values)
    assert n & (n - 1) == 0
    if n <= 1:
        return list(values)
    assert pow(root, n, p) == 1 and pow(root, n // 2, p) != 1
    half = n // 2
    odds = ntt(root * root % p, values[1::2], p)
    evens = ntt(root * root % p, values[0::2], p)
    return [(evens[i % half] + pow(root, i, p) * odds[i % half]) % p for i in range(n)]

def intt(root, values, p):
    if len(values) == 1:
        return list(values)
    ninv = pow(len(values), -1, p)
    return [ninv * v % p for v in ntt(pow(root, -1, p), values, p)]

def poly_eval(coef, x, p):  # polynomial.rs:120-128
    res, tp = 0, 1
    for c in coef:
        res = (res + tp * c) % p
        tp = tp * x % p
    return res

def trim(c):
    c = list(c)
    while c and c[-1] == 0:
        c.pop()
    return c

def poly_mul(a, b, p):  # polynomial.rs:302-316
    a, b = trim(a), trim(b)
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % p
    return trim(out)

def fast_multiply(a, b, root, root_order, p):  # ntt.rs:66-116
    assert pow(root, root_order, p) == 1 and pow(root, root_order // 2, p) != 1
    if not trim(a) or not trim(b):
        return []
    degree = len(trim(a)) - 1 + len(trim(b)) - 1
    if degree < 8:
        return poly_mul(a, b, p)
    order = root_order
    while degree < order // 2:
        root = root * root % p
        order //= 2
    x = list(a) + [0] * (order - len(a))
    y = list(b) + [0] * (order - len(b))
    X, Y = ntt(root, x, p), ntt(root, y, p)
    return intt(root, [u * v % p for u, v in zip(X, Y)], p)

def fft(a, omega, p):  # polynomial.rs:278-300
    n = len(a)
    if n == 1:
        return
    even, odd = a[0::2], a[1::2]
    fft(even, omega * omega % p, p)
    fft(odd, omega * omega % p, p)
    w = 1
    for i in range(n // 2):
        t = w * odd[i] % p
        a[i] = (even[i] + t) % p
        a[i + n // 2] = (even[i] - t) % p
        w = w * omega % p

def fft_multiply(a, b, omega, p):  # polynomial.rs:242-276
    m = len(a) + len(b) - 1
    n = 1
    while n < m:
        n <<= 1
    x = list(a) + [0] * (n - len(a))
    y = list(b) + [0] * (n - len(b))
    fft(x, omega, p); fft(y, omega, p)
    c = [u * v % p for u, v in zip(x, y)]
    fft(c, pow(omega, -1, p), p)
    ninv = pow(n, -1, p)
    return trim([v * ninv % p for v in c][:m])

def fast_coset_evaluate(coef, offset, generator, order, p):  # ntt.rs:254-269 + polynomial.rs:167-174
    scaled = [pow(offset, i, p) * c % p for i, c in enumerate(coef)]
    return ntt(generator, scaled + [0] * (order - len(coef)), p)

# --- curve.rs:44-191 affine group law ----------------------------------------------------------
INF = None
def ec_double(P, a, p):
    if P is INF:
        return INF
    x, y = P
    s = (3 * x * x + a) * pow(2 * y, -1, p) % p
    nx = (s * s - x - x) % p
    return (nx, (-s * nx + s * x - y) % p)

def ec_add(P, Q, a, p):
    if P is INF:
        return Q
    if Q is INF:
        return P
    if P[0] == Q[0] and P[1] == Q[1]:
        return ec_double(P, a, p)
    if P[0] == Q[0]:
        return INF
    s = (Q[1] - P[1]) * pow(Q[0] - P[0], -1, p) % p
    nx = (s * s - P[0] - Q[0]) % p
    return (nx, (-s * nx + (s * P[0] - P[1])) % p)

def ec_mul(P, k, a, p):  # curve.rs:168-191
    assert k >= 0
    res, cur = INF, P
    while k:
        if k & 1:
            res = ec_add(res, cur, a, p)
        cur = ec_double(cur, a, p)
        k >>= 1
    return res

G1 = (1, 2)  # bn128.rs:186-188
def g1_mul(P, k): return ec_mul(P, k, 0, FQ)
def g1_add(P, Q): return ec_add(P, Q, 0, FQ)

def msm(scalars, points):  # polynomial.rs:156-165
    acc = INF
    for s, P in zip(scalars, points):
        acc = g1_add(acc, g1_mul(P, s % FR))
    return acc

def kzg_setup(alpha, max_d):  # kzg.rs:27-40 with a fixed trapdoor
    out, ap = [], 1
    for _ in range(max_d + 1):
        out.append(g1_mul(G1, ap))
        ap = ap * alpha % FR
    return out

def kzg_open(coef, u, powers):  # kzg.rs:61-72 with div_rem_ref (polynomial.rs:371-405)
    y = poly_eval(coef, u, FR)
    rem = trim([(c - (y if i == 0 else 0)) % FR for i, c in enumerate(coef)] if coef else [(-y) % FR])
    quo = [0] * max(len(rem) - 1, 0)
    d0 = (-u) % FR
    while len(rem) >= 2:
        lead = rem[-1]
        dd = len(rem) - 2
        quo[dd] = lead
        rem[dd] = (rem[dd] - lead * d0) % FR
        rem[dd + 1] = (rem[dd + 1] - lead) % FR
        rem = trim(rem)
    quo = trim(quo)
    return y, msm(quo, powers)

def poly_divrem(a, b, p):  # polynomial.rs:371-405
    a, b = trim(a), trim(b)
    if not b or len(a) < len(b):
        return [], a
    inv = pow(b[-1], -1, p)
    quo = [0] * (len(a) - len(b) + 1)
    rem = list(a)
    while len(rem) >= len(b):
        lead = rem[-1] * inv % p
        dd = len(rem) - len(b)
        quo[dd] = lead
        for i in range(len(b)):
            rem[dd + i] = (rem[dd + i] - lead * b[i]) % p
        rem = trim(rem)
    return trim(quo), rem

def from_monomials(xs, p):  # polynomial.rs:202-212
    out = [1]
    for x in xs:
        out = poly_mul(out, [(-x) % p, 1], p)
    return out

def interpolate(xs, ys, p):  # polynomial.rs:177-200
    num = from_monomials(xs, p)
    res = []
    for j in range(len(xs)):
        den = 1
        for i in range(len(xs)):
            if i != j:
                den = den * (xs[j] - xs[i]) % p
        q, _ = poly_divrem(num, [(-xs[j]) * den % p, den], p)
        term = [c * ys[j] % p for c in q]
        res = [((res[i] if i < len(res) else 0) + (term[i] if i < len(term) else 0)) % p for i in range(max(len(res), len(term)))]
    return trim(res)

def kzg_batch_open(coef, us, powers):  # kzg.rs:74-88
    ys = [poly_eval(coef, u, FR) for u in us]
    ip = interpolate(us, ys, FR)
    z = from_monomials(us, FR)
    diff = [((coef[i] if i < len(coef) else 0) - (ip[i] if i < len(ip) else 0)) % FR for i in range(max(len(coef), len(ip)))]
    q, rem = poly_divrem(diff, z, FR)
    assert not rem
    return ys, msm(q, powers)

def kzg_degree_bound(coef, powers, d):  # kzg.rs:121-134
    max_d = len(powers) - 1
    r = [0] * (max_d - d) + list(coef)
    return msm(trim(r), powers)

def fri_fold(cw, alpha, offset, omega, p):  # fri.rs:182-193
    h = len(cw) // 2
    two_inv = pow(2, -1, p)
    out = []
    for i in range(h):
        q = alpha * pow(offset * pow(omega, i, p) % p, -1, p) % p
        out.append(two_inv * ((1 + q) * cw[i] + (1 - q) * cw[h + i]) % p)
    return out

def pt(P):
    return [0, 0] if P is INF else [P[0], P[1]]

def S(x):  # JSON-safe big ints
    if isinstance(x, (list, tuple)):
        return [S(v) for v in x]
    return str(x)

def dump(name, obj):
    with open(os.path.join(HERE, name), "w") as f:
        json.dump(obj, f, indent=1)
        f.write("\n")

def sha_le(values, nbytes):
    h = hashlib.sha256()
    for v in values:
        h.update(int(v).to_bytes(nbytes, "little"))
    return h.hexdigest()

# --- Rescue-Prime over M128 (rescueprime.rs:403-455 hash) ----------------------------------------
def extract_rescue(path):
    src = open(path).read()
    body = src[src.index("pub fn new()"):src.index("pub fn hash(")]
    nums = [int(x) for x in re.findall(r'from_str\("(\d+)"\)', body)]
    # order in the literal: field modulus, alphainv, 4 mds, 4 mdsinv, then the round constants
    assert nums[0] == M128
    alphainv = nums[1]
    mds = [[nums[2], nums[3]], [nums[4], nums[5]]]
    mdsinv = [[nums[6], nums[7]], [nums[8], nums[9]]]
    rc = nums[10:]
    m = int(re.search(r"\bm: (\d+)", body).group(1))
    n = int(re.search(r"\bn: (\d+)", body).group(1))
    alpha = int(re.search(r"\balpha: (\d+)", body).group(1))
    assert len(rc) == 2 * m * n, (len(rc), m, n)
    return dict(m=m, n=n, alpha=alpha, alphainv=alphainv, mds=mds, mdsinv=mdsinv, round_constants=rc)

def rescue_hash(par, x):
    p = M128
    m, N = par["m"], par["n"]
    state = [x] + [0] * (m - 1)
    for r in range(N):
        state = [pow(s, par["alpha"], p) for s in state]
        state = [sum(par["mds"][i][j] * state[j] for j in range(m)) % p for i in range(m)]
        state = [(state[i] + par["round_constants"][2 * r * m + i]) % p for i in range(m)]
        state = [pow(s, par["alphainv"], p) for s in state]
        state = [sum(par["mds"][i][j] * state[j] for j in range(m)) % p for i in range(m)]
        state = [(state[i] + par["round_constants"][2 * r * m + m + i]) % p for i in range(m)]
    return state[0]

# --- rows widened after the hot path: Merkle over bincode leaves, fast_coset_divide, G2 ---------------
def bincode_field(v):   # bincode 1.3 of FiniteFieldElement{value: BigInt}: Sign as i8, u32 digits as a u64-length sequence
    digits = []
    while v:
        digits.append(v & 0xffffffff)
        v >>= 32
    return bytes([1 if digits else 0]) + len(digits).to_bytes(8, "little") + b"".join(d.to_bytes(4, "little") for d in digits)

def merkle_commit(leaves):   # merkle.rs:15-25
    if len(leaves) == 1:
        return leaves[0]
    mid = len(leaves) // 2
    return hashlib.sha3_256(merkle_commit(leaves[:mid]) + merkle_commit(leaves[mid:])).digest()

def merkle_open(index, leaves):   # merkle.rs:28-46
    if len(leaves) == 2:
        return [leaves[1 - index]]
    mid = len(leaves) // 2
    if index < mid:
        return merkle_open(index, leaves[:mid]) + [merkle_commit(leaves[mid:])]
    return merkle_open(index - mid, leaves[mid:]) + [merkle_commit(leaves[:mid])]

def fast_coset_divide(lhs, rhs, offset, root, order, p):   # ntt.rs:271-330
    lhs, rhs = trim(list(lhs)), trim(list(rhs))
    dl, dr = len(lhs) - 1, len(rhs) - 1
    assert pow(root, order, p) == 1 and pow(root, order // 2, p) != 1 and rhs and dr < dl
    if dl < 8:
        return poly_divrem(lhs, rhs, p)[0]
    while dl < order // 2:
        root, order = root * root % p, order // 2
    a = [c * pow(offset, i, p) % p for i, c in enumerate(lhs)] + [0] * (order - dl - 1)
    b = [c * pow(offset, i, p) % p for i, c in enumerate(rhs)] + [0] * (order - dr - 1)
    ea, eb = ntt(root, a, p), ntt(root, b, p)
    q = [x * (pow(y, -1, p) if y else 0) % p for x, y in zip(ea, eb)]      # inverse(0) = 0 (field.rs:209-232)
    sq = intt(root, q, p)[: dl - dr + 1]
    oi = pow(offset, -1, p)
    return [c * pow(oi, i, p) % p for i, c in enumerate(sq)]

G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))   # bn128.rs:190-206
def f2mul(a, b): return ((a[0] * b[0] - a[1] * b[1]) % FQ, (a[0] * b[1] + a[1] * b[0]) % FQ)   # efield.rs:351-353, mod x^2+1
def f2sub(a, b): return ((a[0] - b[0]) % FQ, (a[1] - b[1]) % FQ)
def f2add(a, b): return ((a[0] + b[0]) % FQ, (a[1] + b[1]) % FQ)
def f2inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, FQ)
    return (a[0] * n % FQ, -a[1] * n % FQ)
G2_INF = None
def g2_add(P, Q):   # curve.rs:56-129 over Fq2, a = 0
    if P is None: return Q
    if Q is None: return P
    (x1, y1), (x2, y2) = P, Q
    if x1 == x2:
        if y1 != y2: return None
        lam = f2mul(f2mul((3, 0), f2mul(x1, x1)), f2inv(f2add(y1, y1)))
    else:
        lam = f2mul(f2sub(y2, y1), f2inv(f2sub(x2, x1)))
    x3 = f2sub(f2sub(f2mul(lam, lam), x1), x2)
    return (x3, f2sub(f2mul(lam, f2sub(x1, x3)), y1))
def g2_mul(P, k):   # curve.rs:168-191
    acc, cur = None, P
    while k:
        if k & 1: acc = g2_add(acc, cur)
        cur = g2_add(cur, cur)
        k >>= 1
    return acc
def g2pt(P): return [[0, 0], [0, 0]] if P is None else [list(P[0]), list(P[1])]

def next_rows():
    rng = random.Random(0x6e657874)
    out = {}
    # Merkle roots / paths of field codewords (fri.rs:160-166) and of raw byte leaves (merkle.rs:76-93)
    mk = []
    for name, p in (("M128", M128), ("Fr", FR)):
        for lg in (0, 1, 3, 6):
            vals = [rng.randrange(p) for _ in range(1 << lg)]
            if lg >= 3:
                vals[1], vals[2], vals[5] = 0, 7, 1 << 64
            leaves = [bincode_field(v) for v in vals]
            case = dict(field=name, values=S(vals), root=merkle_commit(leaves).hex())
            if lg >= 1:
                idx = rng.randrange(1 << lg)
                case["open_index"] = idx
                case["path"] = [e.hex() for e in merkle_open(idx, leaves)]
            mk.append(case)
    leaves = [b"leaf1", b"leaf2", b"leaf3", b"leaf4"]
    out["merkle_field"] = mk
    out["merkle_bytes"] = dict(leaves=[l.hex() for l in leaves], root=merkle_commit(leaves).hex(), open_index=2,
                               path=[e.hex() for e in merkle_open(2, leaves)])
    # fast_coset_divide: exact and inexact, both branches (degree < 8 and transform)
    cd = []
    for name, p, rootf, off in (("M128", M128, m128_root, M128_GEN), ("Fr", FR, fr_root, 5)):
        for dq, dr, exact in ((3, 2, True), (20, 9, True), (45, 17, False), (5, 1, False)):
            q = [rng.randrange(p) for _ in range(dq)] + [rng.randrange(1, p)]
            r = [rng.randrange(p) for _ in range(dr)] + [rng.randrange(1, p)]
            lhs = poly_mul(q, r, p) if exact else [rng.randrange(p) for _ in range(dq + dr)] + [rng.randrange(1, p)]
            order = 256
            cd.append(dict(field=name, lhs=S(lhs), rhs=S(r), offset=S(off), root=S(rootf(order)), root_order=order,
                           out=S(fast_coset_divide(lhs, r, off, rootf(order), order, p))))
    out["fast_coset_divide"] = cd
    # G2: multiples of the generator, an MSM, powers_2
    ks = [1, 2, 3, 9, FR - 1, rng.randrange(FR)]
    out["g2_mul"] = [dict(k=S(k), out=S(g2pt(g2_mul(G2_GEN, k)))) for k in ks] + [dict(k=S(FR), out=S(g2pt(g2_mul(G2_GEN, FR))))]
    pts = [g2_mul(G2_GEN, rng.randrange(1, FR)) for _ in range(4)] + [None]
    sc = [rng.randrange(FR) for _ in range(5)]
    acc = None
    for k, P in zip(sc, pts):
        acc = g2_add(acc, g2_mul(P, k) if P is not None else None)
    out["g2_msm"] = dict(scalars=S(sc), points=S([g2pt(P) for P in pts]), out=S(g2pt(acc)))
    out["g2_powers"] = dict(alpha="7", max_d=3, powers=S([g2pt(g2_mul(G2_GEN, pow(7, i, FR))) for i in range(4)]))
    dump("next_rows_vectors.json", out)

def main():
    rng = random.Random(0x4d595a4b50)  # "MYZKP"

    # ---------------- field KATs ----------------
    field = {"moduli": {"Fr": S(FR), "Fq": S(FQ), "M128": S(M128)}, "cases": []}
    for name, p in (("Fr", FR), ("Fq", FQ), ("M128", M128)):
        edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << (p.bit_length() - 1)) - 1, (1 << (p.bit_length() - 1)) % p]
        vals = edge + [rng.randrange(p) for _ in range(24)]
        for i in range(0, len(vals), 2):
            a, b = vals[i], vals[(i * 7 + 3) % len(vals)]
            field["cases"].append(dict(field=name, a=S(a), b=S(b), add=S((a + b) % p), sub=S((a - b) % p),
                                       mul=S(a * b % p), neg=S((-a) % p), inv=S(pow(a, -1, p) if a else 0),
                                       pow_e=S(b >> 3), pow=S(pow(a, b >> 3, p))))
    # reference-held KATs
    field["reference_kats"] = {
        "field_rs_491_497": {"p": 17, "a": 7, "inv": 5},
        "field_rs_544_550": {"p": 31, "a_plus_p_eq": [8, 39]},  # -23 == 8 (mod 31)
        "test_fr_cu_16_17": {"p_minus_2_limbs": ["0x43e1f593efffffff", "0x2833e84879b97091", "0xb85045b68181585d", "0x30644e72e131a029"],
                              "p_minus_12_limbs": ["0x43e1f593effffff5", "0x2833e84879b97091", "0xb85045b68181585d", "0x30644e72e131a029"]},
        "field_hpp_9_31": {"FR_MOD_INV_mod_2_64": S((-pow(FR, -1, 1 << 64)) % (1 << 64)), "R2_mod_N": S(pow(2, 512, FR))},
        "m128_generator": S(M128_GEN), "m128_generator_order_log2": 119,
        "fr_omega_2_28": S(FR_OMEGA28),
    }
    assert FR_OMEGA28 == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert pow(M128_GEN, 1 << 119, M128) == 1 and pow(M128_GEN, 1 << 118, M128) != 1
    dump("field_kats.json", field)

    # ---------------- NTT ----------------
    nt = {"cases": []}
    w8 = m128_root(8)
    out8 = ntt(w8, list(range(1, 9)), M128)
    assert w8 == 131076302407280330469229082343774091404 and out8[0] == 36
    nt["cases"].append(dict(field="M128", kind="ntt", root=S(w8), input=S(list(range(1, 9))), output=S(out8)))
    # ntt.rs:346-374: n = 256, coefficients 1..256
    w256 = m128_root(256)
    out256 = ntt(w256, list(range(1, 257)), M128)
    assert out256 == [poly_eval(list(range(1, 257)), pow(w256, i, M128), M128) for i in range(256)]
    assert intt(w256, out256, M128) == list(range(1, 257))
    assert sha_le(out256, 16) == "8f7102fed15fa3c89551253cd93779f981c69431c46075b5507178c490920e19"
    nt["cases"].append(dict(field="M128", kind="ntt", root=S(w256), input=S(list(range(1, 257))), output=S(out256)))
    for name, p, rootf in (("Fr", FR, fr_root), ("M128", M128, m128_root)):
        for n in (1, 2, 4, 16, 64, 512, 2048):
            v = [rng.randrange(p) for _ in range(n)]
            r = rootf(n) if n > 1 else 1
            nt["cases"].append(dict(field=name, kind="ntt", root=S(r), input=S(v), output=S(ntt(r, v, p))))
            nt["cases"].append(dict(field=name, kind="intt", root=S(r), input=S(v), output=S(intt(r, v, p))))
    # coset LDE (fast_coset_evaluate) incl. the STARK parameters: offset = generator of order 2^119
    for name, p, rootf, off in (("M128", M128, m128_root, M128_GEN), ("Fr", FR, fr_root, 5)):
        for ncoef, order in ((1, 4), (5, 8), (16, 64), (100, 512), (256, 1024)):
            c = [rng.randrange(p) for _ in range(ncoef)]
            g = rootf(order)
            nt["cases"].append(dict(field=name, kind="coset", offset=S(off), generator=S(g), order=order,
                                    input=S(c), output=S(fast_coset_evaluate(c, off, g, order, p))))
    # polynomial products
    for name, p, rootf in (("Fr", FR, fr_root), ("M128", M128, m128_root)):
        for la, lb in ((1, 1), (3, 2), (5, 5), (9, 8), (40, 25), (100, 157)):
            a = [rng.randrange(p) for _ in range(la)]
            b = [rng.randrange(p) for _ in range(lb)]
            m = la + lb - 1
            n = 1
            while n < m:
                n <<= 1
            om = rootf(n) if n > 1 else 1
            nt["cases"].append(dict(field=name, kind="fft_multiply", omega=S(om), a=S(a), b=S(b), output=S(fft_multiply(a, b, om, p))))
            ro = 1 << 10
            nt["cases"].append(dict(field=name, kind="fast_multiply", root=S(rootf(ro)), root_order=ro, a=S(a), b=S(b),
                                    output=S(fast_multiply(a, b, rootf(ro), ro, p))))
    dump("ntt_vectors.json", nt)

    # ---------------- curve / MSM / KZG ----------------
    ec = {}
    # curve.rs:494-495 on y^2 = x^3 + 30x + 34 over F_631
    assert ec_mul((36, 60), 3, 30, 631) == (617, 5) and ec_mul((121, 387), 4, 30, 631) == (121, 244)
    ec["f631"] = {"a": 30, "cases": [dict(P=[36, 60], k=3, out=[617, 5]), dict(P=[121, 387], k=4, out=[121, 244])]}
    G2x = g1_mul(G1, 2)
    assert G2x == (1368015179489954701390400359078579693043519447331113978918064868415326638035,
                   9918110051302171585080402603319702774565515993150576347155970296011118125764)
    assert g1_mul(G1, FR) is INF  # bn128.rs:299-300
    assert g1_add(g1_mul(G1, 9), g1_mul(G1, 5)) == g1_add(g1_mul(G1, 12), g1_mul(G1, 2))  # bn128.rs:296-298
    ks = [1, 2, 3, 5, 9, 12, FR - 1, FR - 2, (FR - 1) // 2] + [rng.randrange(FR) for _ in range(8)]
    ec["g1_mul"] = [dict(k=S(k), out=S(pt(g1_mul(G1, k)))) for k in ks]
    pts = [g1_mul(G1, rng.randrange(1, FR)) for _ in range(24)]
    ec["g1_add"] = []
    pairs = [(pts[0], pts[1]), (pts[2], pts[2]), (pts[3], (pts[3][0], FQ - pts[3][1])), (INF, pts[4]), (pts[5], INF), (INF, INF)]
    for P, Q in pairs:
        ec["g1_add"].append(dict(P=S(pt(P)), Q=S(pt(Q)), out=S(pt(g1_add(P, Q)))))
    # MSM cases, incl. the edge batch of SURVEY 8d-3
    msm_cases = []
    def add_case(tag, scal, points):
        msm_cases.append(dict(tag=tag, scalars=S(scal), points=S([pt(P) for P in points]), out=S(pt(msm(scal, points)))))
    add_case("empty", [], [])
    add_case("single_zero_scalar", [0], [pts[0]])
    add_case("single_one", [1], [pts[0]])
    add_case("single_rm1", [FR - 1], [pts[0]])
    add_case("random8", [rng.randrange(FR) for _ in range(8)], pts[:8])
    add_case("random24", [rng.randrange(FR) for _ in range(24)], pts)
    add_case("all_equal_points_equal_scalars", [12345] * 6, [pts[1]] * 6)
    add_case("p_and_minus_p_adjacent", [77, 77, 5], [pts[2], (pts[2][0], FQ - pts[2][1]), pts[3]])
    add_case("cancels_to_infinity", [9, 9], [pts[4], (pts[4][0], FQ - pts[4][1])])
    add_case("with_infinity_points", [3, 4, 5], [pts[5], INF, pts[6]])
    add_case("repeated_scalars", [0xffff, 0xffff, 0x10000, 0xffff0000ffff] * 2, pts[:8])
    add_case("window_boundary_digits", [(1 << 15), (1 << 15) - 1, (1 << 15) + 1, (1 << 16) - 1, (1 << 253), FR - (1 << 15)], pts[8:14])
    add_case("unsanitized_scalar", [FR + 5, 2 * FR + 1], pts[:2])  # sanitize() first, polynomial.rs:162
    ec["msm"] = msm_cases
    # KZG: (x+1)(x+2)(x+3) = 6 + 11x + 6x^2 + x^3, kzg.rs:157-161, alpha = 7
    f = [6, 11, 6, 1]
    srs = kzg_setup(7, 3)
    com = msm(f, srs)
    assert com == g1_mul(G1, poly_eval(f, 7, FR))
    assert com == (20365992923428316285959523745808258186627862230815949234340139347993644046294,
                   3487465320930240990767468748026125910176748672030068852711792888709826106098)
    y, w = kzg_open(f, 5, srs)
    kz = [dict(alpha="7", coef=S(f), srs=S([pt(P) for P in srs]), commit=S(pt(com)), u="5", y=S(y), w=S(pt(w)))]
    alpha = rng.randrange(FR)
    f2 = [rng.randrange(FR) for _ in range(17)]
    srs2 = kzg_setup(alpha, 16)
    u2 = rng.randrange(FR)
    y2, w2 = kzg_open(f2, u2, srs2)
    # trapdoor identity: commit == [f(alpha)]G, w == [f_u(alpha)]G
    assert msm(f2, srs2) == g1_mul(G1, poly_eval(f2, alpha, FR))
    kz.append(dict(alpha=S(alpha), coef=S(f2), srs=S([pt(P) for P in srs2]), commit=S(pt(msm(f2, srs2))), u=S(u2), y=S(y2), w=S(pt(w2))))
    ec["kzg"] = kz
    # "next" rows: batch_open_kzg and prove_degree_bound on the second SRS
    us3 = [rng.randrange(FR) for _ in range(3)]
    ys3, w3 = kzg_batch_open(f2, us3, srs2)
    ys1, w1 = kzg_batch_open(f, [5, 9], srs)
    ec["kzg_batch_open"] = [dict(coef=S(f2), srs=S([pt(P) for P in srs2]), us=S(us3), ys=S(ys3), w=S(pt(w3))),
                            dict(coef=S(f), srs=S([pt(P) for P in srs]), us=S([5, 9]), ys=S(ys1), w=S(pt(w1)))]
    ec["kzg_degree_bound"] = [dict(coef=S(f2[:9]), srs=S([pt(P) for P in srs2]), d=8, out=S(pt(kzg_degree_bound(f2[:9], srs2, 8)))),
                              dict(coef=S(f2[:5]), srs=S([pt(P) for P in srs2]), d=10, out=S(pt(kzg_degree_bound(f2[:5], srs2, 10))))]
    dump("curve_vectors.json", ec)
    # FRI split-and-fold with the STARK parameters: offset = M128 generator
    fr_cases = []
    for name, p, rootf, off in (("M128", M128, m128_root, M128_GEN), ("Fr", FR, fr_root, 5)):
        for lg in (1, 3, 6, 9):
            n = 1 << lg
            cw = [rng.randrange(p) for _ in range(n)]
            al = rng.randrange(p)
            om = rootf(n)
            fr_cases.append(dict(field=name, alpha=S(al), offset=S(off), omega=S(om), input=S(cw), output=S(fri_fold(cw, al, off, om, p))))
    dump("fri_vectors.json", {"cases": fr_cases})

    next_rows()

    # ---------------- Rescue-Prime KAT (reference-held) ----------------
    rp_src = "/root/reference/myzkp/src/modules/zkstark/rescueprime.rs"
    rp_out = os.path.join(HERE, "rescue_prime_m128.json")
    if os.path.exists(rp_src):
        par = extract_rescue(rp_src)
        kats = [(1, 244180265933090377212304188905974087294),
                (57322816861100832358702415967512842988, 89633745865384635541695204788332415101)]  # rescueprime.rs:606-620
        for x, h in kats:
            assert rescue_hash(par, x) == h, "python transcription disagrees with the reference KAT"
        par = {k: (S(v) if not isinstance(v, int) or v > 1 << 31 else v) for k, v in par.items()}
        par["kats"] = [dict(input=S(x), hash=S(h)) for x, h in kats]
        dump("rescue_prime_m128.json", par)
    else:
        assert os.path.exists(rp_out), "rescue_prime_m128.json missing and /root/reference absent"
    print("golden vectors written to", HERE)

if __name__ == "__main__":
    main()

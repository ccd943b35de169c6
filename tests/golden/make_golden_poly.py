#!/usr/bin/env python3
"""Golden vectors for ntt::fast_zerofier / fast_evaluate / fast_interpolate (myzkp/src/modules/algebra/ntt.rs:118-252):
an independent pure-Python transcription of the three recursions on top of make_golden.py's fast_multiply, poly_mul and
poly_divrem (themselves transcriptions of ntt.rs:66-116 and polynomial.rs:302-316, :371-405).

    python3 tests/golden/make_golden_poly.py        -> tests/golden/poly_tree_vectors.json

The oracle (oracle/mzk_oracle.c) and the GPU library are both checked against this file; values are decimal strings."""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import (FR, M128, m128_root, fr_root, fast_multiply, poly_mul, poly_divrem, poly_eval, trim, S, dump)


def inv0(x, p):          # Field::inverse of the reference: extended Euclid, 0 -> 0 (field.rs:209-232)
    return pow(x, -1, p) if x % p else 0


def fast_zerofier(domain, root, order, p):      # ntt.rs:118-144
    assert pow(root, order, p) == 1 and pow(root, order // 2, p) != 1
    if not domain:
        return []
    if len(domain) == 1:
        return [(-domain[0]) % p, 1]
    half = len(domain) // 2
    return fast_multiply(fast_zerofier(domain[:half], root, order, p), fast_zerofier(domain[half:], root, order, p), root, order, p)


def fast_evaluate(poly, domain, root, order, p):      # ntt.rs:146-189
    assert pow(root, order, p) == 1 and pow(root, order // 2, p) != 1
    if not domain:
        return []
    if len(domain) == 1:
        return [poly_eval(poly, domain[0], p)]
    half = len(domain) // 2
    lz, rz = fast_zerofier(domain[:half], root, order, p), fast_zerofier(domain[half:], root, order, p)
    rem = lambda a, b: poly_divrem(a, b, p)[1] if trim(b) and len(trim(a)) >= len(trim(b)) else list(a)
    return fast_evaluate(rem(poly, lz), domain[:half], root, order, p) + fast_evaluate(rem(poly, rz), domain[half:], root, order, p)


def fast_interpolate(domain, values, root, order, p):      # ntt.rs:191-252
    assert pow(root, order, p) == 1 and pow(root, order // 2, p) != 1
    assert len(domain) == len(values)
    if not domain:
        return []
    if len(domain) == 1:
        return [values[0]]
    half = len(domain) // 2
    lz, rz = fast_zerofier(domain[:half], root, order, p), fast_zerofier(domain[half:], root, order, p)
    lo = fast_evaluate(rz, domain[:half], root, order, p)
    ro = fast_evaluate(lz, domain[half:], root, order, p)
    lt = [v * inv0(d, p) % p for v, d in zip(values[:half], lo)]
    rt = [v * inv0(d, p) % p for v, d in zip(values[half:], ro)]
    li = fast_interpolate(domain[:half], lt, root, order, p)
    ri = fast_interpolate(domain[half:], rt, root, order, p)
    a, b = poly_mul(trim(li), trim(rz), p), poly_mul(trim(ri), trim(lz), p)
    return trim([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % p for i in range(max(len(a), len(b)))])


def main():
    rnd = random.Random(0x504f4c59)
    cases = []
    for name, p, rootf in (("fr", FR, fr_root), ("m128", M128, m128_root)):
        for n in (0, 1, 2, 3, 7, 8, 9, 16, 33, 64, 65, 100, 127, 128, 129, 200):
            order = 1024
            root = rootf(order)
            dom = [rnd.randrange(p) for _ in range(n)]
            if n >= 3:
                dom[1] = 0                       # a zero point among the real ones (the device pads with zeros)
            vals = [rnd.randrange(p) for _ in range(n)]
            m = rnd.choice([0, 1, n, n + 5, 2 * n + 1]) if n else 3
            f = [rnd.randrange(p) for _ in range(m)]
            case = dict(field=name, n=n, root=root, root_order=order, domain=dom, values=vals, poly=f,
                        zerofier=fast_zerofier(dom, root, order, p), evaluate=fast_evaluate(f, dom, root, order, p),
                        interpolate=fast_interpolate(dom, vals, root, order, p))
            # structured domain as FastStark uses it: powers of omicron (fast_stark.rs:197-213), order 256 root
            cases.append(case)
        om = rootf(256)
        dom = [pow(om, i, p) for i in range(150)]
        vals = [rnd.randrange(p) for _ in range(150)]
        cases.append(dict(field=name, n=150, root=om, root_order=256, domain=dom, values=vals, poly=vals[:40],
                          zerofier=fast_zerofier(dom, om, 256, p), evaluate=fast_evaluate(vals[:40], dom, om, 256, p),
                          interpolate=fast_interpolate(dom, vals, om, 256, p)))
        # a repeated point: inverse(0) = 0 zeroes both copies' targets (ntt.rs:233-242)
        dom = [rnd.randrange(p) for _ in range(12)]
        dom[9] = dom[2]
        vals = [rnd.randrange(p) for _ in range(12)]
        cases.append(dict(field=name, n=12, root=rootf(64), root_order=64, domain=dom, values=vals, poly=[1, 2, 3],
                          zerofier=fast_zerofier(dom, rootf(64), 64, p), evaluate=fast_evaluate([1, 2, 3], dom, rootf(64), 64, p),
                          interpolate=fast_interpolate(dom, vals, rootf(64), 64, p)))
    dump("poly_tree_vectors.json", S_cases(cases))


def S_cases(cases):
    out = []
    for c in cases:
        out.append({k: (S(v) if k not in ("field", "n", "root_order") else v) for k, v in c.items()})
    return out


if __name__ == "__main__":
    main()

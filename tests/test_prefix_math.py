"""The two closed forms behind the subgroup-prefix routines of mzk_poly.hip (round 6), checked with Python integers on the CPU -- no GPU, no
library: (1) the interpolant through the first n points of a subgroup of order N is the inverse transform of the values extended by
u_j = sum_i v_i C[j][i], C[j][i] = - sum_t Ainv[j][t] x_i^(1+t), A[t][j] = x_(n+j)^(1+t); (2) the zerofier of those points has the
coefficients z_k = sum_j rho_j w_j^(k+1), w_j = 1 / x_(n+j), rho_j = 1 / prod_(l != j) (x_(n+j) - x_(n+l)).  References: ntt.rs:118-143 (fast_zerofier),
:203-252 (fast_interpolate) define the results; fast_stark.rs:53-57, :197-215 are the callers with such domains."""
import random
import pytest

P = 1 + 407 * (1 << 119)                      # M128 (fri.rs:408)
G = 85408008396924667383611388730472331217    # of order 2^119 (fri.rs:423-447)


def root(lg):
    return pow(G, 1 << (119 - lg), P)


def solve(mat, rhs):
    """Gauss-Jordan mod P; mat square"""
    m = len(mat)
    a = [row[:] + [r] for row, r in zip(mat, rhs)]
    for c in range(m):
        piv = next(r for r in range(c, m) if a[r][c] % P)
        a[c], a[piv] = a[piv], a[c]
        inv = pow(a[c][c], -1, P)
        a[c] = [x * inv % P for x in a[c]]
        for r in range(m):
            if r != c and a[r][c]:
                f = a[r][c]
                a[r] = [(x - f * y) % P for x, y in zip(a[r], a[c])]
    return [a[r][m] for r in range(m)]


def lagrange(xs, vs):
    """coefficients of the interpolant by the textbook formula (O(n^2) per basis polynomial: small n only)"""
    n = len(xs)
    out = [0] * n
    for i in range(n):
        num, den = [1], 1
        for j in range(n):
            if j == i:
                continue
            num = [(a - xs[j] * b) % P for a, b in zip([0] + num, num + [0])]
            den = den * (xs[i] - xs[j]) % P
        s = vs[i] * pow(den, -1, P) % P
        for k in range(n):
            out[k] = (out[k] + s * num[k]) % P
    return out


@pytest.mark.parametrize("lg,m", [(1, 0), (2, 1), (3, 3), (4, 0), (4, 5), (5, 3), (5, 9)])
def test_interpolant_of_a_subgroup_prefix_is_an_inverse_transform_of_extended_values(lg, m):
    N = 1 << lg
    n = N - m
    g = root(lg)
    xs = [pow(g, i, P) for i in range(N)]
    rng = random.Random(100 * lg + m)
    vs = [rng.randrange(P) for _ in range(n)]
    want = lagrange(xs[:n], vs)
    u = []
    if m:
        A = [[pow(xs[n + j], 1 + t, P) for j in range(m)] for t in range(m)]
        ainv_cols = [solve(A, [1 if r == t else 0 for r in range(m)]) for t in range(m)]        # column t of A^-1
        for j in range(m):
            C = [-sum(ainv_cols[t][j] * pow(xs[i], 1 + t, P) for t in range(m)) % P for i in range(n)]
            u.append(sum(v * c for v, c in zip(vs, C)) % P)
    ext = vs + u
    ninv = pow(N, -1, P)
    coef = [ninv * sum(ext[i] * pow(g, -i * k % N, P) for i in range(N)) % P for k in range(N)]
    assert coef[:n] == want and not any(coef[n:])


@pytest.mark.parametrize("lg,m", [(1, 1), (2, 1), (3, 3), (4, 1), (4, 6), (5, 4)])
def test_zerofier_of_a_subgroup_prefix_by_partial_fractions(lg, m):
    N = 1 << lg
    n = N - m
    g = root(lg)
    xs = [pow(g, i, P) for i in range(N)]
    want = [1]
    for x in xs[:n]:
        want = [(a - x * b) % P for a, b in zip([0] + want, want + [0])]
    miss = xs[n:]
    z = []
    for k in range(n + 1):
        acc = 0
        for j in range(m):
            den = 1
            for l in range(m):
                if l != j:
                    den = den * (miss[j] - miss[l]) % P
            acc += pow(den, -1, P) * pow(pow(miss[j], -1, P), k + 1, P)
        z.append(acc % P)
    assert z == want

"""prob_sample's CPU oracle (oracle_cumsum / oracle_prob_sample, restating sampling/tf_sampling_g.cu:7-103).

The reference has no CPU twin and no test for this op (parity unpinned by the reference), so the restatement is held
three ways: (1) a second, independent statement of the scan's ASSOCIATION -- the closed recursion
S[p] = T(aligned 2^u block ending at p) + S[p - 2^u], u = ctz(p + 1), T a balanced tree, evaluated in numpy float32 --
must give the same bits; (2) the search result must be the lower bound of q in the cumsum; (3) float64 agreement."""
import numpy as np
import pytest

from oracle import oracle as O

F = np.float32


def _tree(g, lo, hi):
    """balanced-tree total of g[lo:hi), hi - lo a power of two (the up-sweep, tf_sampling_g.cu:46-55)"""
    if hi - lo == 1:
        return g[lo]
    mid = (lo + hi) // 2
    return F(_tree(g, mid, hi) + _tree(g, lo, mid))


def _scan_totals(g):
    """inclusive scan of the group totals in the association the down-sweep leaves (:56-67)"""
    s = np.zeros(len(g), F)
    for p in range(len(g)):
        u = ((p + 1) & -(p + 1)).bit_length() - 1
        blk = _tree(g, p + 1 - (1 << u), p + 1)
        s[p] = blk if p + 1 == (1 << u) else F(blk + s[p - (1 << u)])
    return s


def cumsum_by_recursion(x):
    """one row, float32, written from the algebra of the kernel rather than from its loops"""
    x = np.asarray(x, F)
    n = len(x)
    out = np.zeros(n, F)
    run, comp = F(0), F(0)
    for j in range(0, n, 8192):
        c = x[j:j + 8192]
        ln = len(c)
        n2 = (ln + 3) // 4
        inner = np.zeros(n2 * 4, F)
        tot = np.zeros(n2, F)
        for g in range(n2):
            e = c[4 * g:4 * g + 4]
            if len(e) == 4:
                a = F(e[0] + e[1])
                inner[4 * g:4 * g + 4] = (e[0], a, F(e[2] + a), F(F(e[3] + e[2]) + a))
            else:
                v = F(0)
                for l in range(4):
                    if l < len(e):
                        v = F(v + e[l])
                    inner[4 * g + l] = v
            tot[g] = inner[4 * g + 3]
        s = _scan_totals(tot)
        for g in range(1, n2):
            inner[4 * g:4 * g + 4] = inner[4 * g:4 * g + 4] + s[g - 1]
        out[j:j + ln] = inner[:ln] + run
        t = F(s[n2 - 1] + comp)
        r2 = F(run + t)
        comp = F(t - F(r2 - run))
        run = r2
    return out


@pytest.mark.parametrize("n", [1, 2, 3, 4, 5, 6, 7, 8, 9, 12, 13, 31, 32, 33, 100, 255, 1000, 1021, 4096, 8191, 8192, 8193,
                               8200, 16384 + 5, 20000])
def test_cumsum_association(n):
    rng = np.random.default_rng(n)
    x = rng.random((2, n)).astype(F)
    x[1] *= rng.integers(0, 2, n).astype(F)             # runs of zero weight
    got = O.cumsum(x)
    for i in range(2):
        np.testing.assert_array_equal(got[i], cumsum_by_recursion(x[i]))
    ref = np.cumsum(x.astype(np.float64), 1)
    assert np.abs(got - ref).max() <= 4e-7 * max(ref.max(), 1.0)


@pytest.mark.parametrize("n,m", [(1, 5), (2, 9), (15, 64), (100, 1000), (1000, 300), (8193, 500), (20000, 200)])
def test_prob_sample_is_the_lower_bound(n, m):
    rng = np.random.default_rng(n * 31 + m)
    p = rng.random((3, n)).astype(F)
    p[1, : n // 2] = 0                                   # leading zero-probability run
    p[2, rng.integers(0, n, n // 3 + 1)] = 0
    r = rng.random((3, m)).astype(F)
    r[:, 0] = 0.0
    r[:, -1] = 1.0
    r[:, 1 % m] = np.nextafter(F(1), F(0))
    out, temp = O.prob_sample(p, r, return_temp=True)
    np.testing.assert_array_equal(temp, O.cumsum(p))
    q = (r * temp[:, -1:]).astype(F)
    for i in range(3):
        exp = np.array([int(np.argmax(temp[i] >= qq)) for qq in q[i]])
        np.testing.assert_array_equal(out[i], exp)
    assert (out[:, 0] == 0).all()                        # q = 0: every cumulative value qualifies, index 0
    assert out.min() >= 0 and out.max() <= n - 1


def test_prob_sample_exact_boundaries():
    """weights whose partial sums are exact in fp32: r on an interval boundary picks the LOWER interval (>=)"""
    p = np.array([[1, 1, 2, 4, 0, 0, 8]], F)             # cumsum 1 2 4 8 8 8 16
    r = np.array([[0, 1 / 16, 1 / 8, 0.126, 1 / 4, 1 / 2, 0.51, 1.0]], F)
    out, temp = O.prob_sample(p, r, return_temp=True)
    np.testing.assert_array_equal(temp, [[1, 2, 4, 8, 8, 8, 16]])
    np.testing.assert_array_equal(out, [[0, 0, 1, 2, 2, 3, 6, 6]])

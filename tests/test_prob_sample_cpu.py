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


# ---- round 6: the kernel's own organisation of the same tree (csrc/sampling.hip cumsum_kernel), restated in numpy float32:
# a lane owns eight consecutive quads (three tree levels in its registers), the lanes' top nodes climb six levels by wave
# shuffles and two over the four waves' tops, and a quad's prefix is the Fenwick query from the LARGEST block down.
def cumsum_lane_layout(x):
    x = np.asarray(x, F)
    n = len(x)
    out = np.zeros(n, F)
    carry, lost = F(0), F(0)
    for c0 in range(0, n, 8192):
        c = x[c0:c0 + 8192]
        ln = len(c)
        nq = (ln + 3) // 4
        part = np.zeros((256, 8, 4), F)
        nd = np.zeros((256, 8), F)
        for lane in range(256):
            for i in range(8):
                e0 = (lane * 8 + i) * 4
                if e0 + 3 < ln:
                    a, b, cc, d = c[e0:e0 + 4]
                    lo, hi = F(b + a), F(d + cc)
                    part[lane, i] = (a, lo, F(cc + lo), F(hi + lo))
                    nd[lane, i] = part[lane, i, 3]
                elif e0 < ln:
                    run = F(0)
                    for l in range(4):
                        if e0 + l < ln:
                            run = F(run + c[e0 + l])
                        part[lane, i, l] = run
                    nd[lane, i] = run
        have = nq - 8 * np.arange(256)
        for a_, b_, need in ((1, 0, 1), (3, 2, 3), (5, 4, 5), (7, 6, 7), (3, 1, 3), (7, 5, 7), (7, 3, 7)):
            m = have > need
            nd[m, a_] = (nd[m, a_] + nd[m, b_]).astype(F)
        top = nd[:, 7].copy()
        whole = have > 7
        for lv in range(6):                                   # inside a wave of 64 lanes
            below = np.empty_like(top)
            for lane in range(256):
                src = lane - (1 << lv)
                below[lane] = top[src] if (lane % 64) >= (1 << lv) else top[lane]
            m = (((np.arange(256) % 64) + 1) & ((2 << lv) - 1)) == 0
            m &= whole
            top[m] = (top[m] + below[m]).astype(F)
        w = [top[63], top[127], top[191], top[255]]
        if whole[127]:
            top[127] = F(w[1] + w[0])
        if whole[255]:
            top[255] = F(F(w[3] + w[2]) + F(w[1] + w[0]))

        def tops_before(lane_ix):
            acc, any_ = F(0), False
            for bit in range(8, -1, -1):
                if (lane_ix >> bit) & 1:
                    at = (lane_ix & ~((1 << bit) - 1)) - 1
                    acc = F(top[at] + acc) if any_ else top[at]
                    any_ = True
            return acc, any_
        total = None
        for lane in range(256):
            if lane * 8 >= nq:
                break
            before, any_ = tops_before(lane)
            pre = np.zeros(8, F)
            pre[0] = F(nd[lane, 0] + before) if any_ else nd[lane, 0]
            pre[1] = F(nd[lane, 1] + before) if any_ else nd[lane, 1]
            pre[2] = F(nd[lane, 2] + pre[1])
            pre[3] = F(nd[lane, 3] + before) if any_ else nd[lane, 3]
            pre[4] = F(nd[lane, 4] + pre[3])
            pre[5] = F(nd[lane, 5] + pre[3])
            pre[6] = F(nd[lane, 6] + pre[5])
            if lane == (nq - 1) // 8:
                li = (nq - 1) % 8
                total = tops_before(lane + 1)[0] if li == 7 else pre[li]
            for i in range(8):
                q = lane * 8 + i
                e0 = q * 4
                if e0 >= ln:
                    continue
                add = before if i == 0 else pre[i - 1]
                for l in range(4):
                    if e0 + l < ln:
                        v = part[lane, i, l] if q == 0 else F(part[lane, i, l] + add)
                        out[c0 + e0 + l] = F(v + carry)
        inc = F(total + lost)
        grown = F(carry + inc)
        lost = F(inc - F(grown - carry))
        carry = grown
    return out


@pytest.mark.parametrize("n", [1, 3, 4, 5, 31, 32, 33, 36, 255, 1000, 2047, 2048, 4100, 8191, 8192, 8193, 8200, 16384 + 29, 20000])
def test_cumsum_kernel_layout_has_the_contract_association(n):
    rng = np.random.default_rng(1000 + n)
    x = rng.random((2, n)).astype(F)
    x[1] *= rng.integers(0, 2, n).astype(F)
    want = O.cumsum(x)
    for i in range(2):
        np.testing.assert_array_equal(cumsum_lane_layout(x[i]), want[i])

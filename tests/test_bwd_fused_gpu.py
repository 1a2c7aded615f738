"""`pcops_mlp_bwd_fused` (include/pcops.h): the data and the weight gradient of a 64-wide layer in one pass -- against
float64 on the same inputs, and against the two-kernel path it replaces (pcops_mlp_gemm_dgrad + pcops_mlp_wgrad), for
the materialised-gradient form and the pooled forms (one group per 32-row stripe; any group size), with a ragged tail."""
import pytest
import torch

from scanobjectnn_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

CASES = [  # (M, K, N, S)   S = 0: dense upstream gradient
    (65536 + 32 * 5 + 13, 64, 128, 0),
    (65536 + 77, 64, 64, 0),
    (32 * 2100, 64, 128, 32),          # SA1's top layer: one pooling group per stripe
    (64 * 1030, 64, 64, 64),
    (20 * 3300, 64, 128, 20),          # any group size
    (16 * 4200, 48, 96, 16),           # narrower than the tile in both directions
    (65536 + 32 * 3 + 5, 64, 96, 0),   # MSG's 64 -> 96 layer: the zero fourth column block is skipped
    (128 * 520, 64, 80, 128),          # ... a ragged third block
]


def _vec(n, g, lo=0.5):
    return ((lo + torch.rand(n, generator=g)) * (1.0 - 2.0 * (torch.arange(n) % 3 == 1))).to(DEV)


@pytest.mark.parametrize("opt", [1, 2])   # the dX half on split operands / the dW half too (128-column tile; the default)
@pytest.mark.parametrize("M,K,N,S", CASES)
def test_bwd_fused_against_float64_and_the_two_kernel_path(M, K, N, S, opt):
    lib = _lib.load()
    prev = _lib.set_option(_lib.OPT_BWD_FUSED_DX_SPLIT_BF16, opt)
    try:
        _bwd_fused_case(lib, M, K, N, S, opt)
    finally:
        _lib.set_option(_lib.OPT_BWD_FUSED_DX_SPLIT_BF16, prev)


def _bwd_fused_case(lib, M, K, N, S, opt):
    groups = lib.pcops_mlp_bwd_fused_groups(M, K, N, S, 1 if S else 0)
    assert groups > 0
    g = torch.Generator().manual_seed(M + N)
    Yprev = torch.randn(M, K, generator=g).to(DEV)
    Y = torch.randn(M, N, generator=g).to(DEV)
    W = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV)
    sc, sh = _vec(K, g), (0.3 * torch.randn(K, generator=g)).to(DEV)
    p, q, t = _vec(N, g), 0.1 * _vec(N, g), (0.05 * torch.randn(N, generator=g)).to(DEV)
    if S:
        G = None
        gpool = torch.randn(M // S, N, generator=g).to(DEV)
        gpool[torch.rand(M // S, N, generator=g).to(DEV) < 0.3] = 0.0            # masked entries (ReLU off at the max)
        argmax = torch.randint(0, S, (M // S, N), generator=g, dtype=torch.int32).to(torch.uint8).to(DEV)
        Gfull = torch.zeros(M // S, S, N, dtype=torch.float64, device=DEV)
        Gfull.scatter_(1, argmax.long().unsqueeze(1), gpool.double().unsqueeze(1))
        Gfull = Gfull.view(M, N)
    else:
        G = torch.randn(M, N, generator=g).to(DEV)
        gpool = argmax = None
        Gfull = G.double()
    # ---- float64 truth
    dY = p.double() * Gfull + q.double() * Y.double() + t.double()
    pre = Yprev.double() * sc.double() + sh.double()
    X = pre.clamp_min(0.0)
    want_dW, want_db = X.t() @ dY, dY.sum(0)
    safe = pre.abs() > 1e-5                                                        # (a sign the fp32 fma may round the other way)
    want_G = (dY @ W.double().t()) * (pre > 0)
    want_s1, want_s2 = want_G.sum(0), (want_G * Yprev.double()).sum(0)

    def ptr(x):
        return None if x is None else x.data_ptr()

    part = torch.empty(groups * (K * N + N), device=DEV)
    dW, db = torch.empty(K, N, device=DEV), torch.empty(N, device=DEV)
    Gprev = torch.full((M, K), float("nan"), device=DEV)
    stats = torch.empty(groups, 2, K, device=DEV)
    _lib.call("pcops_mlp_bwd_fused", M, K, N, Yprev.data_ptr(), sc.data_ptr(), sh.data_ptr(), ptr(G), Y.data_ptr(),
              p.data_ptr(), q.data_ptr(), t.data_ptr(), ptr(gpool), ptr(argmax), S if S else 1, W.data_ptr(),
              part.data_ptr(), dW.data_ptr(), db.data_ptr(), Gprev.data_ptr(), stats.data_ptr())
    # the library's own word on the pipe: 1 = both halves on split operands (only where that variant is built: more than
    # 96 output columns), 2 = the dW half on the fp32 pipe
    assert lib.pcops_last_launch_pipe() == (1 if (opt == 2 and N > 96) else 2)
    torch.cuda.synchronize()
    assert not torch.isnan(Gprev).any()

    def rel(a, b):
        return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    assert rel(dW, want_dW) <= 2e-5 and rel(db, want_db) <= 2e-5
    assert rel(torch.where(safe, Gprev.double(), want_G), want_G) <= 1e-5
    assert rel(stats[:, 0].double().sum(0), want_s1) <= 1e-4 and rel(stats[:, 1].double().sum(0), want_s2) <= 1e-4

    # ---- the two kernels it replaces, same inputs: same numbers up to summation order
    Wt = W.t().contiguous()
    dummy = p.data_ptr() if S else None        # (pool_scale / pool_shift of the older entry points: required, not read)
    P = lib.pcops_mlp_stats_rows(M)
    part2 = torch.empty(P, 2, K, device=DEV)
    Gprev2 = torch.empty(M, K, device=DEV)
    _lib.call("pcops_mlp_gemm_dgrad", M, N, K, ptr(G), Y.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), ptr(gpool),
              ptr(argmax), S if S else 1, dummy, dummy, Wt.data_ptr(), Yprev.data_ptr(), sc.data_ptr(), sh.data_ptr(),
              Gprev2.data_ptr(), part2.data_ptr())
    splits = lib.pcops_mlp_wgrad_splits(M, K, N)
    scratch = torch.empty(splits * (K * N + N), device=DEV)
    dW2, db2 = torch.empty(K, N, device=DEV), torch.empty(N, device=DEV)
    _lib.call("pcops_mlp_wgrad", M, K, N, Yprev.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), ptr(G), Y.data_ptr(),
              p.data_ptr(), q.data_ptr(), t.data_ptr(), ptr(gpool), ptr(argmax), S if S else 1, dummy, dummy,
              scratch.data_ptr(), dW2.data_ptr(), db2.data_ptr())
    torch.cuda.synchronize()
    assert torch.equal(Gprev != 0, Gprev2 != 0)                   # the same mask, element for element (same fmaf)
    assert rel(Gprev, Gprev2.double()) <= 1e-5 and rel(dW, dW2.double()) <= 2e-5 and rel(db, db2.double()) <= 2e-5
    assert rel(stats[:, 1].double().sum(0), part2[:, 1].double().sum(0)) <= 1e-4


GW_CASES = [  # (M, K, N, S, bias)  pooled layers only; Y must be the layer's own forward output
    (32 * 2100, 64, 128, 32, True),     # SA1's top layer: one pooling group per stripe
    (64 * 1030 + 64, 64, 64, 64, True),
    (20 * 3300, 64, 128, 20, True),     # the T-Net's k = 20: up to three groups per stripe
    (16 * 4200, 48, 96, 16, False),     # narrower than the tile in both directions, no bias, zero fourth block skipped
    (12 * 5500 + 12, 64, 128, 12, True),  # four groups per stripe, a ragged last stripe
]


@pytest.mark.parametrize("M,K,N,S,has_bias", GW_CASES)
def test_bwd_fused_gram_form_weight_gradient(M, K, N, S, has_bias):
    """pcops_mlp_bwd_fused_gw (round 6): dW = X^T (p.G) + (X^T X) W diag(q) + (X^T 1)(q.b + t)^T with Y = X W + b -- against
    float64 of the DIRECT form on the same tensors, and against pcops_mlp_bwd_fused (same Gprev bit for bit: the data
    gradient half is the same code)."""
    lib = _lib.load()
    prev = _lib.set_option(_lib.OPT_BWD_FUSED_GRAM_WGRAD, 1)
    try:
        groups = lib.pcops_mlp_bwd_fused_gw_groups(M, K, N, S)
        assert groups > 0 and groups == lib.pcops_mlp_bwd_fused_groups(M, K, N, S, 1)
        g = torch.Generator().manual_seed(M + N + S)
        Yprev = torch.randn(M, K, generator=g).to(DEV)
        W = (torch.randn(K, N, generator=g) / K ** 0.5).to(DEV)
        b = (0.5 * torch.randn(N, generator=g)).to(DEV) if has_bias else None
        sc, sh = _vec(K, g), (0.3 * torch.randn(K, generator=g)).to(DEV)
        p, q, t = _vec(N, g), 0.1 * _vec(N, g), (0.05 * torch.randn(N, generator=g)).to(DEV)
        pre = Yprev.double() * sc.double() + sh.double()
        X = pre.clamp_min(0.0)
        Y = (torch.relu(torch.addcmul(sh, Yprev, sc)) @ W + (b if has_bias else 0.0)).contiguous()   # fp32, as a forward stores it
        gpool = torch.randn(M // S, N, generator=g).to(DEV)
        gpool[torch.rand(M // S, N, generator=g).to(DEV) < 0.3] = 0.0
        argmax = torch.randint(0, S, (M // S, N), generator=g, dtype=torch.int32).to(torch.uint8).to(DEV)
        Gfull = torch.zeros(M // S, S, N, dtype=torch.float64, device=DEV)
        Gfull.scatter_(1, argmax.long().unsqueeze(1), gpool.double().unsqueeze(1))
        dY = p.double() * Gfull.view(M, N) + q.double() * Y.double() + t.double()
        want_dW, want_db = X.t() @ dY, dY.sum(0)

        def run(name, extra):
            part = torch.empty(groups * (K * N + N + K * K + K), device=DEV)
            dW, db = torch.empty(K, N, device=DEV), torch.empty(N, device=DEV)
            Gprev = torch.full((M, K), float("nan"), device=DEV)
            stats = torch.empty(groups, 2, K, device=DEV)
            args = [M, K, N, Yprev.data_ptr(), sc.data_ptr(), sh.data_ptr()] + ([] if extra else [None]) + \
                   [Y.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), gpool.data_ptr(), argmax.data_ptr(), S, W.data_ptr()] + \
                   ([b.data_ptr() if has_bias else None] if extra else []) + \
                   [part.data_ptr(), dW.data_ptr(), db.data_ptr(), Gprev.data_ptr(), stats.data_ptr()]
            _lib.call(name, *args)
            torch.cuda.synchronize()
            return dW, db, Gprev, stats

        dW, db, Gprev, stats = run("pcops_mlp_bwd_fused_gw", True)
        dW0, db0, Gprev0, stats0 = run("pcops_mlp_bwd_fused", False)

        def rel(a, b_):
            return ((a.double() - b_.double()).abs().max() / b_.double().abs().max().clamp_min(1e-30)).item()

        assert rel(dW, want_dW) <= 2e-5 and rel(db, want_db) <= 2e-5
        # (db: the same column sums; in the same order only when both kernels hand rows to lanes alike)
        assert rel(dW, dW0) <= 2e-5 and rel(db, db0) <= 2e-6
        assert torch.equal(Gprev, Gprev0) and torch.equal(stats, stats0)
    finally:
        _lib.set_option(_lib.OPT_BWD_FUSED_GRAM_WGRAD, prev)
    assert lib.pcops_mlp_bwd_fused_gw_groups(M, K, N, S) == (groups if prev else 0)      # the option is read per call


def test_bwd_fused_says_what_it_takes():
    lib = _lib.load()
    assert lib.pcops_mlp_bwd_fused_groups(1 << 22, 64, 128, 32, 1) == 256
    assert lib.pcops_mlp_bwd_fused_groups(1 << 22, 128, 128, 32, 1) == 0      # matrix-pipe bound shapes: two kernels
    assert lib.pcops_mlp_bwd_fused_groups(1 << 22, 64, 256, 0, 0) == 0
    assert lib.pcops_mlp_bwd_fused_groups(1024, 64, 64, 0, 0) == 0


@pytest.mark.parametrize("M,K,ld,bnrelu", [(65536 + 45, 320, 320, False), (70000, 128, 128, True), (66000, 100, 104, True),
                                            (65536, 64, 64, False), (80000, 36, 36, True)])
def test_single_pass_gram_against_float64(M, K, ld, bnrelu):
    """`pcops_mlp_gram` for widths up to 320 on >= 65 536 rows: whole rows staged once, every upper 32 x 32 block in
    accumulators, lower triangle mirrored -- X^T X and X^T 1 against float64 (plain input and relu(bn(.)) input, a row
    stride wider than K, ragged row counts, widths that are not multiples of 32)"""
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    Xs = torch.randn(M, ld, generator=g).to(DEV)
    sc = ((0.5 + torch.rand(K, generator=g)) * (1.0 - 2.0 * (torch.arange(K) % 3 == 1))).to(DEV)
    sh = (0.3 * torch.randn(K, generator=g)).to(DEV)
    X = Xs[:, :K].double()
    if bnrelu:
        X = (X * sc.double() + sh.double()).clamp_min(0.0)
    want, wsum = X.t() @ X, X.sum(0)
    splits = lib.pcops_mlp_wgrad_splits(M, K, K)
    scratch = torch.empty(splits * (K * K + K), device=DEV)
    gram = torch.full((K, K), float("nan"), device=DEV)
    xsum = torch.full((K,), float("nan"), device=DEV)
    _lib.call("pcops_mlp_gram", M, K, Xs.data_ptr(), ld, sc.data_ptr() if bnrelu else None, sh.data_ptr() if bnrelu else None,
              scratch.data_ptr(), gram.data_ptr(), xsum.data_ptr())
    torch.cuda.synchronize()
    assert not torch.isnan(gram).any() and not torch.isnan(xsum).any()
    assert torch.equal(gram, gram.t())                                              # mirrored, bit for bit
    scale = want.abs().max().item()
    assert (gram.double() - want).abs().max().item() <= 2e-5 * scale
    assert (xsum.double() - wsum).abs().max().item() <= 2e-5 * max(wsum.abs().max().item(), M ** 0.5)

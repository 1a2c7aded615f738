"""Round 6: `pcops_mlp_gemm_dgrad` of the matrix-bound layers (128 .. 256 dY columns) on the bf16 matrix pipe with split operands,
in 64-column passes with LDS-resident weight pieces (one pass of three blocks for 65 .. 96 output columns) -- against float64 and
against the fp32-pipe kernel it replaces, selected per call with pcops_set_option(PCOPS_OPT_DGRAD_SPLIT_BF16)."""
import pytest
import torch

from scanobjectnn_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _vec(n, g, lo=0.5):
    return ((lo + torch.rand(n, generator=g)) * (1.0 - 2.0 * (torch.arange(n) % 3 == 1))).to(DEV)


@pytest.mark.parametrize("M,K,Nout", [(65536 + 77, 256, 128), (70000, 128, 128), (66000 + 31, 128, 96), (65536, 192, 64)])
def test_split_operand_data_gradient(M, K, Nout):
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K)
    G = torch.randn(M, K, generator=g).to(DEV)
    Y = torch.randn(M, K, generator=g).to(DEV)
    Yprev = torch.randn(M, Nout, generator=g).to(DEV)
    Wt = (torch.randn(K, Nout, generator=g) / K ** 0.5).to(DEV)
    p, q, t = _vec(K, g), 0.1 * _vec(K, g), (0.05 * torch.randn(K, generator=g)).to(DEV)
    sc, sh = _vec(Nout, g), (0.3 * torch.randn(Nout, generator=g)).to(DEV)
    pre = Yprev.double() * sc.double() + sh.double()
    dY = p.double() * G.double() + q.double() * Y.double() + t.double()
    want = (dY @ Wt.double()) * (pre > 0)
    safe = pre.abs() > 1e-5
    P = lib.pcops_mlp_stats_rows(M)

    def run(mode):
        prev = _lib.set_option(_lib.OPT_DGRAD_SPLIT_BF16, mode)
        try:
            out = torch.full((M, Nout), float("nan"), device=DEV)
            part = torch.empty(P, 2, Nout, device=DEV)
            _lib.call("pcops_mlp_gemm_dgrad", M, K, Nout, G.data_ptr(), Y.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(),
                      None, None, 1, None, None, Wt.data_ptr(), Yprev.data_ptr(), sc.data_ptr(), sh.data_ptr(), out.data_ptr(),
                      part.data_ptr())
            pipe = int(lib.pcops_last_launch_pipe())
            torch.cuda.synchronize()
            return out, part, pipe
        finally:
            _lib.set_option(_lib.OPT_DGRAD_SPLIT_BF16, prev)

    o1, s1, pipe1 = run(1)
    o0, s0, pipe0 = run(0)
    assert (pipe1, pipe0) == (1, 0)                                   # the option is read per call and selects the kernel
    assert not torch.isnan(o1).any() and torch.equal(o1 != 0, o0 != 0)   # the same mask, element for element

    def rel(a):
        return ((torch.where(safe, a.double(), want) - want).abs().max() / want.abs().max()).item()

    e1, e0 = rel(o1), rel(o0)
    assert e1 <= 2e-6 and e1 <= 1.5 * e0 + 1e-7, (e1, e0)             # at least the fp32 chain's accuracy (measured: a third)
    for which in (0, 1):
        a, b = s1[:, which].double().sum(0), s0[:, which].double().sum(0)
        assert ((a - b).abs().max() / b.abs().max()).item() <= 1e-5


TOP_CASES = [  # (M, Kp, masked)   the algebraic top-layer data gradient: Gprev = mask . (X Mq + addend[rowmap] + vconst)
    (40000 + 17, 320, False),     # DGCNN's aggregation layer: the stack's raw input, weights streamed (Kp > 256)
    (33000, 128, False),          # resident weight pieces, one 128-column pass
    (36000 + 5, 512, True),       # SA3's form: masked by the layer below, 64-column passes, streamed
    (34000, 128, True),
]


@pytest.mark.parametrize("M,Kp,masked", TOP_CASES)
def test_dgrad_top_on_split_operands(M, Kp, masked):
    """pcops_mlp_gemm_dgrad_top on the bf16 pipe with split operands (round 6; ws_plan kinds 4 / 5): against float64 on the same
    tensors, the library's own word on the pipe, and -- masked form -- the same mask as float64 away from the fma's rounding edge."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + Kp)
    Yprev = torch.randn(M, Kp, generator=g).to(DEV)
    Mq = (torch.randn(Kp, Kp, generator=g) / Kp ** 0.5).to(DEV)
    vconst = (0.1 * torch.randn(Kp, generator=g)).to(DEV)
    nadd = 777
    addend = torch.randn(nadd, Kp, generator=g).to(DEV)
    rowmap = torch.full((M,), -1, dtype=torch.int32)
    picks = torch.randperm(M, generator=g)[:nadd]
    rowmap[picks] = torch.arange(nadd, dtype=torch.int32)
    rowmap = rowmap.to(DEV)
    sc = ((0.5 + torch.rand(Kp, generator=g)) * (1.0 - 2.0 * (torch.arange(Kp) % 3 == 1))).to(DEV) if masked else None
    sh = (0.3 * torch.randn(Kp, generator=g)).to(DEV) if masked else None
    Gprev = torch.full((M, Kp), float("nan"), device=DEV)
    P = lib.pcops_mlp_stats_rows(M)
    stats = torch.zeros(P, 2, Kp, device=DEV)
    _lib.call("pcops_mlp_gemm_dgrad_top", M, Kp, Yprev.data_ptr(), None if sc is None else sc.data_ptr(),
              None if sh is None else sh.data_ptr(), Mq.data_ptr(), vconst.data_ptr(), addend.data_ptr(), nadd,
              rowmap.data_ptr(), Gprev.data_ptr(), stats.data_ptr() if masked else None)
    assert lib.pcops_last_launch_pipe() == 1
    torch.cuda.synchronize()
    assert not torch.isnan(Gprev).any()
    if masked:
        pre = Yprev.double() * sc.double() + sh.double()
        X = pre.clamp_min(0.0)
    else:
        X = Yprev.double()
    want = X @ Mq.double() + vconst.double()
    want[picks.to(DEV)] += addend.double()
    if masked:
        safe = pre.abs() > 1e-5
        want = want * (pre > 0)
        got = torch.where(safe, Gprev.double(), want)
    else:
        got = Gprev.double()
    err = ((got - want).abs().max() / want.abs().max()).item()
    assert err <= 1e-5, err


@pytest.mark.parametrize("M,K,N,S", [(65536 + 77, 96, 128, 0), (64 * 1100, 96, 128, 64), (20 * 3500 + 20, 80, 128, 20),
                                     (70000, 96, 256, 0), (32 * 2100, 72, 96, 32)])
def test_wgrad_with_65_to_96_input_channels(M, K, N, S):
    """pcops_mlp_wgrad with 65 .. 96 input channels (MSG's 96 -> 128): the 96 x 32 consumer layout of wgrad_bf3_kernel (round 6)
    against float64 -- dense and pooled upstream gradients, a ragged tail, fewer than 96 channels, two column blocks."""
    lib = _lib.load()
    g = torch.Generator().manual_seed(M + K + N)
    X = torch.randn(M, K, generator=g).to(DEV)
    Y = torch.randn(M, N, generator=g).to(DEV)
    sc = ((0.5 + torch.rand(K, generator=g)) * (1.0 - 2.0 * (torch.arange(K) % 3 == 1))).to(DEV)
    sh = (0.3 * torch.randn(K, generator=g)).to(DEV)
    p = (0.5 + torch.rand(N, generator=g)).to(DEV)
    q = (0.1 * torch.randn(N, generator=g)).to(DEV)
    t = (0.05 * torch.randn(N, generator=g)).to(DEV)
    if S:
        G = None
        gpool = torch.randn(M // S, N, generator=g).to(DEV)
        argmax = torch.randint(0, S, (M // S, N), generator=g, dtype=torch.int32).to(torch.uint8).to(DEV)
        Gfull = torch.zeros(M // S, S, N, dtype=torch.float64, device=DEV)
        Gfull.scatter_(1, argmax.long().unsqueeze(1), gpool.double().unsqueeze(1))
        Gfull = Gfull.view(M, N)
    else:
        G = torch.randn(M, N, generator=g).to(DEV)
        gpool = argmax = None
        Gfull = G.double()
    dY = p.double() * Gfull + q.double() * Y.double() + t.double()
    Xa = (X.double() * sc.double() + sh.double()).clamp_min(0.0)
    want_dW, want_db = Xa.t() @ dY, dY.sum(0)

    def ptr(x):
        return None if x is None else x.data_ptr()

    splits = lib.pcops_mlp_wgrad_splits(M, K, N)
    scratch = torch.empty(splits * (K * N + N), device=DEV)
    dW, db = torch.full((K, N), float("nan"), device=DEV), torch.empty(N, device=DEV)
    dummy = p.data_ptr() if S else None
    _lib.call("pcops_mlp_wgrad", M, K, N, X.data_ptr(), K, sc.data_ptr(), sh.data_ptr(), ptr(G), Y.data_ptr(),
              p.data_ptr(), q.data_ptr(), t.data_ptr(), ptr(gpool), ptr(argmax), S if S else 1, dummy, dummy,
              scratch.data_ptr(), dW.data_ptr(), db.data_ptr())
    assert lib.pcops_last_launch_pipe() == 1
    torch.cuda.synchronize()

    def rel(a, b):
        return ((a.double() - b).abs().max() / b.abs().max().clamp_min(1e-30)).item()

    assert rel(dW, want_dW) <= 2e-5 and rel(db, want_db) <= 2e-5

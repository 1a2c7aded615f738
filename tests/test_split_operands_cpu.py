"""The algebra behind the split-operand kernels (csrc/mlp.hip `split3`), checked on the CPU with
torch's bfloat16 (round-to-nearest-even, what v_cvt_pk_bf16_f32 does): an fp32 value is the sum of three bf16 pieces up to
2^-25 of itself with EXACT fp32 residuals, the product of two pieces is exact in fp32, and the six products the kernels
keep differ from the full product by at most ~2^-24 of it."""
import torch


def split3(x):
    h = x.bfloat16().float()
    r1 = x - h
    m = r1.bfloat16().float()
    r2 = r1 - m
    l = r2.bfloat16().float()
    return h, m, l, r1, r2


def _values(n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, generator=g) * torch.exp(8.0 * torch.randn(n, generator=g))       # many binades
    edge = torch.tensor([0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 255.99998, 3.0e38, 1.2e-38, -7.7e-30,
                         2.0 ** -126, 1.17549435e-38 * 3, 0.1, 1.0 / 3.0, 65504.0, 16777215.0])
    return torch.cat([x, edge])


def test_three_bf16_pieces_carry_an_fp32_value():
    x = _values(200000, 1)
    h, m, l, r1, r2 = split3(x)
    xd = x.double()
    # the residuals are exact fp32 subtractions (h is x rounded to 8 bits: x - h has at most 16 significant bits, ...)
    assert torch.equal(r1.double(), xd - h.double())
    assert torch.equal(r2.double(), xd - h.double() - m.double())
    err = (xd - (h.double() + m.double() + l.double())).abs()
    big = xd.abs() >= 2.0 ** -100
    assert bool((err[big] <= 2.0 ** -25 * xd.abs()[big]).all()), (err / xd.abs().clamp_min(1e-300))[big].max()
    # within 2^17 of the smallest normal number the residuals are denormal and lose bits: the error stays below the smallest
    # normal bf16 step there (absolute 2^-133), which no sum of products can notice
    assert bool((err[~big] <= 2.0 ** -133).all())
    # piece magnitudes: each a factor 2^-8 below the one before (round-to-nearest leaves at most half a unit of the 8th bit)
    assert bool((m.abs() <= 2.0 ** -8 * x.abs() * (1 + 2.0 ** -7)).all())
    assert bool((l.abs() <= 2.0 ** -16 * x.abs() * (1 + 2.0 ** -6)).all())


def test_products_of_pieces_are_exact_and_six_of_nine_suffice():
    a = _values(100000, 2)[:100000].clamp(-1e15, 1e15)
    b = _values(100000, 3)[:100000].clamp(-1e15, 1e15)
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    for p, q in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)):
        assert torch.equal((p * q).double(), p.double() * q.double())           # 8 x 8 significant bits: exact in fp32
    six = sum(p.double() * q.double() for p, q in ((ah, bh), (ah, bm), (am, bh), (am, bm), (ah, bl), (al, bh)))
    full = a.double() * b.double()
    ok = full.abs() > 1e-30                                                      # (denormal products aside)
    rel = ((six - full).abs() / full.abs().clamp_min(1e-300))[ok]
    assert rel.max().item() <= 2.0 ** -23, rel.max().item()                     # dropped: m.l + l.m + l.l + the split's own 2^-25 s
    assert rel.mean().item() <= 2.0 ** -26


def test_two_accumulators_beat_one_on_a_long_dot_product():
    """the accumulation order of the kernels, emulated in fp32: h.h products in one accumulator, the five small ones in
    another, against all six in one -- and against a plain fp32 fmaf chain (each step rounded to fp32)"""
    g = torch.Generator().manual_seed(4)
    K, R = 256, 4096
    a = torch.relu(torch.randn(R, K, generator=g))
    w = torch.randn(K, generator=g) / K ** 0.5
    ah, am, al, _, _ = split3(a)
    wh, wm, wl, _, _ = split3(w)
    ref = (a.double() * w.double()).sum(1)
    scale = ref.pow(2).mean().sqrt()

    def chain16(terms):          # one fp32 rounding per 16 k, as one matrix instruction per accumulator makes
        acc = torch.zeros(R)
        for k0 in range(0, K, 16):
            acc = (acc.double() + sum(t[:, k0:k0 + 16].double().sum(1) for t in terms)).float()
        return acc

    small = [al * wh, ah * wl, am * wm, am * wh, ah * wm]
    one = torch.zeros(R)
    for k0 in range(0, K, 16):
        for t in small + [ah * wh]:
            one = (one.double() + t[:, k0:k0 + 16].double().sum(1)).float()
    two = chain16([ah * wh]) + chain16(small)
    plain = torch.zeros(R)
    for k in range(K):
        plain = torch.addcmul(plain, a[:, k], w[k].expand(R))
    e_one = ((one.double() - ref).pow(2).mean().sqrt() / scale).item()
    e_two = ((two.double() - ref).pow(2).mean().sqrt() / scale).item()
    e_plain = ((plain.double() - ref).pow(2).mean().sqrt() / scale).item()
    assert e_two < 0.6 * e_one and e_two < 0.5 * e_plain, (e_two, e_one, e_plain)

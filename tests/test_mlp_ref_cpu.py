"""tests/mlp_ref.chunked_gather_stack -- the hand-written, chunked float64 backward the bench-size GPU tests use where
an autograd graph of the float64 reference does not fit (10.5 M / 16.7 M rows) -- against torch autograd of the same
stack (mlp_ref.run_stack) at sizes where both run, on the CPU."""
import pytest
import torch

import mlp_ref as MR


def _layers(k0, widths, seed):
    g = torch.Generator().manual_seed(seed)
    layers, cin = [], k0
    for w in widths:
        layers.append([torch.randn(cin, w, generator=g, dtype=torch.float64) / cin ** 0.5,
                       0.1 * torch.randn(w, generator=g, dtype=torch.float64),
                       (0.5 + torch.rand(w, generator=g, dtype=torch.float64)) * (1.0 - 2.0 * (torch.arange(w) % 3 == 2)),
                       0.2 * torch.randn(w, generator=g, dtype=torch.float64),
                       torch.zeros(w, dtype=torch.float64), torch.ones(w, dtype=torch.float64)])
        cin = w
    return layers


@pytest.mark.parametrize("form,pool,widths,impose", [
    ("q_ctr", True, [16, 32], False),          # EdgeConv / T-Net form
    ("xyz_bias", True, [16, 24, 32], False),   # coordinate-only first layer (MSG SA1)
    ("q_xyz", True, [16, 32, 32], True),       # feature + coordinate first layer, an imposed activation pattern
    ("q_ctr", False, [16, 16], True),          # unpooled
])
def test_chunked_reference_equals_autograd(form, pool, widths, impose):
    g = torch.Generator().manual_seed(len(form) + len(widths))
    B, N, M, S, C1 = 5, 40, 12, 6, widths[0]
    dt = torch.float64
    idx = torch.randint(0, N, (B, M, S), generator=g).int()
    src = {"Q": torch.randn(B, N, C1, generator=g, dtype=dt) if form in ("q_ctr", "q_xyz") else None,
           "Ctr": torch.randn(B, M, C1, generator=g, dtype=dt) if form == "q_ctr" else None,
           "xyz": torch.randn(B, N, 3, generator=g, dtype=dt) if "xyz" in form else None,
           "new_xyz": torch.randn(B, M, 3, generator=g, dtype=dt) if "xyz" in form else None,
           "wxyz": torch.randn(3, C1, generator=g, dtype=dt) if "xyz" in form else None,
           "bias": 0.1 * torch.randn(C1, generator=g, dtype=dt) if form == "xyz_bias" else None}
    layers = _layers(C1, widths, 7)
    R = B * M * S
    go = torch.randn((B * M if pool else R, widths[-1]), generator=g, dtype=dt)
    pattern = None
    if impose:        # an arbitrary pattern (not the run's own): random masks, random arg-max rows
        masks = [torch.rand(R, w, generator=g) < 0.6 for w in widths]
        pattern = (masks, torch.randint(0, S, (B * M, widths[-1]), generator=g).to(torch.uint8) if pool else None)
    diff = ("Q", "Ctr", "wxyz", "bias")

    s = {k: (v.clone().requires_grad_(k in diff) if v is not None else None) for k, v in src.items()}
    ls = [[t.clone().requires_grad_(True) for t in l[:4]] + l[4:] for l in layers]
    y1 = MR.gather_first_layer(s["Q"], s["Ctr"], s["xyz"], s["new_xyz"], s["wxyz"], s["bias"], idx, dt)
    want_out = MR.run_stack(y1, None, ls, S, pool, True, dt, pattern)
    want_out.backward(go)
    want = [s[k].grad for k in diff if s[k] is not None]
    for li, l in enumerate(ls):
        want += [t.grad for ti, t in enumerate(l[:4]) if not (li == 0 and ti < 2)]

    out, got = MR.chunked_gather_stack(src, idx, layers, pool, go, pattern, dt, clouds_per_chunk=2)
    assert (out - want_out.detach()).abs().max().item() < 1e-12
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert a.shape == b.shape
        assert (a - b).abs().max().item() <= 1e-10 * max(1.0, b.abs().max().item())

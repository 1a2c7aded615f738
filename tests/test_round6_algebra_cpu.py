"""CPU statements of round 6's algebra (no GPU, no library call): the Gram form of a pooled layer's weight gradient
(csrc/mlp.hip bwd_fused_kernel<..., GW> + bwd_fused_gw_finish_kernel) and the first-maximiser rule of
the whole-cloud pool's host-side step (dgcnn/tf_util._FirstMax)."""
import pytest
import torch


@pytest.mark.parametrize("rows,K,N,S,bias", [(640, 64, 128, 32, True), (600, 48, 96, 20, False), (36 * 12, 64, 64, 12, True)])
def test_gram_form_of_the_pooled_weight_gradient(rows, K, N, S, bias):
    """dY = p.G + q.Y + t with ONE non-zero row of G per (group, channel) and Y = X W + b:
       X^T dY = X^T (p.G) + (X^T X) W diag(q) + (X^T 1)(q.b + t)^T   -- what the kernel accumulates (arg rows as vector work, a
       K x K Gram matrix on the matrix pipe, column sums of X) and what the finishing kernel combines."""
    g = torch.Generator().manual_seed(rows + N)
    X = torch.relu(torch.randn(rows, K, generator=g, dtype=torch.float64))
    W = torch.randn(K, N, generator=g, dtype=torch.float64) / K ** 0.5
    b = torch.randn(N, generator=g, dtype=torch.float64) if bias else torch.zeros(N, dtype=torch.float64)
    p, q, t = (torch.randn(N, generator=g, dtype=torch.float64) for _ in range(3))
    G = rows // S
    gpool = torch.randn(G, N, generator=g, dtype=torch.float64)
    arg = torch.randint(0, S, (G, N), generator=g)
    Gfull = torch.zeros(G, S, N, dtype=torch.float64).scatter_(1, arg.unsqueeze(1), gpool.unsqueeze(1)).view(rows, N)
    Y = X @ W + b
    direct = X.t() @ (p * Gfull + q * Y + t)
    # the three pieces the kernel's workgroups leave
    sparse = torch.zeros(K, N, dtype=torch.float64)
    for grp in range(G):
        for c in range(N):
            sparse[:, c] += p[c] * gpool[grp, c] * X[grp * S + arg[grp, c]]
    gram, xsum = X.t() @ X, X.sum(0)
    combined = sparse + (gram @ W) * q + torch.outer(xsum, q * b + t)
    assert (combined - direct).abs().max() <= 1e-10 * direct.abs().max()


def test_first_maximiser_takes_the_gradient_of_the_chunk_maxima():
    """two chunks of a cloud whose maxima agree to the last bit: torch's amax backward halves the gradient between them, the
    contract and every pooling kernel give it to the first -- dgcnn/tf_util._FirstMax"""
    from scanobjectnn_amd.dgcnn.tf_util import _FirstMax
    part = torch.tensor([[[[1.0, 5.0]], [[3.0, 5.0]], [[3.0, 2.0]]]], requires_grad=True)     # (B=1, chunks=3, 1, C=2)
    out = _FirstMax.apply(part, 1)
    assert out.shape == (1, 1, 1, 2) and out.flatten().tolist() == [3.0, 5.0]
    out.backward(torch.tensor([[[[10.0, 20.0]]]]))
    assert part.grad.flatten().tolist() == [0.0, 20.0, 10.0, 0.0, 0.0, 0.0]
    ref = part.detach().clone().requires_grad_(True)
    ref.amax(dim=1, keepdim=True).backward(torch.tensor([[[[10.0, 20.0]]]]))
    assert ref.grad.flatten().tolist() == [0.0, 10.0, 5.0, 10.0, 5.0, 0.0]                  # (what the product used to do)

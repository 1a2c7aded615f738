"""Float64 / float32 torch references of the fused shared-MLP stacks for the GPU parity tests, with the option to
IMPOSE the activation pattern the fused kernels used.

Why: a ReLU whose pre-activation sits within fp32 rounding of 0 (or two neighbourhood members whose activations
tie within rounding under the max-pool) can land on the other side in a float64 run.  That is not an error of
either implementation, but it moves single gradient elements by O(1).  Instead of widening tolerances, the tests
read the pattern the kernels actually used (ReLU masks from the raw layer outputs the autograd node keeps, arg-max
rows of the pooled layer), check that every disagreement with the float64 pattern is a genuine rounding-level tie
(`check_pattern`), and then compare gradients against the float64 reference evaluated WITH THAT SAME PATTERN -- for
which plain tolerances hold.
"""
import torch

EPS = 1e-3


def fused_pattern(out):
    """(masks per layer | None where the layer is never stored, arg-max (G,C) uint8 | None) of the fused stack that
    produced `out` (FusedMLPStack keeps its raw layer outputs and BN coefficients on the autograd node)"""
    node = out.grad_fn
    while node is not None and not hasattr(node, "saved"):
        node = node.next_functions[0][0] if node.next_functions else None
    assert node is not None, "no fused-stack node behind this tensor"
    if len(node.saved) == 11:           # EdgeConvPool: ONE pooled layer, only the arg-max row is a discrete decision
        return [None], node.saved[5]
    Ys, scales, shifts, argmax = node.saved[7], node.saved[10], node.saved[11], node.saved[14]
    expand = None
    rows = getattr(node, "rows", None)
    if rows is not None:
        # compacted rows: the stack ran on 16 * ceil(cnt / 16) rows per group.  Map every row (g, s) of the full
        # (b, m, S) layout to the compacted row that computed it: s itself while it exists, row 0 of the group (whose
        # copy it is) beyond -- then the masks below line up with the uncompacted float64 reference
        idx = node.saved[2]
        S = idx.shape[2]
        bs = rows.block_start.long()
        first = bs[:-1] * 16
        nrow = (bs[1:] - bs[:-1]) * 16
        s_ar = torch.arange(S, device=idx.device).view(1, S)
        expand = (first.view(-1, 1) + torch.where(s_ar < nrow.view(-1, 1), s_ar, torch.zeros_like(s_ar))).reshape(-1)
    masks = []
    for Y, sc, sh in zip(Ys, scales, shifts):
        if Y is None:
            masks.append(None)
        else:
            # the kernels decide with ONE fused multiply-add, fmaf(y, scale, shift) > 0.  In float64 the product of
            # two fp32 numbers is exact and the sum keeps the sign of the exact value, which is the sign the
            # correctly rounded fp32 FMA has: bit-for-bit the kernels' decision.  (A separately rounded fp32
            # multiply + add differs for a handful of the 5e8 elements of a bench-size layer, and a single wrong
            # mask bit moves one row of dx by O(1).)
            n = Y.shape[1]
            m = torch.empty(Y.shape, dtype=torch.bool, device=Y.device)
            step = max(1, (1 << 26) // n)
            for r0 in range(0, Y.shape[0], step):
                m[r0:r0 + step] = (Y[r0:r0 + step].double() * sc[:n].double() + sh[:n].double()) > 0
            masks.append(m if expand is None else m[expand])
    return masks, argmax


def _bn(y, gamma, beta, mm, mv, training):
    if training:
        var, mean = torch.var_mean(y, dim=0, unbiased=False)
    else:
        mean, var = mm.to(y.dtype), mv.to(y.dtype)
    return (y - mean) * torch.rsqrt(var + EPS) * gamma.to(y.dtype) + beta.to(y.dtype)


def run_stack(y_first, a_first, layers, S, pool, training, dtype, pattern=None, report=None):
    """L x [X W + b -> BN -> ReLU] (-> max over S rows).  Either `a_first` (rows, K0) is the input of layer 0, or
    `y_first` (rows, C1) is ALREADY the raw output of layer 0 (gather-first stacks).  pattern = (masks, argmax)
    imposes the activation pattern; report (a dict) receives how far the imposed decisions are from this run's own."""
    masks, argmax = pattern if pattern is not None else (None, None)
    a = a_first
    worst_relu = worst_pool = 0.0
    nflip = 0
    for li, (W, b, gamma, beta, mm, mv) in enumerate(layers):
        y = y_first if (li == 0 and y_first is not None) else a @ W.to(dtype) + b.to(dtype)
        z = _bn(y, gamma, beta, mm, mv, training)
        own = z > 0
        last_pooled = pool and li == len(layers) - 1
        if masks is not None and masks[li] is not None and not last_pooled:
            m = masks[li]
            dis = m != own
            nflip += int(dis.sum().item())
            if dis.any():
                worst_relu = max(worst_relu, z[dis].abs().max().item())
            a = z * m.to(dtype)
        else:
            a = torch.relu(z)
    if pool:
        g = a.view(-1, S, a.shape[1])
        if argmax is not None:
            picked = torch.gather(g, 1, argmax.long().unsqueeze(1)).squeeze(1)
            worst_pool = (g.amax(dim=1) - picked).abs().max().item()
            a = picked
        else:
            a = g.amax(dim=1)
    if report is not None:
        report.update(relu_flips=nflip, worst_relu=worst_relu, worst_pool=worst_pool)
    return a


def check_pattern(report, n_elements, tie=2e-4):
    """every decision of the fused path that differs from the float64 run's own is a rounding-level tie, and there
    are few of them"""
    assert report["worst_relu"] <= tie, report       # flipped ReLUs sit within rounding of 0 in float64
    assert report["worst_pool"] <= tie, report       # a different arg-max row only among (near-)equal maxima
    assert report["relu_flips"] <= max(8, 2e-5 * n_elements), report


def gather_first_layer(Q, Ctr, xyz, new_xyz, wxyz, bias, idx, dtype):
    """Y1[b,j,s,:] = Q[b,idx] + Ctr[b,j] + (xyz[b,idx] - new_xyz[b,j]) wxyz + bias  -> (B*M*S, C1)"""
    B, M, S = idx.shape
    ii = idx.long().reshape(B, M * S, 1)
    y = 0
    if Q is not None:
        y = y + torch.gather(Q.to(dtype), 1, ii.expand(-1, -1, Q.shape[2])).view(B, M, S, -1)
    if Ctr is not None:
        y = y + Ctr.to(dtype).unsqueeze(2)
    if wxyz is not None:
        g = torch.gather(xyz.to(dtype), 1, ii.expand(-1, -1, 3)).view(B, M, S, 3) - new_xyz.to(dtype).unsqueeze(2)
        y = y + g @ wxyz.to(dtype)
    if bias is not None:
        y = y + bias.to(dtype)
    return y.reshape(B * M * S, -1)


def assert_grads_close(names, got, want, plain, rel=1e-3, floor_scale=None):
    """per tensor: max |fused - fp64| <= rel * max|fp64|, or -- where even plain fp32 autograd with the SAME
    activation pattern cannot reach that (heavily cancelling sums behind a batch norm) -- no worse than 1.5x the
    plain-fp32 error, the measured fp32 floor of this very computation"""
    gmax = max(b.abs().max().item() for b in want)
    for name, a, b, c in zip(names, got, want, plain):
        scale = b.abs().max().item()
        if scale < 1e-6 * gmax:
            # analytically zero gradient (a bias in front of a batch norm): rounding noise on every side
            assert a.abs().max().item() <= 1e-3 * (floor_scale if floor_scale is not None else gmax), (name, scale)
            continue
        err = (a.double() - b).abs().max().item()
        err_plain = (c.double() - b).abs().max().item()
        assert err <= max(rel * scale + 1e-6, 1.5 * err_plain), (name, err, err_plain, scale)

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
    return node_pattern(node)


def _fma32(a, b, c):
    """fp32 fused multiply-add of fp32 tensors, via float64: the product is exact, the sum is rounded to 53 bits and
    then to 24 -- equal to the single rounding of a hardware FMA except on double-rounding ties (probability 2^-29
    per element, and then one ulp)"""
    return (a.double() * b.double() + c.double()).float()


def node_pattern(node, virtual_first_layer=False):
    """fused_pattern() of a given FusedMLPStack / EdgeConvPool autograd node.  virtual_first_layer: also rebuild the
    mask of an arithmetic first layer that is never stored (fused_mlp `virt`: y = fma(dz, w2, fma(dy, w1, fma(dx, w0,
    b))) from the centred offsets the node keeps -- csrc/mlp.hip xyz_y) instead of returning None for it."""
    if len(node.saved) == 11:           # EdgeConvPool: ONE pooled layer, only the arg-max row is a discrete decision
        return [None], node.saved[5]
    Ys, scales, shifts, argmax = node.saved[7], node.saved[10], node.saved[11], node.saved[14]
    off4, xyzw = node.saved[16], node.saved[17]
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
    for li, (Y, sc, sh) in enumerate(zip(Ys, scales, shifts)):
        if Y is None and li == 0 and off4 is not None and virtual_first_layer:
            n = xyzw.shape[1]
            m = torch.empty((off4.shape[0], n), dtype=torch.bool, device=off4.device)
            w0, w1, w2, b = (xyzw[i].view(1, n) for i in range(4))
            step = max(1, (1 << 24) // n)
            for r0 in range(0, off4.shape[0], step):
                o = off4[r0:r0 + step]
                y = _fma32(o[:, 2:3], w2, _fma32(o[:, 1:2], w1, _fma32(o[:, 0:1], w0, b)))
                m[r0:r0 + step] = (y.double() * sc[:n].double() + sh[:n].double()) > 0
            masks.append(m if expand is None else m[expand])
        elif Y is None:
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


def chunked_gather_stack(src, idx, layers, pool, go, pattern=None, dtype=torch.float64, clouds_per_chunk=8,
                         report=None):
    """Reference of a TRAINING-mode gather-first stack -- forward output and every gradient -- for row counts whose
    autograd graph does not fit (DGCNN's T-Net at the benchmark size: 10.5 M rows; MSG SA1: 16.7 M rows, float64
    activations of 8-17 GB per layer): the rows are visited in chunks of whole clouds, every chunk recomputes its
    activations, batch-norm statistics and all reductions are accumulated across chunks, and the backward is written
    out by hand:
        zhat = (y - mean) rstd,  z = gamma zhat + beta,  a = mask . z
        dz = da . mask,  dgamma = sum dz zhat,  dbeta = sum dz,  dy = gamma rstd (dz - dbeta / R - zhat dgamma / R)
        dW_l = a_(l-1)^T dy_l,  db_l = sum dy_l,  da_(l-1) = dy_l W_l^T
    dy of a layer needs the FULL-batch dgamma / dbeta of that layer, so the backward takes L + 1 passes over the chunks
    (pass p finishes the sums of layer p and, with them known, the weight gradient of layer p + 1).
    src: Q (B,N,C1) | Ctr (B,M,C1) | xyz (B,N,3) + new_xyz (B,M,3) + wxyz (3,C1) | bias (C1), each optional;
    idx (B,M,S); layers as in run_stack (layer 0 supplies only its BN variables); go: upstream gradient, (B*M, C_L) if
    pool else (B*M*S, C_L); pattern = (masks, argmax) imposes the activation pattern (masks[l] None: the run's own).
    report (a dict) receives how far the imposed decisions are from this run's own, as in run_stack.
    Validated against autograd (run_stack) by tests/test_mlp_ref_cpu.py.
    -> (out, grads) with grads ordered [dQ, dCtr, dwxyz, dbias (those present)] + per layer [gamma, beta] for layer 0
    and [W, b, gamma, beta] above -- the order of test_fused_mlp_gpu._gather_backward_check."""
    B, M, S = idx.shape
    L, R = len(layers), B * M * S
    dev = idx.device
    masks, argmax = pattern if pattern is not None else (None, None)
    W = [l[0].to(dtype) for l in layers]
    b = [l[1].to(dtype) for l in layers]
    gam = [l[2].to(dtype) for l in layers]
    bet = [l[3].to(dtype) for l in layers]
    C = [g.shape[0] for g in gam]
    mean, rstd = [None] * L, [None] * L
    per_cloud = ("Q", "Ctr", "xyz", "new_xyz")
    rep = {"relu_flips": 0, "worst_relu": 0.0, "worst_pool": 0.0, "pass": None} if report is not None else None
    chunks = [(b0, min(B, b0 + clouds_per_chunk)) for b0 in range(0, B, clouds_per_chunk)]

    def forward(b0, b1, upto):
        r0, r1 = b0 * M * S, b1 * M * S
        s = {k: (v[b0:b1] if (v is not None and k in per_cloud) else v) for k, v in src.items()}
        ys, zh, ms, acts = [], [], [], []
        a = None
        for l in range(upto + 1):
            y = (gather_first_layer(s["Q"], s["Ctr"], s["xyz"], s["new_xyz"], s["wxyz"], s["bias"], idx[b0:b1], dtype)
                 if l == 0 else a @ W[l] + b[l])
            ys.append(y)
            if mean[l] is None:
                break                                   # this layer's statistics are what the caller is collecting
            zhat = (y - mean[l]) * rstd[l]
            z = zhat * gam[l] + bet[l]
            if masks is not None and masks[l] is not None and not (pool and l == L - 1):
                m = masks[l][r0:r1]
                if rep is not None and upto == L - 1 and rep["pass"] == L - 1:
                    dis = m != (z > 0)
                    n = int(dis.sum())
                    rep["relu_flips"] += n
                    if n:
                        rep["worst_relu"] = max(rep["worst_relu"], z[dis].abs().max().item())
            else:
                m = z > 0
            a = z * m.to(dtype)
            zh.append(zhat)
            ms.append(m)
            acts.append(a)
        return ys, zh, ms, acts

    for l in range(L):                                  # statistics, layer by layer (shifted by the first chunk's mean)
        s1 = torch.zeros(C[l], dtype=torch.float64, device=dev)
        s2 = torch.zeros(C[l], dtype=torch.float64, device=dev)
        piv = None
        for b0, b1 in chunks:
            y = forward(b0, b1, l)[0][l].double()
            if piv is None:
                piv = y.mean(dim=0)
            d = y - piv
            s1 += d.sum(dim=0)
            s2 += (d * d).sum(dim=0)
        m1 = s1 / R
        mean[l] = (piv + m1).to(dtype)
        rstd[l] = torch.rsqrt(s2 / R - m1 * m1 + EPS).to(dtype)

    G = B * M
    out = torch.empty((G if pool else R, C[-1]), dtype=dtype, device=dev)
    dgam = [torch.zeros(c, dtype=dtype, device=dev) for c in C]
    dbet = [torch.zeros(c, dtype=dtype, device=dev) for c in C]
    dW = [None] + [torch.zeros_like(W[l]) for l in range(1, L)]
    db = [None] + [torch.zeros_like(b[l]) for l in range(1, L)]
    C1 = C[0]
    d_first = {"Q": torch.zeros(src["Q"].shape, dtype=dtype, device=dev) if src["Q"] is not None else None,
               "Ctr": torch.zeros(src["Ctr"].shape, dtype=dtype, device=dev) if src["Ctr"] is not None else None,
               "wxyz": torch.zeros((3, C1), dtype=dtype, device=dev) if src["wxyz"] is not None else None,
               "bias": torch.zeros(C1, dtype=dtype, device=dev) if src["bias"] is not None else None}
    for p in range(L - 1, -2, -1):
        if rep is not None:
            rep["pass"] = p
        for b0, b1 in chunks:
            nb = b1 - b0
            r0, r1, g0, g1 = b0 * M * S, b1 * M * S, b0 * M, b1 * M
            ys, zh, ms, acts = forward(b0, b1, L - 1)
            if pool:
                a3 = acts[-1].view(nb * M, S, C[-1])
                arg = (argmax[g0:g1].long() if argmax is not None else a3.argmax(dim=1)).unsqueeze(1)
                if p == L - 1:
                    out[g0:g1] = torch.gather(a3, 1, arg).squeeze(1)
                    if rep is not None and argmax is not None:
                        rep["worst_pool"] = max(rep["worst_pool"], (a3.amax(dim=1) - out[g0:g1]).abs().max().item())
                da = torch.zeros_like(a3).scatter_(1, arg, go[g0:g1].to(dtype).unsqueeze(1)).view(-1, C[-1])
            else:
                if p == L - 1:
                    out[r0:r1] = acts[-1]
                da = go[r0:r1].to(dtype)
            for l in range(L - 1, max(p, 0) - 1, -1):
                dz = da * ms[l].to(dtype)
                if l == p:
                    dgam[l] += (dz * zh[l]).sum(dim=0)
                    dbet[l] += dz.sum(dim=0)
                    break
                dy = (gam[l] * rstd[l]) * (dz - dbet[l] / R - zh[l] * (dgam[l] / R))
                if l == 0:                              # (p == -1) the gather form's own gradients
                    if d_first["bias"] is not None:
                        d_first["bias"] += dy.sum(dim=0)
                    if d_first["Ctr"] is not None:
                        d_first["Ctr"][b0:b1] = dy.view(nb, M, S, C1).sum(dim=2)
                    ii = idx[b0:b1].long().reshape(nb, M * S, 1)
                    if d_first["wxyz"] is not None:
                        off = (torch.gather(src["xyz"][b0:b1].to(dtype), 1, ii.expand(-1, -1, 3)).view(nb, M, S, 3)
                               - src["new_xyz"][b0:b1].to(dtype).unsqueeze(2))
                        d_first["wxyz"] += off.reshape(-1, 3).t() @ dy
                    if d_first["Q"] is not None:
                        d_first["Q"][b0:b1].scatter_add_(1, ii.expand(-1, -1, C1), dy.view(nb, M * S, C1))
                    break
                if l == p + 1:
                    dW[l] += acts[l - 1].t() @ dy
                    db[l] += dy.sum(dim=0)
                da = dy @ W[l].t()
            del ys, zh, ms, acts, da
    grads = [d_first[k] for k in ("Q", "Ctr", "wxyz", "bias") if d_first[k] is not None]
    for l in range(L):
        if l > 0:
            grads += [dW[l], db[l]]
        grads += [dgam[l], dbet[l]]
    if report is not None:
        report.update({k: rep[k] for k in ("relu_flips", "worst_relu", "worst_pool")})
    return out, [g.double() for g in grads]

"""Fused shared-MLP stack: L x [1x1 conv + bias + BatchNorm + ReLU] (+ max-pool over the neighbourhood),
forward and backward, on the hand-written fp32-MFMA kernels of libpcops (csrc/mlp.hip, csrc/gather.hip).

This is the device-side replacement for the per-layer TensorFlow op chain the reference builds in
`pointnet_sa_module` / `pointnet_fp_module` / EdgeConv (pointnet2/utils/pointnet_util.py:117-127,223-227;
pointnet2/utils/tf_util.py:120-185,512-531; dgcnn/models/dgcnn.py:39-48): each layer is ONE pass over the
activations in each direction (BN+ReLU of the previous layer folded into the operand load, batch statistics
emitted by the GEMM epilogue, BN backward folded into three per-channel vectors).  Same math as the
layer-by-layer path (`tf_util.conv2d` in sequence): batch mean / biased variance in training; eps, decay and
moving-average semantics are parameters so both BN flavours of the reference are covered.

Two kinds of first layer:
  dense   Y1 = X W1 + b1 on a materialised (rows, K) input (FP stacks, group_all, anything already grouped);
  gather  Y1[b,j,s,:] = Q[b, idx[b,j,s], :] + Ctr[b,j,:] + (xyz[b,idx] - new_xyz[b,j]) Wxyz + bias -- the first conv
          of a GROUPED stack applied before the grouping (it is linear; see csrc/gather.hip), so neither the
          grouped input nor a (3+C)-wide concat is ever built and its backward is one scatter-add.
"""
import os

import torch

from . import _lib
from . import dist as _dist


def _f32(n, dev):
    return torch.empty(n, dtype=torch.float32, device=dev)


class _VecArena:
    """per-channel vectors, zero-filled and padded to a multiple of 4 floats (kernels read them 16 bytes at a
    time), carved from ONE zeroed allocation per pass instead of one fill launch per vector"""

    def __init__(self, widths, per_width, dev):
        total = sum(((n + 3) // 4 * 4) * per_width for n in widths)
        # the zero fill only matters for the padding lanes of a width that is not a multiple of 4 (every vector is
        # written in full by the kernel that owns it): the usual case needs no fill launch at all
        alloc = torch.empty if all(n % 4 == 0 for n in widths) else torch.zeros
        self.buf = alloc(total, dtype=torch.float32, device=dev)
        self.off = 0

    def take(self, n):
        n4 = (n + 3) // 4 * 4
        v = self.buf[self.off:self.off + n4]
        self.off += n4
        return v


def _workspace(n, dev):
    lib = _lib.load()
    return torch.empty(int(lib.pcops_mlp_reduce_workspace_bytes(n)) // 8, dtype=torch.float64, device=dev)


def _p(t):
    return t.data_ptr() if t is not None else None


class FusedMLPStack(torch.autograd.Function):
    """apply(a0, ctr, idx, xyz, new_xyz, wxyz, bias, S, pool, training, decay, eps, unbiased_moving_var, rows, L, *per_layer)

    rows: None, or a `_lib.Rows` compacted row set of the grouped stack (ball-query padding left out, pcops.h
          "compacted rows"): gather first layer, max-pooled, no Ctr term
    dense first layer : a0 = x2d (R, K0); ctr .. bias = None
    gather first layer: idx (B, M, S) int32 and any of a0 = Q (B, N, C1), ctr (B, M, C1),
                        xyz (B, N, 3) + new_xyz (B, M, 3) + wxyz (3, C1), bias (C1);  R = B*M*S
    per_layer (6 each): weights (K,N), biases (N), gamma, beta, moving_mean, moving_var -- the first two are None
    for a gather first layer.  Returns (R//S, C_L) if pool else (R, C_L)."""

    @staticmethod
    def forward(ctx, a0, ctr, idx, xyz, new_xyz, wxyz, bias, S, pool, training, decay, eps, unbiased, rows, L, *tensors):
        lib = _lib.load()
        rref = rows.ref if rows is not None else None
        identity = bool(int(pool) & 2)      # gather stack whose idx is 0..n-1 per cloud (group_all): scatter = reshape
        qc = bool(int(pool) & 4)            # a0 is the (B, N, 2 C1) product [Q | Ctr] of ONE GEMM (pcops.h "[Q | Ctr] forms")
        # qc with an input that needs no gradient (DGCNN's T-Net on the raw cloud): xyz = the (B, N, 3) input, wxyz = the
        # layer's (6, C1) weight, bias = its bias -- their gradients come from ONE streaming pass over the masked gradient
        # (pcops.h pcops_edge_first_*), a0 gets none and no scatter runs
        direct = bool(int(pool) & 8)
        pool = bool(int(pool) & 1)
        gather = idx is not None
        need_grad = any(ctx.needs_input_grad)
        sync = training and _dist.sync_bn_active()
        dev = idx.device if gather else a0.device
        layers = [tensors[6 * i:6 * i + 6] for i in range(L)]
        if gather:
            B, M, _ = idx.shape
            Nsrc = a0.shape[1] if a0 is not None else xyz.shape[1]
            C1 = layers[0][2].shape[0]
            R, K0 = B * M * S, None
            if qc:
                assert ctr is None and rows is None and M == Nsrc and new_xyz is None
                assert direct or (xyz is None and wxyz is None and bias is None)
                assert not direct or (xyz is not None and wxyz is not None and tuple(wxyz.shape) == (6, C1) and L >= 2
                                      and not ctx.needs_input_grad[0])
                assert a0.shape[2] == 2 * C1 and a0.is_contiguous()
        else:
            R, K0 = a0.shape
        Ys, means, rstds, scales, shifts, Ws = [], [], [], [], [], []
        src, ld, sc_prev, sh_prev, K = a0, K0, None, None, K0
        vecs = _VecArena([l[2].shape[0] for l in layers], 4, dev)
        ws = _workspace(max(l[2].shape[0] for l in layers), dev) if training else None
        pooled_raw = pooled_parts = None
        pool_top = False          # the pooled top layer takes the algebraic backward: decided ONCE, here (the forward
        #                           drops Y on that decision, so the backward must not come to a different one)
        # a first layer with only the coordinate term is ARITHMETIC in three offsets per row: it is never stored, the
        # next layer and the whole backward rebuild it from off4 (16 bytes per row instead of 4 C1)
        # (with SyncBN, or a backward through eval-mode BN, the layer is materialised: its gradient shortcut assumes
        # rank-local batch statistics)
        virt = (gather and a0 is None and ctr is None and wxyz is not None and (L >= 3 or (L == 2 and not pool))
                and not sync and (training or not need_grad)
                and bool(lib.pcops_mlp_xyz_supported(R, C1, layers[1][0].shape[-1])))
        if rows is not None and gather and a0 is None and not virt:
            rows = rref = None          # a stored coordinate-only first layer has no compacted backward: plain rows
        off4 = xyzw = mom = None
        if virt:
            off4 = _f32((R, 4), dev)
            mom = _f32((lib.pcops_sa_gather_stats_rows(B * M), 9), dev) if training else None
            xyzw = torch.cat([wxyz.detach(), (bias.detach() if bias is not None
                                              else torch.zeros(C1, dtype=torch.float32, device=dev)).view(1, C1)]).contiguous()
        for li, (w, b, gamma, beta, mm, mv) in enumerate(layers):
            # forward statistics are shifted moments around the layer's moving mean (pcops.h pcops_mlp_gemm_fwd): the
            # producer and pcops_mlp_bn_finalize get the same pivot; finalize reads it before it updates the buffer
            piv = mm.data_ptr() if (training and STAT_PIVOT) else None
            if li == 0 and gather and qc:
                N = C1
                Y = _f32((R, N), dev)
                P = lib.pcops_sa_gather_fwd_stats_rows(B, Nsrc, M, S, N, 1, 1, 0, 0)
                part = _f32((P, 2, N), dev) if training else None
                _lib.call("pcops_sa_gather_fwd_ld", B, Nsrc, M, S, N, a0.data_ptr(), 2 * N, a0.data_ptr() + 4 * N, 2 * N,
                          idx.data_ptr(), Y.data_ptr(), _p(part), piv)
                if direct and need_grad:      # the 27 moments of the edge features, for E^T Y1 in the backward
                    mom = _f32((lib.pcops_edge_first_rows(), 27), dev)
                    # ... and, where the layer above takes the one-pass backward, the edge rows themselves (32 bytes each):
                    # E^T Gm is then reduced inside that kernel and the masked gradient is never written
                    N1 = layers[1][0].shape[-1]
                    if (EDGE_DIRECT_FUSED and BWD_FUSED and L == 2 and pool and S % 32 != 0
                            and lib.pcops_mlp_bwd_fused_groups(R, N, N1, S, 1)):
                        ctx.edge_rows = _f32((R, 8), dev)
                    _lib.call("pcops_edge_first_moments", B, Nsrc, M, S, xyz.data_ptr(), idx.data_ptr(), mom.data_ptr(),
                              _p(getattr(ctx, "edge_rows", None)))
                W2 = None
            elif (li == 0 and gather and identity and CLOUD_BIAS and a0 is not None and ctr is not None and xyz is None
                    and wxyz is None and bias is None and rows is None and M == 1 and Nsrc == S and a0.is_contiguous()
                    and lib.pcops_cloud_bias_supported(R, S, C1)):
                # whole clouds in their own order: Y = Q + Ctr[cloud] as one streaming pass (pcops.h pcops_cloud_bias_*)
                N = C1
                Y = _f32((R, N), dev)
                P = lib.pcops_cloud_bias_rows(R)
                part = _f32((P, 2, N), dev) if training else None
                _lib.call("pcops_cloud_bias_fwd", R, S, N, a0.data_ptr(), ctr.data_ptr(), Y.data_ptr(), _p(part), piv)
                W2 = None
                ctx.cloud_bias = True
            elif li == 0 and gather:
                N = C1
                Y = None if virt else _f32((R, N), dev)
                other = wxyz is not None or bias is not None or off4 is not None
                P = lib.pcops_sa_gather_fwd_stats_rows(B, Nsrc, M, S, N, int(a0 is not None), int(ctr is not None),
                                                        int(other), int(rref is not None))
                part = _f32((P, 2, N), dev) if training else None
                _lib.call("pcops_sa_gather_fwd_rows", B, Nsrc, M, S, N, _p(a0), _p(ctr), _p(xyz), _p(new_xyz),
                          _p(wxyz), _p(bias), idx.data_ptr(), _p(Y), _p(off4), _p(part), piv, _p(mom), rref)
                W2 = None
            else:
                N = w.shape[-1]
                W2 = w.detach().reshape(-1, N)
                assert W2.shape[0] == K and W2.is_contiguous()
                Y = _f32((R, N), dev)
                P = lib.pcops_mlp_stats_rows(R)
                part = _f32((P, 2, N), dev) if training else None
                if li == 1 and virt:
                    _lib.call("pcops_mlp_gemm_fwd_xyz_rows", R, K, N, off4.data_ptr(), xyzw.data_ptr(), sc_prev.data_ptr(),
                              sh_prev.data_ptr(), W2.data_ptr(), b.data_ptr(), Y.data_ptr(), _p(part), piv, rref)
                elif (rows is not None and pool and li == L - 1 and sc_prev is not None and ld == K and FUSE_POOL_ROWS
                        and lib.pcops_mlp_gemm_fwd_pool_rows_supported(R, K, N)):
                    # compacted rows: the epilogue emits the extremum of every 16-row block (a block lies inside one
                    # group), a small pass picks per group afterwards
                    nbk = rows.blocks.shape[0]
                    pooled_parts = (_f32((nbk, N), dev), torch.empty((nbk, N), dtype=torch.uint8, device=dev))
                    _lib.call("pcops_mlp_gemm_fwd_pool_rows", R, K, N, src.data_ptr(), ld, sc_prev.data_ptr(),
                              sh_prev.data_ptr(), W2.data_ptr(), b.data_ptr(), gamma.data_ptr(), Y.data_ptr(),
                              _p(part), piv, pooled_parts[0].data_ptr(), pooled_parts[1].data_ptr(), rref)
                elif rows is not None:
                    # compacted rows without the fused epilogue: the max over the groups is its own pass below
                    _lib.call("pcops_mlp_gemm_fwd_rows", R, K, N, src.data_ptr(), ld, _p(sc_prev), _p(sh_prev),
                              W2.data_ptr(), b.data_ptr(), Y.data_ptr(), _p(part), piv, rref)
                elif (pool and li == L - 1 and ld == K and lib.pcops_mlp_gemm_fwd_pool_supported(R, K, N, S)
                        and (sc_prev is not None or li == 0)):
                    # neighbourhood max fused into the GEMM epilogue (raw extrema; resolved after the statistics).
                    # The activation itself is only stored when a backward will read it: not for a forward without
                    # gradient, not when the layer takes the algebraic backward
                    G = R // S
                    pooled_raw = (_f32((G, N), dev), torch.empty((G, N), dtype=torch.uint8, device=dev))
                    pool_top = need_grad and _pool_top_ok(lib, R, K, N, S, li, gather, K0, rows, virt and li == 1,
                                                          b is not None)
                    if not need_grad or pool_top:
                        Y = None
                    _lib.call("pcops_mlp_gemm_fwd_pool", R, K, N, S, src.data_ptr(), ld, _p(sc_prev),
                              _p(sh_prev), W2.data_ptr(), b.data_ptr(), gamma.data_ptr(), _p(Y),
                              _p(part), piv, pooled_raw[0].data_ptr(), pooled_raw[1].data_ptr())
                else:
                    _lib.call("pcops_mlp_gemm_fwd", R, K, N, src.data_ptr(), ld, _p(sc_prev), _p(sh_prev),
                              W2.data_ptr(), b.data_ptr(), Y.data_ptr(), _p(part), piv)
            scale, shift = vecs.take(N), vecs.take(N)
            if training:
                mean, rstd = vecs.take(N), vecs.take(N)
                Pf, Rf, piv_fin = P, R, piv
                if sync:        # SyncBN: the statistics of the global batch (the rank's pivot taken out before the exchange)
                    part, Rf = _dist.allreduce_stat_partials(part, R, mm if piv is not None else None)
                    Pf, piv_fin = part.shape[0], None
                _lib.call("pcops_mlp_bn_finalize", Pf, N, Rf, part.data_ptr(), piv_fin, ws.data_ptr(), gamma.data_ptr(),
                          beta.data_ptr(), float(eps), float(decay), int(unbiased), mm.data_ptr(), mv.data_ptr(),
                          mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
                means.append(mean)
                rstds.append(rstd)
            else:
                _lib.call("pcops_mlp_bn_eval_coeffs", N, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(),
                          mv.data_ptr(), float(eps), scale.data_ptr(), shift.data_ptr())
                if need_grad:   # backward through frozen statistics: y_bn = scale*y + shift with constants
                    means.append(mm.detach().clone())
                    rstds.append(torch.rsqrt(mv.detach() + float(eps)))
            Ys.append(Y)
            scales.append(scale)
            shifts.append(shift)
            Ws.append(W2)
            src, ld, sc_prev, sh_prev, K = Y, N, scale, shift, N

        C = K
        argmax = ysel = None
        if pool:
            G = R // S
            out = _f32((G, C), dev)
            if pooled_parts is not None:
                argmax = torch.empty((G, C), dtype=torch.uint8, device=dev)
                ysel = _f32((G, C), dev)
                _lib.call("pcops_mlp_pool_combine_rows", G, C, pooled_parts[0].data_ptr(), pooled_parts[1].data_ptr(),
                          layers[-1][2].data_ptr(), scales[-1].data_ptr(), shifts[-1].data_ptr(), rref,
                          out.data_ptr(), argmax.data_ptr(), ysel.data_ptr())
            elif pooled_raw is not None:
                ysel, argmax = pooled_raw
                _lib.call("pcops_mlp_pool_select", G, C, ysel.data_ptr(), scales[-1].data_ptr(),
                          shifts[-1].data_ptr(), out.data_ptr())
            else:
                argmax = torch.empty((G, C), dtype=torch.uint8, device=dev) if (training or need_grad) else None
                ysel = _f32((G, C), dev) if (training or need_grad) else None
                if rows is not None:
                    _lib.call("pcops_mlp_bn_relu_maxpool_rows", G, C, Ys[-1].data_ptr(), scales[-1].data_ptr(),
                              shifts[-1].data_ptr(), rref, out.data_ptr(), _p(argmax), _p(ysel))
                else:
                    _lib.call("pcops_mlp_bn_relu_maxpool", G, S, C, Ys[-1].data_ptr(), scales[-1].data_ptr(),
                              shifts[-1].data_ptr(), out.data_ptr(), _p(argmax), _p(ysel))
        else:
            out = _f32((R, C), dev)
            _lib.call("pcops_mlp_bn_relu_apply", R, C, Ys[-1].data_ptr(), scales[-1].data_ptr(),
                      shifts[-1].data_ptr(), out.data_ptr())
        if training or need_grad:
            ctx.saved = (a0, ctr, idx, xyz, new_xyz, wxyz, bias, Ys, means, rstds, scales, shifts, Ws,
                         [l[2] for l in layers], argmax, ysel, off4, xyzw, mom)
            ctx.biases = [l[1] for l in layers]
            ctx.meta = (S, pool, L, R, K0, gather, identity, bool(training), bool(sync))
            ctx.qc = qc
            ctx.direct = gather and qc and direct
            ctx.rows = rows
            ctx.pool_top = pool_top
            if TRACE is not None:
                TRACE.append(ctx)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        lib = _lib.load()
        (a0, ctr, idx, xyz, new_xyz, wxyz, bias, Ys, means, rstds, scales, shifts, Ws, gammas, argmax, ysel, off4,
         xyzw, mom) = ctx.saved
        xstats = None
        virt = off4 is not None
        widths = [g.shape[0] for g in gammas]
        S, pool, L, R, K0, gather, identity, training, sync = ctx.meta
        rows = ctx.rows
        rref = rows.ref if rows is not None else None
        dev = grad_out.device
        grad_out = grad_out.contiguous()
        grads = [None] * (6 * L)
        d0 = d1 = dwxyz = dbias = None

        # ---- top of the stack: statistics of the masked upstream gradient
        C = widths[-1]
        ws = _workspace(max(widths), dev)
        vecs = _VecArena(widths, 3, dev)
        if pool:
            G = R // S
            P = lib.pcops_mlp_bwd_pool_stats_rows(G)
            part = _f32((P, 2, C), dev)
            # gmask = the pooled gradient times the ReLU mask at the pooled rows: what the data / weight gradient kernels
            # place at the arg-max rows (pcops.h, pcops_mlp_pool_bwd_stats)
            gmask = _f32((G, C), dev)
            _lib.call("pcops_mlp_pool_bwd_stats", G, C, grad_out.data_ptr(), ysel.data_ptr(),
                      scales[-1].data_ptr(), shifts[-1].data_ptr(), part.data_ptr(), gmask.data_ptr())
            Gm = None
        else:
            P = lib.pcops_mlp_bwd_stats_rows(R)
            part = _f32((P, 2, C), dev)
            Gm = _f32((R, C), dev)
            _lib.call("pcops_mlp_relu_mask_stats", R, C, grad_out.data_ptr(), Ys[-1].data_ptr(),
                      scales[-1].data_ptr(), shifts[-1].data_ptr(), Gm.data_ptr(), part.data_ptr())

        for l in range(L - 1, -1, -1):
            N = widths[l]
            dgamma, dbeta = _f32(N, dev), _f32(N, dev)
            p, q, t = vecs.take(N), vecs.take(N), vecs.take(N)
            _lib.call("pcops_mlp_bn_bwd_coeffs", P, N, R, part.data_ptr(), ws.data_ptr(), gammas[l].data_ptr(),
                      means[l].data_ptr(), rstds[l].data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                      p.data_ptr(), q.data_ptr(), t.data_ptr())
            if sync:
                # SyncBN: dgamma / dbeta stay rank-local sums (the gradient all-reduce adds the ranks up); the
                # dY = p G + q Y + t coefficients come from the sums over the GLOBAL batch
                gpart, Rg = _dist.allreduce_stat_partials(part, R)
                junk = _f32(2 * N, dev)
                _lib.call("pcops_mlp_bn_bwd_coeffs", gpart.shape[0], N, Rg, gpart.data_ptr(), ws.data_ptr(), gammas[l].data_ptr(),
                          means[l].data_ptr(), rstds[l].data_ptr(), junk.data_ptr(), junk[N:].data_ptr(),
                          p.data_ptr(), q.data_ptr(), t.data_ptr())
            if not training:    # frozen statistics: no mean / variance terms in the BN backward
                q.zero_()
                t.zero_()
            grads[6 * l + 2] = dgamma
            grads[6 * l + 3] = dbeta
            pooled = pool and l == L - 1
            gp = gmask.data_ptr() if pooled else None
            am = argmax.data_ptr() if pooled else None
            psc = scales[l].data_ptr() if pooled else None
            psh = shifts[l].data_ptr() if pooled else None
            Gptr = None if (pooled or Gm is None) else Gm.data_ptr()

            if l == 0 and gather and virt:
                # arithmetic first layer: its gradients are linear in sums the layer above already produced
                dwxyz = _f32((3, N), dev)
                dbias = _f32(N, dev) if bias is not None else None
                _lib.call("pcops_xyz_first_layer_grads", xstats.shape[0], xstats.data_ptr(), mom.shape[0], mom.data_ptr(),
                          N, wxyz.data_ptr(), _p(bias), p.data_ptr(), q.data_ptr(), t.data_ptr(), dbeta.data_ptr(),
                          means[0].data_ptr(), R, dwxyz.data_ptr(), _p(dbias))
                break
            if l == 0 and gather and getattr(ctx, "direct", False):
                # the input needs no gradient: dW (6, C1) / db straight from E^T Gm and the edge moments -- no scatter
                B, M, _ = idx.shape
                if xstats is not None:        # reduced by the one-pass backward of the layer above
                    wpart, P1 = xstats, xstats.shape[0]
                else:
                    P1 = lib.pcops_edge_first_rows()
                    wpart = _f32((P1, 6, N), dev)
                    _lib.call("pcops_edge_first_wgrad", B, a0.shape[1], M, S, N, Gptr, xyz.data_ptr(), idx.data_ptr(),
                              wpart.data_ptr())
                dwxyz = _f32((6, N), dev)
                dbias = _f32(N, dev) if bias is not None else None
                _lib.call("pcops_edge_first_layer_grads", P1, wpart.data_ptr(), mom.shape[0], mom.data_ptr(), N,
                          wxyz.data_ptr(), _p(bias), p.data_ptr(), q.data_ptr(), t.data_ptr(), dbeta.data_ptr(),
                          means[0].data_ptr(), R, dwxyz.data_ptr(), _p(dbias))
                break
            if l == 0 and gather and getattr(ctx, "qc", False):
                # [Q | Ctr] form: dQ and dCtr are the column halves of ONE (B, N, 2 C1) gradient
                B, M, _ = idx.shape
                Nsrc = a0.shape[1]
                d0 = _f32((B, Nsrc, 2 * N), dev)
                wsp = torch.empty(int(lib.pcops_sa_scatter_workspace_bytes(B, Nsrc, M, S)) // 4, dtype=torch.int32, device=dev)
                _lib.call("pcops_sa_scatter_bwd_ld", B, Nsrc, M, S, N, Gptr, p.data_ptr(), q.data_ptr(), t.data_ptr(),
                          idx.data_ptr(), a0.data_ptr(), 2 * N, a0.data_ptr() + 4 * N, 2 * N, d0.data_ptr(), 2 * N,
                          d0.data_ptr() + 4 * N, 2 * N, wsp.data_ptr())
                break
            if l == 0 and gather and getattr(ctx, "cloud_bias", False) and Gptr is not None:
                B, M, _ = idx.shape
                d0 = _f32((B, a0.shape[1], N), dev) if ctx.needs_input_grad[0] else None
                d1 = _f32((B, M, N), dev)
                scratch = _f32((lib.pcops_cloud_bias_rows(R), N), dev)
                _lib.call("pcops_cloud_bias_bwd", R, S, N, Gptr, Ys[0].data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(),
                          _p(d0), d1.data_ptr(), scratch.data_ptr())
                break
            if l == 0 and gather:
                B, M, _ = idx.shape
                Nsrc = a0.shape[1] if a0 is not None else xyz.shape[1]
                d0 = _f32((B, Nsrc, N), dev) if (a0 is not None and ctx.needs_input_grad[0]) else None
                d1 = _f32((B, M, N), dev) if ctr is not None else None
                dwxyz = _f32((3, N), dev) if wxyz is not None else None
                dbias = _f32(N, dev) if bias is not None else None
                wpart = _f32(lib.pcops_sa_scatter_rows(B, M) * 4 * N, dev) if (wxyz is not None or bias is not None) else None
                wsp = None
                d0_out = d0
                if d0 is not None and identity and not pooled:
                    # idx = 0..n-1: the scatter-add is the identity map, dQ = dY row for row
                    if TAIL_FOLD and N % 4 == 0:
                        d0_out = _f32((B, Nsrc, N), dev)
                        _lib.call("pcops_mlp_dy_apply", B * Nsrc, N, Gm.data_ptr(), Ys[0].data_ptr(), p.data_ptr(),
                                  q.data_ptr(), t.data_ptr(), d0_out.data_ptr())
                    else:
                        d0_out = torch.addcmul(t[:N], Gm, p[:N]).addcmul_(Ys[0], q[:N]).view(B, Nsrc, N)
                    d0 = None                # the kernel below then only reduces dWxyz / dbias / dCtr
                if d0 is not None:   # gather formulation over an inverse index
                    wsp = torch.empty(int(lib.pcops_sa_scatter_workspace_bytes(B, Nsrc, M, S)) // 4,
                                      dtype=torch.int32, device=dev)
                _lib.call("pcops_sa_scatter_bwd_rows", B, Nsrc, M, S, N, Gptr, _p(Ys[0]), p.data_ptr(),
                          q.data_ptr(), t.data_ptr(), gp, am, psc, psh, idx.data_ptr(),
                          _p(xyz) if wxyz is not None else None, _p(new_xyz) if wxyz is not None else None,
                          _p(d0), _p(d1), _p(wpart), _p(dwxyz), _p(dbias), _p(a0), _p(ctr), _p(wxyz), _p(bias),
                          _p(wsp), rref)
                d0 = d0_out
                break

            K = Ws[l].shape[0]
            xyz_prev = virt and l == 1          # the layer below is the arithmetic first layer (never stored)
            if pooled and ctx.pool_top:
                # algebraic form (pcops.h "algebraic backward of a pooled top layer"): K x K products instead of K x N
                prev = (Ys[l - 1], scales[l - 1], shifts[l - 1]) if l > 0 else (a0, None, None)
                Gm, part = _pool_top_backward(R, K, N, S, Ws[l], ctx.biases[l].detach(), p, q, t, grad_out, ysel,
                                              argmax, scales[l], shifts[l], prev[0], prev[1], prev[2], grads, l, dev,
                                              l > 0 or ctx.needs_input_grad[0])
                P = lib.pcops_mlp_stats_rows(R)
                if l == 0:
                    d0 = Gm
                continue
            fused_groups = 0
            if BWD_FUSED and l > 0 and Ws[l].data_ptr() % 16 == 0:
                # the bandwidth-bound narrow layers: data and weight gradient in ONE pass over Y / Yprev
                fused_groups = lib.pcops_mlp_bwd_fused_groups(R, K, N, S if pooled else 0, 1 if pooled else 0)
            if fused_groups:
                # pooled layer on uncompacted rows: the weight gradient in its Gram form (pcops.h, round 6) -- a K x K
                # product on the matrix pipe + the arg rows as vector work instead of the K x N product
                gw = (pooled and rows is None and not xyz_prev and
                      lib.pcops_mlp_bwd_fused_gw_groups(R, K, N, S) == fused_groups)
                scratch = _f32(fused_groups * (K * N + N + ((K * K + K) if gw else 0)), dev)
                dW, db = _f32((K, N), dev), _f32(N, dev)
                P = fused_groups
                part = _f32((P, 2, K), dev)
                bl = ctx.biases[l]
                edge_rows = getattr(ctx, "edge_rows", None) if (l == 1 and pooled and getattr(ctx, "direct", False)) else None
                if edge_rows is not None:   # the first EdgeConv layer below, input without gradient: E^T Gm reduced in the kernel
                    xstats = _f32((P, 6, K), dev)
                    if gw:
                        _lib.call("pcops_mlp_bwd_fused_edge_gw", R, K, N, Ys[0].data_ptr(), scales[0].data_ptr(),
                                  shifts[0].data_ptr(), Ys[l].data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), gp, am, S,
                                  Ws[l].data_ptr(), _p(bl), scratch.data_ptr(), dW.data_ptr(), db.data_ptr(), part.data_ptr(),
                                  edge_rows.data_ptr(), xstats.data_ptr())
                    else:
                        _lib.call("pcops_mlp_bwd_fused_edge", R, K, N, Ys[0].data_ptr(), scales[0].data_ptr(),
                                  shifts[0].data_ptr(), Ys[l].data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), gp, am, S,
                                  Ws[l].data_ptr(), scratch.data_ptr(), dW.data_ptr(), db.data_ptr(), part.data_ptr(),
                                  edge_rows.data_ptr(), xstats.data_ptr())
                    grads[6 * l + 0] = dW
                    grads[6 * l + 1] = db
                    Gm = None
                    continue
                if gw:
                    Gprev = _f32((R, K), dev)
                    _lib.call("pcops_mlp_bwd_fused_gw", R, K, N, Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(),
                              shifts[l - 1].data_ptr(), Ys[l].data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(),
                              gp, am, S, Ws[l].data_ptr(), _p(bl), scratch.data_ptr(), dW.data_ptr(), db.data_ptr(),
                              Gprev.data_ptr(), part.data_ptr())
                    grads[6 * l + 0] = dW
                    grads[6 * l + 1] = db
                    Gm = Gprev
                    continue
                if xyz_prev:     # the arithmetic first layer below: its masked gradient is reduced, never written
                    xstats = _f32((P, 3, K), dev)
                    _lib.call("pcops_mlp_bwd_fused_xyz_rows", R, K, N, off4.data_ptr(), xyzw.data_ptr(),
                              scales[0].data_ptr(), shifts[0].data_ptr(), Gptr, Ys[l].data_ptr(), p.data_ptr(),
                              q.data_ptr(), t.data_ptr(), gp, am, S, Ws[l].data_ptr(), scratch.data_ptr(),
                              dW.data_ptr(), db.data_ptr(), part.data_ptr(), xstats.data_ptr(), rref)
                    grads[6 * l + 0] = dW
                    grads[6 * l + 1] = db
                    Gm = None
                    continue
                Gprev = _f32((R, K), dev)
                _lib.call("pcops_mlp_bwd_fused_rows", R, K, N, Ys[l - 1].data_ptr(), scales[l - 1].data_ptr(),
                          shifts[l - 1].data_ptr(), Gptr, Ys[l].data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(),
                          gp, am, S, Ws[l].data_ptr(), scratch.data_ptr(), dW.data_ptr(), db.data_ptr(),
                          Gprev.data_ptr(), part.data_ptr(), rref)
                grads[6 * l + 0] = dW
                grads[6 * l + 1] = db
                Gm = Gprev
                continue
            if l == 0:
                src, ld, asc, ash = a0, K0, None, None
            elif not xyz_prev:
                src, ld, asc, ash = Ys[l - 1], Ys[l - 1].shape[1], scales[l - 1].data_ptr(), shifts[l - 1].data_ptr()
            splits = lib.pcops_mlp_wgrad_splits(R, K, N)
            scratch = _f32(splits * (K * N + N), dev)
            dW, db = _f32((K, N), dev), _f32(N, dev)
            if xyz_prev:
                _lib.call("pcops_mlp_wgrad_xyz_rows", R, K, N, off4.data_ptr(), xyzw.data_ptr(), scales[0].data_ptr(),
                          shifts[0].data_ptr(), Gptr, Ys[l].data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), gp, am,
                          S, psc, psh, scratch.data_ptr(), dW.data_ptr(), db.data_ptr(), rref)
            else:
                _lib.call("pcops_mlp_wgrad_rows", R, K, N, src.data_ptr(), ld, asc, ash, Gptr, Ys[l].data_ptr(),
                          p.data_ptr(), q.data_ptr(), t.data_ptr(), gp, am, S, psc, psh, scratch.data_ptr(),
                          dW.data_ptr(), db.data_ptr(), rref)
            grads[6 * l + 0] = dW
            grads[6 * l + 1] = db
            if l > 0 or ctx.needs_input_grad[0]:
                Wt = _f32((N, K), dev)
                _lib.call("pcops_mlp_transpose", K, N, Ws[l].data_ptr(), Wt.data_ptr())
                Gprev = None if xyz_prev else _f32((R, K), dev)
                if xyz_prev:     # the first layer's masked gradient is reduced in the epilogue, never written
                    P = lib.pcops_mlp_stats_rows(R)
                    part = _f32((P, 2, K), dev)
                    xstats = _f32((P, 3, K), dev)
                    _lib.call("pcops_mlp_gemm_dgrad_xyz_rows", R, N, K, Gptr, Ys[l].data_ptr(), p.data_ptr(), q.data_ptr(),
                              t.data_ptr(), gp, am, S, psc, psh, Wt.data_ptr(), off4.data_ptr(), xyzw.data_ptr(),
                              scales[0].data_ptr(), shifts[0].data_ptr(), None, part.data_ptr(), xstats.data_ptr(),
                              rref)
                elif l > 0:
                    P = lib.pcops_mlp_stats_rows(R)
                    part = _f32((P, 2, K), dev)
                    _lib.call("pcops_mlp_gemm_dgrad_rows", R, N, K, Gptr, Ys[l].data_ptr(), p.data_ptr(), q.data_ptr(),
                              t.data_ptr(), gp, am, S, psc, psh, Wt.data_ptr(), Ys[l - 1].data_ptr(),
                              scales[l - 1].data_ptr(), shifts[l - 1].data_ptr(), Gprev.data_ptr(),
                              part.data_ptr(), rref)
                else:
                    _lib.call("pcops_mlp_gemm_dgrad", R, N, K, Gptr, Ys[l].data_ptr(), p.data_ptr(), q.data_ptr(),
                              t.data_ptr(), gp, am, S, psc, psh, Wt.data_ptr(), None, None, None,
                              Gprev.data_ptr(), None)
                    d0 = Gprev
                Gm = Gprev

        out = [d0 if ctx.needs_input_grad[0] else None, d1, None, None, None, dwxyz, dbias,
               None, None, None, None, None, None, None, None]
        for i in range(L):
            out.extend(grads[6 * i:6 * i + 4])
            out.extend([None, None])
        return tuple(out)


def _pool_top_ok(lib, R, K, N, S, l, gather, K0, rows, xyz_prev, has_bias):
    """the pooled top layer l takes the algebraic backward (pcops.h): K x K products instead of K x N, no use of Y"""
    return bool(POOL_TOP and rows is None and not xyz_prev and S >= 64 and N >= 2 * K
                and (l > 0 or (not gather and K0 == K)) and has_bias
                and lib.pcops_mlp_pool_top_supported(R, K, N, S))


def _pool_top_backward(R, K, N, S, W, b, p, q, t, grad_out, ysel, argmax, sc, sh, Yprev, psc, psh, grads, l, dev,
                       need_dx=True):
    """dW, db (into grads) and the masked data gradient + its statistics of a pooled top layer, from the Kp x Kp
    products of pcops.h's algebraic form.  Returns (Gprev, stats_partial)."""
    lib = _lib.load()
    Wt = _f32((N, K), dev)
    if TAIL_FOLD:       # W^T, W diag(q) and q.b + t out of one launch
        Wq, u = _f32((K, N), dev), _f32(N, dev)
        v = _f32(K, dev) if need_dx else None           # ... and v = W u
        _lib.call("pcops_mlp_pool_top_prep", K, N, W.data_ptr(), b.data_ptr(), q.data_ptr(), t.data_ptr(), Wt.data_ptr(),
                  Wq.data_ptr(), u.data_ptr(), _p(v))
    else:
        _lib.call("pcops_mlp_transpose", K, N, W.data_ptr(), Wt.data_ptr())
        Wq = W * q[:N]                                  # W diag(q)
        u = torch.addcmul(t[:N], q[:N], b)              # q.b + t
    Gprev = part = None
    # the Gram matrix of the input first: with it the layer's two K-sized products (W diag(q) W^T for the data gradient, gram W diag(q)
    # for the weight gradient) are independent of everything else and leave in ONE launch
    splits = lib.pcops_mlp_wgrad_splits(R, K, K)
    scratch = _f32(splits * (K * K + K), dev)
    gram, xsum = _f32((K, K), dev), _f32(K, dev)
    _lib.call("pcops_mlp_gram", R, K, Yprev.data_ptr(), K, _p(psc), _p(psh), scratch.data_ptr(),
              gram.data_ptr(), xsum.data_ptr())
    dW = _f32((K, N), dev)
    paired = TAIL_FOLD and need_dx
    if paired:
        Mq = _f32((K, K), dev)
        _lib.small_gemm_pair((K, N, K, Wq.data_ptr(), N, 0, Wt.data_ptr(), K, 0, None, Mq.data_ptr(), K, None),
                             (K, K, N, gram.data_ptr(), K, 0, Wq.data_ptr(), N, 0, None, dW.data_ptr(), N, None))
    if need_dx:
        if not paired:
            Mq = _f32((K, K), dev)
            _lib.call("pcops_small_gemm", K, N, K, Wq.data_ptr(), N, Wt.data_ptr(), K, Mq.data_ptr(), K)
        if not TAIL_FOLD:
            v = _f32(K, dev)
            _lib.call("pcops_small_gemm", 1, N, K, u.data_ptr(), N, Wt.data_ptr(), K, v.data_ptr(), K)
        G = R // S
        addend = _f32((G * min(S, N), K), dev)
        rowmap = torch.empty(R, dtype=torch.int32, device=dev)
        _lib.call("pcops_mlp_pool_top_addend", R, K, N, S, grad_out.data_ptr(), ysel.data_ptr(), argmax.data_ptr(),
                  sc.data_ptr(), sh.data_ptr(), p.data_ptr(), Wt.data_ptr(), addend.data_ptr(), rowmap.data_ptr())
        Gprev = _f32((R, K), dev)
        part = _f32((lib.pcops_mlp_stats_rows(R), 2, K), dev) if psc is not None else None
        _lib.call("pcops_mlp_gemm_dgrad_top", R, K, Yprev.data_ptr(), _p(psc), _p(psh), Mq.data_ptr(),
                  v.data_ptr(), addend.data_ptr(), addend.shape[0], rowmap.data_ptr(), Gprev.data_ptr(), _p(part))
    # weight gradient
    Ssp, cfsum = _f32((K, N), dev), _f32(N, dev)
    _lib.call("pcops_mlp_pool_top_wsparse", R, K, N, S, grad_out.data_ptr(), ysel.data_ptr(), argmax.data_ptr(),
              sc.data_ptr(), sh.data_ptr(), p.data_ptr(), Yprev.data_ptr(), _p(psc), _p(psh),
              Ssp.data_ptr(), cfsum.data_ptr())
    if not paired:
        _lib.call("pcops_small_gemm", K, K, N, gram.data_ptr(), K, Wq.data_ptr(), N, dW.data_ptr(), N)
    if TAIL_FOLD:       # (dW + Ssp) + xsum u^T in place, db = (cfsum + q.(xsum^T W + R b)) + R t: one launch for eleven
        db = _f32(N, dev)
        _lib.call("pcops_mlp_pool_top_finish", K, N, R, dW.data_ptr(), Ssp.data_ptr(), xsum.data_ptr(), u.data_ptr(),
                  cfsum.data_ptr(), q.data_ptr(), W.data_ptr(), b.data_ptr(), t.data_ptr(), db.data_ptr())
        grads[6 * l + 0] = dW
        grads[6 * l + 1] = db
    else:
        xw = _f32(N, dev)
        _lib.call("pcops_small_gemm", 1, K, N, xsum.data_ptr(), K, W.data_ptr(), N, xw.data_ptr(), N)
        grads[6 * l + 0] = torch.addr(dW.add_(Ssp), xsum, u)
        grads[6 * l + 1] = cfsum + q[:N] * (xw + float(R) * b) + float(R) * t[:N]
    return Gprev, part


class _SmallLinear(torch.autograd.Function):
    """Y = X W + b for a few hundred rows (the classifier head: B x 1024 -> 512 -> 256 -> classes) on
    pcops_small_gemm_ex -- 32 x 32 output tiles with a 4-way K split fill the chip where a library GEMM picks a
    256 x 256 tile and runs on one or two CUs; dX = dY W^T and dW = X^T dY read the operands transposed in place."""

    @staticmethod
    def forward(ctx, x, w, b):
        R, K = x.shape
        N = w.shape[1]
        y = torch.empty((R, N), dtype=torch.float32, device=x.device)
        _lib.call("pcops_small_gemm_ex", R, K, N, x.data_ptr(), K, 0, w.data_ptr(), N, 0, _p(b), y.data_ptr(), N)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable      # a second derivative through this node is an error, not silence
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        R, K = x.shape
        N = w.shape[1]
        gy = gy.contiguous()
        dx = dw = db = None
        want_db = ctx.has_bias and ctx.needs_input_grad[2]
        if TAIL_FOLD and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            # dX = dY W^T and dW = X^T dY (+ db) share dY and nothing else: one launch, both halves of the chip busy
            dx = torch.empty((R, K), dtype=torch.float32, device=x.device)
            dw = torch.empty((K, N), dtype=torch.float32, device=x.device)
            db = torch.empty(N, dtype=torch.float32, device=x.device) if want_db else None
            _lib.small_gemm_pair((R, N, K, gy.data_ptr(), N, 0, w.data_ptr(), N, 1, None, dx.data_ptr(), K, None),
                                 (K, R, N, x.data_ptr(), K, 1, gy.data_ptr(), N, 0, None, dw.data_ptr(), N, _p(db)))
            return dx, dw, db
        if ctx.needs_input_grad[0]:
            dx = torch.empty((R, K), dtype=torch.float32, device=x.device)
            _lib.call("pcops_small_gemm_ex", R, N, K, gy.data_ptr(), N, 0, w.data_ptr(), N, 1, None, dx.data_ptr(), K)
        if ctx.needs_input_grad[1]:
            dw = torch.empty((K, N), dtype=torch.float32, device=x.device)
            if want_db and TAIL_FOLD:     # db = 1^T dY out of the dW = X^T dY launch
                db = torch.empty(N, dtype=torch.float32, device=x.device)
            _lib.call("pcops_small_gemm_colsum", K, R, N, x.data_ptr(), K, 1, gy.data_ptr(), N, 0, None, dw.data_ptr(), N,
                      _p(db))
        if want_db and db is None:
            db = gy.sum(dim=0)
        return dx, dw, db


class _SplitRows(torch.autograd.Function):
    """(w[:k], w[k:]) of a 2-D weight whose gradient comes back as ONE concatenation -- autograd's own slices answer with a
    zero-filled full-size tensor per half plus their sum (five launches where this is one)."""

    @staticmethod
    def forward(ctx, w, k):
        ctx.k, ctx.rows = int(k), w.shape[0]
        return w[:k], w[k:]

    @staticmethod
    def backward(ctx, ga, gb):
        k, n = ctx.k, ctx.rows
        if ga is None and gb is None:
            return None, None
        like = ga if ga is not None else gb
        if ga is None:
            ga = like.new_zeros((k,) + tuple(like.shape[1:]))
        if gb is None:
            gb = like.new_zeros((n - k,) + tuple(like.shape[1:]))
        return torch.cat([ga, gb], dim=0), None


def split_rows(w, k):
    """the first k rows of w and the rest (views), differentiable"""
    if not TAIL_FOLD or not w.requires_grad:
        return w[:k], w[k:]
    return _SplitRows.apply(w, int(k))


class _SoftmaxCE(torch.autograd.Function):
    """mean softmax cross entropy (+ label smoothing) of a batch of logits: loss and gradient out of one launch
    (csrc/head.hip, pcops_softmax_ce); backward scales the saved gradient by the upstream factor."""

    @staticmethod
    def forward(ctx, logits, labels, smoothing):
        R, C = logits.shape
        blocks = int(_lib.load().pcops_softmax_ce_blocks(R))
        loss = torch.empty(blocks, dtype=torch.float32, device=logits.device)
        dl = torch.empty_like(logits)
        _lib.call("pcops_softmax_ce", R, C, logits.data_ptr(), labels.data_ptr(), float(smoothing), loss.data_ptr(),
                  dl.data_ptr())
        ctx.save_for_backward(dl)
        return loss.view(()) if blocks == 1 else loss.sum()

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return dl * g, None, None


def softmax_cross_entropy(logits, labels, label_smoothing=0.0):
    """F.cross_entropy(logits, labels.long(), label_smoothing=...) with mean reduction; on the device one launch per direction
    (+ one to add up the workgroups' shares beyond 4096 rows; torch: 6 launches, 26 with smoothing)"""
    R, C = logits.shape
    if TAIL_FOLD and logits.is_cuda and logits.dtype == torch.float32 and 1 <= R and R * C < 2 ** 31 and 1 <= C <= 4096:
        lab = labels if labels.dtype == torch.int32 else labels.to(torch.int32)
        return _SoftmaxCE.apply(logits.contiguous(), lab.contiguous(), float(label_smoothing))
    import torch.nn.functional as F
    return F.cross_entropy(logits, labels.long(), label_smoothing=float(label_smoothing))


def small_linear(x, w, b):
    """(rows, K) @ (K, N) + b on the small-GEMM kernel, differentiable (rows up to a few thousand)"""
    return _SmallLinear.apply(x.contiguous(), w.contiguous(), b)


class _FcBatchNorm(torch.autograd.Function):
    """BatchNorm (+ ReLU) behind a fully connected layer -- a few hundred rows -- as one launch per direction
    (csrc/head.hip, pcops_fc_bn_fwd / _bwd) instead of F.batch_norm + relu or a dozen elementwise / reduce launches.
    apply(x, gamma, beta, moving_mean, moving_var, training, decay, eps, unbiased_moving_var, relu) -> y"""

    @staticmethod
    def forward(ctx, x, gamma, beta, mm, mv, training, decay, eps, unbiased, relu):
        R, C = x.shape
        dev = x.device
        y = _f32((R, C), dev)
        stat = _f32((2, C), dev)
        _lib.call("pcops_fc_bn_fwd", R, C, x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                  int(bool(training)), float(decay), float(eps), int(bool(unbiased)), int(bool(relu)), y.data_ptr(),
                  stat[0].data_ptr(), stat[1].data_ptr())
        ctx.save_for_backward(x, y, gamma, stat)
        ctx.flags = (bool(training), bool(relu))
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dy):
        x, y, gamma, stat = ctx.saved_tensors
        training, relu = ctx.flags
        R, C = x.shape
        dev = x.device
        dy = dy.contiguous()
        dx = _f32((R, C), dev)
        dgb = _f32((2, C), dev)
        _lib.call("pcops_fc_bn_bwd", R, C, dy.data_ptr(), x.data_ptr(), y.data_ptr(), gamma.data_ptr(), stat[0].data_ptr(),
                  stat[1].data_ptr(), int(training), int(relu), dx.data_ptr(), dgb[0].data_ptr(), dgb[1].data_ptr())
        return dx, dgb[0], dgb[1], None, None, None, None, None, None, None


FC_BN = os.environ.get("PCOPS_FC_BN", "1") != "0"
FC_BN_MAX_ROWS = 8192


def fc_batch_norm_supported(x):
    return (FC_BN and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 1 <= x.shape[0] <= FC_BN_MAX_ROWS
            and not _dist.sync_bn_active())


def fc_batch_norm(x, gamma, beta, mm, mv, training, decay, eps, unbiased_moving_var, relu):
    """(rows, C) -> BN (+ ReLU) with the moving statistics updated in place when training"""
    y = _FcBatchNorm.apply(x.contiguous(), gamma, beta, mm, mv, bool(training), float(decay), float(eps),
                           bool(unbiased_moving_var), bool(relu))
    if TRACE is not None and relu:      # parity tests (tests/decisions.py): the ReLU decision of this layer, as torch.relu's
        from .graph import get_default_graph
        TRACE.append(("relu", get_default_graph().full_name("")[:-1], y.detach()))
    return y


class _EdgeWeights(torch.autograd.Function):
    """apply(W1 (2 c, cp), b1 (cp) or None, kp) -> Wcat (kp, 2 cp) = [W_b | W_a - W_b] (rows >= c zero), bcat (2 cp) = [0 | b1]:
    the concatenated weight of the [Q | Ctr] form as one launch per direction (pcops_edge_weights_fwd / _bwd)"""

    @staticmethod
    def forward(ctx, w1, b1, kp):
        c, cp = w1.shape[0] // 2, w1.shape[1]
        dev = w1.device
        wcat, bcat = _f32((kp, 2 * cp), dev), _f32(2 * cp, dev)
        _lib.call("pcops_edge_weights_fwd", c, cp, int(kp), w1.data_ptr(), _p(b1), wcat.data_ptr(), bcat.data_ptr())
        ctx.dims = (c, cp, b1 is not None)
        return wcat, bcat

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dwcat, dbcat):
        c, cp, has_b = ctx.dims
        dev = dwcat.device
        dwcat = dwcat.contiguous()
        dw1 = _f32((2 * c, cp), dev)
        db1 = _f32(cp, dev) if has_b else None
        _lib.call("pcops_edge_weights_bwd", c, cp, dwcat.data_ptr(), dbcat.contiguous().data_ptr() if has_b else None,
                  dw1.data_ptr(), _p(db1))
        return dw1, db1, None


def edge_weights(w1, b1, kp):
    return _EdgeWeights.apply(w1.contiguous(), b1, int(kp))


_COEF = {}


def _unit_coef(n4, dev):
    """(p, q, t) = (1, 0, 0) for pcops_mlp_wgrad (dY = G): a constant, built once per width and device"""
    key = (n4, str(dev))
    v = _COEF.get(key)
    if v is None:
        v = torch.zeros(3 * n4, dtype=torch.float32, device=dev)
        v[:n4] = 1.0
        _COEF[key] = v
    return v


class _RowsLinear(torch.autograd.Function):
    """Y = X W + b on (rows, K) through the libpcops GEMMs, backward included.  Exists for the per-source-point
    contraction of a grouped first layer (Q = points W_f + b): rows = B*N is large and the weight gradient is a
    (K, N) = (128, 128)-sized reduction over all rows, a shape the library GEMM runs on a handful of CUs."""

    @staticmethod
    def forward(ctx, x, w, b):
        R, K = x.shape
        N = w.shape[1]
        y = _f32((R, N), x.device)
        _lib.call("pcops_mlp_gemm_fwd", R, K, N, x.data_ptr(), K, None, None, w.data_ptr(), _p(b), y.data_ptr(), None, None)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return y

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        R, K = x.shape
        N = w.shape[1]
        dev = g.device
        g = g.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = _f32((N, K), dev)
            _lib.call("pcops_mlp_transpose", K, N, w.data_ptr(), wt.data_ptr())
            dx = _f32((R, K), dev)
            _lib.call("pcops_mlp_gemm_fwd", R, N, K, g.data_ptr(), N, None, None, wt.data_ptr(), None, dx.data_ptr(), None, None)
        if ctx.needs_input_grad[1] or ctx.has_bias:
            n4 = (N + 3) // 4 * 4
            coef = _unit_coef(n4, dev)                                      # p = 1, q = 0, t = 0: dY = G
            scratch = _f32(lib.pcops_mlp_wgrad_splits(R, K, N) * (K * N + N), dev)
            dw, db = _f32((K, N), dev), _f32(N, dev)
            _lib.call("pcops_mlp_wgrad", R, K, N, x.data_ptr(), K, None, None, g.data_ptr(), g.data_ptr(),
                      coef.data_ptr(), coef[n4:].data_ptr(), coef[2 * n4:].data_ptr(), None, None, 1, None, None,
                      scratch.data_ptr(), dw.data_ptr(), db.data_ptr())
            if not ctx.has_bias:
                db = None
        return dx, dw, db


def rows_linear(x2d, w, b=None):
    """x2d (R, K) @ w (K, N) + b through libpcops (forward and backward); K % 4 == 0 and N % 4 == 0"""
    return _RowsLinear.apply(x2d.contiguous(), w.contiguous(), b)


def _alias(buf, col, width):
    """an UNTRACKED alias of buf[..., col:col + width]: same storage, no autograd view relation (the kernels write through
    raw pointers; a tracked view of a buffer other nodes also write into would trip the version counter)"""
    t = torch.empty(0, dtype=buf.dtype, device=buf.device)
    return t.set_(buf.untyped_storage(), buf.storage_offset() + col, tuple(buf.shape[:-1]) + (width,), buf.stride())


class CatBuffer:
    """the (..., C_total) tensor several layers store their column blocks into on their way out (DGCNN's concatenation of
    the four EdgeConv outputs, dgcnn.py:83) -- see pcops_edge_pool_out_ld2"""

    def __init__(self, shape, device):
        self.buf = torch.empty(shape, dtype=torch.float32, device=device)


class _CatAssemble(torch.autograd.Function):
    """apply(cat, *slices) -> the assembled tensor.  Forward: nothing to do (the blocks were stored by their producers);
    backward: the column blocks of the gradient, as strided views"""

    @staticmethod
    def forward(ctx, cat, *slices):
        base = cat.buf.storage_offset()
        ctx.spans = [(s.storage_offset() - base, s.shape[-1], tuple(s.shape)) for s in slices]
        assert sum(w for _, w, _ in ctx.spans) == cat.buf.shape[-1]
        return _alias(cat.buf, 0, cat.buf.shape[-1])

    @staticmethod
    def backward(ctx, g):
        g2 = g.reshape(-1, g.shape[-1])                  # (rows, C_total): a view of a contiguous gradient
        return (None,) + tuple(g2[:, c:c + w].reshape(shape[:-1] + (w,)) if g2[:, c:c + w].shape != shape else g2[:, c:c + w]
                               for c, w, shape in ctx.spans)


def cat_assemble(cat, slices):
    return _CatAssemble.apply(cat, *slices)


def _row_stride(g):
    """floats between consecutive rows of g viewed as (rows, C) when that view exists without a copy (unit stride along C, the
    leading dimensions collapse to one uniform row stride, 16-byte alignment); None otherwise"""
    if g.dtype != torch.float32 or g.dim() < 2 or g.stride(-1) != 1 or g.data_ptr() % 16:
        return None
    ld = span = None
    for d in range(g.dim() - 2, -1, -1):
        if g.shape[d] == 1:
            continue                                        # (a dimension of one element has no say)
        if ld is None:
            ld, span = g.stride(d), g.stride(d) * g.shape[d]
        elif g.stride(d) != span:
            return None
        else:
            span *= g.shape[d]
    ld = g.shape[-1] if ld is None else ld
    return int(ld) if (ld >= g.shape[-1] and ld % 4 == 0) else None


class EdgeConvPool(torch.autograd.Function):
    """apply(Q, Ctr, idx, gamma, beta, mm, mv, training, decay, eps, unbiased[, cat, col]) -> (B*M, C)
    [cat (a CatBuffer over (B, M, C_total)) and col: the output is ALSO stored as columns col..col+C of cat.buf, and a second
    output aliases that block -- for fused_mlp.cat_assemble]
    One pooled layer  y = Q[idx] + Ctr -> BN -> ReLU -> max over the neighbours  without the (B,M,S,C) tensor in either
    direction (csrc/gather.hip, edge_pool_*): the statistics, the pooled value and both gradients are functions of
    per-group sums / extrema of the gathered Q rows."""

    @staticmethod
    def forward(ctx, Q, Ctr, idx, gamma, beta, mm, mv, training, decay, eps, unbiased, cat=None, col=0):
        lib = _lib.load()
        ctx.set_materialize_grads(False)
        B, M, S = idx.shape
        qc = Ctr is None            # Q is the (B, N, 2 C) product [Q | Ctr] of ONE GEMM (pcops.h "[Q | Ctr] forms")
        Nsrc, C = Q.shape[1], (Q.shape[2] // 2 if qc else Q.shape[2])
        dev = Q.device
        G = B * M
        need_grad = any(ctx.needs_input_grad)
        sync = training and _dist.sync_bn_active()
        SQ, qsel = _f32((G, C), dev), _f32((G, C), dev)
        arg = torch.empty((G, C), dtype=torch.uint8, device=dev)
        P = lib.pcops_edge_pool_fwd_stats_rows(B, Nsrc, M, S, C)
        part = _f32((P, 2, C), dev) if training else None
        piv0 = mm.data_ptr() if (training and STAT_PIVOT) else None          # shifted moments around the moving mean
        if qc:
            assert Q.is_contiguous() and M == Nsrc
            _lib.call("pcops_edge_pool_fwd_ld", B, Nsrc, M, S, C, Q.data_ptr(), 2 * C, Q.data_ptr() + 4 * C, 2 * C,
                      idx.data_ptr(), gamma.data_ptr(), SQ.data_ptr(), qsel.data_ptr(), arg.data_ptr(), _p(part), piv0)
        else:
            _lib.call("pcops_edge_pool_fwd", B, Nsrc, M, S, C, Q.data_ptr(), Ctr.data_ptr(), idx.data_ptr(),
                      gamma.data_ptr(), SQ.data_ptr(), qsel.data_ptr(), arg.data_ptr(), _p(part), piv0)
        vecs = _VecArena([C], 4, dev)
        scale, shift = vecs.take(C), vecs.take(C)
        mean = rstd = None
        if training:
            mean, rstd = vecs.take(C), vecs.take(C)
            ws = _workspace(C, dev)
            Pf, Rf, piv_fin = P, G * S, (mm.data_ptr() if STAT_PIVOT else None)
            if sync:
                part, Rf = _dist.allreduce_stat_partials(part, G * S, mm if STAT_PIVOT else None)
                Pf, piv_fin = part.shape[0], None
            _lib.call("pcops_mlp_bn_finalize", Pf, C, Rf, part.data_ptr(), piv_fin,
                      ws.data_ptr(), gamma.data_ptr(),
                      beta.data_ptr(), float(eps), float(decay), int(unbiased), mm.data_ptr(), mv.data_ptr(),
                      mean.data_ptr(), rstd.data_ptr(), scale.data_ptr(), shift.data_ptr())
        else:
            _lib.call("pcops_mlp_bn_eval_coeffs", C, gamma.data_ptr(), beta.data_ptr(), mm.data_ptr(), mv.data_ptr(),
                      float(eps), scale.data_ptr(), shift.data_ptr())
            if need_grad:
                mean, rstd = mm.detach().clone(), torch.rsqrt(mv.detach() + float(eps))
        out = _f32((G, C), dev)
        ysel = _f32((G, C), dev) if (training or need_grad) else None
        sl = None
        if qc and cat is not None:
            sl = _alias(cat.buf.view(G, -1), int(col), C)
            _lib.call("pcops_edge_pool_out_ld2", G, C, qsel.data_ptr(), Q.data_ptr() + 4 * C, 2 * C, scale.data_ptr(),
                      shift.data_ptr(), out.data_ptr(), _p(ysel), sl.data_ptr(), cat.buf.shape[-1])
        elif qc:
            _lib.call("pcops_edge_pool_out_ld", G, C, qsel.data_ptr(), Q.data_ptr() + 4 * C, 2 * C, scale.data_ptr(),
                      shift.data_ptr(), out.data_ptr(), _p(ysel))
        else:
            _lib.call("pcops_edge_pool_out", G, C, qsel.data_ptr(), Ctr.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                      out.data_ptr(), _p(ysel))
        if training or need_grad:
            ctx.saved = (Q, Ctr, idx, gamma, SQ, arg, ysel, mean, rstd, scale, shift)
            ctx.flags = (bool(training), bool(sync))
            if TRACE is not None:
                TRACE.append(ctx)
        ctx.nout = 2 if sl is not None else 1
        return (out, sl) if sl is not None else out

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        # the gradient of the dense output and (cat form) of the block stored into the concatenation
        given = [g for g in grads if g is not None]
        assert given, "EdgeConvPool.backward without any gradient"
        # two pieces, or one that is a column block of a wider tensor: added / gathered inside the statistics pass below
        lds_ = [_row_stride(g) for g in given]
        fold = TAIL_FOLD and all(l is not None for l in lds_) and (len(given) == 2 or lds_[0] != given[0].shape[-1])
        grad_out = None if fold else (given[0] if len(given) == 1 else given[0] + given[1])
        Q, Ctr, idx, gamma, SQ, arg, ysel, mean, rstd, scale, shift = ctx.saved
        training, sync = ctx.flags
        B, M, S = idx.shape
        qc = Ctr is None
        Nsrc, C = Q.shape[1], (Q.shape[2] // 2 if qc else Q.shape[2])
        dev = Q.device
        G = B * M
        P = lib.pcops_mlp_bwd_pool_stats_rows(G)
        part = _f32((P, 2, C), dev)
        if fold:
            grad_out = _f32((G, C), dev)
            gb = given[1] if len(given) == 2 else None
            _lib.call("pcops_mlp_pool_bwd_stats_sum", G, C, given[0].data_ptr(), lds_[0], _p(gb), lds_[1] if gb is not None else 0,
                      ysel.data_ptr(), scale.data_ptr(), shift.data_ptr(), part.data_ptr(), grad_out.data_ptr())
        else:
            grad_out = grad_out.contiguous()
            _lib.call("pcops_mlp_pool_bwd_stats", G, C, grad_out.data_ptr(), ysel.data_ptr(), scale.data_ptr(),
                      shift.data_ptr(), part.data_ptr(), None)
        vecs = _VecArena([C], 3, dev)
        p, q, t = vecs.take(C), vecs.take(C), vecs.take(C)
        dgamma, dbeta = _f32(C, dev), _f32(C, dev)
        ws = _workspace(C, dev)
        _lib.call("pcops_mlp_bn_bwd_coeffs", P, C, G * S, part.data_ptr(), ws.data_ptr(), gamma.data_ptr(),
                  mean.data_ptr(), rstd.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), p.data_ptr(), q.data_ptr(),
                  t.data_ptr())
        if sync:        # see FusedMLPStack.backward
            gpart, Rg = _dist.allreduce_stat_partials(part, G * S)
            junk = _f32(2 * C, dev)
            _lib.call("pcops_mlp_bn_bwd_coeffs", gpart.shape[0], C, Rg, gpart.data_ptr(), ws.data_ptr(), gamma.data_ptr(),
                      mean.data_ptr(), rstd.data_ptr(), junk.data_ptr(), junk[C:].data_ptr(), p.data_ptr(),
                      q.data_ptr(), t.data_ptr())
        if not training:
            q.zero_()
            t.zero_()
        wsp = torch.empty(int(lib.pcops_sa_scatter_workspace_bytes(B, Nsrc, M, S)) // 4, dtype=torch.int32, device=dev)
        if qc:          # dQ and dCtr: the column halves of ONE gradient of the [Q | Ctr] product
            dQC = _f32((B, Nsrc, 2 * C), dev)
            _lib.call("pcops_edge_pool_bwd_ld", B, Nsrc, M, S, C, Q.data_ptr(), 2 * C, Q.data_ptr() + 4 * C, 2 * C,
                      idx.data_ptr(), grad_out.data_ptr(), ysel.data_ptr(), SQ.data_ptr(), arg.data_ptr(), scale.data_ptr(),
                      shift.data_ptr(), p.data_ptr(), q.data_ptr(), t.data_ptr(), dQC.data_ptr(), 2 * C,
                      dQC.data_ptr() + 4 * C, 2 * C, wsp.data_ptr())
            return dQC, None, None, dgamma, dbeta, None, None, None, None, None, None, None, None
        dQ, dCtr = _f32((B, Nsrc, C), dev), _f32((B, M, C), dev)
        _lib.call("pcops_edge_pool_bwd", B, Nsrc, M, S, C, Q.data_ptr(), Ctr.data_ptr(), idx.data_ptr(),
                  grad_out.data_ptr(), ysel.data_ptr(), SQ.data_ptr(), arg.data_ptr(), scale.data_ptr(), shift.data_ptr(),
                  p.data_ptr(), q.data_ptr(), t.data_ptr(), dQ.data_ptr(), dCtr.data_ptr(), wsp.data_ptr())
        return dQ, dCtr, None, dgamma, dbeta, None, None, None, None, None, None, None, None


def edge_conv_pool_supported(C, S):
    return C in (32, 64, 128) or C % 256 == 0 and S <= 256


def fused_supported(x, widths, bn, activation_relu):
    if not (bn and activation_relu and x.is_cuda and x.dtype == torch.float32):
        return False
    if any(w % 32 != 0 for w in widths):
        return False
    return True


def _flat(layer_tensors, first_gather):
    flat = []
    for i, (w, b, gamma, beta, mm, mv) in enumerate(layer_tensors):
        if i == 0 and first_gather:
            flat.extend([None, None, gamma, beta, mm, mv])
        else:
            flat.extend([w.reshape(-1, w.shape[-1]), b, gamma, beta, mm, mv])
    return flat


def mlp_stack(x, S, pool, training, decay, eps, unbiased, layer_tensors):
    """x: (..., K0) channel-last; rows are flattened; S rows per pooling group (contiguous)."""
    x2d = x.reshape(-1, x.shape[-1]).contiguous()
    return FusedMLPStack.apply(x2d, None, None, None, None, None, None, int(S), bool(pool), bool(training),
                               float(decay), float(eps), bool(unbiased), None, len(layer_tensors),
                               *_flat(layer_tensors, False))


# Debug hook of the parity tests (tests/decisions.py): a list that receives the autograd node of every fused stack in
# forward order, so that the discrete decisions the kernels took (ReLU masks from the raw layer outputs and BN
# coefficients the node keeps anyway, arg-max rows of the pooled layer) can be read back.  None: nothing is recorded.
TRACE = None
STAT_PIVOT = os.environ.get("PCOPS_STAT_PIVOT", "1") != "0"   # BN statistics as shifted moments around the moving mean
FUSE_POOL_ROWS = os.environ.get("PCOPS_FUSE_POOL_ROWS", "1") != "0"    # per-block pooled epilogue on compacted rows
BWD_FUSED = os.environ.get("PCOPS_BWD_FUSED", "1") != "0"   # one-pass data + weight gradient of narrow layers (pcops_mlp_bwd_fused)
CLOUD_BIAS = os.environ.get("PCOPS_CLOUD_BIAS", "1") != "0"     # ... its Y = Q + Ctr[cloud] and the backward as streaming passes
CLOUD_POINT = os.environ.get("PCOPS_CLOUD_POINT", "1") != "0"   # dgcnn_bga's head: per-cloud + per-point first conv without the concat
EDGE_DIRECT = os.environ.get("PCOPS_EDGE_DIRECT", "1") != "0"   # first EdgeConv layer of a stack on an input without gradient
EDGE_DIRECT_FUSED = os.environ.get("PCOPS_EDGE_DIRECT_FUSED", "1") != "0"   # ... its E^T Gm inside the one-pass backward above
POOL_TOP = os.environ.get("PCOPS_POOL_TOP", "1") != "0"     # algebraic backward of pooled top layers (fused_mlp._pool_top_backward)
# the step's short generic launches folded into their neighbours (round 6): db out of the FC head's dW launch, the algebraic top
# layer's operand / closing sums as one launch each, one concatenation for the gradient of a split weight (split_rows), the
# whole-cloud group's index built once, the smoothed cross entropy as one launch per direction; "0": the torch forms (A/B, tests)
TAIL_FOLD = os.environ.get("PCOPS_TAIL_FOLD", "1") != "0"
COMPACT_MIN_S = int(os.environ.get("PCOPS_COMPACT_MIN_S", "48"))   # group sizes from which padding is compacted; 0: never
EDGE_QC = os.environ.get("PCOPS_EDGE_QC", "1") != "0"        # EdgeConv's two per-point GEMMs as one [Q | Ctr] product


def _compactable(idx, pool, L, widths, Q, Ctr, xyz, wxyz, identity_idx):
    """ball-query padding can be left out of this stack (pcops.h "compacted rows"): max-pooled gather stack of at
    least two layers without a per-group term, wave-stream sized, group size a multiple of the 16-row block"""
    B, M, S = idx.shape
    # policy: which stacks are worth compacting (and the forms the host code below has a compacted path for)
    if not (COMPACT_MIN_S and pool and not identity_idx and Ctr is None and L >= 2 and S >= COMPACT_MIN_S):
        return False
    if B * M * S < 32768 or any(w % 32 for w in widths) or _dist.sync_bn_active():
        return False
    if Q is None and not (wxyz is not None and L >= 3):
        return False                        # coordinate-only first layer: only as the arithmetic (never stored) form
    # support: ONE answer from the library for every launch of the stack (the *_rows entry points have no fallback)
    import ctypes
    n = Q.shape[1] if Q is not None else xyz.shape[1]
    arr = (ctypes.c_int * L)(*[int(w) for w in widths])
    return bool(_lib.load().pcops_gather_stack_rows_supported(B, n, M, S, 1 if Q is not None else 0, L, arr))


def edge_qc_supported(b, n, s, c):
    """the [Q | Ctr] forms have kernels for this EdgeConv shape (pcops.h): one per-point GEMM instead of two"""
    return bool(EDGE_QC and _lib.load().pcops_edge_ld_supported(int(b), int(n), int(n), int(s), int(c))
                and not _dist.sync_bn_active())


def edge_direct_supported(b, n, k, c_in, c1, n_layers, x):
    """the first EdgeConv layer's weight gradient without a scatter (pcops.h pcops_edge_first_*): 3-channel input that
    needs no gradient, a stack of at least two layers on the one-GEMM path"""
    if not (EDGE_DIRECT and c_in == 3 and n_layers >= 2) or (torch.is_grad_enabled() and x.requires_grad):
        return False
    return bool(_lib.load().pcops_edge_first_supported(b, n, n, k, c1))


def gather_mlp_stack(idx, pool, training, decay, eps, unbiased, layer_tensors, Q=None, Ctr=None, xyz=None,
                     new_xyz=None, wxyz=None, bias=None, identity_idx=False, pts_cnt=None, QC=None, cat_slot=None,
                     direct=None):
    """Grouped stack whose first conv was applied before the grouping:
         Y1[b,j,s,:] = Q[b,idx] + Ctr[b,j] + (xyz[b,idx] - new_xyz[b,j]) wxyz + bias     (terms optional)
    idx (B,M,S) int32, Q (B,N,C1), Ctr (B,M,C1), xyz (B,N,3), new_xyz (B,M,3), wxyz (3,C1), bias (C1);
    layer_tensors[0] supplies only the BN variables of layer 1.  identity_idx: the caller guarantees
    idx[b, 0, s] = s with M = 1 and S = N (group_all), which turns the backward scatter into a reshape.
    Returns (B*M, C_L) if pool else (B*M*S, C_L)."""
    c = lambda t: t.contiguous() if t is not None else None   # noqa: E731
    S = idx.shape[2]
    if QC is not None:
        # QC (B, N, 2 C1) = [Q | Ctr], the product of the layer's input with the concatenated weight (edge_qc_supported)
        assert Q is None and Ctr is None and xyz is None and wxyz is None and bias is None
        if len(layer_tensors) == 1 and pool:
            _w, _b, gamma, beta, mm, mv = layer_tensors[0]
            if cat_slot is not None:         # -> (out, the alias of its block in cat_slot[0].buf)
                return EdgeConvPool.apply(QC.contiguous(), None, idx.contiguous(), gamma, beta, mm, mv, bool(training),
                                          float(decay), float(eps), bool(unbiased), cat_slot[0], int(cat_slot[1]))
            return EdgeConvPool.apply(QC.contiguous(), None, idx.contiguous(), gamma, beta, mm, mv, bool(training),
                                      float(decay), float(eps), bool(unbiased))
        if direct is not None:
            # direct = (x (B, N, 3) without gradient, w1 (6, C1), b1): QC was computed from them OUTSIDE autograd; the layer's
            # weight gradient goes straight to w1 / b1 (edge_direct_supported)
            x3, w1, b1 = direct
            assert not QC.requires_grad and not x3.requires_grad and len(layer_tensors) >= 2
            return FusedMLPStack.apply(QC.contiguous(), None, idx.contiguous(), x3.contiguous(), None, w1.contiguous(), b1,
                                       int(S), int(bool(pool)) | 4 | 8, bool(training), float(decay), float(eps),
                                       bool(unbiased), None, len(layer_tensors), *_flat(layer_tensors, True))
        return FusedMLPStack.apply(QC.contiguous(), None, idx.contiguous(), None, None, None, None, int(S),
                                   int(bool(pool)) | 4, bool(training), float(decay), float(eps), bool(unbiased), None,
                                   len(layer_tensors), *_flat(layer_tensors, True))
    if torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in (xyz, new_xyz)):
        # the fused path has no d(loss)/d(coordinates) (the reference has one through GroupPoint / GatherPoint and
        # the centring subtraction): the callers take the unfused path then -- never detach silently
        raise RuntimeError("gather_mlp_stack: xyz / new_xyz require grad; use the unfused grouped path "
                           "(pointnet_util falls back to it automatically)")
    if (len(layer_tensors) == 1 and pool and Q is not None and Ctr is not None and xyz is None and wxyz is None
            and bias is None and S <= 256 and edge_conv_pool_supported(Q.shape[-1], S)
            and 256 % (Q.shape[-1] // 4) == 0 and Q.shape[1] <= 16384):
        _w, _b, gamma, beta, mm, mv = layer_tensors[0]
        return EdgeConvPool.apply(c(Q), c(Ctr), idx.contiguous(), gamma, beta, mm, mv, bool(training), float(decay),
                                  float(eps), bool(unbiased))
    rows = None
    if pts_cnt is not None and _compactable(idx, pool, len(layer_tensors), [l[2].shape[0] for l in layer_tensors],
                                            Q, Ctr, xyz, wxyz, identity_idx):
        rows = _lib.Rows(pts_cnt.contiguous(), S)
    return FusedMLPStack.apply(c(Q), c(Ctr), idx.contiguous(), c(xyz.detach()) if xyz is not None else None,
                               c(new_xyz.detach()) if new_xyz is not None else None, c(wxyz), c(bias), int(S),
                               int(bool(pool)) | (2 if identity_idx else 0), bool(training), float(decay), float(eps),
                               bool(unbiased), rows, len(layer_tensors), *_flat(layer_tensors, True))

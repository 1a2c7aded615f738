#!/usr/bin/env python3
"""How accurate is a dot product whose fp32 operands are split into bf16 pieces and multiplied on the 16-bit matrix pipe
?  CPU experiment, NumPy: operands ~ the 1x1-conv layers' (post-ReLU activations x Xavier weights),
K = 64 ... 512.  Products of bf16 pieces are exact in fp32 (8 x 8 bits); the matrix pipe accumulates in fp32, modelled here
as an fp32 chain over k per partial product and an fp32 sum of the partials (largest last).  Reported: RMS error of
  fp32 fmaf chain (what v_mfma_f32_32x32x2_f32 does) | bf16x3 with 6 products | bf16x3 with all 9 | bf16x2 with 3
relative to the RMS of the exact result, against float64."""
import numpy as np


def bf16(x):
    """round-to-nearest-even to bfloat16, returned as float32"""
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32)


def split(x, parts):
    out, rest = [], x.astype(np.float32)
    for _ in range(parts):
        p = bf16(rest)
        out.append(p)
        rest = (rest - p).astype(np.float32)
    return out


def chain32(a, b):
    """fp32 fmaf chain over k (a: (M,K), b: (K,N)) -- float64 product rounded to fp32 per step models the fused rounding"""
    acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for k in range(a.shape[1]):
        acc = (acc.astype(np.float64) + a[:, k:k + 1].astype(np.float64) * b[k:k + 1, :].astype(np.float64)).astype(np.float32)
    return acc


def main():
    rng = np.random.default_rng(0)
    print("%-6s %-14s %-14s %-14s %-14s" % ("K", "fp32 chain", "bf16x3, 6 prod", "bf16x3, 9 prod", "bf16x2, 3 prod"))
    for K in (64, 128, 256, 512):
        M, N = 256, 64
        a = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)          # post-ReLU activations
        b = (rng.uniform(-1, 1, (K, N)) * np.sqrt(6.0 / (K + N))).astype(np.float32)
        exact = a.astype(np.float64) @ b.astype(np.float64)
        scale = np.sqrt((exact ** 2).mean())
        err = lambda y: np.sqrt(((y.astype(np.float64) - exact) ** 2).mean()) / scale     # noqa: E731
        e_chain = err(chain32(a, b))
        a3, b3 = split(a, 3), split(b, 3)
        terms6 = [(2, 0), (0, 2), (1, 1), (1, 0), (0, 1), (0, 0)]                    # smallest first
        terms9 = [(2, 2), (2, 1), (1, 2)] + terms6
        def emulate(aa, bb, terms):
            total = np.zeros((M, N), np.float32)
            for i, j in terms:
                total = (total + chain32(aa[i], bb[j])).astype(np.float32)
            return total
        a2, b2 = split(a, 2), split(b, 2)
        print("%-6d %-14.3e %-14.3e %-14.3e %-14.3e" % (K, e_chain, err(emulate(a3, b3, terms6)), err(emulate(a3, b3, terms9)),
                                                        err(emulate(a2, b2, [(1, 0), (0, 1), (0, 0)]))))


if __name__ == "__main__":
    main()

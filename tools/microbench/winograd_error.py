#!/usr/bin/env python3
"""Numerics of ONE 64->64 3x3 layer on an 11x11 board as Winograd F(2x2,3x3) with the tower's split-f16 operands,
next to the direct convolution with the same operands (what k_tower_f16x3_s16 computes), both against a float64
convolution.  Pure numpy emulation of the MFMA arithmetic (f16 x f16 products are exact in fp32; sums in fp32):

  direct   x = hi + lo, w = hi + lo (f16 pairs):  sum over 576 terms of  hi*hi + hi*lo + lo*hi
  winograd V = B^T d B in fp32, split;  U = G g G^T in float64, split;  M[xi] = sum over 64 channels of the same
           three products;  Y = A^T M A in fp32

    python tools/microbench/winograd_error.py
"""
import numpy as np

rng = np.random.RandomState(0)
N, C = 11, 64
x = np.maximum(rng.randn(N, N, C), 0).astype(np.float32)                 # post-ReLU activations
w = (rng.randn(3, 3, C, C) * np.sqrt(2.0 / (9 * C))).astype(np.float32)  # He-scaled, BN folded: [ky][kx][cin][cout]


def split(a):
    hi = a.astype(np.float16)
    lo = (a - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float32), lo.astype(np.float32)


def dot3(a, b):
    """sum_k a[..., k] * b[k, ...] as three f16 MFMA products with fp32 accumulation (a, b fp32 -> split hi/lo)."""
    ah, al = split(a)
    bh, bl = split(b)
    return (ah @ bh).astype(np.float32) + (ah @ bl).astype(np.float32) + (al @ bh).astype(np.float32)


xp = np.zeros((N + 3, N + 3, C), np.float32)      # one row/column of zero halo on top/left, two at the bottom/right
xp[1:N + 1, 1:N + 1] = x
ref = np.zeros((N, N, C))
for ky in range(3):
    for kx in range(3):
        ref += xp[ky:ky + N, kx:kx + N].astype(np.float64) @ w[ky, kx].astype(np.float64)

# direct, split-f16: im2col [121][576] x [576][64]
cols = np.concatenate([xp[ky:ky + N, kx:kx + N].reshape(N * N, C) for ky in range(3) for kx in range(3)], axis=1)
direct = dot3(cols, w.reshape(9 * C, C)).reshape(N, N, C)

# Winograd F(2x2,3x3)
BT = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float32)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
AT = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float32)
U = np.einsum("ay,yxio,bx->abio", G, w.astype(np.float64), G).astype(np.float32)      # [4][4][cin][cout]
T = (N + 1) // 2
wino = np.zeros((2 * T, 2 * T, C), np.float32)
for ty in range(T):
    for tx in range(T):
        d = xp[2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]                                   # [4][4][cin], fp32
        V = np.einsum("ar,rcd,bc->abd", BT, d, BT).astype(np.float32)                  # adds only: exact order irrelevant here
        M = np.stack([np.stack([dot3(V[a, b][None, :], U[a, b])[0] for b in range(4)]) for a in range(4)])   # [4][4][cout]
        Y = np.einsum("ia,abo,jb->ijo", AT, M, AT).astype(np.float32)
        wino[2 * ty:2 * ty + 2, 2 * tx:2 * tx + 2] = Y
wino = wino[:N, :N]
scale = np.abs(ref).max()
print("output scale (max |y|)            %.4f" % scale)
print("direct   split-f16 max |error|    %.3e  (%.2e of scale)" % (np.abs(direct - ref).max(), np.abs(direct - ref).max() / scale))
print("winograd split-f16 max |error|    %.3e  (%.2e of scale)" % (np.abs(wino - ref).max(), np.abs(wino - ref).max() / scale))
print("fp32 direct (numpy) max |error|   %.3e" % np.abs(sum(xp[ky:ky + N, kx:kx + N] @ w[ky, kx] for ky in range(3) for kx in range(3)) - ref).max())

#!/usr/bin/env python
"""CPU emulation of two ways to form fp32 products on 16-bit matrix instructions with fp32 accumulation, against float64:
  bf16 x 3 pieces, 6 partial products (what lsfa_conv_split_fwd runs), and
  fp16 x 2 pieces (hi = fp16(x*s), lo = fp16(x*s - hi)), 3 partial products (hi*hi + hi*lo + lo*hi), s a per-tensor power of two.
Partial products of 16-bit values are exact in fp32; the accumulation is emulated in float32 in k order (numpy float32 dot in blocks of
16 like a k-step).  Prints max / rms error relative to max|y| for a few K, with ReLU-like activations and N(0, 0.01) weights."""
import numpy as np


def bf16_trunc(x):
    return (x.view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def cut_bf16_3(x):
    h = bf16_trunc(x.copy())
    r1 = (x - h).astype(np.float32)
    m = bf16_trunc(r1.copy())
    l = (r1 - m).astype(np.float32)          # at most 8 significant bits: a bf16 value
    return h, m, l


def cut_fp16_2(x, s):
    xs = (x * np.float32(s)).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32)
    l = (xs - h).astype(np.float32).astype(np.float16).astype(np.float32)
    return h, l


def acc32(terms_a, terms_b, step=16):
    """sum over pairs and k-steps, float32 accumulator, each k-step's 16-term partial sum formed in float32 as the MFMA does"""
    M, K = terms_a[0].shape
    N = terms_b[0].shape[1]
    acc = np.zeros((M, N), np.float32)
    for k0 in range(0, K, step):
        for a, b in zip(terms_a, terms_b):
            acc += (a[:, k0:k0 + step].astype(np.float32) @ b[k0:k0 + step].astype(np.float32)).astype(np.float32)
    return acc


def main():
    rs = np.random.RandomState(0)
    print("%8s %26s %26s %26s" % ("K", "fp32 (numpy)", "bf16 x3, 6 products", "fp16 x2, 3 products"))
    for K in (256, 2304, 18432):
        M, N = 48, 32
        a = np.maximum(rs.randn(M, K), 0).astype(np.float32) * np.float32(3.0)
        b = (rs.randn(K, N) * 0.01).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        scale = np.abs(ref).max()
        y32 = acc32([a], [b])
        ah, am, al = cut_bf16_3(a)
        bh, bm, bl = cut_bf16_3(b)
        y6 = acc32([al, ah, am, am, ah, ah], [bh, bl, bm, bh, bm, bh])
        sa = 2.0 ** (14 - np.ceil(np.log2(np.abs(a).max())))
        sb = 2.0 ** (14 - np.ceil(np.log2(np.abs(b).max())))
        fh, fl = cut_fp16_2(a, sa)
        gh, gl = cut_fp16_2(b, sb)
        y3 = acc32([fl, fh, fh], [gh, gl, gh]) / np.float32(sa * sb)
        out = []
        for y in (y32, y6, y3):
            e = (y.astype(np.float64) - ref) / scale
            out.append("max %.2e rms %.2e" % (np.abs(e).max(), np.sqrt((e ** 2).mean())))
        print("%8d %26s %26s %26s" % (K, out[0], out[1], out[2]))


if __name__ == "__main__":
    main()

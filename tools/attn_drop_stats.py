"""Statistics of the attention-dropout mask function of csrc/attention.hip (numpy restatement of attn_keep(); no GPU needed).

    python tools/attn_drop_stats.py

Prints, per key: the drop rate, the correlation of the keep bits of horizontally / vertically / diagonally adjacent elements (inside
a pair that shares its hash word, across pairs, across 4x4 blocks), the variance of row and column sums relative to the binomial
one, the correlation between two (batch, head) keys, and -- exhaustively over the 2^24 values of the shared word y -- the joint
probability of the two decisions a pair takes from one y through the two multipliers KA / KB against p^2.
Expected: every correlation within ~3 / sqrt(T^2) of zero, variance ratios ~1, P(ab) = p^2 to 1e-5.
"""
import numpy as np

M32 = np.uint64(0xFFFFFFFF)
G = np.uint64(0x9E3779B9)
KA, KB = 0x2C1B3D, 0x5A2D39
U = np.uint64


def mix32(x):
    x = x & M32
    x ^= x >> U(16)
    x = (x * U(0x7FEB352D)) & M32
    x ^= x >> U(15)
    x = (x * U(0x846CA68B)) & M32
    x ^= x >> U(16)
    return x


def mul24(a, b):
    return ((a & U(0xFFFFFF)) * (U(b) & U(0xFFFFFF))) & M32


def keep_mask(akey, T, p):
    thr = U(int(p * 2 ** 32))
    i = np.arange(T, dtype=np.uint64)[:, None]
    j = np.arange(T, dtype=np.uint64)[None, :]
    r = mix32(U(akey) + (((i >> U(2)) * U(0x9E3779B9)) & M32))
    c = mix32((((j >> U(2)) * U(0x85EBCA6B)) & M32) + U(0x165667B1))
    x = r ^ c
    seed = (mul24(x, 0x846CA7) + (x >> U(13))) & M32
    t = (seed + ((i & U(3)) * U(2) + ((j & U(3)) >> U(1))) * G) & M32
    y = t ^ (t >> U(15))
    return np.where((j & U(1)) == 0, mul24(y, KA), mul24(y, KB)) >= thr


if __name__ == "__main__":
    T, p = 512, 0.1
    masks = []
    for key in (1, 2, 3, 12345, 0xDEADBEEF):
        m = keep_mask(key, T, p).astype(np.float64)
        k = m - m.mean()
        cor = lambda a, b: (a * b).mean() / k.var()  # noqa: E731
        print(f"key {key:#x}: drop rate {1 - m.mean():.5f}  in-pair {cor(k[:, 0::2], k[:, 1::2]):+.4f}  "
              f"across pairs {cor(k[:, 1:-1:2], k[:, 2::2]):+.4f}  vertical {cor(k[:-1], k[1:]):+.4f}  "
              f"diagonal {cor(k[:-1, :-1], k[1:, 1:]):+.4f}  +4 cols {cor(k[:, :-4], k[:, 4:]):+.4f}  +4 rows {cor(k[:-4], k[4:]):+.4f}  "
              f"row-sum var / binomial {m.sum(1).var() / (T * p * (1 - p)):.3f}  col-sum {m.sum(0).var() / (T * p * (1 - p)):.3f}")
        masks.append(k)
    print(f"two keys: correlation {(masks[0] * masks[1]).mean() / masks[0].var():+.4f}")
    y = np.arange(2 ** 24, dtype=np.uint64)
    for p in (0.1, 0.25, 0.5):
        thr = U(int(p * 2 ** 32))
        a, b = mul24(y, KA) < thr, mul24(y, KB) < thr
        print(f"p = {p}: P(a) {a.mean():.6f}  P(b) {b.mean():.6f}  P(a and b) {(a & b).mean():.6f}  p^2 {p * p:.6f}")

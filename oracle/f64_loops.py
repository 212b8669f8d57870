"""Independent float64 loop derivations of the hot-path ops (TEST INFRASTRUCTURE ONLY).

Second, deliberately naive derivation of every op in oracle/restatement.py, written from the
definitions with explicit index arithmetic instead of torch ops, in float64.
tests/test_oracle.py cross-checks restatement.py against these on small shapes, so that a mistake
in the fp32 restatement (pad split, bilinear taps, variance flavour, Adam epsilon placement ...)
has to be made twice, in two different styles, to go unnoticed.  PARITY UNPINNED wrt TF 1.12.
"""
import math

import numpy as np


def conv_same(x, w, b, stride, pad=0):
    """tf.pad(pad) + conv2d SAME (layers.py:4-10). x [N,H,W,C] w [kh,kw,C,Co]."""
    x = np.asarray(x, np.float64)
    w = np.asarray(w, np.float64)
    n, h, wd, c = x.shape
    kh, kw, _, co = w.shape
    hp, wp = h + 2 * pad, wd + 2 * pad
    ho, wo = math.ceil(hp / stride), math.ceil(wp / stride)
    tot_h = max((ho - 1) * stride + kh - hp, 0)
    tot_w = max((wo - 1) * stride + kw - wp, 0)
    pt, pl = tot_h // 2 + pad, tot_w // 2 + pad      # offset of the real image inside the padded one
    y = np.zeros((n, ho, wo, co), np.float64)
    for oy in range(ho):
        for ox in range(wo):
            acc = np.zeros((n, co), np.float64)
            for r in range(kh):
                iy = oy * stride + r - pt
                if iy < 0 or iy >= h:
                    continue
                for q in range(kw):
                    ix = ox * stride + q - pl
                    if ix < 0 or ix >= wd:
                        continue
                    acc += x[:, iy, ix, :] @ w[r, q]
            y[:, oy, ox, :] = acc
    if b is not None:
        y += np.asarray(b, np.float64)
    return y


def batch_norm_train(x, gamma, beta, eps=1e-5):
    x = np.asarray(x, np.float64)
    c = x.shape[-1]
    flat = x.reshape(-1, c)
    mean = flat.sum(0) / flat.shape[0]
    var = ((flat - mean) ** 2).sum(0) / flat.shape[0]
    y = (x - mean) / np.sqrt(var + eps) * gamma + beta
    return y, mean, var


def resize2x(x):
    """Legacy TF bilinear x2: out[2i]=in[i]; out[2i+1]=in[i]+(in[min(i+1,n-1)]-in[i])*.5; x then y."""
    x = np.asarray(x, np.float64)
    n, h, w, c = x.shape
    tmp = np.zeros((n, h, 2 * w, c))
    for j in range(w):
        jn = min(j + 1, w - 1)
        tmp[:, :, 2 * j] = x[:, :, j]
        tmp[:, :, 2 * j + 1] = x[:, :, j] + (x[:, :, jn] - x[:, :, j]) * 0.5
    out = np.zeros((n, 2 * h, 2 * w, c))
    for i in range(h):
        i_n = min(i + 1, h - 1)
        out[:, 2 * i] = tmp[:, i]
        out[:, 2 * i + 1] = tmp[:, i] + (tmp[:, i_n] - tmp[:, i]) * 0.5
    return out


def get_coord_xy(x):
    """[B,H,W,K] -> [B,K,2] (x,y): utils/model.py:63-70 applied per networks/__init__.py:69-71."""
    x = np.asarray(x, np.float64)
    b, h, w, k = x.shape
    out = np.zeros((b, k, 2))
    for bi in range(b):
        for ki in range(k):
            m = x[bi, :, :, ki]
            rows = m.sum(1) / w            # mean over W  -> profile along H (y)
            cols = m.sum(0) / h            # mean over H  -> profile along W (x)
            for axis, prof, n in ((1, rows, h), (0, cols, w)):
                e = np.exp(prof - prof.max())
                p = e / e.sum()
                grid = -1.0 + (2.0 / (n - 1)) * np.arange(n)
                out[bi, ki, axis] = (p * grid).sum()
    return out


def gaussian_maps(mu, h, w, inv_std=14.3):
    mu = np.asarray(mu, np.float64)
    b, k, _ = mu.shape
    out = np.zeros((b, h, w, k))
    ys = -1.0 + (2.0 / (h - 1)) * np.arange(h)
    xs = -1.0 + (2.0 / (w - 1)) * np.arange(w)
    for bi in range(b):
        for ki in range(k):
            mx, my = mu[bi, ki, 0], mu[bi, ki, 1]
            d = (ys[:, None] - my) ** 2 + (xs[None, :] - mx) ** 2
            out[bi, :, :, ki] = np.exp(-d * inv_std ** 2)
    return out


def maxpool2(x):
    x = np.asarray(x, np.float64)
    n, h, w, c = x.shape
    ho, wo = (h + 1) // 2, (w + 1) // 2
    out = np.full((n, ho, wo, c), -np.inf)
    for i in range(h):
        for j in range(w):
            out[:, i // 2, j // 2] = np.maximum(out[:, i // 2, j // 2], x[:, i, j])
    return out


def sigmoid_xent(x, z):
    x = np.asarray(x, np.float64)
    # -z*log(sigmoid(x)) - (1-z)*log(1-sigmoid(x)), straight from the definition
    s = 1.0 / (1.0 + np.exp(-x))
    return -(z * np.log(s) + (1 - z) * np.log(1 - s))


def adam_tf(p, g, m, v, t, lr, b1=0.5, b2=0.999, eps=1e-8):
    """t = 1-based step count. Returns (p, m, v)."""
    p, g, m, v = (np.asarray(a, np.float64) for a in (p, g, m, v))
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    return p - lr_t * m / (np.sqrt(v) + eps), m, v


def same_pad_table():
    """SURVEY Appendix A samepad column: (in, k, s, explicit pad) -> (before, after, out)."""
    return {(128, 7, 1, 0): (3, 3, 128), (128, 3, 2, 0): (0, 1, 64), (64, 3, 2, 0): (0, 1, 32),
            (32, 3, 2, 0): (0, 1, 16), (128, 4, 2, 1): (1, 1, 65), (65, 4, 2, 1): (1, 2, 34),
            (34, 4, 2, 1): (1, 1, 18), (18, 4, 2, 1): (1, 1, 10), (10, 4, 2, 1): (1, 1, 6),
            (6, 4, 2, 1): (1, 1, 4), (4, 3, 1, 1): (1, 1, 6), (16, 3, 1, 0): (1, 1, 16), (128, 1, 1, 0): (0, 0, 128)}

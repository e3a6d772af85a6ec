"""CPU oracle for the block-wise 8-bit Adam state -- TEST INFRASTRUCTURE ONLY.

Restates what the reference's optimizer does (train.py:254 ``bnb.optim.Adam8bit``): Adam whose two moments are
stored as 8-bit codes per block of 2048 elements with one absolute maximum per block, the first moment on the
signed "dynamic" code table and the second on the unsigned one (bitsandbytes ``create_dynamic_map(signed,
max_exponent_bits=7, total_bits=8)``).  PARITY UNPINNED: bitsandbytes is not installed here and the reference
holds no optimizer fixture; the table below follows the published construction, the update rule is the fp32
Adam of ``csrc/optim.hip`` applied to the de-quantised moments.
"""
import numpy as np

BLOCK = 2048


def dynamic_map(signed, max_exponent_bits=7, total_bits=8):
    data = []
    non_sign_bits = total_bits - 1
    for i in range(max_exponent_bits):
        items = 2 ** (i + non_sign_bits - max_exponent_bits) + 1 if signed else 2 ** (i + non_sign_bits - max_exponent_bits + 1) + 1
        bounds = np.linspace(0.1, 1.0, items, dtype=np.float64)
        means = (bounds[:-1] + bounds[1:]) / 2.0
        scale = 10.0 ** (-(max_exponent_bits - 1) + i)
        data += (scale * means).tolist()
        if signed:
            data += (-scale * means).tolist()
    data += [0.0, 1.0]
    assert len(data) == 2 ** total_bits
    return np.sort(np.asarray(data, dtype=np.float64)).astype(np.float32)


def quantize(x, table):
    """Index of the nearest table entry (ties to the lower one), x already divided by the block's absmax."""
    x = np.maximum(x.astype(np.float32), table[0])
    lo = np.clip(np.searchsorted(table, x, side="right") - 1, 0, 255)
    hi = np.minimum(lo + 1, 255)
    return np.where(x - table[lo] <= table[hi] - x, lo, hi).astype(np.uint8)


class Adam8:
    def __init__(self, n, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, decoupled=True):
        self.lr, self.b1, self.b2, self.eps, self.wd, self.decoupled = lr, betas[0], betas[1], eps, weight_decay, decoupled
        self.t1, self.t2 = dynamic_map(True), dynamic_map(False)
        nb = (n + BLOCK - 1) // BLOCK
        self.q1 = np.full(n, 127, np.uint8)
        self.q2 = np.zeros(n, np.uint8)
        self.a1 = np.zeros(nb, np.float32)
        self.a2 = np.zeros(nb, np.float32)
        self.t = 0

    def step(self, p, g, max_norm=0.0, grad_div=1.0):
        f = np.float32
        self.t += 1
        gscale = f(1.0)
        if max_norm > 0:
            norm = f(np.sqrt(f(np.sum(g.astype(np.float64) ** 2)))) / f(abs(grad_div))
            gscale = min(f(1.0), f(max_norm) / (norm + f(1e-6)))
        gscale = f(gscale / f(grad_div))
        bc1, bc2 = f(1.0 - self.b1 ** self.t), f(1.0 - self.b2 ** self.t)
        blk = np.arange(p.size) // BLOCK
        m = self.t1[self.q1] * self.a1[blk]
        v = self.t2[self.q2] * self.a2[blk]
        gg = (g * gscale).astype(f)
        if not self.decoupled:
            gg = gg + f(self.wd) * p
        m = (f(self.b1) * m + f(1.0 - self.b1) * gg).astype(f)
        v = (f(self.b2) * v + f(1.0 - self.b2) * gg * gg).astype(f)
        pn = p - f(self.lr) * (m / bc1) / (np.sqrt(v / bc2) + f(self.eps))
        if self.decoupled:
            pn = pn * f(1.0 - self.lr * self.wd)
        nb = self.a1.size
        pad = nb * BLOCK - p.size
        a1 = np.abs(np.pad(m, (0, pad))).reshape(nb, BLOCK).max(1).astype(f)
        a2 = np.pad(v, (0, pad)).reshape(nb, BLOCK).max(1).astype(f)
        i1 = np.where(a1 > 0, f(1.0) / np.where(a1 > 0, a1, 1), 0).astype(f)
        i2 = np.where(a2 > 0, f(1.0) / np.where(a2 > 0, a2, 1), 0).astype(f)
        self.q1 = quantize(m * i1[blk], self.t1)
        self.q2 = quantize(v * i2[blk], self.t2)
        self.a1, self.a2 = a1, a2
        return pn.astype(f), m, v

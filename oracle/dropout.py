"""Oracle: the counter-based LoRA dropout mask (test infrastructure).

The reference applies `nn.Dropout(p=0.05)` to the input of every LoRA branch (peft==0.4.0 `Linear.forward`:
`lora_B(lora_A(lora_dropout(x))) * scaling`, wired at reference `training.py:91,218-226`); peft is not installed and torch's
Philox stream cannot be reproduced by a different kernel decomposition anyway, so the HIP path defines its own mask and this file
restates that definition (include/llmseg_hip.h, `llmseg_dropout`) in numpy for bit-exact parity:

    Philox4x32-10, key = (seed_lo, seed_hi), counter = (idx_lo, idx_hi, stream, offset), idx = element // 8;
    element 8 idx + j is decided by 16-bit field j of the 128-bit output (word j >> 1, half j & 1): kept when field >= thr,
    thr = round(p * 65536); kept values are scaled by 65536 / (65536 - thr).

PARITY UNPINNED against peft (absent); the statistical contract (independent Bernoulli(1 - p) keep, 1 / (1 - p) scale) is tested.
"""
import numpy as np
import torch

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays c0..c3; k0, k1 uint32 scalars.  -> four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK32).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def keep_mask(rows, cols, seed, offset, stream, p, seg_rows=0):
    """bool [rows, cols]: True where the element of a dense [rows, cols] activation is kept (cols % 8 == 0).
    seg_rows > 0 (`llmseg_dropout.seg_rows`): consecutive segments of seg_rows rows, segment s with the mask of its own pass at offset + s."""
    assert cols % 8 == 0
    if seg_rows and rows > seg_rows:
        return torch.cat([keep_mask(min(seg_rows, rows - r0), cols, seed, offset + i, stream, p) for i, r0 in enumerate(range(0, rows, seg_rows))], 0)
    thr = int(round(p * 65536))
    n8 = rows * cols // 8
    idx = np.arange(n8, dtype=np.uint64)
    w = philox4x32_10((idx & MASK32).astype(np.uint32), (idx >> np.uint64(32)).astype(np.uint32), np.full(n8, stream, np.uint32),
                      np.full(n8, offset & 0xFFFFFFFF, np.uint32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    fields = np.stack([(w[j >> 1] >> np.uint32(16 * (j & 1))) & np.uint32(0xFFFF) for j in range(8)], 1)      # [n8, 8]
    return torch.from_numpy((fields >= thr).reshape(rows, cols))


def drop_scale(p):
    thr = int(round(p * 65536))
    return 65536.0 / (65536.0 - thr)


def apply(x2d, seed, offset, stream, p, seg_rows=0):
    """x2d [rows, cols] -> x * mask * scale (the HIP kernels' drop(x))."""
    if p <= 0:
        return x2d
    m = keep_mask(x2d.shape[0], x2d.shape[1], seed, offset, stream, p, seg_rows).to(x2d.device)
    return x2d * m.to(x2d.dtype) * drop_scale(p)

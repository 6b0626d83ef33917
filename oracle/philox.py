"""TEST INFRASTRUCTURE — CPU oracle, not product code.

Philox4x32-10 counter-based RNG (Salmon et al., "Parallel random numbers: as
easy as 1, 2, 3", SC'11) in numpy, and the dropout-mask convention that the HIP
kernels (sdumc_amd/csrc/philox.h) implement bit-for-bit.

Why it exists: the reference draws dropout masks with torch's CPU
``bernoulli_`` stream (toolkit/models/wengnet_mosei_mult_views_text_missing.py:54,77,270
-> nn.Dropout), which no GPU kernel can replay.  Parity in train mode is
therefore defined on masks that BOTH sides can compute from
(seed, call, site, sample, row, column) alone.

Mask convention (one Philox call = 4 consecutive columns of one row):
    ctr = ( row * ceil(width/4) + col/4 ,  global sample index ,  site ,  call )
    key = ( seed & 0xffffffff , seed >> 32 )
    keep(col) = word[col % 4] >= floor(p * 2**32)
    value     = keep ? 1/(1-p) (computed in fp32 as 1.0f / (1.0f - p)) : 0
``site`` numbers the nn.Dropout call sites of one reference forward in call
order (see SITE_* in oracle/sdumc_oracle.py); ``call`` numbers forward calls
(stream 0 of step k = 2k, stream 1 = 2k+1).  The sample index is GLOBAL, so a
batch shard draws the same masks as the unsharded batch.
"""
import numpy as np

_M0 = np.uint64(0xD2511F53)
_M1 = np.uint64(0xCD9E8D57)
_W0 = np.uint32(0x9E3779B9)
_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All inputs broadcastable uint32 arrays.
    Returns 4 uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.asarray(c1, dtype=np.uint32)
    c2 = np.asarray(c2, dtype=np.uint32)
    c3 = np.asarray(c3, dtype=np.uint32)
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    with np.errstate(over="ignore"):
        for r in range(10):
            p0 = c0.astype(np.uint64) * _M0
            p1 = c2.astype(np.uint64) * _M1
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK32).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def drop_threshold(p):
    """uint32 threshold: an element is dropped when its word < threshold."""
    return int(float(p) * 4294967296.0)


def drop_scale(p):
    """fp32 scale of kept elements, as ATen computes it (noise.div_(1-p))."""
    return np.float32(1.0) / (np.float32(1.0) - np.float32(p))


def dropout_mask(nsamp, rows, width, p, seed, call, site, sample0=0):
    """float32 [nsamp, rows, width] multiplicative mask (0 or 1/(1-p)).
    A width that is not a multiple of 4 uses ceil(width/4) Philox calls per row (the last one partly unused)."""
    q = (width + 3) // 4
    b = (np.arange(nsamp, dtype=np.uint64) + np.uint64(sample0)).astype(np.uint32)
    rc = np.arange(rows * q, dtype=np.uint32)
    w = philox4x32_10(rc[None, :], b[:, None], np.uint32(site), np.uint32(call),
                      seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    words = np.stack(w, axis=-1).reshape(nsamp, rows, 4 * q)[:, :, :width]
    keep = words >= np.uint32(drop_threshold(p))
    return keep.astype(np.float32) * drop_scale(p)

"""Host (NumPy) model of the device random streams in openmcmc_amd/csrc/omc_common.h.
Philox4x32-10 is integer arithmetic: the device words must match these bit for bit."""

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
MASK = np.uint64(0xFFFFFFFF)
PURPOSE = {"normal": 0, "gamma": 1, "uniform": 2, "raw": 3}


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint32) for v in (c0, c1, c2, c3))
    k0, k1 = np.uint32(k0), np.uint32(k1)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c0.astype(np.uint64)
            p1 = M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & MASK).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + W0), np.uint32(k1 + W1)
    return c0, c1, c2, c3


def rng_blocks(seed, draw_index, purpose, global_chain, blocks):
    """Words of blocks `blocks` (array) for one chain; layout of omc_rng_block()."""
    blocks = np.asarray(blocks, dtype=np.uint32)
    c1 = np.full(blocks.shape, draw_index & 0xFFFFFFFF, dtype=np.uint32)
    c2 = np.full(blocks.shape, global_chain & 0xFFFFFFFF, dtype=np.uint32)
    c3v = (PURPOSE[purpose] << 24) | (((global_chain >> 32) & 0xFF) << 16) | ((draw_index >> 32) & 0xFFFF)
    c3 = np.full(blocks.shape, c3v, dtype=np.uint32)
    return philox4x32_10(blocks, c1, c2, c3, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)


def u53(lo, hi):
    v = lo.astype(np.uint64) ^ (hi.astype(np.uint64) << np.uint64(21))
    return 2.0**-53 + v.astype(np.float64) * 2.0**-53


def normals(seed, draw_index, global_chain, n):
    """First n N(0,1) of the chain's 'normal' stream (Box-Muller as omc_normal_pair)."""
    nb = (n + 1) // 2
    x, y, z, w = rng_blocks(seed, draw_index, "normal", global_chain, np.arange(nb))
    u = u53(x, y)
    v2 = z.astype(np.uint64) ^ (w.astype(np.uint64) << np.uint64(21))
    ang = 2.0**-52 + v2.astype(np.float64) * 2.0**-52
    s = np.sqrt(-2.0 * np.log(u))
    out = np.empty(2 * nb)
    out[0::2] = np.sin(np.pi * ang) * s
    out[1::2] = np.cos(np.pi * ang) * s
    return out[:n]

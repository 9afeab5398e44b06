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


def unit_mantissa(lo, hi20src):
    """float64 in [1, 2): mantissa = top 20 bits of hi20src, then all of lo (omc_unit_mantissa)."""
    bits = (np.uint64(0x3FF) << np.uint64(52)) | ((hi20src.astype(np.uint64) >> np.uint64(12)) << np.uint64(32)) \
        | lo.astype(np.uint64)
    return bits.view(np.float64)


def normal_pairs(x, y, z, w):
    """Words of one Philox block -> two N(0,1) (omc_normal_pair): Box-Muller with the radius uniform
    u = 2 - m(x, y) in (0, 1], the angle t = (m(z, w) - 3/2) pi/2 in [-pi/4, pi/4), a reflection x -> -x on
    bit 11 of w and a swap of the two coordinates on bit 10."""
    x, y, z, w = (np.asarray(v, dtype=np.uint32) for v in (x, y, z, w))
    u = 2.0 - unit_mantissa(x, y)
    r = np.sqrt(np.maximum(-2.0 * np.log(u), 0.0))
    t = (unit_mantissa(z, w) - 1.5) * (np.pi / 2)
    sn, cs = np.sin(t), np.cos(t)
    cs = np.where((w >> np.uint32(11)) & np.uint32(1), -cs, cs)
    swap = ((w >> np.uint32(10)) & np.uint32(1)).astype(bool)
    return np.where(swap, cs, sn) * r, np.where(swap, sn, cs) * r


def normals(seed, draw_index, global_chain, n):
    """First n N(0,1) of the chain's 'normal' stream."""
    nb = (n + 1) // 2
    x, y, z, w = rng_blocks(seed, draw_index, "normal", global_chain, np.arange(nb))
    out = np.empty(2 * nb)
    out[0::2], out[1::2] = normal_pairs(x, y, z, w)
    return out[:n]

"""Philox4x32-10 (Salmon et al., SC'11) vectorised over rays, and the draw conventions of the device."""
import numpy as N

M0 = N.uint64(0xD2511F53)
M1 = N.uint64(0xCD9E8D57)
W0 = 0x9E3779B9
W1 = 0xBB67AE85
MASK32 = N.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """c*: uint64 arrays holding 32-bit words; k*: python ints.  Returns 4 uint64 arrays of 32-bit words."""
    c0 = c0.astype(N.uint64); c1 = c1.astype(N.uint64); c2 = c2.astype(N.uint64); c3 = c3.astype(N.uint64)
    for r in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        n0 = (p1 >> N.uint64(32)) ^ c1 ^ N.uint64(k0)
        n1 = p1 & MASK32
        n2 = (p0 >> N.uint64(32)) ^ c3 ^ N.uint64(k1)
        n3 = p0 & MASK32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def u01(a, b):
    """two 32-bit words -> double in [0,1) with 53 random bits"""
    return ((a >> N.uint64(5)).astype(N.float64) * 67108864.0 + (b >> N.uint64(6)).astype(N.float64)) * \
        (1.0 / 9007199254740992.0)


def uniform_pair(seed, rid, event, block):
    """uniforms (2*block, 2*block+1) of the stream (seed, rid, event); rid: uint64 array"""
    rid = N.asarray(rid, dtype=N.uint64)
    ones = N.ones(rid.shape, dtype=N.uint64)
    o = philox4x32_10(rid & MASK32, rid >> N.uint64(32), ones * N.uint64(event), ones * N.uint64(block),
                      int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    return u01(o[0], o[1]), u01(o[2], o[3])


def uniform_quad(seed, rid, event, block):
    """four uniforms in (0,1) with 32 random bits each from one block: the draws of a source ray (trc_uniform_quad)"""
    rid = N.asarray(rid, dtype=N.uint64)
    ones = N.ones(rid.shape, dtype=N.uint64)
    o = philox4x32_10(rid & MASK32, rid >> N.uint64(32), ones * N.uint64(event), ones * N.uint64(block),
                      int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF)
    return tuple((w.astype(N.float64) + 0.5) * (1.0 / 4294967296.0) for w in o)


def normal_pair(u0, u1):
    """Box-Muller"""
    r = N.sqrt(-2.0 * N.log(1.0 - u0))
    a = 2.0 * N.pi * u1
    return r * N.cos(a), r * N.sin(a)


def child_rid(rid, event):
    rid = N.asarray(rid, dtype=N.uint64)
    with N.errstate(over='ignore'):
        return rid * N.uint64(0x9E3779B97F4A7C15) + N.uint64(0xD1B54A32D192ED03) * N.uint64(event + 1)

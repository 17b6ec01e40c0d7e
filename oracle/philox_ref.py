"""ORACLE (test infrastructure): numpy restatement of Philox4x32-10 (Salmon et al., SC'11; the
published Random123 algorithm) as used by the on-device noise generator (csrc/sde.hip), plus the
Box-Muller mapping.  The reference draws noise with torch.randn_like (utils/sde_utils.py:185), whose
stream differs per device, so parity runs inject host noise; this oracle pins the throughput-mode RNG.
Known-answer vectors from the Random123 distribution (kat_vectors) are checked in tests/test_oracle_philox.py."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)


def philox4x32_10(counter, key):
    """counter: [n,4] uint32, key: [2] uint32 -> [n,4] uint32"""
    c = np.array(counter, dtype=np.uint32).reshape(-1, 4).copy()
    k0, k1 = np.uint32(key[0]), np.uint32(key[1])
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = M0 * c[:, 0].astype(np.uint64)
            p1 = M1 * c[:, 2].astype(np.uint64)
            n0 = (p1 >> np.uint64(32)).astype(np.uint32) ^ c[:, 1] ^ k0
            n1 = p1.astype(np.uint32)
            n2 = (p0 >> np.uint64(32)).astype(np.uint32) ^ c[:, 3] ^ k1
            n3 = p0.astype(np.uint32)
            c = np.stack([n0, n1, n2, n3], axis=1)
            k0 = np.uint32(k0 + W0)
            k1 = np.uint32(k1 + W1)
    return c


def counters(n, seed, offset):
    ctr = np.uint64(offset) + np.arange(n, dtype=np.uint64)
    c = np.zeros((n, 4), dtype=np.uint32)
    c[:, 0] = (ctr & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c[:, 1] = (ctr >> np.uint64(32)).astype(np.uint32)
    key = np.array([seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF], dtype=np.uint32)
    return c, key


def philox_raw(n, seed, offset=0):
    c, key = counters(n, seed, offset)
    return philox4x32_10(c, key)


def u01(x):
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24) + np.float32(2.0 ** -25)


def randn(n, seed, offset=0):
    """n normals; element i = lane i%4 of counter offset + i//4 (matches idiff_randn)."""
    nc = (n + 3) // 4
    w = philox_raw(nc, seed, offset)
    r0 = np.sqrt(np.float32(-2.0) * np.log(u01(w[:, 0])))
    r1 = np.sqrt(np.float32(-2.0) * np.log(u01(w[:, 2])))
    a0 = np.float32(6.283185307179586) * u01(w[:, 1])
    a1 = np.float32(6.283185307179586) * u01(w[:, 3])
    z = np.stack([r0 * np.cos(a0), r0 * np.sin(a0), r1 * np.cos(a1), r1 * np.sin(a1)], axis=1).astype(np.float32)
    return z.reshape(-1)[:n]

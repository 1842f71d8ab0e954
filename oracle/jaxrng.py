"""NumPy emulation of the JAX threefry PRNG and of haiku's initialisers (oracle; test infra only).

The reference's parameter-dependent goldens (tests/test_wf/*.npz, tests/test_hamil/
test_local_energy_*.npz) were produced with `hk.transform(...).init(jax.random.PRNGKey(0), ...)`
(reference tests/conftest.py:116-139).  JAX and haiku cannot be installed here, so their
PUBLISHED algorithms are restated: Threefry-2x32 (Salmon et al. 2011, 20 rounds, as used by
jax._src.prng), `jax.random.split / uniform / truncated_normal`, haiku's `PRNGSequence`
(next key = split(key)[1], carried key = split(key)[0]) and `VarianceScaling` /
`TruncatedNormal`.  Two generations of JAX differ in how counters are laid out
(`jax_threefry_partitionable`, default since JAX 0.5); both are implemented and the caller
selects one.  Validated end to end: the goldens are reproduced only if every step is right.
"""
from __future__ import annotations

import numpy as np
from scipy.special import erf, erfinv

M32 = np.uint64(0xFFFFFFFF)
ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, d):
    return ((x << np.uint64(d)) | (x >> np.uint64(32 - d))) & M32


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32, 20 rounds.  k0,k1 scalars; x0,x1 uint32 arrays (same shape)."""
    k0, k1 = np.uint64(k0), np.uint64(k1)
    x0 = np.asarray(x0, np.uint64).copy()
    x1 = np.asarray(x1, np.uint64).copy()
    ks = (k0, k1, (k0 ^ k1 ^ np.uint64(0x1BD11BDA)) & M32)
    x0 = (x0 + ks[0]) & M32
    x1 = (x1 + ks[1]) & M32
    for r in range(5):
        for d in ROT[r % 2]:
            x0 = (x0 + x1) & M32
            x1 = _rotl(x1, d)
            x1 = x1 ^ x0
        x0 = (x0 + ks[(r + 1) % 3]) & M32
        x1 = (x1 + ks[(r + 2) % 3] + np.uint64(r + 1)) & M32
    return x0.astype(np.uint32), x1.astype(np.uint32)


class JaxRNG:
    def __init__(self, partitionable: bool):
        self.partitionable = partitionable

    @staticmethod
    def key(seed: int):
        return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], np.uint32)

    def _bits_flat(self, key, n):
        """threefry_2x32(key, iota(n)) of the original layout: counters split in halves."""
        odd = n % 2
        cnt = np.arange(n + odd, dtype=np.uint32)
        h = (n + odd) // 2
        a, b = threefry2x32(key[0], key[1], cnt[:h], cnt[h:])
        out = np.concatenate([a, b])
        return out[:n]

    def split(self, key, num=2):
        if self.partitionable:
            a, b = threefry2x32(key[0], key[1], np.zeros(num, np.uint32), np.arange(num, dtype=np.uint32))
            return np.stack([a, b], -1)
        return self._bits_flat(key, 2 * num).reshape(num, 2)

    def random_bits(self, key, bit_width, shape):
        size = int(np.prod(shape)) if len(shape) else 1
        if self.partitionable:
            idx = np.arange(size, dtype=np.uint64)
            a, b = threefry2x32(key[0], key[1], (idx >> np.uint64(32)).astype(np.uint32), (idx & M32).astype(np.uint32))
            if bit_width == 64:
                return ((a.astype(np.uint64) << np.uint64(32)) | b.astype(np.uint64)).reshape(shape)
            return (a ^ b).reshape(shape)
        n32 = size * (bit_width // 32)
        bits = self._bits_flat(key, n32)
        if bit_width == 64:
            hi, lo = bits[:size].astype(np.uint64), bits[size:].astype(np.uint64)
            return ((hi << np.uint64(32)) | lo).reshape(shape)
        return bits.reshape(shape)

    def uniform(self, key, shape, dtype, minval, maxval):
        """jax.random.uniform: mantissa bits | 1.0, minus 1, scaled."""
        if dtype == np.float32:
            bits = self.random_bits(key, 32, shape)
            f = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1)
            mn, mx = np.float32(minval), np.float32(maxval)
            return np.maximum(mn, (f * (mx - mn) + mn).astype(np.float32))
        bits = self.random_bits(key, 64, shape)
        f = ((bits >> np.uint64(12)) | np.uint64(0x3FF0000000000000)).view(np.float64) - 1.0
        return np.maximum(minval, f * (maxval - minval) + minval)

    def truncated_normal(self, key, lower, upper, shape, dtype):
        """jax.random.truncated_normal: sqrt(2) * erfinv(U(erf(l/sqrt2), erf(u/sqrt2))), clipped."""
        if dtype == np.float32:
            s2 = np.float32(np.sqrt(2))
            a, b = np.float32(erf(np.float32(lower) / s2)), np.float32(erf(np.float32(upper) / s2))
            u = self.uniform(key, shape, np.float32, a, b)
            out = (s2 * erfinv(u.astype(np.float64)).astype(np.float32)).astype(np.float32)
            lo = np.nextafter(np.float32(lower), np.float32(np.inf))
            hi = np.nextafter(np.float32(upper), np.float32(-np.inf))
            return np.clip(out, lo, hi)
        s2 = np.sqrt(2.0)
        a, b = erf(lower / s2), erf(upper / s2)
        u = self.uniform(key, shape, np.float64, a, b)
        out = s2 * erfinv(u)
        return np.clip(out, np.nextafter(lower, np.inf), np.nextafter(upper, -np.inf))


    def normal(self, key, shape, dtype=np.float64):
        """jax.random.normal: sqrt(2) * erfinv(U(nextafter(-1, 0), 1))."""
        lo = np.nextafter(dtype(-1.0), dtype(0.0))
        u = self.uniform(key, shape, dtype, lo, dtype(1.0))
        return (np.sqrt(2.0) * erfinv(u.astype(np.float64))).astype(dtype)


class HaikuInit:
    """hk.PRNGSequence + the initialisers the reference's test ansatz uses."""

    TRUNC = 0.87962566103423978

    def __init__(self, rng: JaxRNG, seed: int = 0):
        self.rng = rng
        self.key = rng.key(seed)

    def next_key(self):
        ks = self.rng.split(self.key, 2)
        self.key = ks[0]
        return ks[1]

    def variance_scaling(self, shape, dtype, scale=1.0):
        """VarianceScaling(scale, 'fan_in', 'truncated_normal')."""
        fan_in = shape[-2] if len(shape) >= 2 else shape[0]
        std = np.sqrt(scale / fan_in) / self.TRUNC
        x = self.rng.truncated_normal(self.next_key(), -2.0, 2.0, shape, dtype)
        return (x * dtype(std)).astype(dtype)

    def truncated_normal(self, shape, dtype, stddev=1.0):
        """TruncatedNormal(stddev) (hk.Embed default; hk.Linear default with 1/sqrt(fan_in))."""
        x = self.rng.truncated_normal(self.next_key(), -2.0, 2.0, shape, dtype)
        return (dtype(stddev) * x).astype(dtype)

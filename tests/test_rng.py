"""The keyed random numbers: known answers of the hash (restated here in plain Python integers) and basic
statistics.  The GPU library and the CPU checker carry separate copies of these functions; the same-seed
image tests only make sense if both implement exactly this definition."""
import numpy as np

import oracle_api

M32 = 0xFFFFFFFF


def rotl(x, r):
    return ((x << r) | (x >> (32 - r))) & M32


def fmix(h):
    h ^= h >> 16
    h = (h * 0x85EBCA6B) & M32
    h ^= h >> 13
    h = (h * 0xC2B2AE35) & M32
    h ^= h >> 16
    return h


def mix(h, w):
    w = (w * 0xCC9E2D51) & M32
    w = rotl(w, 15)
    w = (w * 0x1B873593) & M32
    h ^= w
    h = rotl(h, 13)
    return (h * 5 + 0xE6546B64) & M32


def key_root(seed, sample):
    return fmix(mix(mix(seed & M32, seed >> 32), sample))


def rand(seed, sample, pixel, dim):
    return (fmix(mix(mix(key_root(seed, sample), pixel), dim)) >> 8) / 16777216.0


def test_hash_matches_the_definition():
    L = oracle_api.lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed = int(rng.integers(0, 2**63)); sample = int(rng.integers(0, 2**31))
        pixel = int(rng.integers(0, 2**30)); dim = int(rng.integers(0, 2**23))
        assert L.oracle_rand(seed, sample, pixel, dim) == np.float32(rand(seed, sample, pixel, dim))
    for key, slot in [(0, 0), (1, 1 << 24), (0xDEADBEEF, (4 << 24) | 7)]:
        assert L.oracle_key_child(key, slot) == fmix(mix(key, slot) ^ 0x9E3779B9)


def test_root_keys_of_one_pixel_never_collide():
    keys = {key_root(12345, s) for s in range(20000)}
    assert len(keys) == 20000


def test_uniformity_and_independence():
    L = oracle_api.lib()
    u = np.array([[L.oracle_rand(7, s, 1234, d) for d in range(4)] for s in range(20000)])
    assert u.min() >= 0.0 and u.max() < 1.0
    np.testing.assert_allclose(u.mean(0), 0.5, atol=0.01)
    np.testing.assert_allclose(u.var(0), 1 / 12, atol=0.003)
    c = np.corrcoef(u.T)
    assert np.abs(c - np.eye(4)).max() < 0.03
    # neighbouring pixels with the same sample index are unrelated
    a = np.array([L.oracle_rand(7, 3, p, 2) for p in range(20000)])
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 0.03

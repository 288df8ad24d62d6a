"""The numeric contract (include/chroma_math.h) against double-precision NumPy, and the
Philox4x32-10 generator against the published Random123 known-answer vectors."""
import numpy as np
import pytest


def ulp_err(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.spacing(np.abs(ref32)).astype(np.float64)
    return np.abs(got.astype(np.float64) - ref64) / np.maximum(ulp, 1e-45)


@pytest.fixture(scope='module')
def rng():
    return np.random.default_rng(1234)


def test_log_exp(oracle_mod, rng):
    x = rng.uniform(0, 1, 200000).astype(np.float32)
    x[x == 0] = 1e-10
    assert ulp_err(oracle_mod.math_fn('log', x), np.log(x.astype(np.float64))).max() < 1.5
    x = (10 ** rng.uniform(-37, 38, 100000)).astype(np.float32)
    assert ulp_err(oracle_mod.math_fn('log', x), np.log(x.astype(np.float64))).max() < 1.5
    x = rng.uniform(-87, 88, 200000).astype(np.float32)
    assert ulp_err(oracle_mod.math_fn('exp', x), np.exp(x.astype(np.float64))).max() < 1.5
    special = np.array([0.0, -1.0, np.inf, 1.0], dtype=np.float32)
    out = oracle_mod.math_fn('log', special)
    assert out[0] == -np.inf and np.isnan(out[1]) and out[2] == np.inf and out[3] == 0.0
    assert oracle_mod.math_fn('exp', np.array([-200.0, 200.0, 0.0], np.float32)).tolist() == [0.0, np.inf, 1.0]


def test_trig(oracle_mod, rng):
    x = rng.uniform(-20, 20, 200000).astype(np.float32)
    x64 = x.astype(np.float64)
    assert np.abs(oracle_mod.math_fn('sin', x) - np.sin(x64)).max() < 1.5e-7
    assert np.abs(oracle_mod.math_fn('cos', x) - np.cos(x64)).max() < 1.5e-7
    t = oracle_mod.math_fn('tan', x)
    assert np.max(np.abs(t - np.tan(x64)) / np.abs(np.tan(x64))) < 5e-7


def test_inverse_trig(oracle_mod, rng):
    x = rng.uniform(-1, 1, 200000).astype(np.float32)
    x64 = x.astype(np.float64)
    assert ulp_err(oracle_mod.math_fn('asin', x), np.arcsin(x64)).max() < 3.0
    assert ulp_err(oracle_mod.math_fn('acos', x), np.arccos(x64)).max() < 2.0
    # out of range -> NaN: this is how total internal reflection is detected (photon.h:314,335)
    out = oracle_mod.math_fn('asin', np.array([1.0000001, -1.5, 1.0, -1.0], np.float32))
    assert np.isnan(out[0]) and np.isnan(out[1]) and out[2] == np.float32(np.pi / 2) and out[3] == -np.float32(np.pi / 2)
    y = rng.normal(size=100000).astype(np.float32)
    x2 = rng.normal(size=100000).astype(np.float32)
    assert np.abs(oracle_mod.math_fn('atan2', y, x2) - np.arctan2(y.astype(np.float64), x2.astype(np.float64))).max() < 4e-7


def test_contract_vs_libm_variant(oracle_mod, rng):
    x = rng.uniform(1e-6, 1, 10000).astype(np.float32)
    a = oracle_mod.math_fn('log', x)
    b = oracle_mod.math_fn('log', x, variant='libm')
    assert np.max(np.abs(a - b) / np.abs(b)) < 3e-7


def test_philox_known_answers(oracle_mod):
    # Random123 kat_vectors, philox4x32 with 10 rounds
    kat = [
        ([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
        ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
        ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
         [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]),
    ]
    for ctr, key, expect in kat:
        assert oracle_mod.philox(ctr, key).tolist() == expect


def test_uniform_stream(oracle_mod):
    u = oracle_mod.uniform_stream(12345, 77, 4096)
    assert u.min() > 0.0 and u.max() <= 1.0            # curand_uniform's (0, 1]
    assert abs(u.mean() - 0.5) < 0.02
    # resuming mid-stream gives the same numbers: the per-photon counter is all the state
    assert np.array_equal(oracle_mod.uniform_stream(12345, 77, 96, start=1000), u[1000:1096])
    # word -> float mapping of curand_uniform: x * 2^-32 + 2^-33
    words = np.array([0, 1, 0x7fffffff, 0xffffffff], dtype=np.uint32)
    got = oracle_mod.math_fn('uniform', words.view(np.float32))
    expect = (words.astype(np.float32) * np.float32(2.0 ** -32) + np.float32(2.0 ** -33)).astype(np.float32)
    assert np.array_equal(got, expect) and got[-1] == 1.0 and got[0] > 0
    # different photons and seeds decorrelate
    v = oracle_mod.uniform_stream(12345, 78, 4096)
    assert abs(np.corrcoef(u, v)[0, 1]) < 0.05

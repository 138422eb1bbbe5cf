"""The fixed-arithmetic sin/cos/exp/rsqrt/pow5 of the oracle against float64 references."""
import numpy as np


def test_sincos_accuracy(oracle):
    x = np.linspace(0.0, 2 * 3.141592, 200001, dtype=np.float32)
    s, c = oracle.sincosf(x)
    xs = x.astype(np.float64)
    assert np.abs(s - np.sin(xs)).max() < 2.5e-7
    assert np.abs(c - np.cos(xs)).max() < 2.5e-7
    x2 = np.linspace(-50.0, 50.0, 100001, dtype=np.float32)
    s2, c2 = oracle.sincosf(x2)
    assert np.abs(s2 - np.sin(x2.astype(np.float64))).max() < 1e-5
    assert np.abs(c2 - np.cos(x2.astype(np.float64))).max() < 1e-5


def test_sincos_quadrants(oracle):
    s, c = oracle.sincosf(np.array([0.0, np.pi / 2, np.pi, 3 * np.pi / 2, 2 * np.pi], np.float32))
    assert np.allclose(s, [0, 1, 0, -1, 0], atol=1e-6) and np.allclose(c, [1, 0, -1, 0, 1], atol=1e-6)


def test_expf(oracle):
    x = np.linspace(-87.0, 20.0, 200001, dtype=np.float32)
    y = oracle.expf(x)
    ref = np.exp(x.astype(np.float64))
    assert (np.abs(y - ref) / ref).max() < 3e-7
    assert oracle.expf(np.array([-100.0], np.float32))[0] == 0.0
    assert oracle.expf(np.array([0.0], np.float32))[0] == 1.0
    assert np.isinf(oracle.expf(np.array([100.0], np.float32))[0])


def test_rsqrt_pow5(oracle):
    x = np.random.default_rng(0).random(10000).astype(np.float32) + np.float32(1e-3)
    assert np.array_equal(oracle.rsqrtf(x), (np.float32(1.0) / np.sqrt(x)).astype(np.float32))
    x2 = x * x
    assert np.array_equal(oracle.pow5(x), (x2 * x2) * x)

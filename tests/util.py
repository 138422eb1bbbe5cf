import numpy as np


def assert_bits_equal(a, b, what=""):
    """Bit-exact float comparison; NaNs must coincide (payload / sign of a NaN is not compared)."""
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    na, nb = np.isnan(a), np.isnan(b)
    assert np.array_equal(na, nb), "%s: NaN masks differ (%d vs %d)" % (what, na.sum(), nb.sum())
    ia, ib = a.view(np.uint32)[~na], b.view(np.uint32)[~nb]
    bad = np.flatnonzero(ia != ib)
    if bad.size:
        fa, fb = a[~na].ravel()[bad[:5]], b[~nb].ravel()[bad[:5]]
        raise AssertionError("%s: %d of %d values differ bitwise, e.g. %s vs %s" % (what, bad.size, ia.size, fa, fb))


def random_rays(rng, n, box=((-1.8, -1.0, -3.3), (1.8, 1.0, 1.0))):
    lo, hi = np.array(box[0]), np.array(box[1])
    o = lo + (hi - lo) * rng.random((n, 3))
    d = rng.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return np.concatenate([o, d], 1).astype(np.float32)

"""pt_trace.h: slab_hit (round 3) decides `tmax >= tmin && tmax > 0` with ONE compare: tmin's lower clamp -1e30 becomes the smallest
positive float, so tmin' = max(tmin, FLT_TRUE_MIN) and `tmax >= tmin'` is the conjunction. The identity is arithmetic, not a property
of the GPU: checked here in IEEE binary32 with NaN-dropping min / max (numpy fmin / fmax = v_min_f32 / v_max_f32) on random and on
adversarial operands — zeros of both signs, denormals, infinities, NaN products (0 * inf), boxes behind the origin. (The kernels run
with f32 denormals preserved, `.amdhsa_float_denorm_mode_32 3`, which tests/test_isa.py's compile would show otherwise.)"""
import numpy as np

F = np.float32
TRUE_MIN = np.float32(1.401298464e-45)


def _axis(mn, mx, o, inv):
    with np.errstate(all="ignore"):
        t1 = (mn - o) * inv
        t2 = (mx - o) * inv
    return np.fmin(t1, t2), np.fmax(t1, t2)


def _both(mn, mx, o, inv):
    lo = [None] * 3
    hi = [None] * 3
    for a in range(3):
        lo[a], hi[a] = _axis(mn[:, a], mx[:, a], o[:, a], inv[:, a])
    # slab(): aabbIntersect with the clamps of pt_trace.h
    tmn = np.fmax(F(-1e30), lo[0]); tmx = np.fmin(F(1e30), hi[0])
    for a in (1, 2):
        tmn = np.fmax(tmn, lo[a]); tmx = np.fmin(tmx, hi[a])
    ref = (tmx >= tmn) & (tmx > F(0.0))
    # slab_hit(): the lower clamp is the smallest positive float, one compare
    tmn2 = np.fmax(TRUE_MIN, lo[0]); tmx2 = np.fmin(F(1e30), hi[0])
    for a in (1, 2):
        tmn2 = np.fmax(tmn2, lo[a]); tmx2 = np.fmin(tmx2, hi[a])
    return ref, tmx2 >= tmn2


def test_one_compare_equals_the_conjunction_on_random_and_adversarial_operands():
    rng = np.random.default_rng(7)
    n = 400000
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 1e-38, -1e-38, 1e-30, 1.0, -1.0, 1e30, -1e30, 3e38, -3e38, np.inf, -np.inf, np.nan], dtype=F)

    def mix(shape, scale):
        x = (rng.standard_normal(shape) * scale).astype(F)
        m = rng.random(shape) < 0.15
        x[m] = special[rng.integers(0, len(special), int(m.sum()))]
        return x

    c = mix((n, 3), 2.0)
    e = np.abs(mix((n, 3), 1.0))
    with np.errstate(all="ignore"):
        mn, mx = c - e, c + e
    o = mix((n, 3), 3.0)
    d = mix((n, 3), 1.0)
    with np.errstate(all="ignore"):
        inv = (F(1.0) / d).astype(F)
    ref, one = _both(mn.astype(F), mx.astype(F), o, inv)
    assert ref.dtype == bool and np.array_equal(ref, one), int((ref != one).sum())
    assert 0.02 < ref.mean() < 0.98                                       # both outcomes are exercised
    # exactly-zero and denormal tmax: the cases the second compare exists for
    mn1 = np.array([[0.0, -1.0, -1.0], [1e-45, -1.0, -1.0], [-1.0, -1.0, -1.0]], dtype=F)
    mx1 = np.array([[0.0, 1.0, 1.0], [1e-45, 1.0, 1.0], [0.0, 1.0, 1.0]], dtype=F)
    o1 = np.zeros((3, 3), dtype=F)
    inv1 = np.ones((3, 3), dtype=F)
    ref1, one1 = _both(mn1, mx1, o1, inv1)
    assert list(ref1) == [False, True, False] and list(one1) == [False, True, False]


def test_u_and_v_non_negative_as_one_compare_on_v_min():
    """moller_trumbore_sel: `u >= 0 && v >= 0` is `min(u, v) >= 0` although v_min drops a NaN operand — the NaN then reaches
    `u + v <= 1`, which fails. All pairs of a set of special and ordinary values."""
    vals = np.array([0.0, -0.0, 1e-45, -1e-45, 0.25, 0.5, 0.75, 1.0, 1.0000001, -0.25, -1.0, 2.0, 3e38, -3e38, np.inf, -np.inf, np.nan], dtype=F)
    u, v = np.meshgrid(vals, vals)
    with np.errstate(all="ignore"):
        ref = (u >= F(0.0)) & (v >= F(0.0)) & (u + v <= F(1.0))
        one = (np.fmin(u, v) >= F(0.0)) & (u + v <= F(1.0))
    assert np.array_equal(ref, one)
    assert ref.any() and not ref.all()

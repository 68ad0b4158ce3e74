"""Strict mode's perspective divide (tsar_device_math.h persp_divide_exact) against IEEE division, bit for bit.

The oracle divides with `/` (oracle/tsar_oracle.c, getCorrespondingPoint_cu gipuma.cu:161-171).  The HIP tap loops compute the two
quotients of a tap with one v_rcp_f32 + Newton step and one residual correction each, behind an operand guard.  That the short form
is correctly rounded inside the guard was enumerated over all 2^46 mantissa pairs (tools/div_exact.hip,
profiles/r03/div_exact_all_mantissa_pairs.json); here the SHIPPED code path is re-checked through the C ABI:
  * > 2^31 quotients on device-generated operands: drawn like the tap loop's, across the whole guard range (unguarded form too),
    and over every fp32 bit pattern (where the guard has to catch what the short form cannot do);
  * hand-picked edge cases against numpy's float32 division on the host — an IEEE division that is independent of the GPU's.
"""
import numpy as np
import pytest

from tsar_mvs_amd import api

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    m = api.Matcher()
    yield m
    m.close()


def test_tap_loop_operands_2_30_triples(ctx):
    bad, outside = ctx.selftest_divide_random(30, seed=1, mode=0)          # 2^31 quotients
    assert bad == 0
    assert outside < 2 ** 30 * 0.02      # |u| < 2^-16 or so: rare, and handled by the fallback
    bad, _ = ctx.selftest_divide_random(28, seed=2, mode=0)
    assert bad == 0


def test_guard_range_any_mantissa_guarded_and_unguarded(ctx):
    for seed, guarded in ((3, True), (4, False), (5, False)):
        bad, outside = ctx.selftest_divide_random(29, seed=seed, mode=1, guarded=guarded)
        assert outside == 0              # mode 1 draws inside the guard by construction
        assert bad == 0, (seed, guarded, bad)


def test_any_bit_pattern_takes_the_guard(ctx):
    bad, outside = ctx.selftest_divide_random(28, seed=6, mode=2)
    assert bad == 0
    assert outside > 2 ** 28 * 0.5       # most random bit patterns are outside [2^-20, 2^38]: the fallback is what is being tested


def _edge_operands():
    f = np.float32
    tiny, huge = f(1e-45), f(3e38)       # smallest denormal, near FLT_MAX
    specials = [f(0.0), f(-0.0), tiny, -tiny, f(1e-39), f(1.1754944e-38), f(2 ** -21), f(2 ** -20), np.nextafter(f(2 ** -20), f(0)),
                f(2 ** 38), np.nextafter(f(2 ** 38), f(np.inf)), f(2 ** 39), f(1e30), huge, -huge, f(np.inf), f(-np.inf), f(np.nan),
                f(1.0), f(-1.0), f(3.0), f(1.9999999), f(1.0000001), f(6047.5), f(-0.25), f(0.1), f(4031.99)]
    X, Y, Z = [], [], []
    for a in specials:
        for b in specials:
            for c in (f(1.0), f(-3.0), f(0.7), f(1e-7), f(2 ** 38), f(2 ** -20), f(0.0), f(np.inf), tiny, f(2 ** 39)):
                X.append(a); Y.append(c); Z.append(b)
                X.append(c); Y.append(a); Z.append(b)
    # negative Z inside the guard (a plane seen from behind), all sign combinations
    rng = np.random.default_rng(7)
    n = 1 << 16
    zz = -np.exp2(rng.uniform(-19.9, 37.9, n)).astype(f)
    xx = (np.exp2(rng.uniform(-19.9, 37.9, n)) * rng.choice([-1, 1], n)).astype(f)
    yy = (np.exp2(rng.uniform(-19.9, 37.9, n)) * rng.choice([-1, 1], n)).astype(f)
    return (np.concatenate([np.array(X, f), xx]), np.concatenate([np.array(Y, f), yy]), np.concatenate([np.array(Z, f), zz]))


def _same(a, b):
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def test_edge_cases_against_host_division(ctx):
    X, Y, Z = _edge_operands()
    # a wave of the kernel shares one fallback decision: shuffle so that in-guard and out-of-guard triples share waves, and
    # also run the in-guard subset alone so that whole waves take the short form
    perm = np.random.default_rng(8).permutation(X.size)
    X, Y, Z = X[perm], Y[perm], Z[perm]
    with np.errstate(all="ignore"):
        u_ref, v_ref = X / Z, Y / Z
    u, v = ctx.selftest_divide(X, Y, Z)
    assert _same(u, u_ref).all() and _same(v, v_ref).all()
    u_dev, v_dev = ctx.selftest_divide(X, Y, Z, ieee=True)                 # the device's own `/` agrees with the host's as well
    assert _same(u_dev, u_ref).all() and _same(v_dev, v_ref).all()
    mag = np.stack([np.abs(X), np.abs(Y), np.abs(Z)])
    inside = (mag.min(0) >= np.float32(2 ** -20)) & (mag.max(0) <= np.float32(2 ** 38))
    assert inside.sum() > 60000 and (~inside).sum() > 1000
    u, v = ctx.selftest_divide(X[inside], Y[inside], Z[inside])
    assert _same(u, u_ref[inside]).all() and _same(v, v_ref[inside]).all()


def test_cost_tail_square_root_is_correctly_rounded_for_every_mantissa():
    """sqrt_rsq_exact (tsar_device_math.h: v_rsq_f32 + one fused residual correction, the square root of pmCost's tail in both
    arithmetic modes) against sqrtf on the device: every mantissa of two adjacent binades — both exponent parities, 2^24 inputs, the
    enumeration the claim rests on — and 2^24 random inputs per seed with exponents across its range; the control (no correction
    step) must report mismatches, or the comparison would discriminate nothing"""
    m = api.Matcher()
    assert m.selftest_sqrt(0) == 0
    for seed in (1, 2, 3):
        assert m.selftest_sqrt(1, seed) == 0
    assert m.selftest_sqrt(2) > 1000
    for seed in (4, 5):
        assert m.selftest_sqrt(3, seed) == 0            # the production operand range [1e-10, 4.3e9], binade by binade (the per-context probe's second pass)
    m.close()


def test_float_imagery_contexts_are_not_gated_on_the_square_root_probe():
    """only the 8-bit tap loops run sqrt_rsq_exact; a context fed non-integral images keeps sqrtf and must not run (or be refused by)
    the probe: exact_sqrt_probe stays unprobed — observable as set_views succeeding with the probe's kernels absent from the timing"""
    import numpy as np
    from tsar_mvs_amd import synth
    sc = synth.make_scene(96, 64, 2, seed=4)
    m = api.Matcher()
    m.set_params(api.default_params(box_hsize=11, box_vsize=11, n_best=1, depth_min=sc.depth_min, depth_max=sc.depth_max))
    imgs = [im.numpy() + np.float32(0.25) for im in sc.images]           # not an 8-bit decode: the float tap loop
    m.set_views(imgs, sc.K, sc.R, sc.t)
    m.pm_init()
    m.pm_iterate(1)
    assert np.isfinite(m.get_plane()[1]).all()
    m.close()

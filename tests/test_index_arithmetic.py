"""Stage 2 of the K3 tight kernel, restated in NumPy (CPU, no GPU): the pixel index by rounding addition + saturating pack +
16-bit dot product, and the tie / bound test on doubled coordinates with fract -- against the reference's own arithmetic
(IH:362-371: xi = clip(round(u), 0, W - 1), half to even; OPS:285-290 likewise) and against the three-issue form the compacted
set keeps.  float64 addition in NumPy rounds to nearest-even exactly like v_add_f64 / v_fma_f64; v_fract_f64 is x - floor(x)
clamped below 1.  The GPU suite checks the kernels themselves; this pins the reasoning they rest on."""
import numpy as np
import pytest

G = 1e-6                      # kGuardPx
MAGIC = 6755399441055744.0    # 1.5 * 2^52


def _cases(rng, W):
    hw = W // 2
    u = [rng.uniform(-0.5, W + 0.5, 200000)]
    k = rng.integers(-1, W + 1, 20000).astype(np.float64)
    for d in (0.0, 1e-12, 1e-9, 0.9e-6, 1.1e-6, 1e-4, 0.25, 0.49, 0.4999999):
        u += [k + 0.5 + d, k + 0.5 - d, k + d, k - d]
    u += [np.array([-G * 0.999, -1e-300, -0.0, 0.0, 1e-300, W - 0.5, W - 0.5 + 1e-9, W - 1e-9, float(W), W + G * 0.999, W - 0.75])]
    u = np.concatenate(u)
    u = u[(u > -G) & (u < W + G)]                 # what stage 1 lets through ("in view up to the guard")
    return u, u - hw                              # image coordinate, the kernel's centred coordinate (hw an integer: exact)


def _fract(x):
    f = x - np.floor(x)
    return np.where(f >= 1.0, np.nextafter(1.0, 0.0), f)


def _risky_fract(uc2, vc2):
    wu, wv = _fract(uc2) - 0.5, _fract(vc2) - 0.5
    return ~(np.maximum(np.abs(wu), np.abs(wv)) < 0.5 - 2.0 * G)


def _risky_three(uc, vc, ru, rv):
    wu, wv = np.abs(uc - ru) - 0.25, np.abs(vc - rv) - 0.25
    return ~(np.maximum(np.abs(wu), np.abs(wv)) < 0.25 - G)


# (644, 482): 32 768 - H / 2 is odd there -- exact ties round to the OTHER neighbour than rint's, inside the guard band
@pytest.mark.parametrize("W,H", [(640, 480), (128, 96), (1296, 968), (32752, 16), (16, 32764), (644, 482)])
def test_rounding_addition_saturating_pack_and_dot(W, H):
    rng = np.random.default_rng(W * 7 + H)
    u, uc = _cases(rng, W)
    v, vc = _cases(rng, H)
    n = min(u.size, v.size)
    u, uc, v, vc = u[:n], uc[:n], rng.permutation(v)[:n], None
    vc = v - H // 2
    bias_x, bias_y = 32767 - (W - 1), 32767 - (H - 1)
    mu, mv = MAGIC + (W // 2 + bias_x), MAGIC + (H // 2 + bias_y)
    for doubled in (False, True):
        if doubled:                                                   # fma(2u, 0.5, magic): the scaling is exact, one rounding
            tu, tv = (2.0 * uc) * 0.5 + mu, (2.0 * vc) * 0.5 + mv
        else:
            tu, tv = uc + mu, vc + mv
        lo_x = (tu.view(np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        lo_y = (tv.view(np.int64) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        px = np.clip(lo_x, -32768, 32767).astype(np.int64)           # v_cvt_pk_i16_i32
        py = np.clip(lo_y, -32768, 32767).astype(np.int64)
        xi, yi = px - bias_x, py - bias_y                            # pk - pk_bias, no borrow: both halves >= their bias
        assert (px >= bias_x).all() and (py >= bias_y).all()
        ru, rv = tu - mu, tv - mv                                    # exact
        tie_u, tie_v = np.abs(uc - np.floor(uc) - 0.5) == 0.0, np.abs(vc - np.floor(vc) - 0.5) == 0.0
        assert np.array_equal(ru[~tie_u], np.rint(uc)[~tie_u]) and np.array_equal(rv[~tie_v], np.rint(vc)[~tie_v])
        assert (np.abs(uc - ru) <= 0.5).all() and (np.abs(vc - rv) <= 0.5).all()      # ties: either neighbour (guarded below)
        ref_x = np.clip(np.rint(u), 0, W - 1).astype(np.int64)       # the reference's index
        ref_y = np.clip(np.rint(v), 0, H - 1).astype(np.int64)
        risky = _risky_fract(2.0 * uc, 2.0 * vc) if doubled else _risky_three(uc, vc, ru, rv)
        bad = ((xi != ref_x) | (yi != ref_y)) & ~risky
        assert not bad.any(), (u[bad][:5], v[bad][:5])
        # byte offset of the gather: one v_dot2_u32_u16 of the packed pair with (2, 2 W) plus a constant, modulo 2^32
        dw2 = 2 * W
        dot_c = (-(2 * bias_x + dw2 * bias_y)) % (1 << 32)
        off = (px * 2 + py * dw2 + dot_c) % (1 << 32)
        assert np.array_equal(off, yi * dw2 + 2 * xi)
        assert (off < 2 * W * (H + 1)).all()                          # row H (guard band's outer edge) is past the frame: dropped


def test_fract_guard_is_the_three_issue_guard():
    rng = np.random.default_rng(5)
    u, uc = _cases(rng, 640)
    v, vc = _cases(rng, 480)
    n = min(u.size, v.size)
    uc, vc = uc[:n], rng.permutation(vc)[:n]
    a = _risky_fract(2.0 * uc, 2.0 * vc)
    b = _risky_three(uc, vc, np.rint(uc), np.rint(vc))
    du = np.minimum(np.abs(uc - np.rint(uc)), 0.5 - np.abs(uc - np.rint(uc)))    # distance from an integer or a tie
    dv = np.minimum(np.abs(vc - np.rint(vc)), 0.5 - np.abs(vc - np.rint(vc)))
    d = np.minimum(du, dv)
    assert a[d < 0.999 * G].all() and b[d < 0.999 * G].all()         # everything inside the band is re-evaluated
    assert not a[d > 1.001 * G].any() and not b[d > 1.001 * G].any()  # and nothing clear of it
    assert (a == b)[(d < 0.999 * G) | (d > 1.001 * G)].all()

"""Device IEEE behaviour == host IEEE behaviour, bit for bit: correctly rounded divide and sqrt, single-rounding fma,
kept denormals, and the build-defined helper functions (sincos2pi, sky, pack).  This is what makes bit-exact parity of
whole images possible; a difference here would show up as scattered LSB noise in the image tests."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def inputs(n=1 << 18, seed=11):
    rng = np.random.RandomState(seed)
    a = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.randint(-6, 7, n).astype(np.float32)
    b = rng.standard_normal(n).astype(np.float32) * np.float32(10.0) ** rng.randint(-6, 7, n).astype(np.float32)
    # edge values: denormals, huge, tiny, signed zeros, exact powers of two, values around 1
    edge = np.float32([0.0, -0.0, 1e-45, -1e-45, 1e-40, 1.1754942e-38, 1.1754944e-38, 3.4028235e38, -3.4028235e38,
                       1.0, -1.0, 0.99999994, 1.0000001, 0.5, 2.0, 3.0, 1e-20, 1e20, 0.1, 0.7])
    ea, eb = np.meshgrid(edge, edge)
    a[: ea.size] = ea.reshape(-1)
    b[: eb.size] = eb.reshape(-1)
    # x/0, 0/0 stay in: inf and nan are part of the contract
    return a, b


def same_bits(x, y):
    x, y = np.ascontiguousarray(x), np.ascontiguousarray(y)
    nan = np.isnan(x) & np.isnan(y)
    return ((x.view(np.uint32) == y.view(np.uint32)) | nan)


def test_div_sqrt_fma_are_ieee(renderer, oracle):
    a, b = inputs()
    div, sq, fm, cs, sn, sk, pk = renderer.debug_arith(a, b)
    odiv, osq, ofm, ocs, osn, osk, opk = oracle.arith(a, b)
    for name, got, want in (("div", div, odiv), ("sqrt", sq, osq), ("fma", fm, ofm)):
        ok = same_bits(got, want)
        assert ok.all(), "%s: %d / %d differ, e.g. a=%r b=%r got=%r want=%r" % (
            name, (~ok).sum(), ok.size, a[~ok][0], b[~ok][0], got[~ok][0], want[~ok][0])
    # denormal results survive (no flush to zero on the device)
    tiny = np.float32([1e-30, 3e-39]), np.float32([1e10, 2.0])
    d = renderer.debug_arith(*tiny)[0]
    assert d[0] == np.float32(1e-30) / np.float32(1e10) and d[0] != 0 and d[1] != 0


def test_helper_functions_match_the_oracle(renderer, oracle):
    a, b = inputs(seed=12)
    _, _, _, cs, sn, sk, pk = renderer.debug_arith(a, b)
    _, _, _, ocs, osn, osk, opk = oracle.arith(a, b)
    assert same_bits(cs, ocs).all() and same_bits(sn, osn).all()
    finite = np.isfinite(a) & np.isfinite(b)
    assert same_bits(sk[finite], osk[finite]).all()           # float sky == the reference's double-then-float sky
    ok = np.isfinite(a) & np.isfinite(b)
    assert (pk[ok] == opk[ok]).all()

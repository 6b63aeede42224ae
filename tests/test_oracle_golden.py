"""Pin the CPU oracle (oracle/tv_oracle.py) against golden vectors captured from the reference
(tests/golden/make_golden.py) and against the reference's published known answers
(README.md:76-93; examples/b_TV_discretizations_math.ipynb 5x5 impulse)."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN, SCHEMES
from oracle import tv_oracle as orc

TOL = dict(rtol=1e-12, atol=1e-12)


def _load(scheme):
    return np.load(os.path.join(GOLDEN, "ops_%s.npz" % scheme))


def _cases(scheme):
    z = _load(scheme)
    for name in z["case_names"]:
        name = str(name)
        lz, mu, factor = z[name + "/params"]
        mask = z[name + "/mask"]
        mask = False if mask.ndim == 0 else mask
        kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=factor)
        yield name, z, kw


def _tol_for(arr):
    # fp32 cases: the reference itself mixes fp32 storage with fp64 scalars
    return dict(rtol=2e-6, atol=2e-6) if arr.dtype == np.float32 else TOL


@pytest.mark.parametrize("scheme", SCHEMES)
def test_D_matches_reference_golden(scheme):
    for name, z, kw in _cases(scheme):
        x = z[name + "/x"]
        got = orc.D(x, scheme, **kw)
        want = z[name + "/D"]
        assert got.shape == want.shape, (scheme, name)
        np.testing.assert_allclose(got, want, err_msg="%s %s" % (scheme, name), **_tol_for(x))


@pytest.mark.parametrize("scheme", SCHEMES)
def test_DT_matches_reference_golden(scheme):
    for name, z, kw in _cases(scheme):
        y = z[name + "/y"]
        np.testing.assert_allclose(orc.D_T(y, scheme, **kw), z[name + "/DT"],
                                   err_msg="%s %s" % (scheme, name), **_tol_for(y))
        np.testing.assert_allclose(orc.D_T(z[name + "/D"], scheme, **kw), z[name + "/DTD"],
                                   err_msg="%s %s DTD" % (scheme, name), **_tol_for(y))


@pytest.mark.parametrize("scheme", SCHEMES)
def test_L21_matches_reference_golden(scheme):
    for name, z, kw in _cases(scheme):
        l21, norms = orc.compute_L21_norm(z[name + "/D"], return_array=True)
        np.testing.assert_allclose(l21, z[name + "/l21"], rtol=1e-13)
        np.testing.assert_allclose(norms, z[name + "/norms"], **TOL)
        assert np.isclose(orc.compute_L21_norm(z[name + "/D"]), z[name + "/l21"], rtol=1e-13)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_tv_and_subgradient_match_reference_golden(scheme):
    for name, z, kw in _cases(scheme):
        x = z[name + "/x"]
        tv, G, gn = orc.tv(x, scheme, return_grad_norms=True, **kw)
        tol = _tol_for(x)
        np.testing.assert_allclose(tv, z[name + "/tv"], rtol=tol["rtol"])
        np.testing.assert_allclose(G, z[name + "/G"], err_msg="%s %s" % (scheme, name), **tol)
        want_gn = z[name + "/grad_norms"]
        assert np.array_equal(np.isinf(gn), np.isinf(want_gn)), (scheme, name)
        assert np.isinf(want_gn).any(), "fixture should exercise the |D|==0 guard"
        fin = np.isfinite(want_gn)
        np.testing.assert_allclose(gn[fin], want_gn[fin], **tol)


def test_readme_known_answer_hybrid():
    # README.md:88-91 prints 532166.8251801673 for seed 0, rand(20,4,100,100)
    np.random.seed(0)
    x = np.random.rand(20, 4, 100, 100)
    tv, _ = orc.tv(x, "hybrid")
    assert abs(tv - 532166.8251801673) < 1e-6


@pytest.mark.parametrize("scheme", SCHEMES)
@pytest.mark.parametrize("tag,mu", [("mu0", 0.0), ("mu2m5", 2 ** -5)])
def test_readme_input_all_schemes(scheme, tag, mu):
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["readme_%s_%s" % (scheme, tag)]
    np.random.seed(0)
    x = np.random.rand(20, 4, 100, 100)
    tv, G = orc.tv(x, scheme, reg_time=mu)
    assert abs(tv - ka["tv"]) <= 1e-12 * abs(ka["tv"])
    np.testing.assert_allclose(G.sum(), ka["G_sum"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(np.abs(G).sum(), ka["G_abs_sum"], rtol=1e-12)
    np.testing.assert_allclose((G * G).sum(), ka["G_sq_sum"], rtol=1e-12)
    probe = G[[0, 7, 19, 3], [0, 1, 3, 2], [0, 50, 99, 17], [0, 31, 99, 64]]
    np.testing.assert_allclose(probe, ka["G_probe"], rtol=1e-12, atol=1e-13)


@pytest.mark.parametrize("scheme,tv_expected", [("upwind", 2 + np.sqrt(2)), ("downwind", 2 + np.sqrt(2)),
                                                 ("central", 2.0), ("hybrid", 3 * np.sqrt(2))])
def test_impulse_5x5(scheme, tv_expected):
    # examples/b_TV_discretizations_math.ipynb:46-152 (downwind matrix from the code)
    ka = json.load(open(os.path.join(GOLDEN, "known_answers.json")))["impulse5_" + scheme]
    A = np.zeros((1, 1, 5, 5))
    A[0, 0, 2, 2] = 1.0
    tv, G = orc.tv(A, scheme)
    assert abs(tv - tv_expected) < 1e-14
    assert abs(ka["tv"] - tv_expected) < 1e-14
    np.testing.assert_allclose(G[0, 0], np.array(ka["G"]), rtol=1e-14, atol=1e-15)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_readme_loops_2d(scheme):
    z = np.load(os.path.join(GOLDEN, "trajectories_2d.npz"))
    noisy = z["noisy"]
    _, nb_it, reg, step = z["params"]
    # sub-gradient descent is a discontinuous map (sign-like g = D/|D|): round-off level
    # differences in summation order are amplified after ~100 iterations, so the tail of the
    # trajectory is compared loosely and the head tightly
    x, loss = orc.subgradient_descent(noisy, int(nb_it), reg, step, scheme=scheme)
    np.testing.assert_allclose(loss[:60], z["gd_loss_" + scheme][:60], rtol=1e-11)
    np.testing.assert_allclose(loss, z["gd_loss_" + scheme], rtol=1e-3)
    x, loss = orc.chambolle_pock(noisy, int(nb_it), reg, scheme=scheme, tau=1 / 9)
    np.testing.assert_allclose(loss, z["cp_loss_" + scheme], rtol=1e-10)
    np.testing.assert_allclose(x, z["cp_final_" + scheme], rtol=1e-8, atol=1e-8)


def load_trajectory_512():
    """inputs of tests/golden/trajectory_512.npz: the stored phantom plus the seeded noise (legacy RandomState: bit-stable)"""
    z = np.load(os.path.join(GOLDEN, "trajectory_512.npz"))
    truth = z["truth"]
    noise_level, nb_it, reg, step, seed = z["params"]
    noisy = truth + noise_level * np.random.RandomState(int(seed)).rand(*truth.shape)
    assert noisy.shape == (1, 1, 512, 512) and noisy.sum() == float(z["noisy_checksum"])
    return z, noisy, int(nb_it), float(reg), float(step)


def test_readme_subgradient_loop_at_config0_size():
    """BASELINE configs[0] is a 512 x 512 image: the reference's own 300-iteration hybrid loop at that size
    (make_golden.py gen_trajectory_512) against the oracle.  Same head / tail bounds as the 64 x 64 case."""
    z, noisy, nb_it, reg, step = load_trajectory_512()
    x, loss = orc.subgradient_descent(noisy, nb_it, reg, step, scheme="hybrid")
    want = z["gd_loss_hybrid"]
    np.testing.assert_allclose(loss[:60], want[:60], rtol=1e-11)
    np.testing.assert_allclose(loss, want, rtol=1e-3)
    assert np.all(np.diff(loss) < 0)
    assert abs(x.mean() - float(z["gd_final_mean"])) < 1e-3
    np.testing.assert_allclose(x[0, 0, 200], z["gd_final_row"], atol=2.0)       # chaotic tail: pixels move by O(step * reg)


# ---- structural invariants restated from pytv/tests.py ------------------------------------
GEOMS = [((1, 1, 12, 12), 1.0, 0.0), ((6, 1, 10, 10), 1.0, 0.0), ((6, 1, 10, 10), 0.0, 0.0)] + \
        [((1, m, 9, 9), 1.0, 1.0) for m in (2, 3, 4, 8)] + \
        [((5, m, 9, 9), 1.0, 1.0) for m in (2, 3, 4, 8)] + \
        [((5, m, 9, 9), 0.0, 1.0) for m in (2, 3, 4, 8)]


@pytest.mark.parametrize("scheme", SCHEMES)
def test_adjointness(scheme):
    # pytv/tests.py:363-404: <Y, D X> == <X, D^T Y>
    rng = np.random.default_rng(5)
    for shape, lz, mu in GEOMS:
        x = rng.standard_normal(shape)
        Dx = orc.D(x, scheme, lz, mu)
        y = rng.standard_normal(Dx.shape)
        lhs = np.sum(y * Dx)
        rhs = np.sum(x * orc.D_T(y, scheme, lz, mu))
        assert abs(lhs - rhs) <= 1e-11 * max(1.0, abs(lhs)), (scheme, shape, lz, mu)


@pytest.mark.parametrize("scheme", SCHEMES)
def test_2d_vs_3d_tiling(scheme):
    # pytv/tests.py:187-245 with explicit reshapes (the reference's harness builds a ragged list)
    rng = np.random.default_rng(6)
    img = rng.standard_normal((1, 1, 11, 11))
    Nz = 5
    vol = np.tile(img, (Nz, 1, 1, 1))
    tv2, G2 = orc.tv(img, scheme)
    tv3, G3 = orc.tv(vol, scheme)
    assert abs(tv3 / Nz - tv2) < 1e-10
    np.testing.assert_allclose(G3[1], G2[0], rtol=1e-12, atol=1e-12)
    D2 = orc.D(img, scheme, reg_z_over_reg=0)
    D3 = orc.D(vol, scheme, reg_z_over_reg=0)
    np.testing.assert_allclose(D3[1], D2[0], rtol=0, atol=0)
    np.testing.assert_allclose(orc.D_T(D3, scheme, reg_z_over_reg=0)[1],
                               orc.D_T(D2, scheme, reg_z_over_reg=0)[0], rtol=1e-13, atol=1e-13)


def test_channel_counts():
    assert orc.num_channels("hybrid", 1, 1) == 4
    assert orc.num_channels("hybrid", 5, 1) == 6
    assert orc.num_channels("hybrid", 5, 3, 1.0, 1.0) == 8
    assert orc.num_channels("hybrid", 5, 3, 0.0, 1.0) == 6
    assert orc.num_channels("upwind", 5, 3, 1.0, 0.0) == 3
    assert orc.num_channels("central", 1, 3, 1.0, 0.5) == 3


@pytest.mark.parametrize("scheme", SCHEMES)
def test_weight_map_extension_reduces_to_the_reference_mask(scheme):
    """BUILD EXTENSION (the reference's to-do, README.md:258): a float ``mask_static`` is a per-pixel weight map of the time
    regularisation.  W = where(mask, factor, 1) must give the reference's boolean-mask outputs (golden vectors)."""
    z = np.load(os.path.join(GOLDEN, "ops_%s.npz" % scheme))
    done = 0
    for name in z["case_names"]:
        name = str(name)
        mask = z[name + "/mask"]
        if mask.ndim == 0:
            continue
        lz, mu, factor = z[name + "/params"]
        kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=np.where(mask, factor, 1.0))
        x, y = z[name + "/x"].astype(np.float64), z[name + "/y"].astype(np.float64)
        tol = 1e-12 if z[name + "/x"].dtype == np.float64 else 1e-5
        np.testing.assert_allclose(orc.D(x, scheme, **kw), z[name + "/D"], rtol=tol, atol=tol)
        np.testing.assert_allclose(orc.D_T(y, scheme, **kw), z[name + "/DT"], rtol=tol, atol=tol)
        tv, G = orc.tv(x, scheme, **kw)
        np.testing.assert_allclose(tv, z[name + "/tv"], rtol=tol)
        np.testing.assert_allclose(G, z[name + "/G"], rtol=tol, atol=tol)
        done += 1
    assert done >= 2

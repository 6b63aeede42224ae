"""Validate the oracle against the REAL reference, imported from /root/reference.

Runs only in the authoring container (skipped on the GPU box, where the reference does not
exist).  The reference is imported in a subprocess so that its package name (``pytv``) never
collides with the product package of the same name."""
import os
import subprocess
import sys

import pytest

REF = "/root/reference"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, itertools
import numpy as np
sys.path.insert(0, %(root)r)
sys.path.insert(0, %(ref)r)
import pytv
from oracle import tv_oracle as orc
assert pytv.__file__.startswith(%(ref)r)
rng = np.random.default_rng(11)
worst = 0.0
n = 0
geoms = [(1,1,9,9),(4,1,8,8),(3,1,7,7),(5,1,6,6),(1,3,7,7),(4,2,6,6),(3,3,6,6),(5,4,7,7),(6,8,5,5),
         (3,1,2,2),(4,3,2,2),(1,1,3,3),(3,3,4,4)]      # tiny frames: central has no interior row / column at N = 2
for scheme in ("upwind","downwind","central","hybrid"):
    for shape in geoms:
        for lz, mu in ((1.0,0.0),(0.0,0.0),(2.5,1.0),(1.0,2**-5),(0.0,0.7)):
            for use_mask in (False, True):
                if use_mask and not (mu > 0 and shape[1] > 1):
                    continue
                mask = (rng.random((1,1)+shape[2:]) > 0.4) if use_mask else False
                kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=3.0 if use_mask else 0)
                x = rng.standard_normal(shape)
                x[..., :2, :2] = 0.25
                Dr = getattr(pytv.tv_operators_CPU, "D_"+scheme)(x.copy(), **kw)
                Do = orc.D(x, scheme, **kw)
                assert Dr.shape == Do.shape, (scheme, shape, lz, mu)
                assert orc.num_channels(scheme, shape[0], shape[1], lz, mu) == Dr.shape[1]
                worst = max(worst, np.abs(Dr-Do).max())
                y = rng.standard_normal(Dr.shape)
                worst = max(worst, np.abs(getattr(pytv.tv_operators_CPU, "D_T_"+scheme)(y.copy(), **kw) - orc.D_T(y, scheme, **kw)).max())
                tr, Gr, nr = getattr(pytv.tv_CPU, "tv_"+scheme)(x.copy(), return_grad_norms=True, **kw)
                to, Go, no = orc.tv(x, scheme, return_grad_norms=True, **kw)
                worst = max(worst, abs(tr-to)/max(1.0,abs(tr)), np.abs(Gr-Go).max())
                assert np.array_equal(np.isinf(nr), np.isinf(no))
                n += 1
print("CASES", n, "WORST", worst)
assert worst < 1e-12, worst
'''


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "pytv")), reason="reference not mounted")
def test_oracle_equals_imported_reference():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", SCRIPT % dict(root=ROOT, ref=REF)], cwd="/tmp", env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "CASES" in out.stdout


# BASELINE configs[0] on its REAL workload (SURVEY section 8d / BASELINE.md section 2): the README's sub-gradient loop
# (README.md:107-124) on the reference's cameraman image, native 256 x 256 and the 2 x 2 Kronecker 512 x 512 variant,
# hybrid, 300 iterations, np.random.seed(0).  The image is the reference's data and cannot travel: container only.
CONFIG0 = r'''
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from oracle import tv_oracle as orc
cam = np.load(%(ref)r + "/pytv/media/cameraman.npy")
assert cam.shape == (256, 256) and cam.dtype == np.int64 and cam.min() == 7 and cam.max() == 253
want = {256: (97202967.0339645, 39074939.77692743), 512: (373250884.84215766, 139106604.86969954)}
for n, img in ((256, cam), (512, np.kron(cam, np.ones((2, 2), dtype=cam.dtype)))):
    np.random.seed(0)
    truth = np.reshape(img, (1, 1) + img.shape)
    noisy = truth + 100 * np.random.rand(*truth.shape)
    x, loss = orc.subgradient_descent(noisy, 300, 25, 5e-3, scheme="hybrid")
    # the first value to the last digit; the last one to 1e-5: the oracle follows the reference to 1e-15 for the first ~100
    # iterations, then the trajectories part ways (a sub-gradient flips sign at |Dx| ~ 1e-14 under a different summation
    # order: 9e-8 at iteration 200, 1.3e-6 at 299 on the 256 image)
    assert abs(loss[0] - want[n][0]) <= 1e-12 * want[n][0], (n, loss[0])
    assert abs(loss[299] - want[n][1]) <= 1e-5 * want[n][1], (n, loss[299])
    assert np.all(np.diff(loss) < 0)
    print("CONFIG0", n, repr(loss[0]), repr(loss[299]))
# and the reference itself still says the same on the native image (the numbers above are its outputs, BASELINE.md section 2)
sys.path.insert(0, %(ref)r)
import pytv
assert pytv.__file__.startswith(%(ref)r)
np.random.seed(0)
truth = np.reshape(pytv.utils.cameraman(), (1, 1, 256, 256))
noisy = truth + 100 * np.random.rand(*truth.shape)
est = np.copy(noisy)
first = last = None
x, loss = orc.subgradient_descent(noisy, 300, 25, 5e-3, scheme="hybrid")
for it in range(300):
    tv, G = pytv.tv_CPU.tv_hybrid(est)
    est += -5e-3 * ((est - noisy) + 25 * G)
    last = 0.5 * np.sum(np.square(est - noisy)) + 25 * tv
    first = last if first is None else first
    if it <= 100:
        assert abs(last - loss[it]) <= 1e-12 * last, (it, last, loss[it])
assert abs(first - want[256][0]) <= 1e-12 * first and abs(last - want[256][1]) <= 1e-12 * last, (first, last)
print("REFERENCE", repr(first), repr(last))
'''


@pytest.mark.skipif(not os.path.exists(os.path.join(REF, "pytv", "media", "cameraman.npy")), reason="reference not mounted")
def test_config0_cameraman_losses_match_baseline():
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", CONFIG0 % dict(root=ROOT, ref=REF)], cwd="/tmp", env=env,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert out.stdout.count("CONFIG0") == 2 and "REFERENCE" in out.stdout

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

#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (PyTV-4D v1.1.2).

Runs only in the authoring container, where the reference is mounted read-only:

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/reference \
        python3 /root/repo/tests/golden/make_golden.py

Nothing but DATA is written: seeded inputs and the outputs the reference's CPU twin
(pytv.tv_operators_CPU / pytv.tv_CPU) returns for them.  The loops at the bottom are the
README's user-level driver snippets (README.md:107-124 sub-gradient descent, :141-157
Chambolle-Pock) calling the reference's operators.
"""
import json
import os
import sys

import numpy as np

import pytv  # the reference, from PYTHONPATH=/root/reference

assert "/root/reference" in os.path.abspath(pytv.__file__), pytv.__file__
OUT = os.path.dirname(os.path.abspath(__file__))
SCHEMES = ("upwind", "downwind", "central", "hybrid")
ops = pytv.tv_operators_CPU
tvc = pytv.tv_CPU

# name, shape (Nz, M, N, N), reg_z_over_reg, reg_time, mask?, factor, dtype
CASES = [
    ("2d",        (1, 1, 7, 7), 1.0, 0.0,    False, 0, "f8"),
    ("3d",        (5, 1, 6, 6), 1.0, 0.0,    False, 0, "f8"),
    ("3d_noz",    (5, 1, 6, 6), 0.0, 0.0,    False, 0, "f8"),
    ("3d_lz2p5",  (4, 1, 6, 6), 2.5, 0.0,    False, 0, "f8"),
    ("3d_nz6",    (6, 1, 5, 5), 1.0, 0.0,    False, 0, "f8"),
    ("3d_nz3",    (3, 1, 5, 5), 0.3, 0.0,    False, 0, "f8"),
    ("2dt_m4",    (1, 4, 6, 6), 1.0, 1.0,    False, 0, "f8"),
    ("4d_m2",     (3, 2, 6, 6), 1.0, 1.0,    False, 0, "f8"),
    ("4d_m3",     (3, 3, 5, 5), 1.0, 1.0,    False, 0, "f8"),
    ("4d_m4",     (4, 4, 5, 5), 1.0, 1.0,    False, 0, "f8"),
    ("4d_m8",     (3, 8, 5, 5), 1.0, 1.0,    False, 0, "f8"),
    ("4d_mu2m5",  (4, 3, 6, 6), 1.0, 2**-5,  False, 0, "f8"),
    ("4d_noz",    (4, 3, 6, 6), 0.0, 1.0,    False, 0, "f8"),
    ("4d_mask",   (3, 3, 6, 6), 2.5, 0.7,    True,  4, "f8"),
    ("4d_mask_m2", (4, 2, 5, 5), 1.0, 1.5,   True,  0.25, "f8"),
    ("4d_f32",    (3, 4, 8, 8), 1.0, 1.0,    False, 0, "f4"),
    ("4d_f32_mask", (5, 3, 8, 8), 0.5, 2.0,  True,  4, "f4"),
    # round 6: a boolean mask_static with the reference's DEFAULT factor_reg_static=0 (pytv/tv_operators_GPU.py:134): the time
    # differences vanish on the masked pixels (appended: the seeds 1000 + index of the cases above do not move)
    ("4d_mask_f0", (4, 3, 6, 6), 1.0, 1.0,   True,  0, "f8"),
    ("4d_f32_mask_f0", (3, 4, 8, 8), 2.0, 0.5, True, 0, "f4"),
]


def gen_ops():
    for scheme in SCHEMES:
        D = getattr(ops, "D_" + scheme)
        DT = getattr(ops, "D_T_" + scheme)
        tv = getattr(tvc, "tv_" + scheme)
        blob = {}
        names = []
        for ic, (name, shape, lz, mu, use_mask, factor, dt) in enumerate(CASES):
            rng = np.random.default_rng(1000 + ic)
            x = rng.standard_normal(shape).astype(dt)
            # a flat patch so that some gradient norms are exactly zero (sub-gradient guard)
            x[..., :2, :3] = x.dtype.type(0.5)
            mask = (rng.random((1, 1) + shape[2:]) > 0.5) if use_mask else False
            kw = dict(reg_z_over_reg=lz, reg_time=mu, mask_static=mask, factor_reg_static=factor)
            Dx = D(x.copy(), **kw)
            y = rng.standard_normal(Dx.shape).astype(dt)
            DTy = DT(y.copy(), **kw)
            DTDx = DT(Dx.copy(), **kw)
            l21, norms = ops.compute_L21_norm(Dx, return_array=True)
            tvv, G, gn = tv(x.copy(), return_grad_norms=True, **kw)
            names.append(name)
            pre = name + "/"
            blob[pre + "x"] = x
            blob[pre + "y"] = y
            blob[pre + "params"] = np.array([lz, mu, float(factor)], dtype=np.float64)
            blob[pre + "mask"] = np.asarray(mask)
            blob[pre + "D"] = Dx
            blob[pre + "DT"] = DTy
            blob[pre + "DTD"] = DTDx
            blob[pre + "l21"] = np.float64(l21)
            blob[pre + "norms"] = norms
            blob[pre + "tv"] = np.float64(tvv)
            blob[pre + "G"] = G
            blob[pre + "grad_norms"] = gn
        blob["case_names"] = np.array(names)
        np.savez_compressed(os.path.join(OUT, "ops_%s.npz" % scheme), **blob)
        print("wrote ops_%s.npz" % scheme)


def gen_known_answers():
    """README.md:76-93 known answer and its siblings; 5x5 impulse of the maths notebook."""
    ka = {}
    np.random.seed(0)
    x = np.random.rand(20, 4, 100, 100)
    for scheme in SCHEMES:
        tv = getattr(tvc, "tv_" + scheme)
        for tag, mu in (("mu0", 0.0), ("mu2m5", 2**-5)):
            tvv, G = tv(x.copy(), reg_time=mu)
            ka["readme_%s_%s" % (scheme, tag)] = {
                "tv": float(tvv), "G_sum": float(G.sum()), "G_abs_sum": float(np.abs(G).sum()),
                "G_sq_sum": float((G * G).sum()),
                "G_probe": [float(v) for v in G[[0, 7, 19, 3], [0, 1, 3, 2], [0, 50, 99, 17], [0, 31, 99, 64]]],
            }
    A = np.zeros((1, 1, 5, 5))
    A[0, 0, 2, 2] = 1.0
    for scheme in SCHEMES:
        tvv, G = getattr(tvc, "tv_" + scheme)(A.copy())
        ka["impulse5_" + scheme] = {"tv": float(tvv), "G": [[float(v) for v in row] for row in G[0, 0]]}
    with open(os.path.join(OUT, "known_answers.json"), "w") as f:
        json.dump(ka, f, indent=1, sort_keys=True)
    print("wrote known_answers.json")


def phantom2d(n, seed):
    rng = np.random.RandomState(seed)
    img = np.zeros((n, n))
    for _ in range(12):
        r0, c0 = rng.randint(0, n - 8, size=2)
        h, w = rng.randint(6, n // 2, size=2)
        img[r0:r0 + h, c0:c0 + w] += rng.rand() * 255.0 / 3.0
    return img


def gen_trajectories():
    """README driver loops on a seeded synthetic 2-D phantom (the reference's cameraman image is
    package data and is not copied)."""
    noise_level, nb_it, regularization, step_size = 100, 300, 25, 5e-3
    truth = phantom2d(64, 7).reshape(1, 1, 64, 64)
    np.random.seed(0)
    noisy = truth + noise_level * np.random.rand(*truth.shape)
    blob = {"truth": truth, "noisy": noisy,
            "params": np.array([noise_level, nb_it, regularization, step_size], dtype=np.float64)}

    for scheme in SCHEMES:
        tv = getattr(tvc, "tv_" + scheme)
        D = getattr(ops, "D_" + scheme)
        DT = getattr(ops, "D_T_" + scheme)

        # README.md:118-124
        est = np.copy(noisy)
        loss_gd = np.zeros([nb_it, ])
        for it in range(nb_it):
            tvv, G = tv(est)
            est += - step_size * ((est - noisy) + regularization * G)
            loss_gd[it] = 0.5 * np.sum(np.square(est - noisy)) + regularization * tvv
        blob["gd_loss_" + scheme] = loss_gd
        blob["gd_final_" + scheme] = est

        # README.md:141-157
        sigma_D, sigma_A, tau = 0.5, 1.0, 1 / (8 + 1)
        est = np.copy(noisy)
        dual_fid = np.zeros_like(est)
        dual_tv = np.zeros_like(D(est))
        loss_cp = np.zeros([nb_it, ])
        for it in range(nb_it):
            dual_fid = (dual_fid + sigma_A * (est - noisy)) / (1.0 + sigma_A)
            D_x = D(est)
            prox_argument = dual_tv + sigma_D * D_x
            dual_tv = prox_argument / np.maximum(1.0, np.sqrt(np.sum(prox_argument**2, axis=1)) / regularization)
            est = est - tau * dual_fid - tau * DT(dual_tv)
            loss_cp[it] = 0.5 * np.sum(np.square(est - noisy)) + regularization * ops.compute_L21_norm(D_x)
        blob["cp_loss_" + scheme] = loss_cp
        blob["cp_final_" + scheme] = est
    np.savez_compressed(os.path.join(OUT, "trajectories_2d.npz"), **blob)
    print("wrote trajectories_2d.npz")


def gen_trajectory_512():
    """BASELINE configs[0] at its SIZE (512 x 512, hybrid sub-gradient loop, 300 iterations, README.md:107-124) on a
    synthetic phantom -- the cameraman image itself is the reference's data and stays in /root/reference
    (tests/test_oracle_vs_reference.py pins the oracle on it in the authoring container).  Only the phantom (it
    compresses), the seed and the outputs are stored: the noise is np.random.RandomState(0).rand, which NumPy keeps
    bit-stable across versions."""
    noise_level, nb_it, regularization, step_size = 100, 300, 25, 5e-3
    truth = phantom2d(512, 11).reshape(1, 1, 512, 512)
    np.random.seed(0)
    noisy = truth + noise_level * np.random.rand(*truth.shape)
    assert np.array_equal(noisy, truth + noise_level * np.random.RandomState(0).rand(*truth.shape))
    est = np.copy(noisy)
    loss = np.zeros([nb_it, ])
    for it in range(nb_it):
        tvv, G = tvc.tv_hybrid(est)
        est += - step_size * ((est - noisy) + regularization * G)
        loss[it] = 0.5 * np.sum(np.square(est - noisy)) + regularization * tvv
    np.savez_compressed(os.path.join(OUT, "trajectory_512.npz"), truth=truth,
                        params=np.array([noise_level, nb_it, regularization, step_size, 0], dtype=np.float64),
                        gd_loss_hybrid=loss, gd_final_mean=np.array(est.mean()), gd_final_row=est[0, 0, 200].copy(),
                        noisy_checksum=np.array(noisy.sum()))
    print("wrote trajectory_512.npz")


if __name__ == "__main__":
    if "--only-512" not in sys.argv:
        gen_ops()
        gen_known_answers()
        gen_trajectories()
    gen_trajectory_512()
    sys.exit(0)

#!/usr/bin/env python3
"""The two denoising loops of the reference's README (README.md:107-124 sub-gradient descent, :141-157
Chambolle-Pock), first exactly as written there -- numpy in / numpy out through the drop-in modules -- and then
with the state kept on the GPU (pytv.solvers).  The reference's cameraman image is package data and is not shipped
here; a synthetic piecewise-constant phantom stands in for it.

    PYTHONPATH=pytv-4d_amd python examples/denoise_readme_loops.py
"""
import time

import numpy as np
import torch

import pytv

noise_level, nb_it, regularization, step_size = 100, 300, 25, 5e-3
rng = np.random.RandomState(7)
truth = np.zeros((256, 256))
for _ in range(12):
    r0, c0 = rng.randint(0, 248, size=2)
    h, w = rng.randint(6, 128, size=2)
    truth[r0:r0 + h, c0:c0 + w] += rng.rand() * 255.0 / 3.0
np.random.seed(0)
truth = truth.reshape(1, 1, 256, 256)
noisy = truth + noise_level * np.random.rand(*truth.shape)

# --- README.md:118-124, unchanged except for the image -------------------------------------------------------
estimate = np.copy(noisy)
loss_gd = np.zeros(nb_it)
t0 = time.perf_counter()
for it in range(nb_it):
    tv, G = pytv.tv_GPU.tv_hybrid(estimate)
    estimate += - step_size * ((estimate - noisy) + regularization * G)
    loss_gd[it] = 0.5 * np.sum(np.square(estimate - noisy)) + regularization * tv
print("sub-gradient descent, README loop (numpy in/out): %.3f s, loss %.4e -> %.4e" % (time.perf_counter() - t0, loss_gd[0], loss_gd[-1]))

# --- README.md:141-157 ---------------------------------------------------------------------------------------
sigma_D, sigma_A, tau = 0.5, 1.0, 1 / (8 + 1)
estimate = np.copy(noisy)
dual_fid = np.zeros_like(estimate)
dual_tv = np.zeros_like(pytv.tv_operators_GPU.D_hybrid(estimate))
loss_cp = np.zeros(nb_it)
t0 = time.perf_counter()
for it in range(nb_it):
    dual_fid = (dual_fid + sigma_A * (estimate - noisy)) / (1.0 + sigma_A)
    D_x = pytv.tv_operators_GPU.D_hybrid(estimate)
    prox_argument = dual_tv + sigma_D * D_x
    dual_tv = prox_argument / np.maximum(1.0, np.sqrt(np.sum(prox_argument**2, axis=1)) / regularization)
    estimate = estimate - tau * dual_fid - tau * pytv.tv_operators_GPU.D_T_hybrid(dual_tv)
    loss_cp[it] = 0.5 * np.sum(np.square(estimate - noisy)) + regularization * pytv.tv_operators_GPU.compute_L21_norm(D_x)
print("Chambolle-Pock, README loop (numpy in/out):        %.3f s, loss %.4e -> %.4e" % (time.perf_counter() - t0, loss_cp[0], loss_cp[-1]))

# --- the same two algorithms with the state resident on the GPU -----------------------------------------------
x0 = torch.as_tensor(noisy).cuda()
pytv.solvers.ChambollePock(x0, regularization, tau=tau).run(30)          # warm-up (module load, graph capture path)
for name, make in (("sub-gradient descent, device resident", lambda: pytv.solvers.SubgradientDescent(x0, regularization, step_size)),
                   ("Chambolle-Pock, device resident      ", lambda: pytv.solvers.ChambollePock(x0, regularization, tau=tau))):
    solver = make()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = solver.run(nb_it)
    print("%s:        %.3f s, loss %.4e -> %.4e" % (name, time.perf_counter() - t0, loss[0], loss[-1]))
estimate = solver.result().cpu().numpy()
print("PSNR noisy %.2f dB -> denoised %.2f dB" % (
    10 * np.log10(255 ** 2 / np.mean((noisy - truth) ** 2)), 10 * np.log10(255 ** 2 / np.mean((estimate - truth) ** 2))))

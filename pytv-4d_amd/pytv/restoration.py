"""2-D / 3-D ``denoise_tv_chambolle`` front-end with scikit-image's calling convention, on top of the device
resident Chambolle-Pock solver (README.md:260 of the reference lists this as a to-do).

scikit-image minimises  sum |grad u| + (1 / (2 weight)) |u - f|^2  with forward differences, i.e.
1/2 |u - f|^2 + weight * TV_upwind(u): the same objective as ``solvers.ChambollePock(f, weight, scheme="upwind")``.
Its iteration (Chambolle 2004) differs from Chambolle-Pock 2011, so iterates differ; the minimiser is the same.
"""
import torch

from .solvers import ChambollePock
from .tv_operators_GPU import _to_device

__all__ = ["denoise_tv_chambolle"]


def denoise_tv_chambolle(image, weight=0.1, eps=2.0e-4, max_num_iter=200, *, scheme="upwind", check_every=10):
    """Total-variation denoising of a 2-D (rows, cols) or 3-D (planes, rows, cols) image.

    weight : denoising weight (larger = smoother), as in scikit-image.
    eps    : stop when the relative change of the objective over ``check_every`` iterations drops below eps.
    Returns an array of the input's kind (numpy in -> numpy out, torch in -> device tensor), floating point."""
    was_torch = isinstance(image, torch.Tensor)
    x, _ = _to_device(image)
    if x.dim() == 2:
        vol = x.reshape(1, 1, *x.shape)
    elif x.dim() == 3:
        vol = x.reshape(x.shape[0], 1, x.shape[1], x.shape[2])
    else:
        raise ValueError("denoise_tv_chambolle: 2-D or 3-D images only (use pytv.solvers for 4-D data)")
    cp = ChambollePock(vol.contiguous(), float(weight), scheme=scheme, reg_z_over_reg=1.0)
    prev, done = None, 0
    while done < max_num_iter:
        n = min(check_every, max_num_iter - done)
        loss = cp.run(n)
        done += n
        e = float(loss[-1])
        if prev is not None and abs(prev - e) <= eps * max(abs(prev), 1e-30):
            break
        prev = e
    out = cp.result().reshape(x.shape)
    return out if was_torch else out.detach().cpu().numpy()

"""Drop-in replacement of ``pytv.tv_operators_GPU`` (PyTV-4D v1.1.2) on AMD MI355X.

Same function names, keyword arguments, defaults and return conventions as the reference
(pytv/tv_operators_GPU.py:46,92,134,253,362,471,583,719,828,938); the bodies call hand-written HIP
kernels through the C-ABI of ``libpytv4d_hip.so`` (include/pytv4d.h) instead of
``torch.nn.functional.conv3d`` + slice-assign.  PyTorch tensors are only device-memory handles here.

Conventions kept from the reference:
  * numpy in -> numpy out unless ``return_pytorch_tensor``; a torch input FORCES a torch (device)
    output (tv_operators_GPU.py:182,303,412,521,627,763,872,982);
  * float32 stays float32, anything else is computed in float64 (``type_like``, :114-129);
  * ``compute_L21_norm(return_array=True)`` returns ``(0-d numpy, torch tensor)`` unless
    ``return_pytorch_tensor`` (:83-87).
Deliberate differences: the l2,1 value is accumulated and returned in float64; images need not be
square (the reference uses ``N = img.shape[-1]`` for both axes); the GPU twin's minimum sizes
(N >= 3 / 5) do not apply; ``central`` with Nz == 2 uses the forward z stencil instead
of raising.
"""
import os

import numpy as np
import torch

from . import _native as _nv

__all__ = ["compute_L21_norm", "type_like",
           "D_hybrid", "D_downwind", "D_upwind", "D_central",
           "D_T_hybrid", "D_T_downwind", "D_T_upwind", "D_T_central"]


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("pytv.tv_operators_GPU needs a HIP device (torch.cuda.is_available() is False); "
                           "there is no CPU fallback in this package")
    return torch.device("cuda", torch.cuda.current_device())


def _to_device(arr):
    """numpy / torch -> contiguous float32 or float64 device tensor.  Returns (tensor, was_torch)."""
    was_torch = isinstance(arr, torch.Tensor)
    t = arr if was_torch else torch.as_tensor(np.asarray(arr))
    if t.dtype != torch.float32:
        t = t.to(torch.float64)            # integer / other inputs behave as float64
    t = t.to(_device(), non_blocking=False).contiguous()
    return t, was_torch


# results between these sizes come back through pinned host memory; PYTV_PIN_MAX_MB=0 switches the pinned path off
# (page-locked memory is a limited resource in containers / under `ulimit -l`, and the caching host allocator keeps
# the blocks)
_PIN_MIN = 1 << 20
_PIN_MAX = int(float(os.environ.get("PYTV_PIN_MAX_MB", 16 << 10)) * (1 << 20))


def _to_host(t):
    """Device tensor -> numpy array.  The reference's README loops cross the boundary on every call, and a device-to-host
    copy into pageable memory runs at 6 - 10 GB/s on the MI355X box against 53 GB/s into pinned memory (H2D from pageable
    memory is already at 30 - 50 GB/s): results between 1 MiB and 16 GiB land in a pinned buffer of PyTorch's caching host
    allocator (allocated once per size, reused as soon as the previous result has been dropped) and are returned as a view
    of it -- the array owns its buffer like any other."""
    t = t.detach()
    nbytes = t.numel() * t.element_size()
    if t.is_cuda and _PIN_MIN <= nbytes <= _PIN_MAX:
        try:
            h = torch.empty(t.shape, dtype=t.dtype, pin_memory=True)
        except RuntimeError:              # page-locking refused (ulimit -l, container limits, host memory): pageable copy
            h = None
        if h is not None:
            h.copy_(t, non_blocking=True)
            torch.cuda.current_stream(t.device).synchronize()
            return h.numpy()
    return t.cpu().numpy()


def _finish(t, return_pytorch_tensor):
    return t if return_pytorch_tensor else _to_host(t)


def compute_L21_norm(D_img, return_array=False, return_pytorch_tensor=False):
    """|D|_{2,1} = sum_p sqrt(sum_c D[p, c]^2) of an (Nz, Nd, M, N, N) gradient array
    (reference: tv_operators_GPU.py:46-90) in ONE pass over D instead of four."""
    d, _ = _to_device(D_img)
    if d.dim() != 5:
        raise ValueError("D_img must be 5-D (Nz, Nd, M, N, N), got shape %s" % (tuple(d.shape),))
    nz, nd, m, ny, nx = d.shape
    geo = _nv.Geometry((nz, m, ny, nx), "upwind", d.dtype, d.device)
    out = torch.empty((nz, m, ny, nx), dtype=d.dtype, device=d.device) if return_array else None
    res = geo.scalar()
    _nv.check(_nv.lib().tv_l21(geo.ref, _nv.ptr(d), nd, _nv.ptr(out), _nv.ptr(res), _nv.ptr(geo.workspace()),
                               _nv.current_stream(d.device)))
    if return_array:
        if return_pytorch_tensor:
            return res, out
        return res.detach().cpu().numpy(), out
    return res.detach().cpu().numpy()


def type_like(array, array_ref):
    """Return ``array`` with the dtype of ``array_ref`` (reference: tv_operators_GPU.py:92-131):
    numpy/numpy copies the exact dtype, every other pairing yields float32 iff the reference array
    is float32 and float64 otherwise."""
    a_np, r_np = isinstance(array, np.ndarray), isinstance(array_ref, np.ndarray)
    if a_np and r_np:
        return array.astype(array_ref.dtype)
    ref_is_f32 = (array_ref.dtype == np.float32) if r_np else (array_ref.dtype == torch.float32)
    if a_np:
        return array.astype(np.float32 if ref_is_f32 else np.float64)
    return array.type(torch.float32 if ref_is_f32 else torch.float64)


def _apply_D(scheme, img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor):
    x, was_torch = _to_device(img)
    return_pytorch_tensor = return_pytorch_tensor or was_torch
    geo = _nv.Geometry(tuple(x.shape), scheme, x.dtype, x.device, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
    d = torch.empty(geo.grad_shape, dtype=x.dtype, device=x.device)
    _nv.check(_nv.lib().tv_D(geo.ref, _nv.ptr(x), None, None, _nv.ptr(d), _nv.current_stream(x.device)))
    return _finish(d, return_pytorch_tensor)


def _apply_DT(scheme, img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor):
    y, was_torch = _to_device(img)
    return_pytorch_tensor = return_pytorch_tensor or was_torch
    if y.dim() != 5:
        raise ValueError("gradient array must be 5-D (Nz, Nd, M, N, N), got shape %s" % (tuple(y.shape),))
    nz, nd, m, ny, nx = y.shape
    geo = _nv.Geometry((nz, m, ny, nx), scheme, y.dtype, y.device, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
    if nd < geo.nd:
        raise ValueError("D_T_%s: %d gradient channels given, %d needed for these weights" % (scheme, nd, geo.nd))
    if nd != geo.nd:
        # like the reference, only the first geo.nd channels are read (tv_operators_CPU.py:398-448)
        y = y[:, :geo.nd].contiguous()
    out = torch.empty((nz, m, ny, nx), dtype=y.dtype, device=y.device)
    _nv.check(_nv.lib().tv_DT(geo.ref, _nv.ptr(y), None, None, _nv.ptr(out), _nv.current_stream(y.device)))
    return _finish(out, return_pytorch_tensor)


def D_hybrid(img, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0, return_pytorch_tensor=False):
    """D(img), hybrid scheme: (Nz, M, N, N) -> (Nz, Nd, M, N, N), Nd = 4 (+2 z) (+2 time).
    Reference: tv_operators_GPU.py:134-251."""
    return _apply_D("hybrid", img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor)


def D_downwind(img, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0, return_pytorch_tensor=False):
    """D(img), downwind (backward) scheme.  Reference: tv_operators_GPU.py:253-360."""
    return _apply_D("downwind", img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor)


def D_upwind(img, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0, return_pytorch_tensor=False):
    """D(img), upwind (forward) scheme.  Reference: tv_operators_GPU.py:362-469."""
    return _apply_D("upwind", img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor)


def D_central(img, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0, return_pytorch_tensor=False):
    """D(img), central scheme (time axis with M == 2 uses the forward stencil).
    Reference: tv_operators_GPU.py:471-581."""
    return _apply_D("central", img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor)


def D_T_hybrid(img, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0, return_pytorch_tensor=False):
    """D^T(img), hybrid scheme: (Nz, Nd, M, N, N) -> (Nz, M, N, N).  Reference: tv_operators_GPU.py:583-717."""
    return _apply_DT("hybrid", img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor)


def D_T_downwind(img, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0, return_pytorch_tensor=False):
    """D^T(img), downwind scheme.  Reference: tv_operators_GPU.py:719-826."""
    return _apply_DT("downwind", img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor)


def D_T_upwind(img, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0, return_pytorch_tensor=False):
    """D^T(img), upwind scheme.  Reference: tv_operators_GPU.py:828-936."""
    return _apply_DT("upwind", img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor)


def D_T_central(img, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0, return_pytorch_tensor=False):
    """D^T(img), central scheme.  Reference: tv_operators_GPU.py:938-1052."""
    return _apply_DT("central", img, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor)

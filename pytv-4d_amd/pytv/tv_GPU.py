"""Drop-in replacement of ``pytv.tv_GPU`` (PyTV-4D v1.1.2) on AMD MI355X.

``tv_<scheme>(img, ...)`` returns the total variation of ``img`` and the reference's sub-gradient
(pytv/tv_GPU.py:47,142,217,290).  The reference materialises D(img) (Nd x the image), then runs
3..14 sliced ``G[...] += +-D/norm`` updates; here TV, G and -- when asked for -- the per-voxel norms come
from a SINGLE pass over ``img`` (C-ABI ``tv_subgrad_fused`` / ``tv_subgrad_fused_norms``, include/pytv4d.h: fp32 and
fp64, any Nx, any number of frames, with or without a per-voxel weight volume -- since round 3); only ``central`` with a
two-point z or time axis takes the two-pass form (``tv_subgrad``: 1/|D img| per voxel + TV partial sums, then a gather of
G from ``img`` and those norms).  The gradient array is never stored either way.

Conventions kept: ``mask`` zeroes the caller's array in place; the TV
value is always a 0-d numpy array; G (and grad_norms) are numpy unless ``return_pytorch_tensor``;
grad_norms has zeros replaced by +inf; the sub-gradient uses unit weights in the adjoint.

Deliberate differences from the reference's arithmetic (all inside the 1e-5 relative fp32 tolerance of the north star and the
reference's own ``pytv/tests.py:88-109``; fp64: ``v_rsq_f64`` refined by two Newton steps, ~1 ulp, parity at 1e-11):
* the fp32 one-pass kernel computes 1 / |D img| with the hardware reciprocal square root (``v_rsq_f32``, 1 ulp) and multiplies, where
  the reference takes ``sqrt`` and divides (``pytv/tv_GPU.py:86-124``; SURVEY Q12 recommends IEEE sqrt / div).  |D img| itself, where
  it is returned (``return_grad_norms``), is ``|D img|^2 * rsq`` -- within 2 ulp of the IEEE square root;
* a voxel whose squared norm is below the smallest NORMAL fp32 number (1.18e-38, i.e. |D img| < 1.1e-19) counts as a ZERO gradient:
  its terms are dropped and its grad_norm is +inf, exactly like a true zero (``pytv/tv_GPU.py:86-88``).  The reference would divide
  by the sub-normal norm instead; inputs of that magnitude do not occur in image data, and the TV value is unaffected (< 1e-19 per
  voxel).  The fp64 kernels apply the same rule at the smallest normal fp64 number;
* the TV value is accumulated in fp64 over per-block partial sums (the reference: NumPy / torch pairwise sums in the input dtype).
"""
import torch

from . import _native as _nv
from .tv_operators_GPU import _to_device, _to_host

__all__ = ["tv_hybrid", "tv_downwind", "tv_upwind", "tv_central"]


def _has_mask(mask):
    # the reference tests ``mask != []`` (tv_GPU.py:79), which raises for ndarray masks under NumPy 2
    return mask is not None and not isinstance(mask, bool) and len(mask) > 0


def one_pass_ok(geo, itemsize=4):
    """Use the one-pass kernel (``tv_subgrad_fused``)?  Whenever the geometry is supported: it is as fast as or
    faster than the two-pass kernels from 512 x 512 2-D images to the north-star volume (tools/archive/sg_small_bench.py)."""
    return bool(_nv.lib().tv_subgrad_fused_supported(geo.ref))


def tv_subgradient_device(x, scheme, reg_z_over_reg=1.0, reg_time=0.0, mask_static=False, factor_reg_static=0,
                          want_norms=True, one_pass=None):
    """Device-resident core: x is a contiguous fp32/fp64 device tensor (Nz, M, Ny, Nx).
    Returns (tv 0-d fp64 device tensor, G, grad_norms) without any host synchronisation; grad_norms is |Dx| per voxel
    with zeros replaced by +inf (the reference's convention, tv_GPU.py:88), or None with ``want_norms=False``.
    Where the geometry allows, everything comes from ONE pass over x (``one_pass``: None = automatic, True / False =
    force); otherwise from the two-pass kernels (1/|Dx| to memory, then a gather)."""
    geo = _nv.Geometry(tuple(x.shape), scheme, x.dtype, x.device, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
    nz, m, ny, nx = geo.shape
    G = torch.empty_like(x)
    if one_pass is None:
        one_pass = one_pass_ok(geo, x.element_size())
    tv = geo.scalar()
    if one_pass:
        if want_norms:
            norms = torch.empty_like(x)
            _nv.check(_nv.lib().tv_subgrad_fused_norms(geo.ref, _nv.ptr(x), None, None, _nv.ptr(G), _nv.ptr(norms), _nv.ptr(tv),
                                                       _nv.ptr(geo.workspace()), _nv.current_stream(x.device)))
            return tv, G, norms
        _nv.check(_nv.lib().tv_subgrad_fused(geo.ref, _nv.ptr(x), None, None, _nv.ptr(G), _nv.ptr(tv),
                                             _nv.ptr(geo.workspace()), _nv.current_stream(x.device)))
        return tv, G, None
    norms_ext = torch.empty((nz + 2, m, ny, nx), dtype=x.dtype, device=x.device)
    _nv.check(_nv.lib().tv_subgrad(geo.ref, _nv.ptr(x), None, None, _nv.ptr(G), _nv.ptr(norms_ext), _nv.ptr(tv),
                                   _nv.ptr(geo.workspace()), _nv.current_stream(x.device)))
    # the kernels keep 1 / |Dx| (0 where |Dx| counts as zero): the reference's grad_norms is its reciprocal
    return tv, G, (torch.reciprocal(norms_ext[1:nz + 1]) if want_norms else None)


def _tv(scheme, img, mask, reg_z_over_reg, reg_time, mask_static, factor_reg_static, return_pytorch_tensor,
        return_grad_norms):
    if _has_mask(mask):
        img[~mask] = 0                      # in place on the caller's array, as the reference does
    x, _ = _to_device(img)
    if x.dim() != 4:
        raise ValueError("img must be 4-D (Nz, M, N, N), got shape %s" % (tuple(x.shape),))
    tv, G, gn = tv_subgradient_device(x, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static,
                                      want_norms=bool(return_grad_norms))
    tv = tv.detach().cpu().numpy()          # 0-d numpy, always (tv_GPU.py:85 via compute_L21_norm)
    if not return_grad_norms:
        return (tv, G) if return_pytorch_tensor else (tv, _to_host(G))
    if return_pytorch_tensor:
        return tv, G, gn
    return tv, _to_host(G), _to_host(gn)


def tv_hybrid(img, mask=[], reg_z_over_reg=1.0, reg_time=0.0, mask_static=False, factor_reg_static=0,
              return_pytorch_tensor=False, return_grad_norms=False):
    """TV and sub-gradient, hybrid discretisation.  Reference: tv_GPU.py:47-139."""
    return _tv("hybrid", img, mask, reg_z_over_reg, reg_time, mask_static, factor_reg_static,
               return_pytorch_tensor, return_grad_norms)


def tv_downwind(img, mask=[], reg_z_over_reg=1.0, reg_time=0.0, mask_static=False, factor_reg_static=0,
                return_pytorch_tensor=False, return_grad_norms=False):
    """TV and sub-gradient, downwind discretisation.  Reference: tv_GPU.py:142-215."""
    return _tv("downwind", img, mask, reg_z_over_reg, reg_time, mask_static, factor_reg_static,
               return_pytorch_tensor, return_grad_norms)


def tv_upwind(img, mask=[], reg_z_over_reg=1.0, reg_time=0.0, mask_static=False, factor_reg_static=0,
              return_pytorch_tensor=False, return_grad_norms=False):
    """TV and sub-gradient, upwind discretisation.  Reference: tv_GPU.py:217-288."""
    return _tv("upwind", img, mask, reg_z_over_reg, reg_time, mask_static, factor_reg_static,
               return_pytorch_tensor, return_grad_norms)


def tv_central(img, mask=[], reg_z_over_reg=1.0, reg_time=0.0, mask_static=False, factor_reg_static=0,
               return_pytorch_tensor=False, return_grad_norms=False):
    """TV and sub-gradient, central discretisation.  Reference: tv_GPU.py:290-375."""
    return _tv("central", img, mask, reg_z_over_reg, reg_time, mask_static, factor_reg_static,
               return_pytorch_tensor, return_grad_norms)

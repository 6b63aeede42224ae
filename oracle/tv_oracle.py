"""CPU oracle for the PyTV-4D hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The shipped package (``pytv-4d_amd/pytv``) never does: it calls the HIP kernels
through the C-ABI and raises if the native library is missing.

What it is: a scheme-generic NumPy restatement of the reference's CPU twin
(``pytv/tv_operators_CPU.py`` and ``pytv/tv_CPU.py``, PyTV-4D v1.1.2), written from the per-voxel
definitions rather than from the reference's per-scheme slice-assign code:

    fwd_a x [p] = x[p+e_a] - x[p]   if p_a <  n_a-1 else 0        (zero-extended forward difference)
    bwd_a x [p] = x[p] - x[p-e_a]   if p_a >  0     else 0        (= fwd shifted by one)
    cen_a x [p] = x[p+e_a]-x[p-e_a] if 0<p_a<n_a-1  else 0

    upwind  : channel_a = w_a fwd_a x                      tv_operators_CPU.py:264-284
    downwind: channel_a = w_a bwd_a x                      tv_operators_CPU.py:198-218
    central : channel_a = w_a cen_a x, whole array / 2     tv_operators_CPU.py:330-358
    hybrid  : up channels then down channels per axis pair, whole array / sqrt(2)
                                                            tv_operators_CPU.py:117-154

Channel order (axis 1 of the (Nz, Nd, M, N, N) gradient array): rows, cols, [z], [t]; hybrid:
row-up, col-up, row-down, col-down, [z-up, z-down], [t-up, t-down].  The z axis is active iff
``Nz > 1 and reg_z_over_reg > 0``; the time axis iff ``reg_time > 0 and M > 1``
(tv_operators_CPU.py:110-114,190-194,256-260,322-326).

Pinning: ``tests/test_oracle_golden.py`` checks every function here against golden vectors produced
by importing the reference itself (``tests/golden/make_golden.py``, run in the authoring container),
against the README / notebook known answers, and -- when ``/root/reference`` is present -- directly
against the imported reference on fresh random inputs (``tests/test_oracle_vs_reference.py``).

Unpinned (no reference behaviour exists): ``central`` with Nz == 2 (the reference
raises; here z falls back to the upwind stencil, the evident intent of
tv_operators_CPU.py:338-340), Chambolle-Pock beyond 2-D, and ADMM entirely.
"""
import numpy as np

SCHEMES = ("upwind", "downwind", "central", "hybrid")

# axis numbers inside an (Nz, M, N, N) image, in CHANNEL order: rows, cols, z, t
_AX_ROW, _AX_COL, _AX_Z, _AX_T = 2, 3, 0, 1


# --------------------------------------------------------------------------------------------
# one-axis building blocks
# --------------------------------------------------------------------------------------------
def _sl(ndim, axis, s):
    idx = [slice(None)] * ndim
    idx[axis] = s
    return tuple(idx)


def _fwd(x, axis):
    """Zero-extended forward difference along ``axis`` (tv_operators_CPU.py:265,268,273,278)."""
    d = np.zeros_like(x)
    if x.shape[axis] > 1:
        d[_sl(x.ndim, axis, slice(None, -1))] = (
            x[_sl(x.ndim, axis, slice(1, None))] - x[_sl(x.ndim, axis, slice(None, -1))])
    return d


def _bwd(x, axis):
    """Backward difference = the forward difference moved one step up the axis
    (tv_operators_CPU.py:199,202,207,212)."""
    d = np.zeros_like(x)
    if x.shape[axis] > 1:
        d[_sl(x.ndim, axis, slice(1, None))] = (
            x[_sl(x.ndim, axis, slice(1, None))] - x[_sl(x.ndim, axis, slice(None, -1))])
    return d


def _cen(x, axis):
    """x[p+e] - x[p-e] on interior points (tv_operators_CPU.py:331,334); a two-point z or time axis falls back to
    the forward difference (:339-342,347-350).  A two-point ROW or COLUMN axis has no interior point: the channel is
    zero, as the reference's slicing gives for N = 2."""
    n = x.shape[axis]
    if n == 2 and axis in (_AX_Z, _AX_T):
        return _fwd(x, axis)
    d = np.zeros_like(x)
    if n > 2:
        d[_sl(x.ndim, axis, slice(1, -1))] = (
            x[_sl(x.ndim, axis, slice(2, None))] - x[_sl(x.ndim, axis, slice(None, -2))])
    return d


def _fwd_T(y, axis):
    """Adjoint of _fwd: the last sample along ``axis`` is never read
    (tv_operators_CPU.py:555-560,565-566,574-575)."""
    out = np.zeros_like(y)
    if y.shape[axis] > 1:
        core = y[_sl(y.ndim, axis, slice(None, -1))]
        out[_sl(y.ndim, axis, slice(1, None))] += core
        out[_sl(y.ndim, axis, slice(None, -1))] -= core
    return out


def _bwd_T(y, axis):
    """Adjoint of _bwd: the first sample along ``axis`` is never read
    (tv_operators_CPU.py:488-493,498-499,507-508)."""
    out = np.zeros_like(y)
    if y.shape[axis] > 1:
        core = y[_sl(y.ndim, axis, slice(1, None))]
        out[_sl(y.ndim, axis, slice(1, None))] += core
        out[_sl(y.ndim, axis, slice(None, -1))] -= core
    return out


def _cen_T(y, axis):
    """Adjoint of _cen (tv_operators_CPU.py:623-628,633-639,646-651)."""
    n = y.shape[axis]
    if n == 2 and axis in (_AX_Z, _AX_T):
        return _fwd_T(y, axis)
    out = np.zeros_like(y)
    if n > 2:
        core = y[_sl(y.ndim, axis, slice(1, -1))]
        out[_sl(y.ndim, axis, slice(2, None))] += core
        out[_sl(y.ndim, axis, slice(None, -2))] -= core
    return out


# --------------------------------------------------------------------------------------------
# geometry
# --------------------------------------------------------------------------------------------
def active_axes(scheme, Nz, M, reg_z_over_reg, reg_time):
    """(z_active, t_active).  Same rule for every scheme (tv_operators_CPU.py:111-114); the
    reference's CPU ``D_central`` counts z with ``Nz > 2`` (:323) and then raises for Nz == 2,
    the GPU twin uses ``Nz > 1`` (tv_operators_GPU.py:507) -- they agree wherever both run."""
    return bool(Nz > 1 and reg_z_over_reg > 0), bool(reg_time > 0 and M > 1)


def num_channels(scheme, Nz, M, reg_z_over_reg=1.0, reg_time=0.0):
    z, t = active_axes(scheme, Nz, M, reg_z_over_reg, reg_time)
    per_axis = 2 if scheme == "hybrid" else 1
    return per_axis * (2 + int(z) + int(t))


def _axes_and_weights(img_shape, reg_z_over_reg, reg_time, scheme):
    Nz, M = img_shape[0], img_shape[1]
    z, t = active_axes(scheme, Nz, M, reg_z_over_reg, reg_time)
    axes = [(_AX_ROW, None), (_AX_COL, None)]
    if z:
        axes.append((_AX_Z, np.sqrt(reg_z_over_reg)))
    if t:
        axes.append((_AX_T, np.sqrt(reg_time)))
    return axes


def _mask_array(mask_static):
    """bool mask (the reference's semantics), or -- BUILD EXTENSION, the reference's to-do README.md:258 -- a FLOAT
    array: the per-pixel weight of the time regularisation; then the time channels are multiplied by sqrt(weight) and
    factor_reg_static is ignored.  weight = where(mask, factor, 1) is the reference's boolean case."""
    if isinstance(mask_static, bool):
        return None
    m = np.asarray(mask_static)
    if np.issubdtype(m.dtype, np.floating):
        return m
    return m.astype(bool)


def _varies_in_time(mask):
    """is ``mask`` a per-VOXEL weight array (BUILD EXTENSION, README.md:258 "weight matrix of size Nz x M x N x N"),
    i.e. a float array with an extent along z or t?  Then the adjoint scales every time sample by its OWN voxel's
    factor before the difference (the exact adjoint of D); for per-pixel weights that is what the reference does by
    scaling the summed time part at the output voxel (tv_operators_CPU.py:442-446)."""
    return mask is not None and np.issubdtype(mask.dtype, np.floating) and mask.ndim == 4 and (mask.shape[0] > 1 or mask.shape[1] > 1)


def _time_scaled(c, mask, sqrt_factor):
    """time channel / time part of the adjoint, scaled per pixel (tv_operators_CPU.py:148-150,428-446)"""
    if mask is None:
        return c
    if np.issubdtype(mask.dtype, np.floating):
        return c * np.broadcast_to(np.sqrt(mask), c.shape).astype(c.dtype)
    return np.where(np.broadcast_to(mask, c.shape), c * sqrt_factor, c)


# --------------------------------------------------------------------------------------------
# D and D^T
# --------------------------------------------------------------------------------------------
def D(img, scheme, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0):
    """Discrete gradient, (Nz, M, N, N) -> (Nz, Nd, M, N, N).
    Reference: tv_operators_CPU.py D_upwind :220-286, D_downwind :156-218, D_central :288-358,
    D_hybrid :77-154."""
    img = np.asarray(img)
    stencils = {"upwind": (_fwd,), "downwind": (_bwd,), "central": (_cen,), "hybrid": (_fwd, _bwd)}[scheme]
    mask = _mask_array(mask_static)
    chans = []
    for axis, w in _axes_and_weights(img.shape, reg_z_over_reg, reg_time, scheme):
        for st in stencils:
            c = st(img, axis)
            if w is not None:
                c = w * c
            if axis == _AX_T and mask is not None:
                # time channel(s) scaled where the static mask is set (tv_operators_CPU.py:148-150)
                c = _time_scaled(c, mask, np.sqrt(factor_reg_static))
            chans.append(c)
    if scheme == "hybrid":
        # reference channel order: both up channels of the in-plane pair first, then both down
        # channels (tv_operators_CPU.py:117-127); z and t come as (up, down) pairs (:130-146)
        chans[0:4] = [chans[0], chans[2], chans[1], chans[3]]
    out = np.stack(chans, axis=1).astype(img.dtype, copy=False)
    if scheme == "hybrid":
        return out / np.sqrt(2.0)          # tv_operators_CPU.py:154
    if scheme == "central":
        return out / 2.0                   # tv_operators_CPU.py:358
    return out


def _adjoint(y, scheme, z_active, t_active, w_z, w_t, mask, sqrt_factor):
    """Weighted adjoint with an explicit active-axis set (shared by D_T and the sub-gradient)."""
    adj = {"upwind": (_fwd_T,), "downwind": (_bwd_T,), "central": (_cen_T,), "hybrid": (_fwd_T, _bwd_T)}[scheme]
    img_shape = (y.shape[0],) + y.shape[2:]
    out = np.zeros(img_shape, dtype=y.dtype)
    if scheme == "hybrid":
        order = [(_AX_ROW, 0, 0), (_AX_COL, 0, 1), (_AX_ROW, 1, 2), (_AX_COL, 1, 3)]
        c = 4
        if z_active:
            order += [(_AX_Z, 0, c), (_AX_Z, 1, c + 1)]
            c += 2
        if t_active:
            order += [(_AX_T, 0, c), (_AX_T, 1, c + 1)]
    else:
        order = [(_AX_ROW, 0, 0), (_AX_COL, 0, 1)]
        c = 2
        if z_active:
            order.append((_AX_Z, 0, c))
            c += 1
        if t_active:
            order.append((_AX_T, 0, c))
    time_part = None
    per_voxel = _varies_in_time(mask)
    for axis, which, ch in order:
        ych = y[:, ch]
        if axis == _AX_T and per_voxel:
            ych = _time_scaled(ych, mask, sqrt_factor)          # sample by sample, before the adjoint stencil
        term = adj[which](ych, axis)
        if axis == _AX_Z:
            out += w_z * term
        elif axis == _AX_T:
            # the reference gathers the time terms separately so that mask_static scales only
            # them (tv_operators_CPU.py:428-446)
            time_part = w_t * term if time_part is None else time_part + w_t * term
        else:
            out += term
    if time_part is not None:
        if not per_voxel:
            time_part = _time_scaled(time_part, mask, sqrt_factor)
        out += time_part
    if scheme == "hybrid":
        return out / np.sqrt(2.0)          # tv_operators_CPU.py:448
    if scheme == "central":
        return out / 2.0                   # tv_operators_CPU.py:658
    return out


def D_T(y, scheme, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0):
    """Transposed gradient, (Nz, Nd, M, N, N) -> (Nz, M, N, N).
    Reference: tv_operators_CPU.py D_T_hybrid :360-448, D_T_downwind :450-516,
    D_T_upwind :518-583, D_T_central :585-658."""
    y = np.asarray(y)
    Nz, M = y.shape[0], y.shape[2]
    z, t = active_axes(scheme, Nz, M, reg_z_over_reg, reg_time)
    return _adjoint(y, scheme, z, t, np.sqrt(reg_z_over_reg), np.sqrt(reg_time),
                    _mask_array(mask_static), np.sqrt(factor_reg_static))


def compute_L21_norm(D_img, return_array=False):
    """sum_p sqrt(sum_c D[p,c]^2)  (tv_operators_CPU.py:45-75)."""
    norms = np.sqrt(np.sum(np.square(D_img), axis=1))
    total = np.sum(norms)
    return (total, norms) if return_array else total


# --------------------------------------------------------------------------------------------
# TV value + sub-gradient
# --------------------------------------------------------------------------------------------
def tv(img, scheme, reg_z_over_reg=1.0, reg_time=0.0, mask_static=False, factor_reg_static=0,
       return_grad_norms=False):
    """TV value and the reference's sub-gradient.

    The reference assembles G with 3..14 sliced ``G[..] += +-D/norm`` updates (tv_CPU.py:91-126
    hybrid, :176-190 downwind, :239-253 upwind, :302-330 central).  Those updates are exactly the
    adjoint stencil of the scheme applied WITH UNIT WEIGHTS to g = D/|D| (|D| == 0 -> g = 0,
    tv_CPU.py:86): no second sqrt(reg) factor and no mask factor multiply the terms
    (tv_CPU.py:104-122 use ``D_img[:, i_d]`` as is).  That is what is restated here."""
    img = np.asarray(img)
    d = D(img, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
    tv_value, norms = compute_L21_norm(d, return_array=True)
    norms = np.where(norms == 0, np.inf, norms)        # tv_CPU.py:86
    g = d / norms[:, None]
    z, t = active_axes(scheme, img.shape[0], img.shape[1], reg_z_over_reg, reg_time)
    one = d.dtype.type(1)
    G = _adjoint(g, scheme, z, t, one, one, None, one)
    if return_grad_norms:
        return tv_value, G, norms
    return tv_value, G


# --------------------------------------------------------------------------------------------
# driver loops (README.md:107-124 and :141-157), generalised with keepdims so that they are
# defined beyond 2-D; identical to the README in the 2-D case
# --------------------------------------------------------------------------------------------
def time_weight_max(mask_static, factor_reg_static):
    """Largest per-pixel weight on reg_time: the time channels are scaled by sqrt(factor_reg_static) where a boolean
    mask is set, by sqrt(W) for a weight array W."""
    m = _mask_array(mask_static)
    if m is None:
        return 1.0
    if np.issubdtype(m.dtype, np.floating):
        return float(m.max())
    if not m.any():
        return 1.0
    return float(factor_reg_static) if m.all() else max(1.0, float(factor_reg_static))


def cp_step_size(scheme, Nz, M, reg_z_over_reg, reg_time, time_weight_max=1.0):
    """tau = 1 / (1 + L^2), L^2 = 4 (2 + reg_z [z] + reg_time * time_weight_max [t]) >= |D|^2; 1/9 in 2-D
    (README.md:143).  BUILD-DEFINED beyond 2-D (the reference ships only the 2-D snippet)."""
    z, t = active_axes(scheme, Nz, M, reg_z_over_reg, reg_time)
    return 1.0 / (1.0 + 4.0 * (2.0 + (reg_z_over_reg if z else 0.0) + (reg_time * time_weight_max if t else 0.0)))


def chambolle_pock(x0, n_iter, regularization, scheme="hybrid", reg_z_over_reg=1.0, reg_time=0.0,
                   sigma_D=0.5, sigma_A=1.0, tau=None, mask_static=False, factor_reg_static=0,
                   return_state=False):
    """README.md:141-157: dual fidelity update, dual TV update = projection on the l2,inf ball of
    radius ``regularization``, primal update, loss.  ``np.sum(.., axis=1)`` gets keepdims here."""
    x0 = np.asarray(x0)
    kw = dict(reg_z_over_reg=reg_z_over_reg, reg_time=reg_time, mask_static=mask_static,
              factor_reg_static=factor_reg_static)
    if tau is None:
        tau = cp_step_size(scheme, x0.shape[0], x0.shape[1], reg_z_over_reg, reg_time,
                           time_weight_max(mask_static, factor_reg_static))
    x = x0.copy()
    p = np.zeros_like(x0)
    q = np.zeros_like(D(x0, scheme, **kw))
    loss = np.zeros(n_iter)
    for it in range(n_iter):
        p = (p + sigma_A * (x - x0)) / (1.0 + sigma_A)
        Dx = D(x, scheme, **kw)
        v = q + sigma_D * Dx
        q = v / np.maximum(1.0, np.sqrt(np.sum(v ** 2, axis=1, keepdims=True)) / regularization)
        x = x - tau * p - tau * D_T(q, scheme, **kw)
        loss[it] = 0.5 * np.sum(np.square(x - x0)) + regularization * compute_L21_norm(Dx)
    if return_state:
        return x, loss, p, q
    return x, loss


def subgradient_descent(x0, n_iter, regularization, step_size, scheme="hybrid", reg_z_over_reg=1.0,
                        reg_time=0.0, mask_static=False, factor_reg_static=0):
    """README.md:118-124."""
    x0 = np.asarray(x0)
    x = x0.copy()
    loss = np.zeros(n_iter)
    for it in range(n_iter):
        tv_value, G = tv(x, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
        x = x - step_size * ((x - x0) + regularization * G)
        loss[it] = 0.5 * np.sum(np.square(x - x0)) + regularization * tv_value
    return x, loss


def group_soft_threshold(v, thresh):
    """prox of thresh*||.||_{2,1}: v * max(0, 1 - thresh/|v|_2), 0 where |v| == 0."""
    n = np.sqrt(np.sum(v ** 2, axis=1, keepdims=True))
    scale = np.where(n > 0, np.maximum(0.0, 1.0 - thresh / np.where(n > 0, n, 1.0)), 0.0)
    return v * scale


def normal_spectral_bound(scheme, shape, reg_z_over_reg=1.0, reg_time=0.0, mask_static=False, factor_reg_static=0):
    """L >= lambda_max(D^T D): a forward / backward difference has norm^2 <= 4, a halved central one <= 1, each axis weighted
    by the square of its weight (the time axis by the largest per-pixel factor); hybrid is the mean of the two one-sided
    operators.  The same rule as pytv.solvers (and as the Chambolle-Pock step size 1 / (1 + L), README.md:141-143)."""
    nz, m = shape[0], shape[1]
    twmax = 1.0
    if not isinstance(mask_static, bool):
        mk = np.asarray(mask_static)
        if mk.dtype == bool:
            twmax = float(factor_reg_static) if bool(mk.all()) else max(1.0, float(factor_reg_static))
        else:
            twmax = float(mk.max())
    s = 2.0 + (reg_z_over_reg if (nz > 1 and reg_z_over_reg > 0) else 0.0) + (reg_time * twmax if (m > 1 and reg_time > 0) else 0.0)
    return (1.0 if scheme == "central" else 4.0) * s


def chebyshev_coefficients(lmax, n):
    """(alpha_k, beta_k), k = 0 .. n-1, of the Chebyshev iteration e_{k+1} = e_k + alpha_k (b - A e_k) + beta_k (e_k - e_{k-1})
    (e_0 = e_{-1} = 0) for a symmetric A with spectrum in [1, lmax] (Saad, Iterative Methods, alg. 12.1, written as a
    three-term recurrence in e)."""
    theta, delta = 0.5 * (lmax + 1.0), 0.5 * (lmax - 1.0)
    if delta <= 1e-14 * theta:
        return [(1.0 / theta, 0.0)] * n
    sigma = theta / delta
    out, rho_prev = [(1.0 / theta, 0.0)], 1.0 / sigma
    for _ in range(1, n):
        rho_k = 1.0 / (2.0 * sigma - rho_prev)
        out.append((2.0 * rho_k / delta, rho_k * rho_prev))
        rho_prev = rho_k
    return out[:n]


def admm(x0, n_outer, regularization, rho, n_cg, scheme="hybrid", reg_z_over_reg=1.0, reg_time=0.0,
         mask_static=False, factor_reg_static=0, return_state=False, single_reduction=False, x_solver="cg"):
    """Scaled-form ADMM for min 1/2|x-x0|^2 + reg |z|_{2,1} s.t. Dx = z.  NOT in the reference
    (README.md:26,135 only mention it): build-defined, parity unpinned; this NumPy version pins
    the HIP implementation to the same arithmetic.
      x-step : (I + rho D^T D) x = x0 + rho D^T (z - u), n_cg conjugate-gradient steps, warm start
      z-step : z = group_soft_threshold(Dx + u, reg/rho)
      u-step : u += Dx - z
    loss[k] = 1/2|x-x0|^2 + reg |Dx|_{2,1} after the x-step.
    single_reduction: the Chronopoulos-Gear form of the same CG recurrence (w = A r, gamma = <r,r>, delta = <r,w>
    reduced together: one all-reduce per step on a sharded volume) -- what pytv.solvers.ADMM(single_reduction=True)
    runs; identical to the textbook form in exact arithmetic.
    x_solver="chebyshev": n_cg steps of the Chebyshev iteration on A e = b - A x (spectrum in [1, 1 + rho L],
    normal_spectral_bound) instead of CG -- what pytv.solvers.ADMM(x_solver="chebyshev") runs (no dot products)."""
    x0 = np.asarray(x0)
    kw = dict(reg_z_over_reg=reg_z_over_reg, reg_time=reg_time, mask_static=mask_static,
              factor_reg_static=factor_reg_static)
    x = x0.copy()
    z = np.zeros_like(D(x0, scheme, **kw))
    u = np.zeros_like(z)
    loss = np.zeros(n_outer)

    def A(v):
        return v + rho * D_T(D(v, scheme, **kw), scheme, **kw)

    for k in range(n_outer):
        b = x0 + rho * D_T(z - u, scheme, **kw)
        r = b - A(x)
        d = r.copy()
        rs = float(np.sum(r.astype(np.float64) ** 2))
        if x_solver == "chebyshev":
            L = normal_spectral_bound(scheme, x0.shape, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
            coef = chebyshev_coefficients(1.0 + rho * L, n_cg)
            e_prev, e = np.zeros_like(r), np.zeros_like(r)
            for a_k, b_k in coef:
                e, e_prev = e + e.dtype.type(a_k) * (r - A(e)) + e.dtype.type(b_k) * (e - e_prev), e
            x = x + e
        elif single_reduction:
            w = A(r)
            gamma, delta = rs, float(np.sum(r.astype(np.float64) * w))
            alpha, beta = (gamma / delta if delta > 0 else 0.0), 0.0
            d, s = np.zeros_like(r), np.zeros_like(r)
            for c in range(n_cg):
                d = r + d.dtype.type(beta) * d
                s = w + s.dtype.type(beta) * s              # s = A d
                x = x + x.dtype.type(alpha) * d
                r = r - r.dtype.type(alpha) * s
                if c + 1 == n_cg:
                    break
                w = A(r)
                gamma_new = float(np.sum(r.astype(np.float64) ** 2))
                delta = float(np.sum(r.astype(np.float64) * w))
                beta = gamma_new / gamma if gamma > 0 else 0.0
                den = delta - beta * gamma_new / alpha if alpha != 0.0 else 0.0
                alpha = gamma_new / den if den > 0 else 0.0
                gamma = gamma_new
        for _ in range(0 if (single_reduction or x_solver == "chebyshev") else n_cg):
            Ad = A(d)
            dAd = float(np.sum(d.astype(np.float64) * Ad))
            alpha = rs / dAd if dAd > 0 else 0.0
            x = x + x.dtype.type(alpha) * d
            r = r - r.dtype.type(alpha) * Ad
            rs_new = float(np.sum(r.astype(np.float64) ** 2))
            beta = rs_new / rs if rs > 0 else 0.0
            d = r + d.dtype.type(beta) * d
            rs = rs_new
        Dx = D(x, scheme, **kw)
        z = group_soft_threshold(Dx + u, regularization / rho)
        u = u + Dx - z
        loss[k] = 0.5 * np.sum(np.square(x - x0)) + regularization * compute_L21_norm(Dx)
    if return_state:
        return x, loss, z, u
    return x, loss


# --------------------------------------------------------------------------------------------
# synthetic phantom used by bench.py and the parity tests; pure function of the
# global voxel index so that any z-slab of it can be produced independently
# --------------------------------------------------------------------------------------------
def phantom(shape, z0=0, nz_local=None, seed=1234, n_boxes=32, dtype=np.float32):
    """Piecewise-constant 4-D phantom: sum of ``n_boxes`` axis-aligned boxes with amplitudes in
    [0, 255/4), each box drifting by one column every time frame."""
    Nz, M, Ny, Nx = shape
    nz_local = Nz if nz_local is None else nz_local
    rng = np.random.RandomState(seed)
    boxes = []
    for _ in range(n_boxes):
        c = rng.rand(3)
        h = 0.05 + 0.25 * rng.rand(3)
        amp = rng.rand() * 255.0 / 4.0
        boxes.append((c, h, amp))
    zz = (np.arange(z0, z0 + nz_local) + 0.5) / Nz
    yy = (np.arange(Ny) + 0.5) / Ny
    xx = (np.arange(Nx) + 0.5) / Nx
    out = np.zeros((nz_local, M, Ny, Nx), dtype=np.float64)
    for c, h, amp in boxes:
        mz = (np.abs(zz - c[0]) < h[0])[:, None, None]
        my = (np.abs(yy - c[1]) < h[1])[None, :, None]
        for t in range(M):
            mx = (np.abs(xx - c[2] - t / Nx) < h[2])[None, None, :]
            out[:, t] += amp * (mz & my & mx)
    return out.astype(dtype)

"""ctypes front-end of the C / OpenMP oracle (oracle/tv_oracle_c.c).  TEST INFRASTRUCTURE ONLY."""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "tv_oracle_c.c")
INC = os.path.join(HERE, "tv_oracle_c.inc")
OUT = os.path.join(HERE, "_build", "libtv_oracle_c.so")
SCHEMES = {"upwind": 0, "downwind": 1, "central": 2, "hybrid": 3}


class Geom(ctypes.Structure):
    _fields_ = [("nz", ctypes.c_long), ("m", ctypes.c_long), ("ny", ctypes.c_long), ("nx", ctypes.c_long),
                ("scheme", ctypes.c_int), ("nd", ctypes.c_int), ("za", ctypes.c_int), ("ta", ctypes.c_int),
                ("wz", ctypes.c_double), ("wt", ctypes.c_double), ("sf", ctypes.c_double), ("mask", ctypes.c_void_p)]


def build(force=False):
    override = os.environ.get("TV_ORACLE_C_LIB")        # a sanitizer build made by tools/sanitize.py
    if override:
        return override
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= max(os.path.getmtime(SRC), os.path.getmtime(INC)):
        return OUT
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    subprocess.check_call(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-std=c11", SRC, "-o", OUT, "-lm"])
    return OUT


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        for suf in ("_f32", "_f64"):
            getattr(_lib, "tvc_l21" + suf).restype = ctypes.c_double
    return _lib


def _geom(shape, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static):
    nz, m, ny, nx = shape
    g = Geom()
    g.nz, g.m, g.ny, g.nx = nz, m, ny, nx
    g.scheme = SCHEMES[scheme]
    g.za = int(nz > 1 and reg_z_over_reg > 0)
    g.ta = int(m > 1 and reg_time > 0)
    g.nd = (2 if scheme == "hybrid" else 1) * (2 + g.za + g.ta)
    g.wz, g.wt, g.sf = np.sqrt(reg_z_over_reg), np.sqrt(reg_time), np.sqrt(factor_reg_static)
    keep = None
    if not isinstance(mask_static, bool):
        keep = np.ascontiguousarray(np.broadcast_to(np.asarray(mask_static, dtype=bool), (1, 1, ny, nx)).reshape(ny, nx).astype(np.uint8))
        g.mask = keep.ctypes.data
    return g, keep


def _suf(a):
    return "_f32" if a.dtype == np.float32 else "_f64"


def D(img, scheme, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0):
    x = np.ascontiguousarray(img, dtype=np.float32 if np.asarray(img).dtype == np.float32 else np.float64)
    g, keep = _geom(x.shape, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
    d = np.empty((x.shape[0], g.nd) + x.shape[1:], dtype=x.dtype)
    getattr(lib(), "tvc_D" + _suf(x))(ctypes.byref(g), x.ctypes.data_as(ctypes.c_void_p), d.ctypes.data_as(ctypes.c_void_p))
    return d


def D_T(y, scheme, reg_z_over_reg=1.0, reg_time=0, mask_static=False, factor_reg_static=0):
    y = np.ascontiguousarray(y, dtype=np.float32 if np.asarray(y).dtype == np.float32 else np.float64)
    shape = (y.shape[0],) + y.shape[2:]
    g, keep = _geom(shape, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
    assert g.nd == y.shape[1], (g.nd, y.shape)
    out = np.empty(shape, dtype=y.dtype)
    getattr(lib(), "tvc_DT" + _suf(y))(ctypes.byref(g), y.ctypes.data_as(ctypes.c_void_p), out.ctypes.data_as(ctypes.c_void_p))
    return out


def compute_L21_norm(d, shape_img, scheme="upwind"):
    d = np.ascontiguousarray(d)
    g, _ = _geom(shape_img, scheme, 1.0, 1.0, False, 0)
    g.nd = d.shape[1]
    return getattr(lib(), "tvc_l21" + _suf(d))(ctypes.byref(g), d.ctypes.data_as(ctypes.c_void_p))


def chambolle_pock(x0, n_iter, regularization, scheme="hybrid", reg_z_over_reg=1.0, reg_time=0.0, sigma_D=0.5, sigma_A=1.0,
                   tau=None, mask_static=False, factor_reg_static=0, numa=False):
    """Same iteration as tv_oracle.chambolle_pock, in C with OpenMP (all host cores unless OMP_NUM_THREADS says otherwise).
    numa=True: the working arrays are allocated and first touched by the worker threads (bench.py's multi-core baseline);
    returns (x, loss, seconds of the iterations alone)."""
    x0 = np.ascontiguousarray(x0, dtype=np.float32 if np.asarray(x0).dtype == np.float32 else np.float64)
    g, keep = _geom(x0.shape, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
    if tau is None:
        from . import tv_oracle as _orc
        tau = _orc.cp_step_size(scheme, x0.shape[0], x0.shape[1], reg_z_over_reg, reg_time,
                                _orc.time_weight_max(mask_static, factor_reg_static))
    loss = np.zeros(n_iter)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    if numa:
        x, secs = np.empty_like(x0), ctypes.c_double(0.0)
        getattr(lib(), "tvc_cp_numa" + _suf(x0))(ctypes.byref(g), vp(x), vp(x0), ctypes.c_int(n_iter), ctypes.c_double(regularization),
                                                  ctypes.c_double(sigma_D), ctypes.c_double(sigma_A), ctypes.c_double(tau), vp(loss),
                                                  ctypes.byref(secs))
        return x, loss, secs.value
    x, p = x0.copy(), np.zeros_like(x0)
    q = np.zeros((x0.shape[0], g.nd) + x0.shape[1:], dtype=x0.dtype)
    d, dt = np.empty_like(q), np.empty_like(x0)
    getattr(lib(), "tvc_cp" + _suf(x0))(ctypes.byref(g), vp(x), vp(x0), vp(p), vp(q), vp(d), vp(dt), ctypes.c_int(n_iter),
                                         ctypes.c_double(regularization), ctypes.c_double(sigma_D), ctypes.c_double(sigma_A),
                                         ctypes.c_double(tau), vp(loss))
    return x, loss


def admm(x0, n_outer, regularization, rho, n_cg, scheme="hybrid", reg_z_over_reg=1.0, reg_time=0.0, mask_static=False,
         factor_reg_static=0, single_reduction=False, return_state=False):
    """Same iteration as tv_oracle.admm, in C with OpenMP.  single_reduction: the Chronopoulos-Gear form of the CG
    recurrence (one reduction per step), otherwise the textbook one."""
    x0 = np.ascontiguousarray(x0, dtype=np.float32 if np.asarray(x0).dtype == np.float32 else np.float64)
    g, keep = _geom(x0.shape, scheme, reg_z_over_reg, reg_time, mask_static, factor_reg_static)
    x = x0.copy()
    z = np.zeros((x0.shape[0], g.nd) + x0.shape[1:], dtype=x0.dtype)
    u = np.zeros_like(z)
    img = np.empty((4,) + x0.shape, dtype=x0.dtype)
    dw = np.empty((2,) + z.shape, dtype=x0.dtype)
    loss = np.zeros(n_outer)
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    getattr(lib(), "tvc_admm" + _suf(x0))(ctypes.byref(g), vp(x), vp(x0), vp(z), vp(u), vp(img), vp(dw), ctypes.c_int(n_outer),
                                           ctypes.c_int(n_cg), ctypes.c_double(regularization), ctypes.c_double(rho),
                                           ctypes.c_int(int(bool(single_reduction))), vp(loss))
    if return_state:
        return x, loss, z, u
    return x, loss

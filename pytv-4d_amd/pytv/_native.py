"""ctypes binding of libpytv4d_hip.so (C-ABI: include/pytv4d.h).

The HIP library is the ONLY compute path of this package.  If it cannot be loaded the import
fails loudly -- there is no CPU or PyTorch fallback.  PyTorch is used for device memory, the
current HIP stream and (in ``slab.py``) ``torch.distributed``; it is imported BEFORE the library
so that the library binds to the same ``libamdhip64`` runtime PyTorch already loaded (one HIP
runtime per process).
"""
import ctypes
import os

import numpy as np
import torch  # noqa: F401  (must come first: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# PYTV4D_LIB: developer override (A/B of an experimental build of the same C-ABI: pytv-4d_amd/build.py TV_VARIANT=...)
LIB_PATH = os.environ.get("PYTV4D_LIB") or os.path.join(_HERE, "libpytv4d_hip.so")

SCHEMES = {"upwind": 0, "downwind": 1, "central": 2, "hybrid": 3}
TV_F32, TV_F64 = 0, 1

_c_void_p = ctypes.c_void_p
_c_double_p = ctypes.c_void_p   # device pointers to fp64 scalars are passed as raw addresses


class TvGeom(ctypes.Structure):
    """struct tv_geom of include/pytv4d.h"""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("abi_version", ctypes.c_uint32),
        ("nz", ctypes.c_int64), ("m", ctypes.c_int64), ("ny", ctypes.c_int64), ("nx", ctypes.c_int64),
        ("nz_global", ctypes.c_int64), ("z0", ctypes.c_int64),
        ("scheme", ctypes.c_int32), ("dtype", ctypes.c_int32),
        ("reg_z_over_reg", ctypes.c_double), ("reg_time", ctypes.c_double), ("factor_reg_static", ctypes.c_double),
        ("mask_static", ctypes.c_void_p),
        ("time_factor", ctypes.c_void_p),
        ("time_weight_vol", ctypes.c_void_p),
        ("time_weight_prev", ctypes.c_void_p),
        ("time_weight_next", ctypes.c_void_p),
        ("row_pitch", ctypes.c_int64), ("frame_pitch", ctypes.c_int64),
    ]


ABI_VERSION = 5      # TV_ABI_VERSION of include/pytv4d.h this binding was written against


def new_geom():
    """A zeroed ``tv_geom`` stamped with this binding's struct size and interface version (tv_geom_init of the header)."""
    g = TvGeom()
    g.struct_size = ctypes.sizeof(TvGeom)
    g.abi_version = ABI_VERSION
    return g


_G = ctypes.POINTER(TvGeom)
_SIGNATURES = {
    # name: (restype, argtypes)
    "tv_last_error": (ctypes.c_char_p, []),
    "tv_version": (ctypes.c_int, []),
    "tv_abi_version": (ctypes.c_int, []),
    "tv_set_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "tv_unset_option": (ctypes.c_int, [ctypes.c_char_p]),
    "tv_get_option": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
    "tv_ctx_unique_id": (ctypes.c_int, [_c_void_p]),
    "tv_ctx_create": (ctypes.c_int, [ctypes.POINTER(_c_void_p), ctypes.c_int, ctypes.c_int, _c_void_p, ctypes.c_int]),
    "tv_ctx_destroy": (ctypes.c_int, [_c_void_p]),
    "tv_ctx_rank": (ctypes.c_int, [_c_void_p]),
    "tv_ctx_size": (ctypes.c_int, [_c_void_p]),
    "tv_halo_exchange": (ctypes.c_int, [_c_void_p, ctypes.c_int32, ctypes.c_int64, ctypes.c_int, ctypes.c_int] + [_c_void_p] * 5),
    "tv_allreduce_f64": (ctypes.c_int, [_c_void_p, _c_void_p, ctypes.c_int64, ctypes.c_int32, _c_void_p]),
    "tv_num_channels": (ctypes.c_int, [_G]),
    "tv_workspace_bytes": (ctypes.c_size_t, [_G]),
    "tv_small_supported": (ctypes.c_int, [_G]),
    "tv_small_workspace_bytes": (ctypes.c_size_t, [_G, ctypes.c_int64]),
    "tv_small_cp": (ctypes.c_int, [_G] + [_c_void_p] * 4 + [ctypes.c_double] * 4 + [ctypes.c_int64, _c_double_p, ctypes.c_int64, ctypes.c_int64, _c_void_p, _c_void_p]),
    "tv_small_subgrad_descent": (ctypes.c_int, [_G] + [_c_void_p] * 4 + [ctypes.c_double] * 2 + [ctypes.c_int64, _c_double_p, ctypes.c_int64, ctypes.c_int64, _c_void_p, _c_void_p]),
    "tv_D": (ctypes.c_int, [_G] + [_c_void_p] * 5),
    "tv_DT": (ctypes.c_int, [_G] + [_c_void_p] * 5),
    "tv_l21": (ctypes.c_int, [_G, _c_void_p, ctypes.c_int32, _c_void_p, _c_double_p, _c_void_p, _c_void_p]),
    "tv_subgrad": (ctypes.c_int, [_G] + [_c_void_p] * 5 + [_c_double_p, _c_void_p, _c_void_p]),
    "tv_subgrad_fused_supported": (ctypes.c_int, [_G]),
    "tv_subgrad_fused": (ctypes.c_int, [_G] + [_c_void_p] * 4 + [_c_double_p, _c_void_p, _c_void_p]),
    "tv_subgrad_fused_norms": (ctypes.c_int, [_G] + [_c_void_p] * 5 + [_c_double_p, _c_void_p, _c_void_p]),
    "tv_subgrad_step_fused": (ctypes.c_int, [_G] + [_c_void_p] * 5 + [ctypes.c_double, ctypes.c_double, _c_double_p, _c_double_p,
                                                                      _c_void_p, _c_void_p]),
    "tv_cp_dual": (ctypes.c_int, [_G] + [_c_void_p] * 4 + [ctypes.c_double, ctypes.c_double, _c_double_p, _c_void_p, _c_void_p]),
    "tv_cp_primal": (ctypes.c_int, [_G] + [_c_void_p] * 6 + [ctypes.c_double, ctypes.c_double, _c_double_p, _c_void_p, _c_void_p]),
    "tv_cp_fused_supported": (ctypes.c_int, [_G]),
    "tv_cp_zchunk": (ctypes.c_int, [_G]),
    "tv_cp_fused": (ctypes.c_int, [_G] + [_c_void_p] * 7 + [ctypes.c_double] * 4 + [ctypes.c_int64] * 2 + [_c_double_p] * 2
                    + [_c_void_p, _c_void_p]),
    "tv_cp_sweep": (ctypes.c_int, [_G] + [_c_void_p] * 8 + [ctypes.c_double] * 4 + [ctypes.c_int32] + [ctypes.c_int64] * 2 + [_c_double_p] * 2
                    + [_c_void_p, _c_void_p]),
    "tv_cp_fixup": (ctypes.c_int, [_G] + [_c_void_p] * 5 + [ctypes.c_double] + [ctypes.c_int64] * 2 + [_c_double_p, _c_void_p, _c_void_p]),
    "tv_admm_zu": (ctypes.c_int, [_G] + [_c_void_p] * 5 + [ctypes.c_double, _c_double_p, _c_void_p, _c_void_p]),
    "tv_cpop_fused": (ctypes.c_int, [_G] + [_c_void_p] * 6 + [ctypes.c_double] * 3 + [ctypes.c_int64] * 2 + [_c_double_p, _c_void_p, _c_void_p]),
    "tv_cpop_fixup": (ctypes.c_int, [_G] + [_c_void_p] * 4 + [ctypes.c_double] + [ctypes.c_int64] * 2 + [_c_void_p, _c_void_p]),
    "tv_cheb_step": (ctypes.c_int, [_G] + [_c_void_p] * 3 + [ctypes.c_double] + [_c_void_p] * 2 + [ctypes.c_double] + [_c_void_p] * 2
                     + [ctypes.c_double] * 2 + [_c_void_p, _c_double_p, _c_void_p, _c_void_p]),
    "tv_axpby": (ctypes.c_int, [_G, ctypes.c_double, _c_void_p, ctypes.c_double, _c_void_p, _c_void_p, _c_void_p, _c_double_p, _c_void_p, _c_void_p]),
    "tv_admm_fused": (ctypes.c_int, [_G] + [_c_void_p] * 7 + [ctypes.c_double, ctypes.c_double, ctypes.c_int32] + [ctypes.c_int64] * 2
                      + [_c_double_p] * 2 + [_c_void_p, _c_void_p]),
    "tv_admm_sweep": (ctypes.c_int, [_G] + [_c_void_p] * 8 + [ctypes.c_double, ctypes.c_double, ctypes.c_int32] + [ctypes.c_int64] * 2
                      + [_c_double_p] * 2 + [_c_void_p, _c_void_p]),
    "tv_admm_fixup": (ctypes.c_int, [_G] + [_c_void_p] * 4 + [ctypes.c_double] + [ctypes.c_int64] * 2 + [_c_double_p, _c_void_p, _c_void_p]),
    "tv_DT_axpy": (ctypes.c_int, [_G] + [_c_void_p] * 5 + [ctypes.c_double, _c_void_p, _c_void_p]),
    "tv_DT_axpy2": (ctypes.c_int, [_G] + [_c_void_p] * 6 + [ctypes.c_double, ctypes.c_double, _c_void_p, _c_void_p]),
    "tv_cpop_p": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, _c_void_p, _c_void_p, ctypes.c_double, _c_void_p]),
    "tv_cpop_residual": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64, _c_void_p, _c_void_p, _c_void_p, _c_double_p, _c_void_p, _c_void_p]),
    "tv_normal_op": (ctypes.c_int, [_G] + [_c_void_p] * 3 + [ctypes.c_double, _c_void_p, _c_double_p, _c_void_p, _c_void_p]),
    "tv_normal_op2": (ctypes.c_int, [_G] + [_c_void_p] * 3 + [ctypes.c_double] + [_c_void_p] * 3 + [_c_double_p, _c_void_p, _c_void_p]),
    "tv_cg_update": (ctypes.c_int, [_G] + [_c_void_p] * 5 + [_c_double_p, _c_void_p, _c_double_p, _c_void_p, _c_void_p]),
    "tv_admm_tu": (ctypes.c_int, [_G] + [_c_void_p] * 5 + [ctypes.c_double, _c_double_p, _c_void_p, _c_void_p]),
    "tv_cg_step1": (ctypes.c_int, [_G] + [_c_void_p] * 4 + [_c_double_p] * 3 + [_c_void_p, _c_void_p]),
    "tv_cg_step2": (ctypes.c_int, [_G] + [_c_void_p] * 2 + [_c_double_p] * 2 + [_c_void_p]),
    "tv_sub": (ctypes.c_int, [ctypes.c_int32, ctypes.c_int64] + [_c_void_p] * 4),
    "tv_dot": (ctypes.c_int, [_G, _c_void_p, _c_void_p, _c_double_p, _c_void_p, _c_void_p]),
    "tv_subgrad_step": (ctypes.c_int, [_G, _c_void_p, _c_void_p, _c_void_p, ctypes.c_double, ctypes.c_double,
                                       _c_double_p, _c_void_p, _c_void_p]),
}

_lib = None


def lib():
    """The loaded C-ABI library; raises ImportError if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "pytv: native HIP library %s is missing. Build it with "
                "`python pytv-4d_amd/build.py` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        if handle.tv_abi_version() != ABI_VERSION:
            raise ImportError("pytv: %s implements interface version %d, this binding %d -- rebuild the library "
                              "(python pytv-4d_amd/build.py --force)" % (LIB_PATH, handle.tv_abi_version(), ABI_VERSION))
        _lib = handle
    return _lib


def check(rc):
    """Map a C-ABI status to a Python exception (0 ok, <0 argument error, >0 hipError_t)."""
    if rc == 0:
        return
    msg = lib().tv_last_error().decode("utf-8", "replace")
    if rc < 0:
        raise ValueError("pytv native: %s (code %d)" % (msg, rc))
    raise RuntimeError("pytv native: HIP error %d: %s" % (rc, msg))


def set_option(name, value):
    """Set (value is an int) or unset (value is None) one of the library's tuning options (include/pytv4d.h)."""
    if value is None:
        check(lib().tv_unset_option(name.encode()))
    else:
        check(lib().tv_set_option(name.encode(), int(value)))


def get_option(name, default):
    return int(lib().tv_get_option(name.encode(), int(default)))


def dtype_code(torch_dtype):
    if torch_dtype == torch.float32:
        return TV_F32
    if torch_dtype == torch.float64:
        return TV_F64
    raise ValueError("pytv native: only float32 / float64 arrays are supported, got %s" % torch_dtype)


def ptr(t):
    """Device address of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def current_stream(device):
    return torch.cuda.current_stream(device).cuda_stream


class Geometry:
    """Python-side owner of a ``tv_geom``: keeps the device mask alive and caches the workspace."""

    def __init__(self, shape, scheme, dtype, device, reg_z_over_reg=1.0, reg_time=0.0, mask_static=False,
                 factor_reg_static=0.0, nz_global=None, z0=0, weight_halo=None, weight_dev=None, row_pitch=0, frame_pitch=0):
        """mask_static: ``False``, a boolean mask broadcastable from (1, 1, Ny, Nx) (the reference's form, with
        ``factor_reg_static``), or a FLOAT array of weights on the time regularisation -- per pixel (broadcastable from
        (1, 1, Ny, Nx)) or per voxel (broadcastable to (Nz, M, Ny, Nx): the reference's to-do "weight matrix of size
        Nz x M x N x N", README.md:258).  The time channels are multiplied by sqrt(weight).
        weight_halo: (plane z0-1, plane z0+nz) of a per-voxel weight on a z-slab, each (M, Ny, Nx) or None (only the
        sub-gradient on a slab needs them).  weight_dev: internal -- (vol, prev, next) device tensors that already hold
        sqrt(weight) for exactly these planes (sub-slab geometries of the solvers).
        row_pitch / frame_pitch (elements, 0 = dense): the arrays of this geometry are PITCHED (tv_geom::row_pitch / frame_pitch of
        include/pytv4d.h; pads hold zeros) -- allocate them with ``new_image`` / ``new_grad``, which return (Nz, M, Ny, Nx) /
        (Nz, Nd, M, Ny, Nx) VIEWS of zero-filled padded storage."""
        if scheme not in SCHEMES:
            raise ValueError("unknown TV scheme %r" % (scheme,))
        if len(shape) != 4:
            raise ValueError("image must be 4-D (Nz, M, N, N), got shape %s" % (tuple(shape),))
        nz, m, ny, nx = (int(v) for v in shape)
        self.shape = (nz, m, ny, nx)
        self.scheme = scheme
        self.dtype = dtype
        self.device = torch.device(device)
        self.mask_dev = None
        self.factor_dev = None
        self.weight_vol = None          # (sqrt-weight volume, plane z0-1 or None, plane z0+nz or None), device tensors
        # largest per-pixel weight on reg_time: the time channels are scaled by sqrt(factor_reg_static) where the mask is
        # set and by sqrt(weight) of a weight map, so |D|^2 <= 4 (2 + reg_z + reg_time * time_weight_max)
        self.time_weight_max = 1.0
        if not isinstance(mask_static, bool):
            mk = torch.as_tensor(np.asarray(mask_static.detach().cpu()) if isinstance(mask_static, torch.Tensor)
                                 else np.asarray(mask_static))
            if mk.dtype.is_floating_point:
                # a FLOAT array is the weight of the time regularisation (the reference's to-do, README.md:258: "replace
                # mask_static, factor_reg_static with a weight matrix"): the time channels are multiplied by
                # sqrt(weight); where(mask, factor, 1) reproduces the boolean mask exactly
                if bool((mk < 0).any()):
                    raise ValueError("weights must be non-negative")
                per_pixel = mk.dim() <= 2 or all(int(v) == 1 for v in mk.shape[:-2])
                if per_pixel:
                    wm = torch.broadcast_to(mk.to(torch.float64), (1, 1, ny, nx)).reshape(ny, nx)
                    self.factor_dev = torch.sqrt(wm).to(dtype).contiguous().to(self.device)
                else:       # per voxel: (Nz, M, Ny, Nx)
                    wm = torch.broadcast_to(mk.to(torch.float64), (nz, m, ny, nx))
                    halo = [None, None]
                    for k in (0, 1):
                        if weight_halo is not None and weight_halo[k] is not None:
                            hk = torch.as_tensor(np.asarray(weight_halo[k].detach().cpu()) if isinstance(weight_halo[k], torch.Tensor)
                                                 else np.asarray(weight_halo[k])).to(torch.float64)
                            halo[k] = torch.sqrt(torch.broadcast_to(hk, (m, ny, nx))).to(dtype).contiguous().to(self.device)
                    self.weight_vol = (torch.sqrt(wm).to(dtype).contiguous().to(self.device), halo[0], halo[1])
                self.time_weight_max = float(wm.max())
            else:
                mk = torch.broadcast_to(mk.to(torch.bool), (1, 1, ny, nx)).reshape(ny, nx)
                self.mask_dev = mk.to(torch.uint8).contiguous().to(self.device)
                if bool(mk.any()):
                    self.time_weight_max = float(factor_reg_static) if bool(mk.all()) else max(1.0, float(factor_reg_static))
        g = new_geom()
        g.nz, g.m, g.ny, g.nx = nz, m, ny, nx
        g.nz_global = nz if nz_global is None else int(nz_global)
        g.z0 = int(z0)
        g.scheme = SCHEMES[scheme]
        g.dtype = dtype_code(dtype)
        g.reg_z_over_reg = float(reg_z_over_reg)
        g.reg_time = float(reg_time)
        g.factor_reg_static = float(factor_reg_static)
        g.mask_static = ptr(self.mask_dev)
        g.time_factor = ptr(self.factor_dev)
        g.row_pitch, g.frame_pitch = int(row_pitch), int(frame_pitch)
        self.row_pitch = int(row_pitch) if row_pitch else nx
        self.frame_pitch = int(frame_pitch) if frame_pitch else ny * self.row_pitch
        self.pitched = bool(row_pitch or frame_pitch)
        if weight_dev is not None:
            self.weight_vol = weight_dev
        if self.weight_vol is not None:
            if tuple(self.weight_vol[0].shape) != (nz, m, ny, nx):
                raise ValueError("weight volume has shape %s, the image %s" % (tuple(self.weight_vol[0].shape), (nz, m, ny, nx)))
            if self.pitched and not self.has_layout(self.weight_vol[0]):
                w = self.new_image()
                w.copy_(self.weight_vol[0])
                self.weight_vol = (w,) + tuple(self._plane_like(h) for h in self.weight_vol[1:])
            g.time_weight_vol = ptr(self.weight_vol[0])
            g.time_weight_prev = ptr(self.weight_vol[1])
            g.time_weight_next = ptr(self.weight_vol[2])
        self.c = g
        nd = lib().tv_num_channels(ctypes.byref(g))
        if nd < 0:
            check(nd)
        self.nd = nd
        self.z_active = g.nz_global > 1 and g.reg_z_over_reg > 0
        self.t_active = m > 1 and g.reg_time > 0
        self._ws = None
        self._scalars = None

    @property
    def ref(self):
        return ctypes.byref(self.c)

    @property
    def grad_shape(self):
        nz, m, ny, nx = self.shape
        return (nz, self.nd, m, ny, nx)

    @property
    def plane(self):
        """storage elements of one image plane (M frames; pads included when the geometry is pitched)"""
        return self.shape[1] * self.frame_pitch

    @property
    def image_elems(self):
        """storage elements of an image-like array of this geometry (pads included): what the flat vector helpers are given"""
        return self.shape[0] * self.plane

    # ---- pitched storage (tv_geom::row_pitch / frame_pitch) --------------------------------------------------------------
    def image_strides(self):
        return (self.shape[1] * self.frame_pitch, self.frame_pitch, self.row_pitch, 1)

    def new_image(self, planes=None, dtype=None):
        """(planes, M, Ny, Nx) view of ZERO-filled storage with this geometry's pitches (dense geometry: a plain zeros tensor)."""
        nz, m, ny, nx = self.shape
        n = nz if planes is None else int(planes)
        dt = self.dtype if dtype is None else dtype
        if not self.pitched:
            return torch.zeros((n, m, ny, nx), dtype=dt, device=self.device)
        buf = torch.zeros(n * m * self.frame_pitch, dtype=dt, device=self.device)
        return buf.as_strided((n, m, ny, nx), self.image_strides())

    def new_grad(self, planes=None):
        """(planes, Nd, M, Ny, Nx) view of zero-filled storage: one image per channel, channels between z and time."""
        nz, m, ny, nx = self.shape
        n = nz if planes is None else int(planes)
        if not self.pitched:
            return torch.zeros((n, self.nd, m, ny, nx), dtype=self.dtype, device=self.device)
        fp, rp = self.frame_pitch, self.row_pitch
        buf = torch.zeros(n * self.nd * m * fp, dtype=self.dtype, device=self.device)
        return buf.as_strided((n, self.nd, m, ny, nx), (self.nd * m * fp, m * fp, fp, rp, 1))

    def has_layout(self, t):
        """does the image-like tensor t (…, M, Ny, Nx) already have this geometry's pitches?"""
        want = self.image_strides()
        return tuple(t.stride()[-3:]) == want[-3:] and (t.dim() < 4 or t.shape[0] == 1 or t.stride(0) == want[0])

    def _plane_like(self, h):
        if h is None or not self.pitched:
            return h
        w = self.new_image(1)[0]
        w.copy_(h)
        return w

    def workspace(self):
        if self._ws is None:
            nbytes = lib().tv_workspace_bytes(self.ref)
            self._ws = torch.empty((nbytes + 7) // 8, dtype=torch.float64, device=self.device)
        return self._ws

    def scalar(self):
        """A fresh device fp64 scalar (0-d tensor)."""
        return torch.zeros((), dtype=torch.float64, device=self.device)

"""CPU-side checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol
include/pytv4d.h declares; argument validation that needs no GPU; the Python shim mirrors the
reference's public names and signatures (pytv/tv_operators_GPU.py, pytv/tv_GPU.py)."""
import ctypes
import inspect
import os
import re

import pytest

from conftest import ROOT, SCHEMES
from oracle import tv_oracle as orc

HEADER = os.path.join(ROOT, "include", "pytv4d.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+char\s*\*|int|size_t)\s+(tv_\w+)\s*\(", src, flags=re.M)
    return sorted(set(names))


def test_header_declares_the_contract():
    names = _declared_functions()
    for must in ("tv_D", "tv_DT", "tv_l21", "tv_subgrad", "tv_cp_dual", "tv_cp_primal", "tv_admm_zu",
                 "tv_DT_axpy", "tv_normal_op", "tv_cg_step1", "tv_cg_step2", "tv_num_channels",
                 "tv_workspace_bytes", "tv_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from pytv import _native as nv
    lib = ctypes.CDLL(nv.LIB_PATH)
    missing = [n for n in _declared_functions() if not hasattr(lib, n)]
    assert not missing, missing
    # and the ctypes table binds exactly the declared set
    assert sorted(nv._SIGNATURES) == _declared_functions()


def test_struct_layout_matches_header():
    from pytv import _native as nv
    # 2 x uint32 (struct_size, abi_version) + 6 x int64 + 2 x int32 + 3 x double + 5 pointers + 2 x int64 (row_pitch, frame_pitch:
    # interface version 4), no padding surprises
    assert ctypes.sizeof(nv.TvGeom) == 8 + 6 * 8 + 2 * 4 + 3 * 8 + 8 + 8 + 3 * 8 + 2 * 8
    assert nv.TvGeom.row_pitch.offset == 128 and nv.TvGeom.frame_pitch.offset == 136 and nv.ABI_VERSION == 5
    assert nv.TvGeom.struct_size.offset == 0 and nv.TvGeom.abi_version.offset == 4 and nv.TvGeom.nz.offset == 8
    assert nv.TvGeom.scheme.offset == 56 and nv.TvGeom.reg_z_over_reg.offset == 64 and nv.TvGeom.mask_static.offset == 88 and nv.TvGeom.time_factor.offset == 96 and nv.TvGeom.time_weight_vol.offset == 104 and nv.TvGeom.time_weight_next.offset == 120


def test_struct_version_stamp_is_enforced():
    """round-2 advice: the C-ABI changed incompatibly while tv_version() stood still.  tv_geom now starts with its own
    size and the interface version; a struct from another header is refused by every entry point, not mis-read."""
    from pytv import _native as nv
    lib = nv.lib()
    assert lib.tv_version() >= 300 and lib.tv_abi_version() == nv.ABI_VERSION
    # the header and the binding agree on the number
    hdr = open(os.path.join(ROOT, "include", "pytv4d.h")).read()
    assert "#define TV_ABI_VERSION %d" % nv.ABI_VERSION in hdr
    g = nv.new_geom()
    g.nz, g.m, g.ny, g.nx, g.nz_global, g.z0 = 4, 1, 8, 8, 4, 0
    assert lib.tv_num_channels(ctypes.byref(g)) == 2
    for field, val in (("struct_size", ctypes.sizeof(nv.TvGeom) - 24), ("struct_size", 0), ("abi_version", nv.ABI_VERSION - 1), ("abi_version", 0)):
        h = nv.new_geom()
        h.nz, h.m, h.ny, h.nx, h.nz_global, h.z0 = 4, 1, 8, 8, 4, 0
        setattr(h, field, val)
        assert lib.tv_num_channels(ctypes.byref(h)) == -1
        assert b"pytv4d.h" in lib.tv_last_error()
        assert lib.tv_workspace_bytes(ctypes.byref(h)) == 0
        assert lib.tv_D(ctypes.byref(h), None, None, None, None, None) == -1


@pytest.mark.parametrize("scheme", SCHEMES)
def test_channel_count_rule_matches_oracle(scheme):
    from pytv import _native as nv
    lib = nv.lib()
    for nz in (1, 3, 7):
        for m in (1, 2, 5):
            for lz in (0.0, 1.0, 2.5):
                for mu in (0.0, 0.7):
                    g = nv.new_geom()
                    g.nz, g.m, g.ny, g.nx, g.nz_global, g.z0 = nz, m, 6, 6, nz, 0
                    g.scheme, g.dtype = nv.SCHEMES[scheme], 0
                    g.reg_z_over_reg, g.reg_time = lz, mu
                    assert lib.tv_num_channels(ctypes.byref(g)) == orc.num_channels(scheme, nz, m, lz, mu)


def test_argument_errors_are_reported_not_thrown():
    from pytv import _native as nv
    lib = nv.lib()
    g = nv.new_geom()
    g.nz, g.m, g.ny, g.nx, g.nz_global, g.z0 = 4, 1, 8, 8, 4, 0
    g.scheme, g.dtype = 9, 0
    assert lib.tv_num_channels(ctypes.byref(g)) == -1
    assert b"scheme" in lib.tv_last_error()
    g.scheme = 0
    g.nz = 0
    assert lib.tv_num_channels(ctypes.byref(g)) == -1
    g.nz, g.z0 = 4, 2            # slab sticks out of the volume
    assert lib.tv_num_channels(ctypes.byref(g)) == -1
    g.z0 = 0
    # NULL arrays are rejected before anything touches the device
    assert lib.tv_D(ctypes.byref(g), None, None, None, None, None) == -1
    with pytest.raises(ValueError):
        nv.check(-1)


def test_shim_mirrors_reference_signatures():
    import pytv
    ops, tvg = pytv.tv_operators_GPU, pytv.tv_GPU
    op_params = ["img", "reg_z_over_reg", "reg_time", "mask_static", "factor_reg_static", "return_pytorch_tensor"]
    for s in SCHEMES:
        for prefix in ("D_", "D_T_"):
            sig = inspect.signature(getattr(ops, prefix + s))
            assert list(sig.parameters) == op_params
            assert sig.parameters["reg_z_over_reg"].default == 1.0 and sig.parameters["reg_time"].default == 0
            assert sig.parameters["mask_static"].default is False
        sig = inspect.signature(getattr(tvg, "tv_" + s))
        assert list(sig.parameters) == ["img", "mask", "reg_z_over_reg", "reg_time", "mask_static", "factor_reg_static",
                                        "return_pytorch_tensor", "return_grad_norms"]
    assert list(inspect.signature(ops.compute_L21_norm).parameters) == ["D_img", "return_array", "return_pytorch_tensor"]
    assert list(inspect.signature(ops.type_like).parameters) == ["array", "array_ref"]


def test_type_like_dtype_contract():
    import numpy as np
    import torch
    from pytv.tv_operators_GPU import type_like
    a = np.arange(4, dtype=np.int64)
    assert type_like(a, np.zeros(1, np.float32)).dtype == np.float32
    assert type_like(a, np.zeros(1, np.int16)).dtype == np.int16          # numpy/numpy copies the dtype
    assert type_like(a, torch.zeros(1, dtype=torch.float32)).dtype == np.float32
    assert type_like(a, torch.zeros(1, dtype=torch.float16)).dtype == np.float64
    t = torch.arange(4)
    assert type_like(t, np.zeros(1, np.float32)).dtype == torch.float32
    assert type_like(t, np.zeros(1, np.float64)).dtype == torch.float64
    assert type_like(t, torch.zeros(1, dtype=torch.float32)).dtype == torch.float32
    assert type_like(t, torch.zeros(1, dtype=torch.int32)).dtype == torch.float64


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pytv-4d_amd", "pytv")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "import oracle" not in src and "from oracle" not in src and "tv_oracle" not in src.replace("oracle/tv_oracle.py", ""), fn


def test_package_surface_says_what_is_missing(monkeypatch, tmp_path):
    """pytv.utils.cameraman() exists (clear message when the image is not available); the CPU twins are named
    as absent instead of failing with a bare AttributeError."""
    import numpy as np
    import pytv
    with pytest.raises(AttributeError, match="not part of the MI355X build"):
        pytv.tv_CPU
    p = tmp_path / "cam.npy"
    np.save(p, np.arange(256 * 256).reshape(256, 256))
    monkeypatch.setenv("PYTV_CAMERAMAN", str(p))
    assert pytv.utils.cameraman().shape == (256, 256)
    monkeypatch.delenv("PYTV_CAMERAMAN")
    try:
        import skimage  # noqa: F401
    except ImportError:
        with pytest.raises(FileNotFoundError, match="PYTV_CAMERAMAN"):
            pytv.utils.cameraman()


def test_auto_pitch_rule():
    """The solvers' padding rule (host logic, no GPU): whole-cache-line rows stay dense, rows rounded up to 128 bytes when that costs at
    most 8 %, short ragged rows only up to the 16-byte lane (pytv/solvers.py: auto_pitch; measured in profiles/r4_pitch_bench.txt)."""
    import torch
    from pytv.solvers import auto_pitch
    f32, f64 = torch.float32, torch.float64
    assert auto_pitch(1024, 1024, f32) is None and auto_pitch(512, 512, f64) is None            # every BASELINE configuration: dense
    assert auto_pitch(1000, 1000, f32) == (1024, 1000 * 1024)                                   # 1000 -> 1024 columns: + 2.4 %
    assert auto_pitch(1001, 1001, f32) == (1024, 1001 * 1024)
    assert auto_pitch(100, 100, f32) is None                                                    # 100 -> 128 would be + 28 %: stays dense (lane-aligned)
    assert auto_pitch(7, 9, f32) == (12, 7 * 12)                                                # ragged and short: up to the 16-byte lane only
    assert auto_pitch(9, 13, f64) == (14, 9 * 14)
    assert auto_pitch(1000, 1000, f64) == (1008, 1000 * 1008)                                   # 8000-byte rows -> 8064 (128-byte multiple)
    rp, fp = auto_pitch(16, 128, f32, frame_pad_bytes=4352)                                     # experiments: a frame pad on dense rows
    assert rp == 128 and fp == 16 * 128 + 1088


def test_one_sweep_path_refuses_frames_of_two_gib_and_more():
    """round-5 advice: the one-sweep kernel addresses x0 / p of the lagged primal update through a per-frame buffer descriptor whose
    size is the frame's byte count as a 32-bit number, and marks absent lanes with the offset 0x80000000: frames must stay below 2^31
    bytes.  tv_cp_fused_supported is a host-side predicate (no device call)."""
    from pytv import _native as nv
    lib = nv.lib()
    for dtype, nx_lim in ((0, 32768), (1, 16384)):          # ny = 16384 rows: 2^31 bytes per frame at this Nx
        for nx, want in ((nx_lim, 0), (2 * nx_lim, 0), (nx_lim - 64, 1), (1024, 1)):
            g = nv.new_geom()
            g.nz, g.m, g.ny, g.nx, g.nz_global, g.z0 = 2, 2, 16384, nx, 2, 0
            g.scheme, g.dtype = nv.SCHEMES["hybrid"], dtype
            g.reg_z_over_reg, g.reg_time = 1.0, 1.0
            assert lib.tv_cp_fused_supported(ctypes.byref(g)) == want, (dtype, nx)

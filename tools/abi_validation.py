#!/usr/bin/env python3
"""Drive every C-ABI entry point through the paths it takes BEFORE its first HIP call: argument validation, geometry
rules, workspace sizing, the option table.  No GPU, no torch: plain ctypes against the library named by PYTV4D_LIB (or
the in-tree one).  Run under ASan / UBSan by tools/sanitize.py; on its own it is a quick CPU check of the error contract
(0 ok, < 0 argument error with a message, nothing thrown across the ABI)."""
import ctypes
import itertools
import os
import sys

import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the interface version of the header in the tree (a struct stamped with another number is refused by every entry point: the checks
# below would then "pass" for the wrong reason -- round 6 found this file still stamping 4 after the bump to 5)
ABI_VERSION = int(re.search(r"#define TV_ABI_VERSION (\d+)", open(os.path.join(ROOT, "include", "pytv4d.h")).read()).group(1))
LIB = os.environ.get("PYTV4D_LIB") or os.path.join(ROOT, "pytv-4d_amd", "pytv", "libpytv4d_hip.so")


class TvGeom(ctypes.Structure):
    _fields_ = [("struct_size", ctypes.c_uint32), ("abi_version", ctypes.c_uint32), ("nz", ctypes.c_int64), ("m", ctypes.c_int64), ("ny", ctypes.c_int64), ("nx", ctypes.c_int64),
                ("nz_global", ctypes.c_int64), ("z0", ctypes.c_int64), ("scheme", ctypes.c_int32), ("dtype", ctypes.c_int32),
                ("reg_z_over_reg", ctypes.c_double), ("reg_time", ctypes.c_double), ("factor_reg_static", ctypes.c_double),
                ("mask_static", ctypes.c_void_p), ("time_factor", ctypes.c_void_p), ("time_weight_vol", ctypes.c_void_p),
                ("time_weight_prev", ctypes.c_void_p), ("time_weight_next", ctypes.c_void_p),
                ("row_pitch", ctypes.c_int64), ("frame_pitch", ctypes.c_int64)]


def geom(nz=4, m=3, ny=8, nx=16, scheme=3, dtype=0, nzg=None, z0=0, lz=1.0, mu=1.0, factor=0.0):
    g = TvGeom()
    g.struct_size, g.abi_version = ctypes.sizeof(TvGeom), ABI_VERSION
    g.nz, g.m, g.ny, g.nx, g.nz_global, g.z0 = nz, m, ny, nx, (nz if nzg is None else nzg), z0
    g.scheme, g.dtype, g.reg_z_over_reg, g.reg_time, g.factor_reg_static = scheme, dtype, lz, mu, factor
    return g


def main():
    lib = ctypes.CDLL(LIB)
    lib.tv_last_error.restype = ctypes.c_char_p
    lib.tv_workspace_bytes.restype = ctypes.c_size_t
    P = ctypes.c_void_p
    n_checks = 0

    def expect_neg(rc, what):
        nonlocal n_checks
        n_checks += 1
        msg = lib.tv_last_error()
        assert rc < 0, "%s: expected an argument error, got %d" % (what, rc)
        assert msg and len(msg) > 3, what

    assert lib.tv_version() >= 300 and lib.tv_abi_version() == ABI_VERSION
    # ---- geometry rules: channel counts and workspace for many shapes / schemes / weights
    for nz, m, ny, nx, scheme, dtype, lz, mu in itertools.product((1, 2, 7, 300), (1, 2, 9, 16), (1, 5, 1024), (2, 64, 1028),
                                                                  range(4), (0, 1), (0.0, 1.5), (0.0, 0.3)):
        g = geom(nz, m, ny, nx, scheme, dtype, lz=lz, mu=mu)
        nd = lib.tv_num_channels(ctypes.byref(g))
        per = 2 if scheme == 3 else 1
        assert nd == per * (2 + (1 if nz > 1 and lz > 0 else 0) + (1 if m > 1 and mu > 0 else 0)), (nz, m, scheme, lz, mu, nd)
        assert lib.tv_workspace_bytes(ctypes.byref(g)) >= 8 * 2048
        for f in (lib.tv_cp_fused_supported, lib.tv_subgrad_fused_supported):
            assert f(ctypes.byref(g)) in (0, 1)
        assert lib.tv_cp_zchunk(ctypes.byref(g)) >= 1
        n_checks += 1
    # ---- broken geometries
    bad = [geom(nz=0), geom(m=0), geom(ny=-3), geom(nx=0), geom(scheme=7), geom(scheme=-1), geom(dtype=5), geom(nzg=2),
           geom(nz=4, nzg=8, z0=6), geom(z0=-1, nzg=10), geom(lz=-1.0), geom(mu=float("nan")), geom(factor=-2.0),
           geom(nz=70000, nzg=70000), geom(m=70000), geom(ny=1 << 20, nx=1 << 20)]
    for g in bad:
        expect_neg(lib.tv_num_channels(ctypes.byref(g)), "tv_num_channels(bad geometry)")
        assert lib.tv_workspace_bytes(ctypes.byref(g)) == 0
        assert lib.tv_cp_fused_supported(ctypes.byref(g)) == 0 and lib.tv_subgrad_fused_supported(ctypes.byref(g)) == 0
    expect_neg(lib.tv_num_channels(None), "tv_num_channels(NULL)")
    # ---- every operator: NULL geometry, NULL arrays, missing halos (all rejected before any HIP call)
    g = geom()
    gs = geom(nz=2, nzg=6, z0=2)              # an interior slab: halos are mandatory
    one = ctypes.c_double(0.0)
    dp = ctypes.cast(ctypes.pointer(one), P)
    buf = (ctypes.c_char * 64)()
    a = ctypes.cast(buf, P)                   # a non-NULL (host!) pointer: must never be dereferenced by the library
    N = None
    calls = {
        "tv_D": lambda G, x: lib.tv_D(G, x, N, N, x, N),
        "tv_DT": lambda G, x: lib.tv_DT(G, x, N, N, x, N),
        "tv_DT_axpy": lambda G, x: lib.tv_DT_axpy(G, x, N, N, N, N, ctypes.c_double(1.0), x, N),
        "tv_l21": lambda G, x: lib.tv_l21(G, x, 8, N, dp if x else N, x, N),
        "tv_subgrad": lambda G, x: lib.tv_subgrad(G, x, N, N, x, x, dp if x else N, x, N),
        "tv_subgrad_fused": lambda G, x: lib.tv_subgrad_fused(G, x, N, N, x, dp if x else N, x, N),
        "tv_subgrad_fused_norms": lambda G, x: lib.tv_subgrad_fused_norms(G, x, N, N, x, x, dp if x else N, x, N),
        "tv_subgrad_step_fused": lambda G, x: lib.tv_subgrad_step_fused(G, x, N, N, x, N, ctypes.c_double(.1), ctypes.c_double(1.), dp, dp, x, N),
        "tv_cp_dual": lambda G, x: lib.tv_cp_dual(G, x, N, N, x, ctypes.c_double(.5), ctypes.c_double(25.), dp if x else N, x, N),
        "tv_cp_primal": lambda G, x: lib.tv_cp_primal(G, x, N, N, x, x, x, ctypes.c_double(.1), ctypes.c_double(1.), dp if x else N, x, N),
        "tv_cp_fused": lambda G, x: lib.tv_cp_fused(G, x, N, N, x, x, x, x, ctypes.c_double(.5), ctypes.c_double(25.), ctypes.c_double(.1),
                                                    ctypes.c_double(1.), ctypes.c_int64(0), ctypes.c_int64(-1), dp, dp, x, N),
        "tv_cp_fixup": lambda G, x: lib.tv_cp_fixup(G, x, N, N, x, x, ctypes.c_double(.1), ctypes.c_int64(0), ctypes.c_int64(-1), dp if x else N, x, N),
        "tv_cpop_fused": lambda G, x: lib.tv_cpop_fused(G, x, N, N, x, x, x, ctypes.c_double(.5), ctypes.c_double(25.), ctypes.c_double(.1),
                                                        ctypes.c_int64(0), ctypes.c_int64(-1), dp, x, N),
        "tv_cpop_fixup": lambda G, x: lib.tv_cpop_fixup(G, x, N, N, x, ctypes.c_double(.1), ctypes.c_int64(0), ctypes.c_int64(-1), x, N),
        "tv_cheb_step": lambda G, x: lib.tv_cheb_step(G, x, N, N, ctypes.c_double(.1), x, N, ctypes.c_double(0.), N, N, ctypes.c_double(.5), ctypes.c_double(.1), x, dp if x else N, x, N),
        "tv_axpby": lambda G, x: lib.tv_axpby(G, ctypes.c_double(1.), x, ctypes.c_double(1.), N, N, x, N, N, N),
        "tv_admm_fused": lambda G, x: lib.tv_admm_fused(G, x, N, N, x, x, x, x, ctypes.c_double(1.), ctypes.c_double(.1), ctypes.c_int32(0),
                                                        ctypes.c_int64(0), ctypes.c_int64(-1), dp, dp, x, N),
        "tv_admm_fixup": lambda G, x: lib.tv_admm_fixup(G, x, N, N, x, ctypes.c_double(.1), ctypes.c_int64(0), ctypes.c_int64(-1), dp if x else N, x, N),
        "tv_admm_zu": lambda G, x: lib.tv_admm_zu(G, x, N, N, x, x, ctypes.c_double(1.), dp if x else N, x, N),
        "tv_normal_op": lambda G, x: lib.tv_normal_op(G, x, N, N, ctypes.c_double(.1), x, dp if x else N, x, N),
        "tv_admm_tu": lambda G, x: lib.tv_admm_tu(G, x, N, N, x, x, ctypes.c_double(1.), dp if x else N, x, N),
        "tv_normal_op2": lambda G, x: lib.tv_normal_op2(G, x, N, N, ctypes.c_double(.1), N, x, N, dp if x else N, x, N),
        "tv_cg_update": lambda G, x: lib.tv_cg_update(G, x, x, x, x, x, dp if x else N, N, N, x, N),
        "tv_cg_step1": lambda G, x: lib.tv_cg_step1(G, x, x, x, x, dp if x else N, dp, dp, x, N),
        "tv_cg_step2": lambda G, x: lib.tv_cg_step2(G, x, x, dp if x else N, dp, N),
        "tv_dot": lambda G, x: lib.tv_dot(G, x, x, dp if x else N, x, N),
        "tv_small_cp": lambda G, x: lib.tv_small_cp(G, x, x, x, x, ctypes.c_double(.5), ctypes.c_double(25.), ctypes.c_double(.1), ctypes.c_double(1.), ctypes.c_int64(4),
                                                    dp if x else N, ctypes.c_int64(2), ctypes.c_int64(1), x, N),
        "tv_small_subgrad_descent": lambda G, x: lib.tv_small_subgrad_descent(G, x, N, x, x, ctypes.c_double(.01), ctypes.c_double(25.), ctypes.c_int64(4), dp if x else N,
                                                                              ctypes.c_int64(2), ctypes.c_int64(1), x, N),
        "tv_subgrad_step": lambda G, x: lib.tv_subgrad_step(G, x, x, x, ctypes.c_double(.1), ctypes.c_double(1.), dp if x else N, x, N),
    }
    halo_ops = ("tv_D", "tv_DT", "tv_DT_axpy", "tv_subgrad", "tv_subgrad_fused", "tv_subgrad_fused_norms", "tv_cp_dual", "tv_cp_primal",
                "tv_admm_zu", "tv_admm_tu", "tv_normal_op", "tv_normal_op2", "tv_cp_fixup", "tv_admm_fixup", "tv_cheb_step", "tv_cpop_fixup")
    for name, call in calls.items():
        expect_neg(call(None, a), name + "(NULL geometry)")
        expect_neg(call(ctypes.byref(g), None), name + "(NULL arrays)")
        for gb in bad[:6]:
            expect_neg(call(ctypes.byref(gb), a), name + "(bad geometry)")
        if name in halo_ops:
            rc = call(ctypes.byref(gs), a)
            expect_neg(rc, name + "(interior slab without halos)")
    # the persistent small-volume loops: unsharded volumes only, sane history layout, volume bound (all before any HIP call)
    lib.tv_small_workspace_bytes.restype = ctypes.c_size_t
    assert lib.tv_small_supported(ctypes.byref(g)) == 1 and lib.tv_small_supported(ctypes.byref(gs)) == 0 and lib.tv_small_supported(None) == 0
    assert lib.tv_small_workspace_bytes(ctypes.byref(g), ctypes.c_int64(0)) == 0 and lib.tv_small_workspace_bytes(ctypes.byref(g), ctypes.c_int64(16)) > (1 << 20)
    big = geom(nz=64, m=8, ny=1024, nx=1024)
    assert lib.tv_small_supported(ctypes.byref(big)) == 0
    for gg, what in ((gs, "a z-slab"), (big, "a volume beyond TV_SMALL_MAX_KVOXELS")):
        expect_neg(lib.tv_small_cp(ctypes.byref(gg), a, a, a, a, ctypes.c_double(.5), ctypes.c_double(25.), ctypes.c_double(.1), ctypes.c_double(1.), ctypes.c_int64(4), dp,
                                   ctypes.c_int64(2), ctypes.c_int64(1), a, N), "tv_small_cp(%s)" % what)
    for stride, off in ((0, 1), (2, 0), (2, 2), (2, -1)):
        expect_neg(lib.tv_small_cp(ctypes.byref(g), a, a, a, a, ctypes.c_double(.5), ctypes.c_double(25.), ctypes.c_double(.1), ctypes.c_double(1.), ctypes.c_int64(4), dp,
                                   ctypes.c_int64(stride), ctypes.c_int64(off), a, N), "tv_small_cp(history layout %d, %d)" % (stride, off))
    expect_neg(lib.tv_small_cp(ctypes.byref(g), a, a, a, a, ctypes.c_double(.5), ctypes.c_double(25.), ctypes.c_double(.1), ctypes.c_double(1.), ctypes.c_int64(0), dp,
                               ctypes.c_int64(2), ctypes.c_int64(1), a, N), "tv_small_cp(n_iter = 0)")
    expect_neg(lib.tv_small_subgrad_descent(ctypes.byref(g), a, a, a, a, ctypes.c_double(.01), ctypes.c_double(25.), ctypes.c_int64(4), dp, ctypes.c_int64(2), ctypes.c_int64(1), a, N),
               "tv_small_subgrad_descent(x == x_alt)")
    expect_neg(lib.tv_cp_dual(ctypes.byref(g), a, N, N, a, ctypes.c_double(.5), ctypes.c_double(0.0), dp, a, N), "lambda = 0")
    expect_neg(lib.tv_cp_fused(ctypes.byref(g), a, N, N, a, a, a, a, ctypes.c_double(.5), ctypes.c_double(25.), ctypes.c_double(.1),
                               ctypes.c_double(1.), ctypes.c_int64(0), ctypes.c_int64(-1), dp, dp, a, N), "x_in == x_out")
    expect_neg(lib.tv_sub(0, ctypes.c_int64(-4), a, a, a, N), "tv_sub(n < 0)")
    expect_neg(lib.tv_sub(3, ctypes.c_int64(4), a, a, a, N), "tv_sub(dtype)")
    assert lib.tv_sub(0, ctypes.c_int64(0), a, a, a, N) == 0
    expect_neg(lib.tv_l21(ctypes.byref(g), a, 0, N, dp, a, N), "tv_l21(nd = 0)")
    # ---- option table
    assert lib.tv_set_option(b"TV_ZCHUNK", 5) == 0 and lib.tv_get_option(b"TV_ZCHUNK", 0) == 5
    assert lib.tv_cp_zchunk(ctypes.byref(geom(nz=40))) == 5
    assert lib.tv_unset_option(b"TV_ZCHUNK") == 0 and lib.tv_get_option(b"TV_ZCHUNK", -7) == -7
    expect_neg(lib.tv_set_option(b"TV_NO_SUCH_OPTION", 1), "unknown option")
    expect_neg(lib.tv_set_option(None, 1), "NULL option")
    assert lib.tv_get_option(None, 3) == 3
    # ---- multi-GPU surface: argument errors only (no RCCL is initialised here)
    expect_neg(lib.tv_ctx_unique_id(None), "tv_ctx_unique_id(NULL)")
    ctx = P()
    expect_neg(lib.tv_ctx_create(ctypes.byref(ctx), 3, 2, a, 0), "rank >= nranks")
    expect_neg(lib.tv_ctx_create(ctypes.byref(ctx), 0, 1, None, 0), "NULL unique id")
    expect_neg(lib.tv_ctx_create(None, 0, 1, a, 0), "NULL ctx out")
    expect_neg(lib.tv_halo_exchange(None, 0, ctypes.c_int64(4), -1, -1, N, N, N, N, N), "NULL ctx")
    expect_neg(lib.tv_allreduce_f64(None, a, ctypes.c_int64(1), 0, N), "NULL ctx")
    assert lib.tv_ctx_destroy(None) == 0
    print("abi_validation: %d checks ok (%s)" % (n_checks, os.path.basename(LIB)))


if __name__ == "__main__":
    main()

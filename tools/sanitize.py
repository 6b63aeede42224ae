#!/usr/bin/env python3
"""AddressSanitizer / UndefinedBehaviorSanitizer on the CPU side (the GPU pool has no device ASan).

  1. oracle/tv_oracle_c.c built with gcc -fsanitize=address,undefined; tests/test_oracle_c.py runs against that build
     (D, D^T, l2,1, Chambolle-Pock, ADMM on ragged shapes: every index computation of the C restatement).
  2. The HOST side of the C-ABI: every translation unit of libpytv4d_hip.so compiled with hipcc --cuda-host-only
     -fsanitize=address,undefined (no device code: nothing can launch) and driven through its argument validation,
     geometry, workspace and option paths by tools/abi_validation.py -- everything an entry point does before its
     first HIP call.

usage: python tools/sanitize.py [--keep]        exit code 0 = clean.  Runs in ~2 minutes on 8 cores, no GPU needed."""
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pytv-4d_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "pytv-4d_amd"))
import build as _product_build  # noqa: E402  (the product's own unit list, so that the two cannot drift)
UNITS = [u[:-len(".hip")] for u in _product_build.UNITS]
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g", "-O1"]


def run(cmd, **kw):
    print("[sanitize] " + " ".join(cmd), flush=True)
    return subprocess.run(cmd, **kw)


def oracle_leg(tmp):
    out = os.path.join(tmp, "libtv_oracle_c_asan.so")
    run(["gcc"] + SAN + ["-fopenmp", "-shared", "-fPIC", "-std=c11", os.path.join(ROOT, "oracle", "tv_oracle_c.c"), "-o", out, "-lm"],
        check=True)
    rt = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               TV_ORACLE_C_LIB=out, OMP_NUM_THREADS="4")
    p = run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", os.path.join(ROOT, "tests", "test_oracle_c.py")], env=env, cwd=ROOT)
    return p.returncode


def host_leg(tmp):
    hipcc = "/opt/rocm/bin/hipcc"
    flags = ["--cuda-host-only"] + SAN + ["-std=c++20", "-fPIC", "-Wno-unused-value"]

    def comp(u):
        obj = os.path.join(tmp, u + ".o")
        subprocess.check_call([hipcc] + flags + ["-c", os.path.join(CSRC, u + ".hip"), "-o", obj])
        return obj
    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(comp, UNITS))
    # a host-only object refers to the device image of its translation unit (__hip_fatbin_<hash>), which does not exist
    # here: define each one as an empty bundle so that the library loads; nothing can be launched from it
    syms = set()
    for o in objs:
        for line in subprocess.check_output(["nm", "-u", o], text=True).splitlines():
            name = line.split()[-1]
            if name.startswith("__hip_fatbin_"):
                syms.add(name)
    stub = os.path.join(tmp, "fatbin_stub.c")
    with open(stub, "w") as f:
        for s in sorted(syms):
            f.write('const char %s[64] __attribute__((aligned(4096))) = "__CLANG_OFFLOAD_BUNDLE__";\n' % s)
    subprocess.check_call(["gcc", "-c", "-fPIC", stub, "-o", stub[:-2] + ".o"])
    out = os.path.join(tmp, "libpytv4d_host_asan.so")
    run([hipcc, "-fsanitize=address,undefined", "-shared", "-fPIC", "-o", out] + objs + [stub[:-2] + ".o"], check=True)
    rt = subprocess.check_output(["/opt/rocm/lib/llvm/bin/clang", "--print-file-name=libclang_rt.asan-x86_64.so"], text=True).strip()
    env = dict(os.environ, LD_PRELOAD=rt, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="print_stacktrace=1",
               PYTV4D_LIB=out)
    p = run([sys.executable, os.path.join(ROOT, "tools", "abi_validation.py")], env=env, cwd=ROOT)
    return p.returncode


def main():
    keep = "--keep" in sys.argv
    tmp = tempfile.mkdtemp(prefix="tv_sanitize_")
    rc1 = oracle_leg(tmp)
    rc2 = host_leg(tmp)
    print("[sanitize] oracle leg rc=%d, C-ABI host leg rc=%d%s" % (rc1, rc2, "  (build kept in %s)" % tmp if keep else ""))
    if not keep:
        import shutil
        shutil.rmtree(tmp, ignore_errors=True)
    sys.exit(0 if rc1 == 0 and rc2 == 0 else 1)


if __name__ == "__main__":
    main()

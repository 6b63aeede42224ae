import os, sys
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "pytv-4d_amd"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import pytv
from pytv import _native as nv
from oracle import tv_oracle as orc
lib = nv.lib()
for shape in [(7, 3, 9, 256), (3, 16, 5, 128)]:
    rng = np.random.default_rng(5)
    kw = dict(reg_z_over_reg=1.3, reg_time=0.5)
    x = torch.as_tensor((rng.standard_normal(shape) * 10).astype(np.float32)).cuda()
    g = nv.Geometry(shape, "central", x.dtype, x.device, **kw)
    st, ws = nv.current_stream(x.device), g.workspace()
    x64 = x.double().cpu().numpy()
    want = x64 + 0.3 * orc.D_T(orc.D(x64, "central", **kw), "central", **kw)
    out = torch.empty_like(x); dots = torch.zeros(2, dtype=torch.float64, device="cuda")
    nv.check(lib.tv_normal_op2(g.ref, nv.ptr(x), None, None, 0.3, None, nv.ptr(out), None, dots.data_ptr(), nv.ptr(ws), st))
    torch.cuda.synchronize()
    bad = np.argwhere(np.abs(out.cpu().numpy() - want) > 2e-3 + 1e-5 * np.abs(want))
    print(shape, "bad", len(bad))
    for ax, name in enumerate("ztyx"):
        u, c = np.unique(bad[:, ax], return_counts=True)
        print("  ", name, dict(zip(u.tolist(), c.tolist())))

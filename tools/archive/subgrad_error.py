import sys, os
sys.path.insert(0, "pytv-4d_amd"); sys.path.insert(0, ".")
os.environ["TV_MARCH_MIN_PLANE_KB"] = "0"
import numpy as np, torch, pytv
from oracle import tv_oracle as orc
np.random.seed(0)
img = np.random.rand(20, 4, 100, 100).astype(np.float32)
x = torch.as_tensor(img).cuda()
for scheme in ("hybrid", "upwind", "downwind", "central"):
    kw = dict(reg_z_over_reg=1.0, reg_time=2.0 ** -5)
    tv_ref, G_ref = orc.tv(img.astype(np.float64), scheme, **kw)
    tv1, G1, _ = pytv.tv_GPU.tv_subgradient_device(x, scheme, want_norms=False, one_pass=True, **kw)
    tv2, G2, _ = pytv.tv_GPU.tv_subgradient_device(x, scheme, one_pass=False, **kw)
    e1 = np.abs(G1.cpu().numpy() - G_ref); e2 = np.abs(G2.cpu().numpy() - G_ref)
    print("%-9s one-pass: max %.2e rms %.2e tv rel %.1e | two-pass: max %.2e rms %.2e tv rel %.1e" % (
        scheme, e1.max(), np.sqrt((e1 ** 2).mean()), abs(float(tv1) - tv_ref) / tv_ref, e2.max(), np.sqrt((e2 ** 2).mean()), abs(float(tv2) - tv_ref) / tv_ref))

"""Where does k_chol_panel spend its time?  (diagnostic; stamps of chain 0 per panel through the context's stamps buffer)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from openmcmc_amd.engine import Engine
p, C = 1000, 256
rng = np.random.default_rng(0)
X = rng.standard_normal((4 * p, p))
G = X.T @ X / (4 * p)
eng = Engine(C, seed=1)
dG = eng.to_device(G)
lam, tau = eng.full((C,), 0.5), eng.full((C,), 1.0)
terms = eng.dense_terms([{"mat": None, "scale": lam}, {"mat": dG, "scale": tau}], p)
b = eng.empty(C, p)
stamps = torch.zeros(16 * 16 * 16, dtype=torch.int64, device="cuda")
ref = None
for old, ovl in ((0, 1), (1, 1), (0, 0), (1, 0)):
    eng.set_option("dense_panel_old", old)
    eng.set_option("dense_overlap", ovl)
    for it in range(3):
        eng.dense_sample_canonical(p, terms, b, draw_index=it)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(5):
        eng.dense_sample_canonical(p, terms, b, draw_index=9)
    e1.record()
    torch.cuda.synchronize()
    print("dense_panel_old =", old, "dense_overlap =", ovl, ": draw ms", e0.elapsed_time(e1) / 5)
    if ref is None:
        ref = b.clone()
    else:
        print("   max |x_new - x_old| =", (b - ref).abs().max().item(), "of", ref.abs().max().item())
eng.set_option("stamps_ptr", stamps.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
eng.dense_sample_canonical(p, terms, b, draw_index=9)
e1.record()
torch.cuda.synchronize()
print("draw ms", e0.elapsed_time(e1))
st = stamps.cpu().numpy()[:16 * 8].reshape(16, 8).astype(np.float64)
names = ["load", "chol 64", "inverse", "write block", "rows below"]
print("panel   " + "  ".join(f"{n:>12s}" for n in names) + "   total (ticks of s_memtime)")
for j in range(16):
    d = np.diff(st[j, :6])
    print(f"{j:5d}   " + "  ".join(f"{v:12.0f}" for v in d) + f"   {st[j,5]-st[j,0]:10.0f}")
eng.check_status()

import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from test_tridiag_joins_gpu import problem, relerr
from openmcmc_amd.engine import Engine

def go(n, lam, tau, seg, newton_max, ppb, generic, algo=2, irregular=True):
    pd, po, y, z, a, b, r = problem(n, lam, tau, seed=9)
    if not irregular:
        pd = np.full(n, 2.0); pd[0] = pd[-1] = 1.0; pd[0] += 1e-3
    eng = Engine(2)
    eng.set_option("tridiag_algo", algo); eng.set_option("tridiag_seg", seg)
    if newton_max is not None: eng.set_option("tridiag_newton_max", newton_max)
    eng.set_option("tridiag_perturb_ppb", ppb); eng.set_option("tridiag_generic", generic)
    terms = [{"diag": eng.to_device(pd), "off": eng.to_device(po), "scale": eng.full((2,), lam)},
             {"rhs": eng.to_device(y), "center": eng.to_device(y), "scale": eng.full((2,), tau)}]
    x, ld = eng.empty(2, n), eng.empty(2)
    eng.tridiag_sample_canonical(n, terms, x, z=eng.to_device(np.tile(z, (2, 1))), logdet_out=ld)
    try:
        eng.check_status(); st = "ok"
    except Exception as e:
        st = "NOTPD"
    fb = eng.counter("tridiag_join_fallbacks")
    xs = x.cpu().numpy()
    nan = np.where(~np.isfinite(xs[0]))[0]
    print(f"n {n} lam {lam:g} seg {seg} algo {algo} nmax {newton_max} ppb {ppb} irr {irregular}: {st} fb {fb} nan {nan.size} first {nan[:1]} last {nan[-1:]} ld {ld.cpu().numpy()[0]:.6f}", flush=True)
    eng.close()

for lam in (1e3, 1e4, 1e5, 1e6, 1e7, 3e7):
    go(10000, lam, 1.0, 10, None, 0, 0)
for n in (6400, 7000, 8000, 9000, 9600, 9601, 9700, 10000, 10240):
    go(n, 3e7, 1.0, 10, None, 0, 0)
go(10000, 3e7, 1.0, 10, None, 0, 0, irregular=False)
go(10000, 3e7, 1.0, 10, 0, 1000, 0)
go(10000, 3e7, 1.0, 32, None, 0, 0)
go(10000, 3e7, 1.0, 16, None, 0, 0)
go(10000, 3e7, 1.0, 10, None, 0, 0, algo=1)
print("---- extremes")
for lam in (1e9, 1e12, 1e15):
    go(10000, lam, 1.0, 10, None, 0, 0)
    go(10000, lam, 1.0, 10, None, 0, 0, irregular=False)
go(10000, 1e-9, 1.0, 10, None, 0, 0)
go(10000, 3e7, 1e-6, 10, None, 0, 0)
go(10000, 3e7, 1.0, 10, None, 1000, 0)
go(10000, 3e7, 1.0, 10, 1, 1000000, 0)

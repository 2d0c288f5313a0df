import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests._util import dev, engine_from_oracle, make_oracle, relerr
m, eps = make_oracle(kind="rbf", W=60, H=50, V=50, K=10, n_points=(32, 16), dtype=torch.float64, jitter=1e-6, lengthscale=0.1, s_perturb=0.02, trained_scale=0.3)
with torch.no_grad():
    for p in m.params.values(): p.copy_(p.float().double())
    m.Z = m.Z.float().double()
eng = engine_from_oracle(m, dtype=torch.float32, mfma_mode="f16x3")
xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
lvl = eng.factorize()
lay = eng.red_layout; Mp = (eng.M + 31) // 32 * 32; mm = Mp * Mp
outs = []
for it in range(3):
    eng.loss_and_grads(xs, ws, dev(eps, eng), force_level=lvl)
    eng.read_out()
    A = eng.red_T[lay["A"]:lay["A"] + eng.K * mm].view(eng.K, Mp, Mp).clone()
    outs.append(A)
    bad = ~torch.isfinite(A)
    print(os.environ.get("TAG", ""), "iter", it, "non-finite", int(bad.sum()), "per topic", bad.view(eng.K, -1).sum(1).tolist(), "equal to iter 0:", bool(torch.equal(A, outs[0])) if not bad.any() and it else "-")
vb = eng.workspace("vbar", m.N)
print("vbar finite:", bool(torch.isfinite(vb).all()), "max per topic", vb.abs().max(1).values.tolist())
sc = torch.empty(6 + 6 * eng.K)

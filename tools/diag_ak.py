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
eng.loss_and_grads(xs, ws, dev(eps, eng), force_level=lvl)
eng.read_out()
lay = eng.red_layout; Mp = (eng.M + 31) // 32 * 32; mm = Mp * Mp
A = eng.red_T[lay["A"]:lay["A"] + eng.K * mm].view(eng.K, Mp, Mp).cpu().double().numpy()
bad = ~np.isfinite(A)
print("non-finite entries:", bad.sum(), "of", A.size, "per topic", bad.reshape(eng.K, -1).sum(1))
if bad.any():
    idx = np.argwhere(bad)[:10]; print(idx)
We = eng.workspace("W", m.N).cpu().double().numpy(); ve = eng.workspace("vbar", m.N).cpu().double().numpy()
Aiso = np.einsum("ni,kn,nj->kij", We, ve, We)
Af = np.where(bad[:, :eng.M, :eng.M], 0, A[:, :eng.M, :eng.M])
print("err on finite entries", np.abs(Af - np.where(bad[:, :eng.M, :eng.M], 0, Aiso)).max() / np.abs(Aiso).max(), "max |vbar|", np.abs(ve).max(), "scales", eng.K)
D = np.abs(Af - np.where(bad[:, :eng.M, :eng.M], 0, Aiso)) / np.abs(Aiso).max()
print("per-topic max err:", ["%.1e" % D[k].max() for k in range(eng.K)])
T = D.max(0)
blk = T.reshape(eng.M // 32, 32, eng.M // 32, 32).max(axis=(1, 3))
np.set_printoptions(linewidth=250, precision=0)
print("max err per 32x32 block (lower triangle computed, upper mirrored), in units of 1e-6:")
print((blk * 1e6).astype(int))

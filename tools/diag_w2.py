"""Diagnostic: where does the default A_k kernel differ from the 64-row form?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests._util import dev, engine_from_oracle, make_oracle
from tests.test_gpu_ak_forms import SHAPES, _env

def A_of(m, eps, f):
    with _env(GDRF_TNT_W1=f):
        eng = engine_from_oracle(m, mfma_mode="f16x3", store_t=False)
        xs, ws, e = dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng)
        eng.loss_and_grads(xs, ws, e); torch.cuda.synchronize()
        Mp, lay = (m.M + 31) // 32 * 32, eng.red_layout
        return eng.red_T[lay["A"]:lay["A"] + m.K * Mp * Mp].view(m.K, Mp, Mp).cpu().double().numpy()

for name in SHAPES:
    m, eps = make_oracle(dtype=torch.float32, jitter=1e-4, **SHAPES[name])
    A1, A2 = A_of(m, eps, 1), A_of(m, eps, 2)
    bad = ~np.isfinite(A2) | (np.abs(A2 - A1) > 1e-5 * np.abs(A1).max())
    print(name, "bad entries", bad.sum(), "of", bad.size, "nan", np.isnan(A2).sum())
    if bad.any():
        k, i, j = np.nonzero(bad)
        print("  topics", np.unique(k), " 32-blocks (i,j):", sorted(set(zip((i // 32).tolist(), (j // 32).tolist())))[:40])

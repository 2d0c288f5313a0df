"""GPU: the three forms of the A_k = W^T diag(vbar_k) W kernel (csrc/gemm_tn_topics.h: two waves per SIMD; csrc/gemm_tn_topics1.h: one wave
per SIMD with 64 or - the default - 128 rows per wave, inline-asm MFMAs, AGPR accumulators, 3-deep LDS-DMA ring) and the two forms of the
tt kernel (16 waves / the opt-in one-wave form of csrc/gemm_fwd_t1.h), each against the fp64 product of the engine's OWN float32 inputs, on
shapes that reach their edges: rows that are no multiple of the 64-row chunk, splits with no rows at all, more topics than one
accumulator group holds (K = 12 -> groups of 10 + 2), fewer (K = 3), a padded inducing count (M = 100 -> Mp = 128), an inducing count that
is no multiple of the tiles (M = 480), and the headline's M = 512.  The forms are selected per call through the library's environment knobs (GDRF_TNT_W1, GDRF_FWDT_W1)."""
import os

import numpy as np
import pytest
import torch

from tests._util import dev, engine_from_oracle, make_oracle, relerr

pytestmark = pytest.mark.gpu

SHAPES = {
    "m256_k10_ragged_rows": dict(kind="rbf", W=41, H=25, V=20, K=10, n_points=(16, 16), lengthscale=0.08),       # N = 1025
    "m100_k12_two_groups": dict(kind="matern52", W=40, H=25, V=12, K=12, n_points=(10, 10), lengthscale=0.12),   # Mp = 128
    "m512_k3": dict(kind="rbf", W=60, H=50, V=20, K=3, n_points=(32, 16), lengthscale=0.1),
    "m480_k10_partial_tiles": dict(kind="rbf", W=50, H=40, V=20, K=10, n_points=(24, 20), lengthscale=0.1),       # Mp = 480: the last 128 / 64 tiles are partial
}


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update({k: str(v) for k, v in self.kv.items()})

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _run(m, eps, **env):
    with _env(**env):
        eng = engine_from_oracle(m, mfma_mode="f16x3", store_t=False)
        xs, ws, e = dev(m.xs, eng), dev(m.ws, eng, torch.int32), dev(eps, eng)
        eng.loss_and_grads(xs, ws, e)
        torch.cuda.synchronize()
        n = m.N
        Wm = eng.workspace("W", n).cpu().double().numpy()
        vbar = eng.workspace("vbar", n).cpu().double().numpy()
        S = eng.workspace("S").cpu().double().numpy()
        Mp, lay = (m.M + 31) // 32 * 32, eng.red_layout
        A = eng.red_T[lay["A"]:lay["A"] + m.K * Mp * Mp].view(m.K, Mp, Mp)[:, :m.M, :m.M].cpu().double().numpy()
        tt = eng.workspace("tt", n).cpu().double().numpy()
        A_ref = np.stack([Wm.T @ (vbar[k][:, None] * Wm) for k in range(m.K)])
        tt_ref = np.stack([((Wm @ S[k]) ** 2).sum(1) for k in range(m.K)])
        grads = eng.grads.clone().cpu()
        # second call on the same engine: bit-identical (fixed accumulation order)
        eng.loss_and_grads(xs, ws, e)
        torch.cuda.synchronize()
        A2 = eng.red_T[lay["A"]:lay["A"] + m.K * Mp * Mp].view(m.K, Mp, Mp)[:, :m.M, :m.M].cpu().double().numpy()
        assert np.array_equal(A, A2), (env, float(np.abs(A - A2).max()), float(np.abs(A).max()))
    return dict(A=np.tril(A), A_ref=np.tril(A_ref), tt=tt, tt_ref=tt_ref, grads=grads)


@pytest.mark.parametrize("name", list(SHAPES))
def test_ak_kernel_forms_agree_with_the_fp64_product(name):
    m, eps = make_oracle(dtype=torch.float32, jitter=1e-4, **SHAPES[name])
    res = {f: _run(m, eps, GDRF_TNT_W1=f) for f in (0, 1, 2)}
    errs = {f: relerr(r["A"], r["A_ref"]) for f, r in res.items()}
    print(name, "A_k vs the fp64 product of the same inputs, by kernel form:", {f: "%.2e" % e for f, e in errs.items()})
    for f, e in errs.items():
        assert e < 2e-6, (name, f, e)                        # 22-bit arithmetic: the split forms measure 2 - 5e-7 here
    # every block of the gradient depends on A_k (through Sbar): the forms must agree to float32 rounding of the sums
    for f in (1, 2):
        d = float((res[f]["grads"] - res[0]["grads"]).abs().max() / res[0]["grads"].abs().max())
        assert d < 2e-5, (name, f, d)


@pytest.mark.parametrize("name", ["m256_k10_ragged_rows", "m512_k3"])
def test_tt_one_wave_form_agrees_with_the_fp64_product(name):
    """csrc/gemm_fwd_t1.h (opt-in): Mp a multiple of 256; the triangle cut at 16 columns instead of 64."""
    m, eps = make_oracle(dtype=torch.float32, jitter=1e-4, **SHAPES[name])
    res = {f: _run(m, eps, GDRF_FWDT_W1=f) for f in (0, 1)}
    errs = {f: relerr(r["tt"], r["tt_ref"]) for f, r in res.items()}
    print(name, "tt vs the fp64 product of the same inputs, by kernel form:", {f: "%.2e" % e for f, e in errs.items()})
    assert errs[1] < 1.5 * errs[0] + 1e-7 and errs[1] < 2e-6, errs
    d = float((res[1]["grads"] - res[0]["grads"]).abs().max() / res[0]["grads"].abs().max())
    assert d < 2e-5, d

"""GPU: BASELINE.json's configs[1] (N = 100 000, M = 256, K = 10, RBF, fp64) and configs[4] (Matern-5/2, K = 20, M = 1024,
N = 250 000, fp32 arrays) AT THEIR FULL N, the way tests/test_gpu_headline.py checks configs[3]: 64-bit offsets, row-split and
slab counts, XCD maps and the K = 20 LDS budgets of the Wbar / A_k kernels only take their real values here.

  * known answers at initialisation over all rows;
  * the ELBO and the u_loc / phi / noise gradients of one full step against the fp64 oracle evaluated in row chunks on the host
    (oracle/gdrf_oracle.py::loss_chunked, restating gdrf/models/sparse_gdrf.py:323-409);
  * size-independent property: payload(rows [0, h)) + payload(rows [h, N)) == payload(all rows) for every block of the
    all-reduce payload, including the K x M x M contraction no host evaluation of this size can afford.
"""
import os
import time

import numpy as np
import pytest
import torch

from gdrf_amd.data import synth_circles
from oracle.gdrf_oracle import RefShapedGDRF

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (W, H, V, K, n_points, kernel, dtype, loss tol, grad tol, split-sum tol, oracle chunk)
    "config1_fp64_n100k": dict(W=400, H=250, V=50, K=10, npts=(16, 16), kind="rbf", dtype=torch.float64, ltol=1e-7, gtol=1e-7,
                               stol=1e-10, chunk=25000, jitter=1e-6),
    "config4_matern52_m1024_k20_n250k": dict(W=500, H=500, V=50, K=20, npts=(32, 32), kind="matern52", dtype=torch.float32,
                                             ltol=1e-4, gtol=1e-3, stol=2e-5, chunk=10000, jitter=1e-6),
}


def _build(cfg):
    from gdrf_amd.kernels import RBF, Matern52
    from gdrf_amd.models import SparseMultinomialGDRF
    dev = "cuda:0"
    xs_np, ws_np, _ = synth_circles(cfg["W"], cfg["H"], cfg["V"], cfg["K"], seed=4242)
    xs = torch.from_numpy(xs_np).to(dev, cfg["dtype"]).contiguous()
    ws = torch.from_numpy(ws_np).to(dev).contiguous()
    kcls = RBF if cfg["kind"] == "rbf" else Matern52
    model = SparseMultinomialGDRF(xs=xs, ws=ws, world=[(0.0, 1.0)] * 2,
                                  kernel=kcls(input_dim=2, lengthscale=torch.tensor(0.1), variance=torch.tensor(25.0)),
                                  num_observation_categories=cfg["V"], num_topic_categories=cfg["K"], dirichlet_param=0.01,
                                  n_points=list(cfg["npts"]), fixed_inducing_points=True, inducing_init="grid", maxjitter=15,
                                  jitter=cfg["jitter"], device=dev, dtype=cfg["dtype"], seed=4242)
    return model, xs, ws, xs_np, ws_np


def _perturb(eng, seed=5):
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        eng.view("u_loc").add_((0.3 * torch.randn(eng.K, eng.M, generator=g)).to(eng.device, eng.dtype))
        eng.view("u_scale_tril_unc").add_((0.03 * torch.randn(eng.K, eng.M, eng.M, generator=g)).tril().to(eng.device, eng.dtype))
        eng.view("phi_unc").add_((0.5 * torch.randn(eng.K, eng.V, generator=g)).to(eng.device, eng.dtype))
        eng.view("log_noise").add_(0.2)


@pytest.mark.parametrize("name", list(CONFIGS))
def test_full_size_configuration(name):
    cfg = CONFIGS[name]
    K, V = cfg["K"], cfg["V"]
    model, xs, ws, xs_np, ws_np = _build(cfg)
    N = xs.shape[0]
    assert N == cfg["W"] * cfg["H"] and model.M == cfg["npts"][0] * cfg["npts"][1]
    # ---- known answers at initialisation
    tp = model.topic_probs(xs)
    assert tp.shape == (N, K) and float((tp - 1.0 / K).abs().max()) < 1e-6
    assert abs(float(model.perplexity(xs, ws).item()) - V) < 1e-3 * V
    del tp
    # ---- one full evaluation through the C ABI at perturbed parameters
    eng = model._engine_for(N)
    eng.pg = None
    _perturb(eng)
    eps = eng.fill_eps(321, 0, 0, N)
    eng.loss_and_grads(xs, ws, eps)
    out = eng.read_out()
    assert out["chol_failed"] == 0
    level = eng.last_jitter_level
    full_T, full_d = eng.red_T.clone(), eng.red_d.clone()
    got = {n_: eng.view(n_, eng.grads).cpu().double() for n_ in ("u_loc", "phi_unc", "log_noise")}
    # ---- the oracle at the same parameter values, in float64, chunked over the rows
    ref = RefShapedGDRF(xs_np[:16], ws_np[:16], kind=cfg["kind"], K=K, n_points=cfg["npts"], lengthscale=0.1, variance=25.0,
                        dirichlet_param=0.01, jitter=cfg["jitter"], maxjitter=15, dtype=torch.float64, force_jitter_level=level)
    with torch.no_grad():
        for pname in eng.PARAM_NAMES:
            ref.params[pname].copy_(eng.view(pname).cpu().double())
        ref.Z = eng.Z.cpu().double()            # the engine's inducing inputs (float32-rounded grid in float32 contexts), not the oracle's own grid
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    t0 = time.time()
    loss_ref, grads_ref = ref.loss_chunked(eps.cpu().double(), torch.from_numpy(xs_np).double(), torch.from_numpy(ws_np),
                                           n_global=N, chunk=cfg["chunk"], grad_names=("u_loc", "phi_unc", "log_noise"))
    print(f"{name}: chunked fp64 oracle {time.time() - t0:.1f} s, loss {loss_ref:.10f} vs HIP {out['loss']:.10f}, jitter level {level}")
    assert abs(out["loss"] - loss_ref) <= cfg["ltol"] * abs(loss_ref), (out["loss"], loss_ref)
    # float32 arrays: mu = loc + v eps is stored with 24 bits, so the softmax weights carry a relative error of |mu| 2^-24 whatever
    # computes them (tests/test_gpu_round3.py, regime "init": u_scale_tril = L_uu + noise makes |mu| as large as cond(K_uu))
    gtol = cfg["gtol"]
    if cfg["dtype"] == torch.float32:
        mu_res = float(eng.workspace("mu", N).abs().max()) * 2.0 ** -24
        gtol = max(gtol, 6.0 * mu_res)
        print(f"  float32 resolution of mu: {mu_res:.2e} -> gradient bound {gtol:.2e}")
    for pname, g in got.items():
        r = grads_ref[pname].double()
        err = float((g - r).abs().max() / r.abs().max())
        print(f"  grad {pname}: rel err {err:.2e}")
        assert err <= gtol, (pname, err)
    # ---- linearity over a ragged row split
    h = N // 2 - 32 + 77
    parts_T, parts_d = torch.zeros_like(full_T), torch.zeros_like(full_d)
    for lo, hi in ((0, h), (h, N)):
        eng.loss_and_grads(xs[lo:hi], ws[lo:hi], eps[:, lo:hi].contiguous(), n_global=N, force_level=level)
        torch.cuda.synchronize()
        parts_T += eng.red_T
        parts_d += eng.red_d
    lay = eng.red_layout
    mm = ((eng.M + 31) // 32 * 32) ** 2
    blocks = {"ubar": (lay["ubar"], lay["phibar"]), "phibar": (lay["phibar"], lay["A"]), "A": (lay["A"], lay["A"] + K * mm),
              "GT": (lay["GT"], lay["GT"] + mm)}
    for bname, (a, b) in blocks.items():
        x, y = parts_T[a:b].double(), full_T[a:b].double()
        err = float((x - y).abs().max() / y.abs().max())
        print(f"  split-sum {bname}: rel err {err:.2e}")
        assert err < cfg["stol"], (bname, err)
    dtol = 1e-11 if cfg["dtype"] == torch.float64 else 2e-6
    for i, dname in enumerate(["site", "llw", "noise_g", "var_direct", "knm_k", "knm_dls"]):
        x, y = float(parts_d[i]), float(full_d[i])
        assert abs(x - y) <= dtol * max(abs(y), 1.0), (dname, x, y)
    assert abs(float(parts_d[7]) - float(full_d[7])) <= 1e-12 * abs(float(full_d[7]))
    del model, eng, xs, ws, eps, full_T, parts_T
    torch.cuda.empty_cache()

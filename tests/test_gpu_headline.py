"""GPU: the HEADLINE workload itself (BASELINE.json: N = 1e6, M = 512, K = 10, V = 50, 2-D, RBF, fp32 arrays) is checked,
not a scaled-down stand-in: tile maps, row-split counts, XCD maps and 64-bit offsets only take their headline values here.

  * known answers at initialisation over all 1e6 rows;
  * the ELBO and the u_loc / phi gradients of one full step against the fp64 oracle evaluated in row chunks on the host
    (every term but the Dirichlet one is a sum over observations; ~5e12 f64 flop, about a minute of host time);
  * size-independent property: the all-reduce payload of the two halves of the rows adds up to the payload of all rows
    (every block, including the M x M x K contractions no host evaluation of this size can afford).
"""
import os
import time

import numpy as np
import pytest
import torch

from gdrf_amd.data import synth_circles
from oracle.gdrf_oracle import RefShapedGDRF

pytestmark = pytest.mark.gpu

N_SIDE, V, K, NPTS = 1000, 50, 10, (32, 16)


@pytest.fixture(scope="module")
def headline():
    from gdrf_amd.kernels import RBF
    from gdrf_amd.models import SparseMultinomialGDRF
    dev = "cuda:0"
    xs_np, ws_np, _ = synth_circles(N_SIDE, N_SIDE, V, K, seed=777)           # bench.py's workload, same seed
    xs = torch.from_numpy(xs_np).to(dev, torch.float32).contiguous()
    ws = torch.from_numpy(ws_np).to(dev).contiguous()
    model = SparseMultinomialGDRF(xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=RBF(input_dim=2, lengthscale=torch.tensor(0.1), variance=torch.tensor(25.0)),
                                  num_observation_categories=V, num_topic_categories=K, dirichlet_param=0.01, n_points=list(NPTS),
                                  fixed_inducing_points=True, inducing_init="grid", maxjitter=15, jitter=1e-6, device=dev,
                                  dtype=torch.float32, seed=777)
    yield dict(model=model, xs=xs, ws=ws, xs_np=xs_np, ws_np=ws_np)
    del model, xs, ws
    torch.cuda.empty_cache()


def test_known_answers_at_initialisation_at_full_size(headline):
    model, xs, ws = headline["model"], headline["xs"], headline["ws"]
    N = xs.shape[0]
    assert N == 1_000_000 and model.M == 512
    tp = model.topic_probs(xs)                                  # u_loc = 0 => loc = 0 => 1/K everywhere
    assert tp.shape == (N, K) and float((tp - 1.0 / K).abs().max()) < 1e-6
    assert abs(float(model.perplexity(xs, ws).item()) - V) < 1e-3 * V          # uniform initial word-topic matrix
    wp = model.word_probs(xs[-5:])
    assert float((wp - 1.0 / V).abs().max()) < 1e-6


def _perturb(eng, seed=5):
    g = torch.Generator(device="cpu").manual_seed(seed)
    with torch.no_grad():
        eng.view("u_loc").add_(0.3 * torch.randn(eng.K, eng.M, generator=g).to(eng.device))
        eng.view("u_scale_tril_unc").add_((0.03 * torch.randn(eng.K, eng.M, eng.M, generator=g)).tril().to(eng.device))
        eng.view("phi_unc").add_(0.5 * torch.randn(eng.K, eng.V, generator=g).to(eng.device))
        eng.view("log_noise").add_(0.2)


def test_full_size_step_against_the_chunked_fp64_oracle_and_row_split_linearity(headline):
    model, xs, ws = headline["model"], headline["xs"], headline["ws"]
    N = xs.shape[0]
    eng = model._engine_for(N)
    eng.pg = None
    _perturb(eng)
    eps = eng.fill_eps(123, 0, 0, N)
    # ---- one full evaluation through the C ABI
    eng.loss_and_grads(xs, ws, eps)
    out = eng.read_out()
    assert out["chol_failed"] == 0
    level = eng.last_jitter_level
    full_T, full_d = eng.red_T.clone(), eng.red_d.clone()
    g_uloc = eng.view("u_loc", eng.grads).cpu().double()
    g_phi = eng.view("phi_unc", eng.grads).cpu().double()
    # ---- the oracle at the same (float32-valued) parameters, in float64, chunked over the rows
    ref = RefShapedGDRF(headline["xs_np"][:16], headline["ws_np"][:16], kind="rbf", K=K, n_points=NPTS, lengthscale=0.1, variance=25.0,
                        dirichlet_param=0.01, jitter=1e-6, maxjitter=15, dtype=torch.float64, force_jitter_level=level)
    with torch.no_grad():
        for name in eng.PARAM_NAMES:
            ref.params[name].copy_(eng.view(name).cpu().double())
        ref.Z = eng.Z.cpu().double()          # the float32-rounded inducing grid the engine (and the all-float32 reference) holds
    torch.set_num_threads(max(1, min(os.cpu_count() or 1, 32)))
    t0 = time.time()
    loss_ref, grads_ref = ref.loss_chunked(eps.cpu().double(), torch.from_numpy(headline["xs_np"]).double(),
                                           torch.from_numpy(headline["ws_np"]), n_global=N, chunk=25000)
    print(f"chunked fp64 oracle: {time.time() - t0:.1f} s, loss {loss_ref:.9f} vs HIP {out['loss']:.9f}, jitter level {level}")
    assert abs(out["loss"] - loss_ref) <= 1e-4 * abs(loss_ref), (out["loss"], loss_ref)
    for name, got in (("u_loc", g_uloc), ("phi_unc", g_phi)):
        err = float((got - grads_ref[name]).abs().max() / grads_ref[name].abs().max())
        print(f"  grad {name}: rel err {err:.2e}")
        assert err <= 1e-3, (name, err)
    # ---- linearity over a row split: payload(rows [0, h)) + payload(rows [h, N)) == payload(all rows), every block
    h = 499_968 + 77                                            # ragged: not a multiple of any tile
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
    for name, (a, b) in blocks.items():
        x, y = parts_T[a:b].double(), full_T[a:b].double()
        err = float((x - y).abs().max() / y.abs().max())
        print(f"  split-sum {name}: rel err {err:.2e}")
        assert err < 2e-5, (name, err)
    for i, name in enumerate(["site", "llw", "noise_g", "var_direct", "knm_k", "knm_dls"]):
        x, y = float(parts_d[i]), float(full_d[i])
        assert abs(x - y) <= 2e-6 * max(abs(y), 1.0), (name, x, y)
    assert abs(float(parts_d[7]) - float(full_d[7])) <= 1e-12 * abs(float(full_d[7]))       # the data constant

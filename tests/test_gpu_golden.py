"""GPU: the HIP path (through the C ABI and the Python surface) against the committed golden fixtures."""
import os

import numpy as np
import pytest
import torch

from tests._util import relerr


def close(a, b, rtol, atol=1e-13):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() <= rtol * np.abs(b).max() + atol

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
NAMES = ("log_lengthscale", "log_variance", "log_noise", "u_loc", "phi_unc", "u_scale_tril_unc")


def _engine(g, kind, dtype=torch.float64):
    from gdrf_amd.engine import Engine
    K, M = g["p0_u_loc"].shape if "p0_u_loc" in g else g["p_u_loc"].shape
    V = g["ws"].shape[1]
    eng = Engine(g["xs"].shape[0], M, K, V, g["xs"].shape[1], dtype=dtype, kernel=kind, jitter=float(g["jitter"]), maxjitter=15,
                 process_group=None)
    eng.set_inducing_points(torch.from_numpy(g["Z"]))
    eng.set_dirichlet(torch.from_numpy(g["alpha"]))
    return eng


@pytest.mark.parametrize("name", ["g1_artificial2d_rbf.npz", "g2_synth1d_matern52.npz"])
def test_five_adam_steps_match_golden(name):
    g = np.load(os.path.join(GOLD, name))
    kind = "rbf" if "rbf" in name else "matern52"
    eng = _engine(g, kind)
    for n in NAMES:
        eng.view(n).copy_(torch.from_numpy(g["p0_" + n]))
    xs = torch.from_numpy(g["xs"]).to(eng.device, eng.dtype).contiguous()
    ws = torch.from_numpy(g["ws"]).to(eng.device).contiguous()
    eps = torch.from_numpy(g["eps"]).to(eng.device, eng.dtype)
    eng.loss_and_grads(xs, ws, eps[0].contiguous())
    out = eng.read_out()
    assert eng.last_jitter_level == int(g["level0"])
    assert abs(out["loss"] - float(g["loss0"])) < 1e-6 * abs(float(g["loss0"]))      # fp32-lgamma quirk of the oracle
    gv = eng.named_views(eng.grads)
    for n in NAMES:
        assert close(gv[n].cpu().numpy(), g["g0_" + n], 1e-7), n
    losses = []
    for s in range(eps.shape[0]):
        eng.loss_and_grads(xs, ws, eps[s].contiguous())
        eng.adam("adam", 1e-2)
        losses.append(eng.read_out()["loss"])
        if s == 0:
            for n in NAMES:
                assert close(eng.view(n).cpu().numpy(), g["p1_" + n], 1e-9), n
    assert np.allclose(losses, g["losses"], rtol=1e-6)
    for n in NAMES:
        assert close(eng.view(n).cpu().numpy(), g[f"p{eps.shape[0]}_" + n], 1e-8), n
    assert np.abs(eng.predict(xs, 1).cpu().numpy() - g["topic_probs"]).max() < 1e-8
    s = eng.predict(xs, 3, ws).cpu().numpy()
    assert abs(np.exp(-s[0] / s[1]) - float(g["perplexity"])) < 1e-8 * float(g["perplexity"])


@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_stage_values_match_golden(dtype):
    g = np.load(os.path.join(GOLD, "g3_stages_64x16.npz"))
    eng = _engine(g, "rbf", dtype)
    for n in NAMES:
        eng.view(n).copy_(torch.from_numpy(g["p_" + n]).to(dtype))
    xs = torch.from_numpy(g["xs"]).to(eng.device, dtype).contiguous()
    ws = torch.from_numpy(g["ws"]).to(eng.device).contiguous()
    eps = torch.from_numpy(g["eps"]).to(eng.device, dtype).contiguous()
    eng.loss_and_grads(xs, ws, eps, force_level=0)
    tol = 1e-9 if dtype == torch.float64 else 5e-4
    assert relerr(eng.knm(xs).cpu().numpy(), g["a_Knm"]) < (1e-12 if dtype == torch.float64 else 5e-6)
    stages = ["Kuu", "L", "W", "loc", "tt", "mu"]
    # with the initial (uniform) word-topic matrix the softmax Jacobian cancels thetabar exactly; in fp32 the
    # cancellation leaves rounding noise larger than the remaining signal, so vbar / Wbar are checked in fp64 only
    if dtype == torch.float64:
        stages += ["vbar", "Wbar"]
    for ws_name in stages:
        # vbar / Wbar: the fixture's softmax pull-back theta_k (thetabar_k - sum_j theta_j thetabar_j) cancels thetabar ~ 1e2 down to
        # the remaining 1e-6 in fp64 (1e-9 relative noise IN THE FIXTURE); the kernels' cancellation-free form returns the exact zero
        assert relerr(eng.workspace(ws_name, 64).cpu().numpy(), g["a_" + ws_name]) < (1e-8 if ws_name in ("vbar", "Wbar") else tol), ws_name
    assert abs(eng.read_out()["loss"] - float(g["loss"])) < max(tol, 1e-9) * abs(float(g["loss"]))


def test_duplicate_inducing_point_needs_the_golden_jitter_level():
    g = np.load(os.path.join(GOLD, "g4_jitter_duplicate.npz"))
    from gdrf_amd.engine import Engine
    eng = Engine(8, g["Z"].shape[0], 2, 5, 2, dtype=torch.float64, kernel="rbf", jitter=float(g["jitter"]), maxjitter=20,
                 process_group=None)
    eng.set_inducing_points(torch.from_numpy(g["Z"]))
    eng.view("log_lengthscale").fill_(float(np.log(0.4)))
    eng.view("log_variance").fill_(float(np.log(25.0)))
    # K_uu is exactly singular; which attempt first sees a positive pivot depends on rounding (LAPACK needed
    # level_fp64 attempts).  The HIP factorisation must succeed no later than one level after it, use the same
    # cumulative schedule, and return a finite factor that reproduces K_uu + jitter.
    lvl = eng.factorize()
    assert lvl <= int(g["level_fp64"]) + 1
    assert abs(eng.jitter_total(int(g["level_fp64"])) - float(g["total"])) < 1e-12 * float(g["total"])
    L = eng.workspace("L").cpu().numpy()
    Kuu = eng.workspace("Kuu").cpu().numpy()
    assert np.isfinite(L).all() and np.abs(L @ L.T - Kuu).max() < 1e-6 * np.abs(Kuu).max()

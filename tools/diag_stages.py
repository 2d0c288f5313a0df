"""Diagnostic (GPU): stage-by-stage error of the fp32 build against the fp64 numpy restatement at the headline workload's
conditioning, per arithmetic mode and for several sizes of the perturbation of u_scale_tril.  Usage: python tools/diag_stages.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests._util import dev, engine_from_oracle, make_oracle, relerr
from tests.test_gpu_parity import _aux

def run(sp, mode, npts=(32, 16), ls=0.1, W=60, H=50, trained=None):
    m, eps = make_oracle(kind="rbf", W=W, H=H, V=50, K=10, n_points=npts, dtype=torch.float64, jitter=1e-6, lengthscale=ls, s_perturb=sp, trained_scale=trained)
    with torch.no_grad():
        for p in m.params.values(): p.copy_(p.float().double())
        m.Z = m.Z.float().double()
    eng = engine_from_oracle(m, dtype=torch.float32, mfma_mode=mode)
    xs, ws = dev(m.xs, eng), dev(m.ws, eng, torch.int32)
    lvl = eng.factorize(); m.force_jitter_level = lvl
    eng.loss_and_grads(xs, ws, dev(eps, eng), force_level=lvl)
    out = eng.read_out()
    loss_np, g_np, aux = _aux(m, eps, lvl)
    rep = {}
    for name in ["W", "q", "loc", "tt", "mu", "vbar", "locbar", "Wbar"]:
        rep[name] = relerr(eng.workspace(name, m.N).cpu().double().numpy(), aux[name])
    lay = eng.red_layout; Mp = (eng.M + 31) // 32 * 32; mm = Mp * Mp
    A = eng.red_T[lay["A"]:lay["A"] + eng.K * mm].view(eng.K, Mp, Mp)[:, :eng.M, :eng.M].cpu().double().numpy()
    GT = eng.red_T[lay["GT"]:lay["GT"] + mm].view(Mp, Mp)[:eng.M, :eng.M].cpu().double().numpy()
    rep["A"] = relerr(A, aux["A"]); rep["GT"] = relerr(GT, aux["G"])          # aux G = Wbar^T W; the engine stores W^T Wbar
    rep["GT_t"] = relerr(GT, aux["G"].T)
    vb = aux["vbar"]; rep["vbar_range"] = float(np.abs(vb).max() / np.median(np.abs(vb)))
    # the A_k kernel in isolation: fp64 product of the engine's OWN float32 inputs
    We = eng.workspace("W", m.N).cpu().double().numpy(); ve = eng.workspace("vbar", m.N).cpu().double().numpy()
    Aiso = np.einsum("ni,kn,nj->kij", We, ve, We)
    Aabs = np.einsum("ni,kn,nj->kij", np.abs(We), np.abs(ve), np.abs(We))
    rep["A_iso"] = relerr(A, Aiso); rep["A_cancel"] = float(Aabs.max() / np.abs(Aiso).max())
    k = int(np.argmax(np.abs(A - Aiso).reshape(eng.K, -1).max(1))); d = np.abs(A[k] - Aiso[k]); ij = np.unravel_index(np.argmax(d), d.shape)
    rep["A_worst_at"] = "k%d i%d j%d |A|=%.2e max|A_k|=%.2e" % (k, ij[0], ij[1], abs(Aiso[k][ij]), np.abs(Aiso[k]).max())
    gv = eng.named_views(eng.grads)
    for name in eng.PARAM_NAMES:
        rep["g_" + name] = relerr(gv[name].cpu().double().numpy(), g_np[name])
    v = aux["tt"] + np.maximum(25.0 - aux["q"], 0)[None, :] if aux["q"].ndim == 1 else None
    print(f"s_perturb={sp} mode={mode} level={lvl} max tt={aux['tt'].max():.1f} max|mu|={np.abs(aux['mu']).max():.1f}", {k: (x if isinstance(x, str) else f"{x:.1e}") for k, x in rep.items()}, flush=True)

def run_t(mode):
    m, eps = make_oracle(kind="rbf", W=60, H=50, V=50, K=10, n_points=(32, 16), dtype=torch.float64, jitter=1e-6, lengthscale=0.1, s_perturb=0.02, trained_scale=0.3)
    return m, eps

for mode in ("f16x3", "bf16x6", "f32"):
    run(0.02, mode, trained=0.3)

"""Generates the committed golden fixtures (tests/golden/*.npz) from the CPU oracle.

Run in the build container:  python tests/golden/make_golden.py
The reference itself cannot be imported (pyro-ppl is absent) and holds no fixtures for
this path, so these vectors pin the ORACLE (regression) and the HIP path against it; they are not outputs
of the reference ("parity unpinned", oracle/gdrf_oracle.py header).

G1: first 256 rows of the reference's data/data_2d_artificial.csv (read as DATA, normalised as
    gdrf/train_script.py:261-268), K=3, M=[8,4], RBF l=0.3 var=25, alpha=0.01, fixed eps: loss, site sums,
    unconstrained grads, params after 1 and 5 Adam steps, topic_probs.   (fp64)
G2: 1-D synthetic N=512, M=32, K=4, V=20, Matern52, same outputs.
G3: stage values (Knm, Kuu, L, W, loc, var) for a 64 x 16 case.
G4: jitter-retry case (duplicate inducing point): the level the fp64 oracle needs from jitter=1e-12.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle.gdrf_oracle import RefShapedGDRF, fused_elbo_and_grads, jitter_total, jittercholesky, kernel_matrix  # noqa: E402
from gdrf_amd.data import normalise_index, synth_circles  # noqa: E402


def run_case(xs, ws, name, steps=5, **kw):
    m = RefShapedGDRF(xs, ws, dtype=torch.float64, optimizer="adam", lr=1e-2, **kw)
    g = torch.Generator().manual_seed(11)
    K = m.K
    # move off the symmetric initial point (u_loc = 0, uniform word-topic matrix): there the exact gradient of
    # u_loc is 0 and Adam would normalise pure rounding noise to +-lr, which no two implementations share
    with torch.no_grad():
        m.params["u_loc"].add_(0.3 * torch.randn(m.params["u_loc"].shape, generator=g, dtype=torch.float64))
        m.params["u_scale_tril_unc"].add_(0.1 * torch.randn(m.params["u_scale_tril_unc"].shape, generator=g, dtype=torch.float64).tril())
        m.params["phi_unc"].add_(0.5 * torch.randn(m.params["phi_unc"].shape, generator=g, dtype=torch.float64))
        m.params["log_noise"].add_(0.2)
    out = dict(xs=np.asarray(xs, dtype=np.float64), ws=np.asarray(ws, dtype=np.int32), Z=m.Z.numpy(), alpha=m.alpha.numpy(),
               jitter=np.float64(m.jitter))
    for k, v in m.params.items():
        out["p0_" + k] = v.detach().numpy().copy()
    eps_all = torch.randn(steps, K, m.N, generator=g, dtype=torch.float64)
    out["eps"] = eps_all.numpy()
    loss, grads = m.loss_and_grads(eps_all[0])
    out["loss0"] = np.float64(loss)
    out["level0"] = np.int64(m.last_jitter_level)
    for k, v in m.last_terms.items():
        out["term_" + k] = np.float64(v)
    for k, v in grads.items():
        out["g0_" + k] = v.numpy()
    losses = []
    for s in range(steps):
        losses.append(m.step(eps_all[s]))
        if s in (0, steps - 1):
            for k, v in m.params.items():
                out[f"p{s + 1}_" + k] = v.detach().numpy().copy()
    out["losses"] = np.asarray(losses)
    out["topic_probs"] = m.topic_probs().numpy()
    out["perplexity"] = np.float64(m.perplexity())
    np.savez_compressed(os.path.join(HERE, name), **out)
    print(name, "loss0", loss, "levels", m.last_jitter_level, "perplexity", float(out["perplexity"]))


def main():
    # G1: reference data file, read as data
    csv = "/root/reference/data/data_2d_artificial.csv"
    import pandas as pd
    df = pd.read_csv(csv, index_col=[0, 1], header=0).fillna(0).astype(int)
    idx = np.array(df.index.to_list())
    xs_all = normalise_index(idx)
    sel = slice(0, 256)
    # rows 0..255 cover x in {0..7}: renormalise the selection as train() would for that file slice
    xs = normalise_index(idx[sel])
    run_case(xs, df.values[sel].astype(np.int32), "g1_artificial2d_rbf.npz", kind="rbf", K=3, n_points=(8, 4), lengthscale=0.3,
             variance=25.0, dirichlet_param=0.01, jitter=1e-6)
    # G2: 1-D synthetic, Matern52
    xs2, ws2, _ = synth_circles(512, 1, 20, 4, seed=5, one_d=True)
    run_case(xs2, ws2, "g2_synth1d_matern52.npz", kind="matern52", K=4, n_points=(32,), lengthscale=0.05, variance=4.0,
             dirichlet_param=0.1, jitter=1e-6)
    # G3: stage values
    xs3, ws3, _ = synth_circles(8, 8, 10, 3, seed=9)
    m = RefShapedGDRF(xs3, ws3, dtype=torch.float64, kind="rbf", K=3, n_points=(4, 4), lengthscale=0.4, jitter=1e-6)
    g = torch.Generator().manual_seed(2)
    with torch.no_grad():
        m.params["u_loc"].add_(0.2 * torch.randn(m.params["u_loc"].shape, generator=g, dtype=torch.float64))
    eps = torch.randn(3, 64, generator=g, dtype=torch.float64)
    P = {k: v.detach().numpy().copy() for k, v in m.params.items()}
    loss, grads, aux = fused_elbo_and_grads("rbf", m.xs.numpy(), m.ws.numpy(), m.Z.numpy(), P, m.alpha.numpy(), eps.numpy(),
                                            jitter_total(1e-6, 0))
    np.savez_compressed(os.path.join(HERE, "g3_stages_64x16.npz"), xs=m.xs.numpy(), ws=m.ws.numpy(), Z=m.Z.numpy(),
                        alpha=m.alpha.numpy(), eps=eps.numpy(), loss=np.float64(loss), jitter=np.float64(1e-6),
                        **{"p_" + k: v for k, v in P.items()}, **{"g_" + k: v for k, v in grads.items()},
                        **{"a_" + k: aux[k] for k in ["Knm", "Kuu", "L", "W", "loc", "var", "tt", "mu", "vbar", "Wbar"]})
    print("g3 loss", loss)
    # G4: duplicate inducing point -> cumulative jitter schedule
    Z = m.Z.clone()
    Z[1] = Z[0]
    Kuu = kernel_matrix("rbf", Z, Z, torch.tensor(0.4, dtype=torch.float64), torch.tensor(25.0, dtype=torch.float64))
    L, lvl = jittercholesky(Kuu.clone(), Z.shape[0], 1e-18, 20)
    np.savez_compressed(os.path.join(HERE, "g4_jitter_duplicate.npz"), Z=Z.numpy(), level_fp64=np.int64(lvl),
                        jitter=np.float64(1e-18), total=np.float64(jitter_total(1e-18, lvl)))
    print("g4 level", lvl)


if __name__ == "__main__":
    main()

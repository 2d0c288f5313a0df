"""Synthetic GDRF workloads and the reference's CSV normalisation.

``synth_circles`` follows the recipe of the reference's ``generate_data_2d_circles``
(gdrf/models/utils.py:106-193: one-topic discs over a uniform background, Girdhar-style
block word distributions, U{V..10V-1} words per cell), vectorised with numpy so that the
N = 1e6 benchmark lattice is generated in seconds.  ``normalise_index`` is the index -> [0,1]^D
map of gdrf/train_script.py:261-267.
"""
from __future__ import annotations

import numpy as np


def normalise_index(index: np.ndarray) -> np.ndarray:
    index = np.asarray(index, dtype=np.float64)
    if index.ndim == 1:
        index = index[:, None]
    index = index - index.min(axis=0, keepdims=True)
    return index / index.max(axis=0, keepdims=True)


def synth_circles(W: int, H: int, V: int, K: int, *, n_discs=8, R_frac=0.1, eta=0.1, seed=777,
                  one_d: bool = False):
    """Regular lattice xs in [0,1]^D (train_script.py:261-267 normalisation), K-1 disc
    topics over a uniform background, per-cell total count ~ U{V..10V-1}, ws ~ Multinomial."""
    rng = np.random.default_rng(seed)
    if one_d:
        N = W
        idx = np.arange(N, dtype=np.float64)[:, None]
        xs = idx / idx.max()
        centers = rng.uniform(0.1, 0.9, size=(n_discs, 1))
    else:
        gx, gy = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
        idx = np.stack([gx.ravel(), gy.ravel()], 1).astype(np.float64)
        xs = idx / idx.max(0, keepdims=True)
        centers = rng.uniform(0.1, 0.9, size=(n_discs, 2))
    N = xs.shape[0]
    K_obj = K - 1
    obj_topics = rng.integers(0, K_obj, size=n_discs) if K_obj > 0 else np.full(n_discs, K - 1)
    p_v_z = np.full((K, V), eta)
    for k in range(K_obj):
        lo, hi = V * k / K_obj, V * (k + 1) / K_obj
        vs = np.arange(V)
        p_v_z[k, (vs >= lo) & (vs < hi)] += 1.0
    p_v_z[K - 1, :] = 1.0 / V
    p_v_z /= p_v_z.sum(-1, keepdims=True)
    topic = np.full(N, K - 1)
    for i in range(n_discs - 1, -1, -1):
        d2 = ((xs - centers[i]) ** 2).sum(-1)
        topic[d2 <= R_frac * R_frac] = obj_topics[i]
    counts = rng.integers(V, 10 * V, size=N)
    ws = np.empty((N, V), dtype=np.int32)
    for k in range(K):
        sel = np.nonzero(topic == k)[0]
        if sel.size:
            ws[sel] = rng.multinomial(counts[sel], p_v_z[k]).astype(np.int32)
    return xs.astype(np.float32), ws, topic

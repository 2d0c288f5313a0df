"""Minimal training driver for the hot path, mirroring the parts of ``gdrf/train_script.py::train`` that feed it:
CSV -> tensors (``:251-273``), kernel / model / optimizer / objective construction (``:289-335``), the SVI object
(``:365-371``), the epoch loop with optional streaming inference (``:389-508``) and the per-epoch perplexity.
Logging to W&B, plots, checkpoints-to-disk and early stopping are outside this build's scope;
the function returns what the loggers would have recorded.

The keyword names are the reference's (``train_script.py:102-145``) for the options that reach the hot path.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch

from . import poutine
from .data import normalise_index
from .infer import OBJECTIVE_DICT, SVI
from .kernels import KERNEL_DICT
from .models import GDRF_MODEL_DICT
from .optim import OPTIMIZER_DICT

STREAMING_MODES = ("uniform", "now", "exp", "uniform_now", "exp_now", "uniform_exp")


def streaming_probabilities(mode: str, n_stream: int, streaming_weight: float = 0.1, streaming_exp: float = 1.0) -> np.ndarray:
    """The selection probabilities over the first ``n_stream`` observations (train_script.py:406-450), normalised."""
    if mode == "uniform":
        p = [1.0] * n_stream
    elif mode == "now":
        p = [0.0] * n_stream
        p[-1] = 1.0
    elif mode == "exp":
        p = [np.exp(-streaming_exp * (n_stream - i)) for i in range(n_stream)]
    elif mode == "uniform_now":
        p = [streaming_weight / n_stream] * n_stream
        p[-1] += 1 - streaming_weight
    elif mode == "exp_now":
        p = [np.exp(-streaming_exp * (n_stream - i)) for i in range(n_stream)]
        tot = sum(p)
        p = [streaming_weight * q / tot for q in p]
        p[-1] += 1 - streaming_weight
    elif mode == "uniform_exp":
        p = [np.exp(-streaming_exp * (n_stream - i)) for i in range(n_stream)]
        tot = sum(p)
        p = [(1 - streaming_weight) * q / tot for q in p]
        p = [q + streaming_weight / (n_stream + 1) for q in p]
    else:
        raise ValueError("streaming_inference should be one of 'uniform', 'now', 'exp', 'uniform_now, 'exp_now', or "
                         f"'uniform_exp'; you passed {mode}")
    p = np.asarray(p, dtype=np.float64)
    return p / p.sum()


def load_csv(path: str, dimensions: int):
    """train_script.py:251-273: the first ``dimensions`` columns are the index; counts are filled/cast to int."""
    import pandas as pd
    df = pd.read_csv(filepath_or_buffer=path, index_col=list(range(dimensions)), header=0, parse_dates=True).fillna(0).astype(int)
    index = df.index.values if dimensions == 1 else np.array(df.index.to_list())
    xs = normalise_index(index.astype(np.float64) if index.dtype.kind in "iuf" else (index - index.min()) / np.timedelta64(1, "s"))
    return xs.astype(np.float32), df.values.astype(np.int32)


def train(data: Union[str, None] = None, xs: Optional[np.ndarray] = None, ws: Optional[np.ndarray] = None, device: str = "cuda:0",
          dimensions: int = 1, epochs: int = 3000, model_type: str = "sparsemultinomialgdrf", num_topics: int = 1,
          dirichlet_param: float = 0.01, num_inducing_points: Union[int, Sequence[int]] = 25, fixed_inducing_points: bool = True,
          inducing_initialization_method: str = "random", jitter: float = 1e-8, max_jitter: int = 15, kernel_type: str = "rbf",
          kernel_lengthscale: float = 0.1, kernel_variance: float = 25.0, optimizer_type: str = "adamw", optimizer_lr: float = 0.001,
          objective_type: str = "graphelbo", objective_num_particles: int = 1, streaming_inference: str = "",
          streaming_weight: float = 0.1, streaming_exp: float = 1.0, streaming_truncate: int = -1, streaming_size: int = 1,
          streaming_subepochs: int = 1, streaming_batch_splits: int = -1, randomize_wt_matrix: bool = False, seed: int = 1,
          dtype: torch.dtype = torch.float32, perplexity_every: int = 1) -> Dict[str, object]:
    if data is not None:
        xs, ws = load_csv(data, dimensions)
    xs_t = torch.as_tensor(xs).float().to(device)
    if xs_t.dim() == 1:
        xs_t = xs_t.unsqueeze(-1)
    ws_t = torch.as_tensor(ws).int().to(device)
    world = list(zip(xs_t.min(dim=0).values.cpu().numpy().tolist(), xs_t.max(dim=0).values.cpu().numpy().tolist()))
    n_data = len(xs_t)
    streaming = streaming_inference != ""
    streaming_batch = streaming and streaming_batch_splits > 0
    if streaming and not streaming_batch:
        epochs = n_data
    elif streaming_batch:
        epochs = streaming_batch_splits
    rng = np.random.RandomState(seed)                                   # init_seeds(1) seeds numpy (utils/general.py:103-108)

    kernel = KERNEL_DICT[kernel_type](input_dim=xs_t.shape[1], lengthscale=torch.tensor(kernel_lengthscale),
                                      variance=torch.tensor(kernel_variance)).to(device)
    model = GDRF_MODEL_DICT[model_type](
        xs=xs_t, ws=ws_t, world=world, kernel=kernel, num_observation_categories=ws_t.shape[1], device=device,
        num_topic_categories=num_topics, dirichlet_param=dirichlet_param, n_points=num_inducing_points,
        fixed_inducing_points=fixed_inducing_points, inducing_init=inducing_initialization_method, maxjitter=max_jitter,
        jitter=jitter, randomize_wt_matrix=randomize_wt_matrix, dtype=dtype, seed=seed)
    optimizer = OPTIMIZER_DICT[optimizer_type]({"lr": optimizer_lr})
    objective = OBJECTIVE_DICT[objective_type](max_plate_nesting=1, vectorize_particles=True, num_particles=objective_num_particles)
    scale = poutine.scale(scale=1.0 / len(xs_t))
    svi = SVI(model=scale(model.model), guide=scale(model.guide), optim=optimizer, loss=objective)

    history: List[List[float]] = []
    for epoch in range(epochs):
        model.train()
        if streaming:
            for _ in range(streaming_subepochs):
                effective_epoch = epoch * n_data // epochs if streaming_batch else epoch
                n_stream = min(effective_epoch + 1, streaming_truncate) if streaming_truncate > 0 else effective_epoch + 1
                p = streaming_probabilities(streaming_inference, n_stream, streaming_weight, streaming_exp)
                selection = rng.choice(n_stream, size=streaming_size if streaming_size > 1 else None, p=p)
                if streaming_truncate > 0:
                    selection = selection + epoch + 1 - n_stream
                sel = np.atleast_1d(selection)
                loss = svi.step(xs=xs_t[sel, ...], ws=ws_t[sel, ...], subsample=False)
        else:
            loss = svi.step(xs=xs_t, ws=ws_t, subsample=False)
        model.eval()
        perplexity = float(model.perplexity(xs_t, ws_t).item()) if (epoch % perplexity_every == 0 or epoch == epochs - 1) else float("nan")
        history.append([float(loss), perplexity, float(model.kernel_lengthscale), float(model.kernel_variance)])
    return {"model": model, "svi": svi, "optimizer": optimizer, "history": np.asarray(history),
            "keys": ["train/loss", "metrics/perplexity", "x/kernel.lengthscale", "x/kernel.variance"]}

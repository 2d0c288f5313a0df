"""world_size=2 (gloo, CPU): the observation-sharded evaluation -- each rank reduces its own rows,
ONE all-reduce of the flat payload, identical replicated epilogue -- reproduces the unsharded loss and
gradients.  Uses the oracle's local/finish split, which is the same
factorisation gdrf_step_local / gdrf_step_finish implement on the GPU."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pack(payload):
    keys = sorted(payload)
    flat = np.concatenate([np.atleast_1d(np.asarray(payload[k], dtype=np.float64)).ravel() for k in keys])
    return keys, [np.asarray(payload[k]).shape for k in keys], flat


def _unpack(keys, shapes, flat):
    out, o = {}, 0
    for k, s in zip(keys, shapes):
        n = int(np.prod(s)) if len(s) else 1
        out[k] = flat[o:o + n].reshape(s) if len(s) else float(flat[o])
        o += n
    return out


def _worker(rank, world, port, tmp):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle.gdrf_oracle import fused_finish, fused_local, jitter_total
    from tests._util import make_oracle
    m, eps = make_oracle(kind="rbf", W=14, H=9, V=15, K=3, n_points=(4, 3))      # same seed on every rank
    P = {k: v.detach().numpy().copy() for k, v in m.params.items()}
    N = m.N
    lo, hi = rank * N // world, (rank + 1) * N // world                          # contiguous row blocks
    jt = jitter_total(m.jitter, 0)
    payload, _ = fused_local("rbf", m.xs.numpy()[lo:hi], m.ws.numpy()[lo:hi], m.Z.numpy(), P, eps.numpy()[:, lo:hi], jt)
    keys, shapes, flat = _pack(payload)
    t = torch.from_numpy(flat)
    dist.all_reduce(t)                                                           # the step's single collective
    red = _unpack(keys, shapes, t.numpy())
    loss, grads, _ = fused_finish("rbf", m.Z.numpy(), P, m.alpha.numpy(), red, float(N), jt)
    np.savez(os.path.join(tmp, f"rank{rank}.npz"), loss=loss, **grads)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_sharded_step_equals_unsharded(tmp_path):
    from oracle.gdrf_oracle import fused_elbo_and_grads, jitter_total
    from tests._util import make_oracle
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    m, eps = make_oracle(kind="rbf", W=14, H=9, V=15, K=3, n_points=(4, 3))
    P = {k: v.detach().numpy().copy() for k, v in m.params.items()}
    loss, grads, _ = fused_elbo_and_grads("rbf", m.xs.numpy(), m.ws.numpy(), m.Z.numpy(), P, m.alpha.numpy(), eps.numpy(),
                                          jitter_total(m.jitter, 0))
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert float(r0["loss"]) == float(r1["loss"])                                # replicated epilogue: bit-identical
    assert abs(float(r0["loss"]) - loss) < 1e-12 * abs(loss)
    for k, g in grads.items():
        assert np.array_equal(r0[k], r1[k]), k
        assert np.abs(r0[k] - g).max() <= 1e-11 * max(1.0, np.abs(g).max()), k

"""Diagnostic (not part of the product or the tests): per-workgroup phase stamps of the f64 W = K_nm L^-T launch.

Needs a library built with -DGDRF_NT_TRACE (make -C gdrf_amd/csrc EXTRA=-DGDRF_NT_TRACE).  Runs one headline step with
GDRF_NT_TRACE_FILE set, reads the stamps (wall_clock64, 100 MHz) and prints, per column tile: the time from workgroup start to the
first staged chunk, the reduction loop, the epilogue, and the number of workgroups resident per CU over the launch.
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

OUT = "gpurun_out/nt_trace.bin"
os.makedirs("gpurun_out", exist_ok=True)


def build_step():
    from gdrf_amd.data import synth_circles
    from gdrf_amd.infer import SVI, Trace_ELBO
    from gdrf_amd.kernels import RBF
    from gdrf_amd.models import SparseMultinomialGDRF
    from gdrf_amd.optim import Adam
    from gdrf_amd import poutine
    xs_np, ws_np, _ = synth_circles(1000, 1000, 50, 10, seed=0)
    N = xs_np.shape[0]
    xs = torch.from_numpy(xs_np).to("cuda:0", torch.float32).contiguous()
    ws = torch.from_numpy(ws_np).to("cuda:0").contiguous()
    model = SparseMultinomialGDRF(xs=xs, ws=ws, world=[(0.0, 1.0)] * 2, kernel=RBF(input_dim=2, lengthscale=torch.tensor(0.1), variance=torch.tensor(25.0)),
                                  num_observation_categories=50, device="cuda:0", num_topic_categories=10, dirichlet_param=0.01, n_points=[32, 16],
                                  fixed_inducing_points=True, inducing_init="grid", maxjitter=15, jitter=1e-6, randomize_wt_matrix=False,
                                  dtype=torch.float32, seed=0)
    scale = poutine.scale(scale=1.0 / N)
    svi = SVI(model=scale(model.model), guide=scale(model.guide), optim=Adam({"lr": 1e-3}), loss=Trace_ELBO(max_plate_nesting=1, vectorize_particles=True, num_particles=1))
    eng = model._engine_for(N)

    def step():
        return svi.step(xs=xs, ws=ws, subsample=False)
    step.eng = eng
    return step


def main():
    step = build_step()
    eng = step.eng

    class _S:
        pass
    model = _S()
    model.step = step
    for _ in range(2):
        model.step()
    torch.cuda.synchronize()
    os.environ["GDRF_NT_TRACE_FILE"] = OUT
    model.step()
    torch.cuda.synchronize()
    del os.environ["GDRF_NT_TRACE_FILE"]
    a = np.fromfile(OUT, dtype=np.uint64).reshape(-1, 8)
    a = a[a[:, 3] > 0]
    t0 = a[:, 3].min()
    tick = 10.0e-3                                            # us per tick (100 MHz)
    st, fc, le, te, en = [(a[:, i] - t0).astype(np.float64) * tick for i in (3, 4, 5, 6, 7)]
    M = eng.M
    nct = (((M + 31) // 32 * 32) + 63) // 64
    bid = (a[:, 0] & np.uint64(0xFFFFFFFF)).astype(np.int64)
    ct = (a[:, 0] >> np.uint64(32)).astype(np.int64)
    print(f"workgroups {len(a)}, launch span {en.max():.1f} us, column tiles {nct}")
    print("ct  count  prologue  loop   epilogue  finish  total   (mean us)")
    for c in range(nct):
        m = ct == c
        print(f"{c:2d} {m.sum():6d}  {np.mean(fc[m] - st[m]):7.2f} {np.mean(le[m] - fc[m]):7.2f} {np.mean(te[m] - le[m]):7.2f}"
              f" {np.mean(en[m] - te[m]):7.2f} {np.mean(en[m] - st[m]):7.2f}")
    tot = en - st
    print(f"sum of workgroup lifetimes {tot.sum():.0f} us over {en.max():.1f} us x 256 CUs = {tot.sum() / en.max() / 256:.2f} resident per CU")
    for nm, x in (("prologue", fc - st), ("loop", le - fc), ("epilogue", te - le), ("finish", en - te)):
        print(f"  {nm}: {x.sum() / tot.sum():.3f} of the lifetime")
    hw = a[:, 1].astype(np.int64)
    xcc = a[:, 2].astype(np.int64) & 0xF
    cu = (hw >> 8) & 0xF
    sh = (hw >> 12) & 0x1
    se = (hw >> 13) & 0x7
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ose = np.array([[np.sum((xcc == x) & (se == e)) for e in range(4)] for x in range(8)])
    print("work units (ct + 1) per (XCD, shader engine):")
    print(np.array([[np.sum((ct + 1)[(xcc == x) & (se == e)]) for e in range(4)] for x in range(8)]))
    print("distinct (xcc, se, sh, cu):", len(np.unique(cuid)), " xcc values:", np.unique(xcc), " block&7 == xcc for", np.mean((bid & 7) == xcc))
    # residency over time for a few CUs
    grid = np.linspace(0, en.max(), 400)
    occ_all = []
    for u in np.unique(cuid):
        m = cuid == u
        occ = ((st[m][None, :] <= grid[:, None]) & (en[m][None, :] > grid[:, None])).sum(1)
        occ_all.append(occ)
    occ_all = np.array(occ_all)
    print("resident workgroups per CU over time (mean over CUs), 20 bins:", np.round(occ_all.mean(0).reshape(20, -1).mean(1), 2))
    print("histogram of per-CU residency samples:", np.bincount(occ_all.ravel().astype(int), minlength=5))
    # gaps: time between a workgroup ending on a CU and the next one starting there
    gaps = []
    for u in np.unique(cuid)[:64]:
        m = cuid == u
        s_, e_ = np.sort(st[m]), np.sort(en[m])
        # k-th start (k >= slots) vs (k - slots)-th end with 3 slots
        for slots in (3,):
            if len(s_) > slots:
                gaps.append(s_[slots:] - e_[:-slots])
    g = np.concatenate(gaps)
    print(f"start[k+3] - end[k] per CU: mean {g.mean():.2f} us, median {np.median(g):.2f}, p10 {np.percentile(g, 10):.2f}, p90 {np.percentile(g, 90):.2f}")
    # xcd-level: order of starts by block id
    o = np.argsort(bid)
    d = np.diff(st[o])
    print(f"start time vs block order: monotone fraction {np.mean(d >= 0):.3f}")
    np.save("gpurun_out/nt_trace.npy", a)


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Benchmark of the GDRF SVI ELBO hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

A "step" is one full ``SVI.step(xs, ws)`` of ``SparseMultinomialGDRF`` (jitter-retry Cholesky,
forward, backward, all-reduce, epilogue, Adam, loss read-back) on the BASELINE.json headline
workload: N = 1e6 synthetic 2-D observations, V = 50 taxa, K = 10 topics, M = 512 inducing points
(32 x 16 grid), RBF kernel, fp32.  With N GPUs the 1e6 observations are sharded over the ranks
(strong scaling: the total work is fixed).  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  ``roofline`` is the dominant kernel of the step (an f32-MFMA GEMM);
``roofline_knm`` is the standalone K_nm kernel the metric names (HBM-bound).  ``cpu_baseline``
times the reference-shaped torch-CPU oracle (oracle/gdrf_oracle.py) on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_* dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: bf16 dense peak (16x the f32 MFMA rate)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", type=int, default=1_000_000, help="global number of observations (lattice side^2)")
    ap.add_argument("--n-points", type=int, nargs="+", default=[32, 16])
    ap.add_argument("--topics", type=int, default=10)
    ap.add_argument("--vocab", type=int, default=50)
    ap.add_argument("--kernel", default="rbf")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--jitter", type=float, default=1e-6)
    ap.add_argument("--cpu-baseline-n", type=int, default=20000, help="rows of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--cpu-baseline-steps", type=int, default=3)
    ap.add_argument("--knm-iters", type=int, default=50)
    ap.add_argument("--seed", type=int, default=777)
    ap.add_argument("--mfma-mode", default="auto", choices=["auto", "f32", "bf16x6"],
                    help="arithmetic of the f32 GEMM-shaped contractions: exact-split emulation on bf16 MFMA (the default for "
                         "float32), or native f32 MFMA")
    return ap.parse_args()


def lattice_shape(n):
    w = int(round(n ** 0.5))
    while n % w:
        w -= 1
    return n // w, w


def host_cores() -> int:
    """CPUs this process may actually use (cgroup quota / affinity), not the machine's core count."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return min(n, int(os.environ.get("GDRF_BENCH_CPU_THREADS", 16)))


def cpu_baseline(args, M_points):
    """Reference-shaped oracle (2x conditional, materialised W.S, autograd, per-parameter Adam), fp32,
    all host threads, on a bounded sample of the same workload; extrapolated linearly in N."""
    from oracle.gdrf_oracle import RefShapedGDRF
    from gdrf_amd.data import synth_circles
    ns = args.cpu_baseline_n
    W, H = lattice_shape(ns)
    xs, ws, _ = synth_circles(W, H, args.vocab, args.topics, seed=args.seed)
    cores = host_cores()
    torch.set_num_threads(cores)
    m = RefShapedGDRF(xs, ws, kind=args.kernel, K=args.topics, n_points=tuple(args.n_points), dtype=torch.float32,
                      jitter=args.jitter, maxjitter=15, optimizer="adam", lr=1e-3)
    g = torch.Generator().manual_seed(1)
    times = []
    for i in range(1 + args.cpu_baseline_steps):
        eps = torch.randn(args.topics, m.N, generator=g)
        t0 = time.perf_counter()
        m.step(eps, n_global=args.n)
        times.append(time.perf_counter() - t0)
    t = float(np.median(times[1:]))
    return {
        "value": (1.0 / t) * (m.N / args.n), "unit": "steps/s", "cores": cores, "kind": "port",
        "sample": f"reference-shaped torch-CPU oracle, fp32, N={m.N} rows of the same workload (M={M_points}, K={args.topics}, "
                  f"V={args.vocab}), median of {args.cpu_baseline_steps} steps after 1 warm-up = {t:.3f} s/step at N={m.N}; value is the "
                  f"linear-in-N extrapolation to N={args.n} (the reference shape needs > 100 GB of host RAM at N=1e6)",
        "sec_per_step_at_sample": t,
    }


def main():
    args = parse()
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    world = int(os.environ.get("WORLD_SIZE", 1))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        dist.init_process_group("nccl", device_id=device)

    from gdrf_amd.data import synth_circles
    from gdrf_amd.infer import SVI, Trace_ELBO
    from gdrf_amd.kernels import KERNEL_DICT
    from gdrf_amd.models import SparseMultinomialGDRF
    from gdrf_amd.optim import Adam
    from gdrf_amd import poutine

    dtype = torch.float32 if args.dtype == "f32" else torch.float64
    W, H = lattice_shape(args.n)
    xs_np, ws_np, _ = synth_circles(W, H, args.vocab, args.topics, seed=args.seed)
    N = xs_np.shape[0]
    lo, hi = rank * N // world, (rank + 1) * N // world          # contiguous row blocks
    xs = torch.from_numpy(xs_np[lo:hi]).to(device=device, dtype=dtype).contiguous()
    ws = torch.from_numpy(ws_np[lo:hi]).to(device=device).contiguous()
    del xs_np, ws_np
    D = xs.shape[1]
    kernel = KERNEL_DICT[args.kernel](input_dim=D, lengthscale=torch.tensor(0.1), variance=torch.tensor(25.0))
    model = SparseMultinomialGDRF(
        xs=xs, ws=ws, world=[(0.0, 1.0)] * D, kernel=kernel, num_observation_categories=args.vocab, device=str(device),
        num_topic_categories=args.topics, dirichlet_param=0.01, n_points=list(args.n_points), fixed_inducing_points=True,
        inducing_init="grid", maxjitter=15, jitter=args.jitter, randomize_wt_matrix=False, dtype=dtype, seed=args.seed,
        mfma_mode=args.mfma_mode)
    M = model.M
    optimizer = Adam({"lr": 1e-3})
    objective = Trace_ELBO(max_plate_nesting=1, vectorize_particles=True, num_particles=1)
    scale = poutine.scale(scale=1.0 / N)
    svi = SVI(model=scale(model.model), guide=scale(model.guide), optim=optimizer, loss=objective)
    svi.row_offset = lo
    eng = model._engine_for(xs.shape[0])

    def barrier():
        torch.cuda.synchronize(device)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)

    loss = None
    for _ in range(args.warmup):
        loss = svi.step(xs=xs, ws=ws, subsample=False)
    eng.set_timing(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = svi.step(xs=xs, ws=ws, subsample=False)
    barrier()
    dt = time.perf_counter() - t0
    timing = eng.get_timing()
    eng.set_timing(False)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    perplexity = float(model.perplexity(xs, ws).item()) if world == 1 else None

    # ---- standalone K_nm kernel (HBM roofline), same shard
    knm_out = torch.empty(xs.shape[0], M, dtype=dtype, device=device)
    for _ in range(5):                       # untimed: the first launches after the step loop run 10-15 % slow
        eng.knm_into(xs, knm_out)
    eng.set_timing(True)
    for _ in range(args.knm_iters):
        eng.knm_into(xs, knm_out)
    torch.cuda.synchronize(device)
    t_knm = eng.get_timing()["k_nm"]
    eng.set_timing(False)
    del knm_out

    if rank == 0:
        esz = 4 if dtype == torch.float32 else 8
        n_loc = xs.shape[0]
        K = args.topics
        ms = {k: (v["ms"] / v["count"] if v["count"] else 0.0) for k, v in timing.items()}
        per_step = {k: v["ms"] / args.steps for k, v in timing.items()}
        flops = {  # useful flops per launch (triangular/symmetric halves skipped, nothing counted twice)
            "fwd_w": 1.0 * n_loc * M * M, "loc": 2.0 * n_loc * M * K, "fwd_t": 1.0 * n_loc * M * M * K,
            "bwd_wbar": (1.0 if eng.stores_t else 2.0) * n_loc * M * M * K,   # triangular T_k S_k^T vs dense W B_k
            "bwd_knm": 1.0 * n_loc * M * M, "tn_sym": 1.0 * n_loc * M * M * K, "tn_gt": 2.0 * n_loc * M * M,
        }
        dom = max(flops, key=lambda k: per_step[k])
        dom_t = ms[dom] * 1e-3
        ach = flops[dom] / dom_t / 1e12 if dom_t > 0 else 0.0
        peak = PEAK_F32_MFMA_TFLOPS if dtype == torch.float32 else PEAK_F32_MFMA_TFLOPS / 2
        if dom in ("fwd_w", "bwd_knm") and not eng.pure_fp32:
            peak = PEAK_F32_MFMA_TFLOPS / 2          # the solve-side GEMMs run on f64 MFMA in every mode but the all-fp32 one
        emulated = eng.mfma_mode == "bf16x6" and dom in ("fwd_t", "bwd_wbar", "tn_sym", "tn_gt")
        if emulated:     # six bf16 MFMA products per f32 multiply-add (csrc/gemm_bf16x6.h); tiles are computed whole
            issued = {"fwd_t": 1.25, "bwd_wbar": 1.0, "tn_sym": 1.25, "tn_gt": 1.0}[dom] * 6.0 * ach
            emu = {"mfma_dtype": "bf16", "products_per_f32_mac": 6, "issued_tflops": issued, "issued_peak": PEAK_BF16_MFMA_TFLOPS,
                   "issued_frac": issued / PEAK_BF16_MFMA_TFLOPS,
                   "note": "achieved/peak above are algorithmic f32 flops over the f32 MFMA dense peak (the dtype of the path); "
                           "the kernel issues them as 6 bf16 MFMA products each, issued_* is the bf16 matrix-pipe view"}
        kname = {"fwd_t": "fwd_t_bf16x6_kernel", "bwd_wbar": "bwd_wbar_bf16x6_kernel", "tn_sym": "gemm_tn_bf16x6_kernel<A_k>",
                 "tn_gt": "gemm_tn_bf16x6_kernel<GT>"}[dom] if emulated else (
            f"gemm_nt<{dom}>" if not dom.startswith("tn") else f"gemm_tn<{dom}>")
        knm_bytes = n_loc * M * esz + n_loc * D * esz + M * D * esz
        knm_t = t_knm["ms"] / max(t_knm["count"], 1) * 1e-3
        knm_gbs = knm_bytes / knm_t / 1e9 if knm_t > 0 else 0.0
        survey_flops = 3 * (2.0 * N * M * M * K) + 3 * (2.0 * N * M * M) + 2 * (2.0 * N * M * K) + 2 * (2.0 * N * K * args.vocab)
        out = {
            "metric": "ELBO steps/sec (+ achieved HBM GB/s on K_nm) at N=1e6,M=512,K=10,V=50",
            "value": args.steps / dt, "unit": "steps/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": f"configs[3] at {world} GPU(s): N={N} synthetic 2-D lattice ({W}x{H}), V={args.vocab}, K={K}, "
                                   f"M={M} ({'x'.join(map(str, args.n_points))} grid inducing points), {args.kernel} kernel, "
                                   f"Adam lr=1e-3, Trace_ELBO, 1 particle, jitter={args.jitter}, observations sharded over ranks",
                       "N": N, "M": M, "K": K, "V": args.vocab, "D": D, "rows_per_rank": n_loc,
                       "jitter_level": eng.last_jitter_level, "stores_T": eng.stores_t, "mfma_mode": eng.mfma_mode},
            "roofline": {"bound": "mfma", "kernel": kname,
                         "achieved": ach, "peak": peak, "unit": "TFLOP/s", "frac": ach / peak, "traffic": None,
                         "flops_per_launch": flops[dom], "avg_ms": ms[dom], **({"emulation": emu} if emulated else {})},
            "roofline_knm": {"bound": "hbm", "kernel": "knm_kernel", "achieved": knm_gbs, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                             "frac": knm_gbs / PEAK_HBM_GBS, "traffic": None, "bytes_per_launch": knm_bytes,
                             "avg_ms": knm_t * 1e3},
            "step_mfma_frac_survey_flops": survey_flops / world / (dt / args.steps) / (peak * 1e12),
            "kernel_ms_per_step": per_step,
            "final_loss": loss, "perplexity": perplexity,
        }
        if world == 1 and args.cpu_baseline_n > 0:
            out["cpu_baseline"] = cpu_baseline(args, M)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

#!/usr/bin/env python
"""Benchmark of the GDRF SVI ELBO hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by torch.distributed.run, one rank per GPU over RCCL)

A "step" is one full ``SVI.step(xs, ws)`` of ``SparseMultinomialGDRF`` (jitter-retry Cholesky,
forward, backward, all-reduce, epilogue, Adam, loss read-back) on the BASELINE.json headline
workload: N = 1e6 synthetic 2-D observations, V = 50 taxa, K = 10 topics, M = 512 inducing points
(32 x 16 grid), RBF kernel, fp32.  With N GPUs the 1e6 observations are sharded over the ranks
(strong scaling: the total work is fixed).  Inputs are resident in HBM before the timed region.

Rank 0 prints ONE JSON line.  ``value`` comes from a timed region that runs with the library's per-kernel event timing OFF;
per-kernel times are collected in a second, untimed pass of the same steps.  ``roofline`` is the dominant kernel of the
step, priced on the matrix pipe it EXECUTES on (the f32 contractions are issued as split fp16 / bf16 products:
``achieved`` counts the issued MFMA flops, ``peak`` is that pipe's dense peak, the f32-equivalent view is an extra key);
``roofline.traffic`` is that kernel's HBM bytes per launch from the rocprofv3 PMC passes of the same command, read from
``profiles/<round>/pmc_hbm.json`` (tools/profile_round.sh).  ``roofline_knm`` is the K_nm kernel the metric names
(HBM-bound): the standalone f32 launch AND the solve-precision launch the step itself runs.  ``cpu_baseline`` times the
reference-shaped torch-CPU oracle (oracle/gdrf_oracle.py) on bounded samples of the same workload.
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
PEAK_F64_MFMA_TFLOPS = 78.6      # v_mfma_f64_16x16x4_f64: half the f32 MFMA rate
PEAK_16BIT_MFMA_TFLOPS = 2500.0  # MI355X_MICROARCH.md: bf16 / f16 dense peak (16x the f32 MFMA rate)
PEAK_HBM_GBS = 8000.0            # MI355X_MICROARCH.md: HBM3E spec
PROFILE_ROUND = "r03"            # profiles/<round>/pmc_hbm.json: HBM bytes per launch from the PMC passes of this command
SPLIT_PRODUCTS = {"bf16x6": 6, "f16x3": 3}     # MFMA products issued per f32 multiply-add (csrc/gemm_split.h)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--rows", dest="n", type=int, default=1_000_000,
                    help="global number of observations (lattice side^2); --rows is the spelling to use under torch.distributed.run, whose own parser claims --n")
    ap.add_argument("--n-points", type=int, nargs="+", default=[32, 16])
    ap.add_argument("--topics", type=int, default=10)
    ap.add_argument("--vocab", type=int, default=50)
    ap.add_argument("--kernel", default="rbf")
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--jitter", type=float, default=1e-6)
    ap.add_argument("--cpu-baseline-n", type=int, nargs="+", default=[25000, 50000, 100000],
                    help="rows of the CPU-baseline samples (0 = skip); BASELINE.md section 3: three sizes, linearity checked")
    ap.add_argument("--cpu-baseline-steps", type=int, default=5)
    ap.add_argument("--kernel-pass-steps", type=int, default=5, help="steps of the second (untimed) pass that collects per-kernel times")
    ap.add_argument("--knm-iters", type=int, default=50)
    ap.add_argument("--seed", type=int, default=777)
    ap.add_argument("--mfma-mode", default="auto", choices=["auto", "f32", "bf16x6", "f16x3"],
                    help="arithmetic of the f32 GEMM-shaped contractions: split emulation on the 16-bit matrix pipe (f16x3: two "
                         "fp16 pieces, 3 products, block-scaled - the default for float32; bf16x6: three bf16 pieces, 6 products), "
                         "or native f32 MFMA")
    return ap.parse_args()


def csrc_sha():
    """Fingerprint of the kernel sources (as tools/pmc_to_json.py writes it into pmc_hbm.json): PMC traffic collected on other sources
    is not reported."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "gdrf_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


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
    return int(os.environ.get("GDRF_BENCH_CPU_THREADS", n))


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(args, M_points):
    """Reference-shaped oracle (2x conditional, materialised W.S, autograd, per-parameter Adam), fp32, every host thread
    this process may use, on bounded samples N in args.cpu_baseline_n of the same workload (same M, K, V, kernel); the
    headline figure is the linear-in-N extrapolation from the largest sample, the smaller ones state how linear it is."""
    from oracle.gdrf_oracle import RefShapedGDRF
    from gdrf_amd.data import synth_circles
    cores = host_cores()
    torch.set_num_threads(cores)
    per = []
    for ns in sorted(args.cpu_baseline_n):
        W, H = lattice_shape(ns)
        xs, ws, _ = synth_circles(W, H, args.vocab, args.topics, seed=args.seed)
        m = RefShapedGDRF(xs, ws, kind=args.kernel, K=args.topics, n_points=tuple(args.n_points), dtype=torch.float32,
                          jitter=args.jitter, maxjitter=15, optimizer="adam", lr=1e-3)
        g = torch.Generator().manual_seed(1)
        times = []
        for i in range(1 + args.cpu_baseline_steps):
            eps = torch.randn(args.topics, m.N, generator=g)
            t0 = time.perf_counter()
            m.step(eps, n_global=args.n)
            times.append(time.perf_counter() - t0)
        per.append({"N": m.N, "sec_per_step": float(np.median(times[1:])), "us_per_row": float(np.median(times[1:])) / m.N * 1e6})
        del m
    big = per[-1]
    lin = max(p["us_per_row"] for p in per) / min(p["us_per_row"] for p in per)
    return {
        "value": 1.0 / (big["us_per_row"] * 1e-6 * args.n), "unit": "steps/s", "cores": cores, "kind": "port",
        "cpu_model": cpu_model(), "os_cpu_count": os.cpu_count(),
        "sample": f"reference-shaped torch-CPU oracle, fp32, {cores} threads, N in {[p['N'] for p in per]} rows of the same workload "
                  f"(M={M_points}, K={args.topics}, V={args.vocab}), median of {args.cpu_baseline_steps} steps after 1 warm-up each; value = "
                  f"linear-in-N extrapolation of the largest sample to N={args.n} (the reference shape needs > 100 GB of host RAM "
                  f"at N=1e6); time per row varies {lin:.2f}x over the samples",
        "samples": per, "linearity_max_over_min": lin,
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
    # rehearsal knobs (never set by the driver): GDRF_BENCH_ONE_GPU=1 puts every rank on cuda:0 and GDRF_BENCH_BACKEND=gloo replaces RCCL, so that
    # the N > 1 code path of this file can be run on a one-GPU box (RCCL refuses two ranks on one device)
    if os.environ.get("GDRF_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        backend = os.environ.get("GDRF_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

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
    optimizer = Adam({"lr": float(os.environ.get("GDRF_BENCH_LR", 1e-3))})      # the env override is for timing-only kernel ablations
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
    # ---- the timed region: per-kernel event timing OFF
    eng.set_timing(False)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = svi.step(xs=xs, ws=ws, subsample=False)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    # ---- second pass (untimed): per-kernel times from HIP events recorded on the launch streams inside the library
    kp = max(1, args.kernel_pass_steps)
    eng.set_timing(True)
    for _ in range(kp):
        svi.step(xs=xs, ws=ws, subsample=False)
    barrier()
    timing = eng.get_timing()
    eng.set_timing(False)
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
        ssz = 4 if eng.pure_fp32 else 8                      # solve precision (K_nm of the step, W = K_nm L^-T, its backward)
        n_loc = xs.shape[0]
        K = args.topics
        Mp = (M + 31) // 32 * 32
        ms = {k: (v["ms"] / v["count"] if v["count"] else 0.0) for k, v in timing.items()}
        per_step = {k: v["ms"] / kp for k, v in timing.items()}
        flops = {  # useful flops per launch (triangular/symmetric halves skipped, nothing counted twice)
            "fwd_w": 1.0 * n_loc * M * M, "loc": 2.0 * n_loc * M * K, "fwd_t": 1.0 * n_loc * M * M * K,
            "bwd_wbar": (1.0 if eng.stores_t else 2.0) * n_loc * M * M * K,   # triangular T_k S_k^T vs dense W B_k
            "bwd_knm": 1.0 * n_loc * M * M, "tn_sym": 1.0 * n_loc * M * M * K, "tn_gt": 2.0 * n_loc * M * M,
        }
        # whole 128-wide tiles are computed: executed / useful MFMA work of the triangular and symmetric products
        tile_factor = {"fwd_t": 1.25, "tn_sym": 1.25, "fwd_w": 1.25, "bwd_knm": 1.25}
        if eng.mfma_mode == "f16x3":
            tile_factor["tn_sym"] = 1.25         # tn_topics_w2_kernel multiplies whole 128 x 64 tiles (20 of them at M = 512: 160 of 256 blocks for 128 needed)
            tile_factor["fwd_t"] = 1.125         # fwd_t_split_q4_kernel: the triangle at 64-column granularity per wave
        dom = max(flops, key=lambda k: per_step[k])
        dom_t = ms[dom] * 1e-3
        f32_equiv = flops[dom] / dom_t / 1e12 if dom_t > 0 else 0.0
        split = eng.mfma_mode in SPLIT_PRODUCTS and dom in ("fwd_t", "bwd_wbar", "tn_sym", "tn_gt")
        if dom in ("fwd_w", "bwd_knm") and not eng.pure_fp32:
            pipe, peak, nprod = "f64", PEAK_F64_MFMA_TFLOPS, 1
        elif split:
            pipe, peak, nprod = ("bf16" if eng.mfma_mode == "bf16x6" else "f16"), PEAK_16BIT_MFMA_TFLOPS, SPLIT_PRODUCTS[eng.mfma_mode]
        else:
            pipe, peak, nprod = ("f32" if dtype == torch.float32 else "f64"), (PEAK_F32_MFMA_TFLOPS if dtype == torch.float32 else PEAK_F64_MFMA_TFLOPS), 1
        issued = nprod * f32_equiv        # MFMA flops the algorithm needs on this pipe per second (products per multiply-add x useful f32 flops), TFLOP/s
        issued_tiles = tile_factor.get(dom, 1.0) * issued      # including the work above the diagonal that whole tiles execute
        f16 = eng.mfma_mode == "f16x3"
        kname = {"fwd_t": "fwd_t_split_q4_kernel" if f16 else "fwd_t_split_cc_kernel",
                 "bwd_wbar": "bwd_wbar_f16_k64_kernel" if f16 else "bwd_wbar_split_kernel",
                 "tn_sym": "tn_topics_w2_kernel" if f16 else "gemm_tn_split_kernel<A_k>",
                 "tn_gt": "gemm_tn_split_kernel<GT>"}[dom] if split else (
            f"gemm_nt<{dom}>" if not dom.startswith("tn") else f"gemm_tn<{dom}>")
        # HBM bytes per launch of the dominant kernel from the PMC passes of this command (tools/profile_round.sh -> pmc_hbm.json)
        traffic, traffic_note = None, "no profiles/%s/pmc_hbm.json" % PROFILE_ROUND
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", PROFILE_ROUND, "pmc_hbm.json")))
            ent = pm["kernels"].get(dom)
            if pm.get("csrc_sha") != csrc_sha():
                traffic_note = "profiles/%s/pmc_hbm.json was collected on other kernel sources (csrc_sha differs): stale, not reported" % PROFILE_ROUND
            elif ent and pm.get("mfma_mode") == eng.mfma_mode and pm.get("N") == N and world == 1:
                traffic = ent["traffic_bytes"]
                traffic_note = pm.get("note", "")
            else:
                traffic_note = "pmc_hbm.json was collected for another configuration"
        except Exception:
            pass
        knm_bytes = n_loc * M * esz + n_loc * D * esz + M * D * esz
        knm_t = t_knm["ms"] / max(t_knm["count"], 1) * 1e-3
        knm_gbs = knm_bytes / knm_t / 1e9 if knm_t > 0 else 0.0
        knm_step_bytes = n_loc * Mp * ssz + n_loc * D * esz + M * D * ssz      # the step's own launch: solve precision, padded rows
        knm_step_t = ms["k_nm"] * 1e-3
        knm_step_gbs = knm_step_bytes / knm_step_t / 1e9 if knm_step_t > 0 else 0.0
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
            "roofline": {"bound": "mfma", "kernel": kname, "pipe": pipe,
                         "achieved": issued, "peak": peak, "unit": "TFLOP/s", "frac": issued / peak, "traffic": traffic,
                         "traffic_note": traffic_note,
                         "algorithmic_flops_per_launch": flops[dom], "issued_flops_per_launch": tile_factor.get(dom, 1.0) * nprod * flops[dom],
                         "products_per_f32_mac": nprod, "tile_granularity_factor": tile_factor.get(dom, 1.0),
                         "frac_including_whole_tile_overcompute": issued_tiles / peak, "avg_ms": ms[dom],
                         "f32_equivalent_tflops": f32_equiv, "f32_mfma_peak": PEAK_F32_MFMA_TFLOPS,
                         "note": "achieved = USEFUL MFMA flops per second on the pipe the kernel runs on (algorithmic f32 flops x products per "
                                 "multiply-add; work above the diagonal that whole tiles execute is NOT counted - frac_including_whole_tile_overcompute "
                                 "has it); f32_equivalent_tflops = algorithmic f32 flops / time"},
            "roofline_knm": {"bound": "hbm", "kernel": "knm_kernel<f32> (standalone, the metric's K_nm kernel)", "achieved": knm_gbs,
                             "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": knm_gbs / PEAK_HBM_GBS, "traffic": knm_bytes,
                             "traffic_note": "WRITE_SIZE of this kernel equals its algorithmic bytes (profiles/%s/pmc_hbm.json, k_nm_f32)" % PROFILE_ROUND,
                             "bytes_per_launch": knm_bytes, "avg_ms": knm_t * 1e3,
                             "in_step": {"kernel": ("knm_rbf_f64_kernel" if (ssz == 8 and args.kernel == "rbf") else f"knm_kernel<f{8 * ssz}>")
                                                   + " (the launch inside the step: solve precision, feeds W = K_nm L^-T)",
                                         "achieved": knm_step_gbs, "peak": PEAK_HBM_GBS, "frac": knm_step_gbs / PEAK_HBM_GBS,
                                         "bytes_per_launch": knm_step_bytes, "avg_ms": ms["k_nm"]}},
            "step_f32_equivalent_tflops": survey_flops / world / (dt / args.steps) / 1e12,
            "kernel_ms_per_step": per_step, "kernel_pass_steps": kp,
            "final_loss": loss, "perplexity": perplexity,
        }
        if world == 1 and args.cpu_baseline_n and min(args.cpu_baseline_n) > 0:
            out["cpu_baseline"] = cpu_baseline(args, M)
            out["gpu_over_cpu"] = out["value"] / out["cpu_baseline"]["value"]
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Host-side driver of libgdrf_hip: owns the device context, the flat unconstrained-parameter /
gradient / optimizer-state buffers (PyTorch-ROCm tensors used as storage only) and sequences one
SVI step:  factorize (jitter retry) -> step_local -> [RCCL all-reduce] -> step_finish -> adam.

Reference behaviour mirrored here (paths under /root/reference):
  jittercholesky retry schedule        gdrf/models/utils.py:27-40
  SVI.step / 1/N scaling               gdrf/train_script.py:365-371,467
  parameter initialisation             gdrf/models/sparse_gdrf.py:96-122, abstract_gdrf.py:57-84
"""
from __future__ import annotations

import ctypes as C
import os
import math
from typing import Dict, Optional

import torch

from . import _lib

KERNEL_IDS = {"rbf": 0, "matern52": 1, "matern32": 2, "exponential": 3, "rationalquadratic": 4}
OPT_MODES = {"adam": 0, "adamw": 1, "clippedadam": 2}

_WS_IDS = dict(W=0, Wbar=1, q=2, loc=3, tt=4, vbar=5, locbar=6, asum=7, Kuu=8, L=9, Linv=10, S=11, B=12, phi=13,
               mu=14, LinvT=15, ST=16, Knm=17)


def _stream_ptr(device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


class Engine:
    """One device context for fixed (n_cap, M, K, V, D, dtype, kernel)."""

    def __init__(self, n_cap: int, M: int, K: int, V: int, D: int, *, dtype=torch.float32, kernel: str = "rbf",
                 device="cuda:0", jitter: float = 1e-8, maxjitter: int = 15, process_group="auto", pure_fp32: bool = False,
                 store_t="auto", mfma_mode: str = "auto", learn_inducing: bool = False, whiten: bool = True,
                 hyper_backward: str = "auto", allreduce_fn=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.GdrfHipError("gdrf_amd needs a HIP device (torch.cuda.is_available() is False); there is no CPU path")
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.GdrfHipError(f"gdrf_amd runs on a HIP device only, got device={device!r}")
        if dtype not in (torch.float32, torch.float64):
            raise ValueError("dtype must be torch.float32 or torch.float64")
        self.dtype = dtype
        self.n_cap, self.M, self.K, self.V, self.D = int(n_cap), int(M), int(K), int(V), int(D)
        self.kernel = kernel
        self.jitter, self.maxjitter = float(jitter), int(maxjitter)
        self.pg = process_group          # "auto": default group when torch.distributed is initialised; None: never reduce
        dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.device = torch.device("cuda", dev_index)
        torch.cuda.set_device(dev_index)
        ctx = C.c_void_p()
        _lib.check(self.lib.gdrf_ctx_create_ex(C.byref(ctx), dev_index, self.n_cap, self.M, self.K, self.V, self.D,
                                               (2 if pure_fp32 else 0) if dtype == torch.float32 else 1, KERNEL_IDS[kernel],
                                               {"auto": 2, True: 1, False: 0}[store_t]), "gdrf_ctx_create_ex")
        self.pure_fp32 = bool(pure_fp32) and dtype == torch.float32
        self.ctx = ctx
        self.stores_t = bool(self.lib.gdrf_stores_t(self.ctx))
        # arithmetic of the f32 GEMM-shaped contractions (csrc/gemm_split.h): "f16x3" = two block-scaled fp16 pieces per operand,
        # 3 products; "bf16x6" = three bf16 pieces, 6 products (both: f32 accumulation on the 16-bit matrix path, error held to
        # the native f32 MFMA kernels'); "f32" = native f32 MFMA.  "auto" = f16x3 wherever it applies (float32 arrays with the
        # f64 solve - which bounds |W| by sqrt(variance) -, dense Wbar), bf16x6 for the all-fp32 arithmetic, else f32.
        if mfma_mode not in ("auto", "f32", "bf16x6", "f16x3"):
            raise ValueError("mfma_mode must be 'auto', 'f32', 'bf16x6' or 'f16x3'")
        if mfma_mode == "auto":
            mfma_mode = "f32" if (dtype != torch.float32 or self.stores_t) else ("bf16x6" if self.pure_fp32 else "f16x3")
        self.mfma_mode = mfma_mode
        if mfma_mode != "f32":
            _lib.check(self.lib.gdrf_set_mfma_mode(self.ctx, {"bf16x6": 1, "f16x3": 2}[mfma_mode]), "gdrf_set_mfma_mode")
        # K_nm parts of the kernel hyper-parameter gradients: "f64" (= "auto") = Kbar = Wbar L^-1 on the f64 matrix pipe; "tn" = Hd = dK^T Wbar
        # on the split TN kernel + an M x M contraction with L^-1 in double (csrc/hyper_tn.h; no f64 backward GEMM: faster, but d / d log
        # lengthscale then carries float32-level rounding through the cancelling contraction: 3.6e-4 instead of 1e-7 at the headline grid)
        if hyper_backward not in ("auto", "tn", "f64"):
            raise ValueError("hyper_backward must be 'auto', 'tn' or 'f64'")
        _lib.check(self.lib.gdrf_set_hyper_backward(self.ctx, 1 if hyper_backward == "tn" else 0), "gdrf_set_hyper_backward")
        self._hyper_backward_request = hyper_backward
        # a caller-owned collective behind the C ABI (gdrf_set_allreduce): fn(buf_ptr, count, is_double, stream_ptr) -> 0 sums the flat payload in
        # place over the caller's ranks (e.g. a ctypes wrapper of ncclAllReduce on its RCCL communicator); None = torch.distributed (default)
        self._allreduce_cb = None
        if allreduce_fn is not None:
            self.set_allreduce(allreduce_fn)
        lay = (C.c_int64 * 7)()
        _lib.check(self.lib.gdrf_param_layout(self.ctx, lay), "gdrf_param_layout")
        zl = (C.c_int64 * 2)()
        _lib.check(self.lib.gdrf_inducing_layout(self.ctx, zl), "gdrf_inducing_layout")
        self.layout = dict(log_lengthscale=lay[0], log_variance=lay[1], log_noise=lay[2], log_scale_mixture=3, u_loc=lay[3],
                           phi_unc=lay[4], u_scale_tril_unc=lay[5], inducing_unc=zl[0], total=lay[6])
        # fixed_inducing_points=False of the reference: Z = sigmoid(unconstrained block), refreshed before every evaluation
        self.learn_inducing = bool(learn_inducing)
        if self.learn_inducing:
            _lib.check(self.lib.gdrf_set_learn_inducing(self.ctx, 1), "gdrf_set_learn_inducing")
        self.whiten = bool(whiten)
        if not self.whiten:
            _lib.check(self.lib.gdrf_set_whiten(self.ctx, 0), "gdrf_set_whiten")
        red = (C.c_int64 * 6)()
        _lib.check(self.lib.gdrf_red_layout(self.ctx, red), "gdrf_red_layout")
        self.red_layout = dict(ubar=red[0], phibar=red[1], A=red[2], GT=red[3], total_T=red[4], total_d=red[5])
        z = lambda n, dt: torch.zeros(int(n), dtype=dt, device=self.device)
        self.params = z(lay[6], dtype)
        self.grads = z(lay[6], dtype)
        self.exp_avg = z(lay[6], dtype)
        self.exp_avg_sq = z(lay[6], dtype)
        self.red_T = z(red[4], dtype)
        self.red_d = z(red[5], torch.float64)
        self.out_d = z(8, torch.float64)
        self.Z: Optional[torch.Tensor] = None
        self.opt_step = 0
        self.last_jitter_level = 0
        self._ll_cache = None
        self._guess_level: Optional[int] = None      # jitter level of the previous step: this step starts on it speculatively
        self._probe_stream = torch.cuda.Stream(device=self.device)
        self.speculate = True
        self.prefactorize = os.environ.get("GDRF_PREFACTORIZE", "1") != "0"      # factorise for the next step right behind the optimizer update
        # a caller-supplied link (the reference's `link_function`, abstract_gdrf.py:34-50): a callable on the (K, n) tensor mu returning the
        # (K, n) topic weights; None = the softmax link fused into the row kernel.  Evaluated with torch between three library calls.
        self.link_function = None

    def set_allreduce(self, fn):
        """Register the step's collective behind the C ABI (gdrf_set_allreduce): ``fn(buf_ptr, count, is_double, stream_ptr)`` sums the flat
        payload in place over the caller's ranks and returns 0 / None; None unregisters (back to torch.distributed, or one rank)."""
        if fn is None:
            self._allreduce_cb = None
            _lib.check(self.lib.gdrf_set_allreduce(self.ctx, None, None), "gdrf_set_allreduce")
            return

        def _cb(buf, count, is_double, stream, user, _fn=fn):
            try:
                return int(_fn(buf, count, bool(is_double), stream) or 0)
            except Exception:
                return 1
        self._allreduce_cb = _lib.ALLREDUCE_FN(_cb)              # kept alive with the engine
        _lib.check(self.lib.gdrf_set_allreduce(self.ctx, C.cast(self._allreduce_cb, C.c_void_p), None), "gdrf_set_allreduce")

    @property
    def hyper_backward(self) -> str:
        """The form the next step uses ("tn" needs f16x3, fixed inducing inputs, a kernel other than RationalQuadratic)."""
        return "tn" if self.lib.gdrf_get_hyper_backward(self.ctx) else "f64"

    def __del__(self):
        try:
            if getattr(self, "ctx", None):
                self.lib.gdrf_ctx_destroy(self.ctx)
                self.ctx = None
        except Exception:
            pass

    # ---- parameter views (natural shapes) ---------------------------------------------------------
    def view(self, name: str, buf: Optional[torch.Tensor] = None) -> torch.Tensor:
        buf = self.params if buf is None else buf
        o = self.layout[name]
        K, M, V = self.K, self.M, self.V
        if name in ("log_lengthscale", "log_variance", "log_noise", "log_scale_mixture"):
            return buf[o:o + 1].view(())
        if name == "u_loc":
            return buf[o:o + K * M].view(K, M)
        if name == "phi_unc":
            return buf[o:o + K * V].view(K, V)
        if name == "u_scale_tril_unc":
            return buf[o:o + K * M * M].view(K, M, M)
        if name == "inducing_unc":
            return buf[o:o + M * self.D].view(M, self.D)
        raise KeyError(name)

    PARAM_NAMES = ("log_lengthscale", "log_variance", "log_noise", "u_loc", "phi_unc", "u_scale_tril_unc")

    @property
    def param_names(self):
        """PARAM_NAMES plus the blocks only some configurations learn (RationalQuadratic's scale_mixture, inducing inputs)."""
        extra = (("log_scale_mixture",) if self.kernel == "rationalquadratic" else ()) + \
                (("inducing_unc",) if self.learn_inducing else ())
        return self.PARAM_NAMES + extra

    def named_views(self, buf: Optional[torch.Tensor] = None) -> Dict[str, torch.Tensor]:
        return {n: self.view(n, buf) for n in self.param_names}

    def set_inducing_points(self, Z: torch.Tensor):
        Z = Z.to(device=self.device, dtype=self.dtype).contiguous()
        assert Z.shape == (self.M, self.D), f"inducing points must be ({self.M},{self.D}), got {tuple(Z.shape)}"
        self.Z = Z.clone()
        if self.learn_inducing:
            # transform_to(interval(0, 1)).inv: logit of the value clamped to [tiny, 1 - eps] (torch SigmoidTransform._inverse)
            fi = torch.finfo(self.dtype)
            y = self.Z.clamp(min=fi.tiny, max=1.0 - fi.eps)
            self.view("inducing_unc").copy_(y.log() - (-y).log1p())

    def refresh_inducing(self):
        """Z = sigmoid(unconstrained block) when the inducing inputs are learnable (no-op otherwise)."""
        if self.learn_inducing:
            torch.sigmoid(self.view("inducing_unc"), out=self.Z)

    def set_dirichlet(self, alpha: torch.Tensor):
        a = alpha.detach().to("cpu", torch.float64).contiguous()
        assert a.shape == (self.K, self.V)
        arr = (C.c_double * (self.K * self.V))(*a.flatten().tolist())
        _lib.check(self.lib.gdrf_set_dirichlet(self.ctx, arr), "gdrf_set_dirichlet")

    # ---- workspace access for parity tests ----------------------------------------------------------
    def workspace(self, name: str, n_rows: Optional[int] = None) -> torch.Tensor:
        """Copy of a workspace buffer in its natural (unpadded) shape."""
        ptr, cnt = C.c_void_p(), C.c_int64()
        _lib.check(self.lib.gdrf_ws_ptr(self.ctx, _WS_IDS[name], C.byref(ptr), C.byref(cnt)), "gdrf_ws_ptr")
        esz = self.lib.gdrf_ws_elem_size(self.ctx, _WS_IDS[name])
        flat = torch.empty(cnt.value, dtype=torch.float32 if esz == 4 else torch.float64, device=self.device)
        _lib.check(self.lib.gdrf_ws_copy(self.ctx, _WS_IDS[name], flat.data_ptr(), cnt.value, _stream_ptr(self.device)),
                   "gdrf_ws_copy")
        torch.cuda.synchronize(self.device)
        Mp = (self.M + 31) // 32 * 32
        ldk = (self.n_cap + 3) // 4 * 4
        n = self.n_cap if n_rows is None else n_rows
        if name in ("W", "Wbar", "Knm"):
            return flat.view(self.n_cap, Mp)[:n, :self.M].clone()
        if name in ("q", "asum"):
            return flat[:n].clone()
        if name in ("loc", "tt", "vbar", "locbar", "mu"):
            return flat.view(self.K, ldk)[:, :n].clone()
        if name in ("Kuu", "L", "Linv", "LinvT"):
            return flat.view(Mp, Mp)[:self.M, :self.M].clone()
        if name in ("S", "B", "ST"):
            return flat.view(self.K, Mp, Mp)[:, :self.M, :self.M].clone()
        if name == "phi":
            return flat.view(self.K, self.V).clone()
        raise KeyError(name)

    TIMING_SLOTS = ("probe", "k_nm", "transforms", "fwd_w", "loc", "fwd_t", "elbo_rows", "bwd_wbar", "bwd_knm",
                    "tn_sym", "tn_gt", "slab_reduce", "ubar", "step_finish", "adam", "factorize")

    def set_timing(self, enable: bool):
        _lib.check(self.lib.gdrf_set_timing(self.ctx, 1 if enable else 0), "gdrf_set_timing")

    def get_timing(self) -> Dict[str, Dict[str, float]]:
        n = len(self.TIMING_SLOTS)
        ms, cnt = (C.c_double * n)(), (C.c_int64 * n)()
        _lib.check(self.lib.gdrf_get_timing(self.ctx, ms, cnt, n), "gdrf_get_timing")
        return {name: dict(ms=ms[i], count=cnt[i]) for i, name in enumerate(self.TIMING_SLOTS)}

    # ---- primitives ------------------------------------------------------------------------------
    def _chk_rows(self, xs: torch.Tensor, ws: Optional[torch.Tensor] = None):
        if xs.device != self.device or xs.dtype != self.dtype or not xs.is_contiguous() or xs.dim() != 2 or xs.shape[1] != self.D:
            raise ValueError(f"xs must be a contiguous ({'n'},{self.D}) {self.dtype} tensor on {self.device}")
        if ws is not None:
            if ws.device != self.device or ws.dtype != torch.int32 or not ws.is_contiguous() or ws.shape != (xs.shape[0], self.V):
                raise ValueError(f"ws must be a contiguous (n,{self.V}) int32 tensor on {self.device}")

    def knm(self, xs: torch.Tensor) -> torch.Tensor:
        """K_nm = k(xs, Z) (n, M) row-major: the HBM-roofline kernel."""
        self._chk_rows(xs)
        out = torch.empty(xs.shape[0], self.M, dtype=self.dtype, device=self.device)
        self.knm_into(xs, out)
        return out

    def knm_into(self, xs: torch.Tensor, out: torch.Tensor):
        _lib.check(self.lib.gdrf_knm(self.ctx, xs.data_ptr(), xs.shape[0], self.Z.data_ptr(), self.params.data_ptr(),
                                     out.data_ptr(), out.shape[1], _stream_ptr(self.device)), "gdrf_knm")

    def fill_eps(self, seed: int, step: int, n_offset: int, n: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty(self.K, n, dtype=self.dtype, device=self.device)
        _lib.check(self.lib.gdrf_fill_eps(self.ctx, seed & (2 ** 64 - 1), step & 0xFFFFFFFF, n_offset, n, out.data_ptr(),
                                          _stream_ptr(self.device)), "gdrf_fill_eps")
        return out

    def ll_const_dev(self, ws: torch.Tensor) -> torch.Tensor:
        """Device scalar holding the data-only constant of Multinomial.log_prob for ``ws`` (no host sync).  Cached for the
        IDENTICAL tensor object only (held here, so its storage cannot be handed to another mini-batch) at the same
        ``_version``: a fresh same-shaped mini-batch (train_script.py:461-465) is always recomputed."""
        c = self._ll_cache
        if c is not None and c[0] is ws and c[1] == ws._version:
            return c[2]
        out = torch.empty(1, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.gdrf_ll_const_dev(self.ctx, ws.data_ptr(), ws.shape[0], out.data_ptr(), _stream_ptr(self.device)),
                   "gdrf_ll_const_dev")
        self._ll_cache = (ws, ws._version, out)
        return out

    def ll_const(self, ws: torch.Tensor) -> float:
        return float(self.ll_const_dev(ws).item())

    def jitter_total(self, level: int) -> float:
        return sum(self.jitter * (10 ** n) for n in range(level + 1))

    def factorize(self, force_level: Optional[int] = None) -> int:
        """jittercholesky: smallest level whose cumulative jitter lets the Cholesky factorisation succeed IN THE
        ARRAY PRECISION (every call starts from level 0, as the reference rebuilds K_uu each time; up to 8 levels
        are attempted concurrently per launch), then the factor and its inverse in the solve precision."""
        s = _stream_ptr(self.device)
        failed = C.c_int()
        level = 0 if force_level is None else force_level
        while level < self.maxjitter:
            if force_level is None:
                nlev = min(4 if level == 0 else 8, self.maxjitter - level)
                jit = (C.c_double * nlev)(*[self.jitter_total(level + l) for l in range(nlev)])
                flags = (C.c_int * nlev)()
                _lib.check(self.lib.gdrf_probe(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), jit, nlev, flags, s), "gdrf_probe")
                ok = [l for l in range(nlev) if not flags[l]]
                if not ok:
                    level += nlev
                    continue
                level += ok[0]
            _lib.check(self.lib.gdrf_factorize(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), self.jitter_total(level), s),
                       "gdrf_factorize")
            _lib.check(self.lib.gdrf_chol_failed(self.ctx, C.byref(failed), s), "gdrf_chol_failed")
            if not failed.value:
                self.last_jitter_level = level
                return level
            if force_level is not None:
                break
            level += 1
        raise RuntimeError("reached max jitter, covariance is unstable")

    def loss_and_grads(self, xs, ws, eps, n_global: Optional[int] = None, ll_const: Optional[float] = None,
                       force_level: Optional[int] = None, renyi_alpha: Optional[float] = None, mean: Optional[torch.Tensor] = None,
                       xs_guide: Optional[torch.Tensor] = None, mean_guide: Optional[torch.Tensor] = None):
        """One ELBO evaluation + backward.  Leaves d loss/d unconstrained in self.grads (device) and
        returns nothing host-side; call read_out() for the loss.  ``mean``: the values of the model's mean_function on these
        rows, broadcastable to (K, n) (gdrf/models/sparse_gdrf.py:346,395); None = zero_mean.  ``xs_guide``: the inputs the
        GUIDE's predictive is evaluated at when they differ from the model's (quirk Q3: the reference's guide scales twice,
        sparse_gdrf.py:376-380), with ``mean_guide`` the mean_function values there; None = the same inputs (one evaluation)."""
        self._chk_rows(xs, ws)
        n = xs.shape[0]
        self._set_mean(mean, n)
        self._xs_guide = None
        if xs_guide is not None:
            self._chk_rows(xs_guide)
            if xs_guide.shape != xs.shape:
                raise ValueError("xs_guide must have the shape of xs")
            self._xs_guide = xs_guide
            self._set_mean(mean_guide, n, guide=True)
        if eps.dim() == 2:
            eps = eps.unsqueeze(0)
        P = eps.shape[0]                          # particles (Trace_ELBO num_particles): the estimator is their mean
        if tuple(eps.shape[1:]) != (self.K, n) or eps.dtype != self.dtype or not eps.is_contiguous() or eps.device != self.device:
            raise ValueError(f"eps must be a contiguous ([P,]{self.K},{n}) {self.dtype} tensor on {self.device}")
        s = _stream_ptr(self.device)
        self.refresh_inducing()
        if ll_const is None:
            llc = self.ll_const_dev(ws)
        else:
            llc = torch.full((1,), float(ll_const), dtype=torch.float64, device=self.device)
        ng = float(n if n_global is None else n_global)
        guess = self._guess_level if (force_level is None and self.speculate) else None
        if guess is None:
            self.factorize(force_level)
            self._local_and_finish(xs, ws, eps, P, n, ng, llc, s, renyi_alpha)
        else:
            # Speculate on the previous step's jitter level: the solve-precision factorisation and the whole step go onto
            # the main stream at once, the array-precision probe (which decides the level, as the reference's fp32
            # jittercholesky would) runs beside them on a second stream, and the host only waits for the probe and for the
            # factorisation flag while the GPU is busy with the N-side kernels.  A wrong guess (the level moved, or the
            # f64 factorisation failed where the f32 one passed) redoes the step on the right level; every rank holds the
            # same parameters, so every rank takes the same branch.
            main = torch.cuda.current_stream(self.device)
            self._probe_stream.wait_stream(main)
            ps = self._probe_stream.cuda_stream
            # the probe of the first levels goes out FIRST, without waiting: it runs while the host enqueues the step
            nlev0 = min(4, self.maxjitter)
            jit0 = (C.c_double * nlev0)(*[self.jitter_total(l) for l in range(nlev0)])
            _lib.check(self.lib.gdrf_probe_launch(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), jit0, nlev0, ps), "gdrf_probe_launch")
            # mode 2: a factorisation made ahead of this step (behind the previous optimizer update, see adam()) is reused after a
            # device-side check that its inputs are still the current ones; gdrf_chol_failed reports a mismatch like a failure
            _lib.check(self.lib.gdrf_factorize_mode(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), self.jitter_total(guess), s, 2),
                       "gdrf_factorize_mode")
            fact_done = torch.cuda.Event()
            fact_done.record(main)
            self._local_and_finish(xs, ws, eps, P, n, ng, llc, s, renyi_alpha)
            flags0 = (C.c_int * nlev0)()
            _lib.check(self.lib.gdrf_probe_read(self.ctx, nlev0, flags0, ps), "gdrf_probe_read")
            ok0 = [l for l in range(nlev0) if not flags0[l]]
            level = ok0[0] if ok0 else self._probe_level(ps, start=nlev0)
            failed = C.c_int()
            self._probe_stream.wait_event(fact_done)
            _lib.check(self.lib.gdrf_chol_failed(self.ctx, C.byref(failed), ps), "gdrf_chol_failed")
            if level == guess and not failed.value:
                self.last_jitter_level = level
            else:
                torch.cuda.synchronize(self.device)
                self.factorize(None)
                self._local_and_finish(xs, ws, eps, P, n, ng, llc, s, renyi_alpha)
        self._guess_level = self.last_jitter_level if force_level is None else None

    def _set_mean(self, mean, n: int, guide: bool = False):
        fn = self.lib.gdrf_set_mean_guide if guide else self.lib.gdrf_set_mean
        if mean is None:
            setattr(self, "_mean_g" if guide else "_mean", None)
            _lib.check(fn(self.ctx, None, 0, 0), "gdrf_set_mean")
            return
        mean = torch.as_tensor(mean).detach().to(device=self.device, dtype=self.dtype)
        try:
            mean = mean.expand(self.K, n)                       # (n,), (K, 1), (K, n), scalars: torch broadcasting, as f_loc + mean
        except RuntimeError:
            raise ValueError(f"mean_function returned shape {tuple(mean.shape)}, not broadcastable to ({self.K}, {n})") from None
        setattr(self, "_mean_g" if guide else "_mean", mean)    # keeps the storage alive while the context borrows it
        _lib.check(fn(self.ctx, mean.data_ptr(), mean.stride(0), mean.stride(1)), "gdrf_set_mean")

    def _probe_level(self, stream_ptr: int, start: int = 0) -> int:
        """First cumulative-jitter level >= start whose array-precision Cholesky succeeds (probe only; raises past maxjitter)."""
        level = start
        while level < self.maxjitter:
            nlev = min(4 if level == 0 else 8, self.maxjitter - level)
            jit = (C.c_double * nlev)(*[self.jitter_total(level + l) for l in range(nlev)])
            flags = (C.c_int * nlev)()
            _lib.check(self.lib.gdrf_probe(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), jit, nlev, flags, stream_ptr), "gdrf_probe")
            ok = [l for l in range(nlev) if not flags[l]]
            if ok:
                return level + ok[0]
            level += nlev
        raise RuntimeError("reached max jitter, covariance is unstable")

    def _local_and_finish(self, xs, ws, eps, P, n, ng, llc, s, renyi_alpha=None):
        """The N-side kernels for every particle, the particle combination, the all-reduce and the replicated epilogue.
        Trace_ELBO averages the P payloads (every entry is linear in the per-particle sums).  RenyiELBO(alpha) weights them
        with w_p = softmax_p((1 - alpha) e_p), e_p = the particle-varying part of the scaled ELBO (site + likelihood sums
        over ALL ranks; the terms shared by the particles factor out of the logsumexp), and reports
        -(logsumexp((1 - alpha) e_p) - log P) / (1 - alpha) through the payload's site slot.  All on the device."""
        dist_on = self._distributed()
        if dist_on:
            import torch.distributed as dist
            pg = None if isinstance(self.pg, str) else self.pg
        if renyi_alpha is not None and float(renyi_alpha) == 1.0:
            raise ValueError("RenyiELBO: alpha must differ from 1")
        Ts, ds = [], []
        xg = getattr(self, "_xs_guide", None)
        for p in range(P):
            if self.link_function is not None:
                if xg is not None:
                    raise NotImplementedError("a custom link_function together with a non-unit world's doubly scaled guide (quirk Q3)")
                self._step_local_link(xs, ws, eps[p], n, s)
            elif xg is None:
                _lib.check(self.lib.gdrf_step_local(self.ctx, xs.data_ptr(), ws.data_ptr(), eps[p].data_ptr(), n, self.Z.data_ptr(),
                                                    self.params.data_ptr(), self.red_T.data_ptr(), self.red_d.data_ptr(), s),
                           "gdrf_step_local")
            else:
                _lib.check(self.lib.gdrf_step_local2(self.ctx, xs.data_ptr(), xg.data_ptr(), ws.data_ptr(), eps[p].data_ptr(), n,
                                                     self.Z.data_ptr(), self.params.data_ptr(), self.red_T.data_ptr(),
                                                     self.red_d.data_ptr(), s), "gdrf_step_local2")
            if renyi_alpha is not None:
                Ts.append(self.red_T.clone()); ds.append(self.red_d.clone())
            elif P > 1:                                           # Trace_ELBO: a running sum of the payloads, no per-particle copies
                if p == 0:
                    accT, accd = self.red_T.clone(), self.red_d.clone()
                else:
                    accT.add_(self.red_T); accd.add_(self.red_d)
        if renyi_alpha is not None:
            T_all, d_all = torch.stack(Ts), torch.stack(ds)
            e = d_all[:, 0] + d_all[:, 1]                         # this rank's site + likelihood sums per particle
            if dist_on:
                dist.all_reduce(e, group=pg)
            logw = (1.0 - float(renyi_alpha)) * e / ng
            w = torch.softmax(logw, 0)
            self.red_T.copy_((w.to(T_all.dtype)[:, None] * T_all).sum(0))
            self.red_d.copy_((w[:, None] * d_all).sum(0))
            bound = (torch.logsumexp(logw, 0) - math.log(P)) / (1.0 - float(renyi_alpha))      # scaled by 1/N already
            # step_finish forms loss = -(red_d[0] + red_d[1] + constants) / N from the REDUCED payload: one rank carries it
            first = (not dist_on) or dist.get_rank(pg) == 0
            self.red_d[0] = bound * ng if first else 0.0
            self.red_d[1] = 0.0
        elif P > 1:
            self.red_T.copy_(accT.div_(P))
            self.red_d.copy_(accd.div_(P))
        self.red_d[7:8].copy_(llc)                   # the data constant is a sum over observations too; stays on the device
        if self._allreduce_cb is not None:           # the caller's collective, behind the C ABI (pack, its sum over the ranks, unpack)
            _lib.check(self.lib.gdrf_payload_allreduce(self.ctx, self.red_T.data_ptr(), self.red_d.data_ptr(), s), "gdrf_payload_allreduce")
        elif dist_on:
            # ONE collective per step: the doubles of red_d ride in the tail of the flat payload (exactly in float64 contexts; as four
            # float pieces each in float32 ones: exact over <= 8 ranks for the loss sums, whose per-rank values have similar magnitude,
            # and to 2^-24 of the largest summand for entries that differ by orders of magnitude between ranks - csrc/kernels_n.h)
            _lib.check(self.lib.gdrf_payload_pack(self.ctx, self.red_T.data_ptr(), self.red_d.data_ptr(), s), "gdrf_payload_pack")
            dist.all_reduce(self.red_T, group=pg)    # RCCL over xGMI (backend "nccl" on ROCm)
            _lib.check(self.lib.gdrf_payload_unpack(self.ctx, self.red_T.data_ptr(), self.red_d.data_ptr(), s), "gdrf_payload_unpack")
        self._finish(ng, None)

    def _step_local_link(self, xs, ws, eps_p, n: int, s: int):
        """gdrf_step_local with the link and its Jacobian evaluated here: theta = link(mu) and mubar = J^T thetabar by autograd
        (sparse_gdrf.py:361: `topic_probs = self._link_function(mu).transpose(-2, -1)`)."""
        args = (self.ctx, xs.data_ptr(), ws.data_ptr(), eps_p.data_ptr(), n, self.Z.data_ptr(), self.params.data_ptr(),
                self.red_T.data_ptr(), self.red_d.data_ptr(), s)
        _lib.check(self.lib.gdrf_step_local_link(*args, 0, None, 0), "gdrf_step_local_link(0)")
        mu = self.workspace("mu", n).requires_grad_(True)                      # (K, n)
        with torch.enable_grad():
            theta = self.link_function(mu)
        if tuple(theta.shape) != (self.K, n):
            raise ValueError(f"link_function returned shape {tuple(theta.shape)}, expected ({self.K}, {n}) like its argument")
        th = theta.detach().to(self.dtype).contiguous()
        _lib.check(self.lib.gdrf_step_local_link(*args, 1, th.data_ptr(), n), "gdrf_step_local_link(1)")
        thbar = self.workspace("locbar", n)
        if theta.requires_grad:
            (mubar,) = torch.autograd.grad(theta, mu, grad_outputs=thbar.to(theta.dtype))
        else:                                                                  # a link that ignores mu
            mubar = torch.zeros_like(mu)
        mubar = mubar.detach().to(self.dtype).contiguous()
        _lib.check(self.lib.gdrf_step_local_link(*args, 2, mubar.data_ptr(), n), "gdrf_step_local_link(2)")

    def _distributed(self) -> bool:
        if self.pg is None:
            return False
        import torch.distributed as dist
        if not (dist.is_available() and dist.is_initialized()):
            return False
        pg = None if isinstance(self.pg, str) else self.pg
        return dist.get_world_size(pg) > 1

    def _finish(self, n_global: float, ll_const: Optional[float]):
        s = _stream_ptr(self.device)
        if ll_const is None:                         # the reduced copy travels in red_d[7]; the kernel reads it there (no host sync)
            ll_const = float("nan")
        _lib.check(self.lib.gdrf_step_finish(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), self.red_T.data_ptr(),
                                             self.red_d.data_ptr(), n_global, ll_const, self.grads.data_ptr(),
                                             self.out_d.data_ptr(), s), "gdrf_step_finish")

    def adam(self, mode: str, lr: float, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, clip=10.0):
        self.opt_step += 1
        _lib.check(self.lib.gdrf_adam(self.ctx, OPT_MODES[mode], self.params.data_ptr(), self.grads.data_ptr(),
                                      self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.opt_step, lr, betas[0], betas[1],
                                      eps, weight_decay, clip, _stream_ptr(self.device)), "gdrf_adam")
        # The next step's factorisation depends only on what this update just wrote (kernel hyper-parameters, inducing inputs): start it now,
        # on the guessed jitter level, so that its serial chain runs while the host reads the loss and enqueues the step.  The step checks
        # on the device that the inputs are still the same (anything may write the parameters in between) and redoes itself otherwise.
        if self.speculate and self.prefactorize and self._guess_level is not None:
            self.refresh_inducing()
            _lib.check(self.lib.gdrf_factorize_mode(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), self.jitter_total(self._guess_level),
                                                    _stream_ptr(self.device), 1), "gdrf_factorize_mode")

    def read_out(self) -> Dict[str, float]:
        o = self.out_d.cpu().tolist()          # synchronises
        return dict(loss=o[0], chol_failed=o[1], site=o[2], loglik=o[3], lp_phi=o[4])

    def predict(self, xs: torch.Tensor, mode: int, ws: Optional[torch.Tensor] = None):
        """Predictive path (gdrf_predict): mode 0 f_loc (K, n), 1 topic_probs (n, K), 2 word_probs (n, V), 3 {sum w log p, sum w},
        4 (f_loc, f_var) as (2, K, n).  Like the step, it starts on the previous jitter level (and on the factorisation adam() queued
        ahead, if its inputs still match) while the array-precision probe that decides the level runs on the second stream; a wrong
        guess redoes the evaluation on the right level."""
        self._chk_rows(xs, ws)
        n = xs.shape[0]
        self.refresh_inducing()
        if mode == 4 and n > self.n_cap:
            raise ValueError(f"predict(mode=4) needs n <= n_cap ({n} > {self.n_cap})")

        def run():
            out = None
            if mode == 0:
                out = torch.empty(self.K, n, dtype=self.dtype, device=self.device)
            elif mode == 1:
                out = torch.empty(n, self.K, dtype=self.dtype, device=self.device)
            elif mode == 2:
                out = torch.empty(n, self.V, dtype=self.dtype, device=self.device)
            elif mode == 4:          # (f_loc, f_var) of gp.util.conditional(full_cov=False): sparse_gdrf.py:277-319
                out = torch.empty(2, self.K, n, dtype=self.dtype, device=self.device)
            _lib.check(self.lib.gdrf_predict(self.ctx, xs.data_ptr(), n, self.Z.data_ptr(), self.params.data_ptr(),
                                             ws.data_ptr() if ws is not None else None, mode,
                                             out.data_ptr() if out is not None else None, self.out_d.data_ptr(),
                                             _stream_ptr(self.device)), "gdrf_predict")
            return self.out_d[:2].clone() if mode == 3 else out

        guess = self._guess_level if self.speculate else None
        if guess is None:
            self.factorize()
            return run()
        main = torch.cuda.current_stream(self.device)
        self._probe_stream.wait_stream(main)
        ps = self._probe_stream.cuda_stream
        nlev0 = min(4, self.maxjitter)
        jit0 = (C.c_double * nlev0)(*[self.jitter_total(l) for l in range(nlev0)])
        _lib.check(self.lib.gdrf_probe_launch(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), jit0, nlev0, ps), "gdrf_probe_launch")
        _lib.check(self.lib.gdrf_factorize_mode(self.ctx, self.Z.data_ptr(), self.params.data_ptr(), self.jitter_total(guess),
                                                main.cuda_stream, 2), "gdrf_factorize_mode")
        fact_done = torch.cuda.Event()
        fact_done.record(main)
        out = run()
        flags0 = (C.c_int * nlev0)()
        _lib.check(self.lib.gdrf_probe_read(self.ctx, nlev0, flags0, ps), "gdrf_probe_read")
        ok0 = [l for l in range(nlev0) if not flags0[l]]
        level = ok0[0] if ok0 else self._probe_level(ps, start=nlev0)
        failed = C.c_int()
        self._probe_stream.wait_event(fact_done)
        _lib.check(self.lib.gdrf_chol_failed(self.ctx, C.byref(failed), ps), "gdrf_chol_failed")
        if level == guess and not failed.value:
            self.last_jitter_level = level
            return out
        torch.cuda.synchronize(self.device)
        self.factorize(None)
        self._guess_level = self.last_jitter_level
        return run()

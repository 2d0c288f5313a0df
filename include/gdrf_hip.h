/* libgdrf_hip: C ABI of the MI355X-native GDRF SVI ELBO hot path.
 *
 * The reference (san-soucie/gdrf v0.1.3) is pure Python and has no FFI; the seam this library
 * sits behind is the duck-typed Python surface of SURVEY.md 8(b).  Each entry point below names
 * the reference code (paths under /root/reference) whose work it replaces; the Python mirror in
 * gdrf_amd/ binds them with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions: every function returns 0 on success, < 0 on a HIP/argument error (message from
 * gdrf_last_error()); pointers named *_dev are borrowed device pointers that the caller keeps
 * alive until the stream has been synchronised; `stream` is a hipStream_t passed as void*;
 * no exceptions cross the ABI; one host thread per context.  kernel_id: 0 RBF, 1 Matern52, 2 Matern32, 3 Exponential,
 * 4 RationalQuadratic.
 * dtype fixes the element type of every "void*" real array below:
 *   GDRF_F32 (0)      float arrays.  The K-fold contractions are f32 GEMMs evaluated on the matrix cores in the arithmetic
 *                     gdrf_set_mfma_mode() selects (native f32 MFMA, or split operands on the 16-bit matrix path with f32
 *                     accumulation and f32-level error); the ill-conditioned pieces (K_uu, its Cholesky factor and inverse,
 *                     the solve W = K_nm L^-T, its backward and the M x M epilogue) run in f64, because the fp32 solve
 *                     cancels terms |L^-1||k| >> |w| (DESIGN.md "precision").
 *   GDRF_F64 (1)      double arrays, everything f64.
 *   GDRF_F32_PURE (2) float arrays, everything f32 (the reference's literal .float() arithmetic; for A/B runs).
 */
#ifndef GDRF_HIP_H
#define GDRF_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct gdrf_ctx gdrf_ctx;

enum { GDRF_F32 = 0, GDRF_F64 = 1, GDRF_F32_PURE = 2 };
enum { GDRF_RBF = 0, GDRF_MATERN52 = 1, GDRF_MATERN32 = 2, GDRF_EXPONENTIAL = 3, GDRF_RATIONALQUADRATIC = 4 };
enum { GDRF_ADAM = 0, GDRF_ADAMW = 1, GDRF_CLIPPED_ADAM = 2 };
enum { GDRF_PRED_LOC = 0, GDRF_PRED_TOPIC_PROBS = 1, GDRF_PRED_WORD_PROBS = 2, GDRF_PRED_PERPLEXITY = 3, GDRF_PRED_LOC_VAR = 4 };

const char* gdrf_last_error(void);
int gdrf_version(void);

/* Workspaces for at most n_cap local observations.  Replaces the tensors pyro/autograd allocate
 * per step inside SVI.step (gdrf/train_script.py:365-371,467). */
int gdrf_ctx_create(gdrf_ctx** out, int device, int64_t n_cap, int M, int K, int V, int D, int dtype, int kernel_id);
/* As above with an explicit choice for the backward of the variance term: GDRF_STORE_T_ON keeps T_k = W S_k
 * (K*n_cap*M elements) from the forward so that Wbar is the triangular product sum_k diag(2 vbar_k) T_k S_k^T;
 * GDRF_STORE_T_OFF (what gdrf_ctx_create uses) takes the dense form sum_k diag(2 vbar_k) W (S_k S_k^T): twice the
 * MFMA flops, no extra memory, but its A operand is reused across the K topics from LDS -- measured faster on MI355X
 * (43 vs 47 ms at the headline size).  GDRF_STORE_T_AUTO currently resolves to OFF. */
enum { GDRF_STORE_T_OFF = 0, GDRF_STORE_T_ON = 1, GDRF_STORE_T_AUTO = 2 };
int gdrf_ctx_create_ex(gdrf_ctx** out, int device, int64_t n_cap, int M, int K, int V, int D, int dtype, int kernel_id, int store_t);
int gdrf_stores_t(const gdrf_ctx* ctx);
/* Arithmetic of the four f32 GEMM-shaped contractions of the step (tt = |S_k^T w|^2, Wbar, A_k = W^T diag(vbar_k) W,
 * G^T = W^T Wbar) in float contexts (csrc/gemm_split.h):
 *   0  v_mfma_f32_16x16x4_f32 (native f32 MFMA);
 *   1  "bf16x6": each f32 operand as 3 bf16 pieces, the 6 cross products of weight >= 2^-16 on v_mfma_f32_16x16x32_bf16,
 *      f32 accumulation: 16/6 of the f32 MFMA rate;
 *   2  "f16x3": each f32 operand, times one power-of-two block scale that puts its largest magnitude below 2^15, as 2 fp16
 *      pieces (22-23 significand bits), the 3 cross products hh + hl + lh on v_mfma_f32_16x16x32_f16, f32 accumulation, the
 *      scales undone in the epilogues: 16/3 of the f32 MFMA rate.
 * Modes 1 and 2 are held to the error of mode 0 against fp64 products of the same inputs (tests/test_gpu_parity.py).  They
 * need float arrays and the dense Wbar form (GDRF_STORE_T_OFF); a fresh context is in mode 0, gdrf_amd.Engine selects 2 for
 * float32 with the f64 solve by default. */
int gdrf_set_mfma_mode(gdrf_ctx* ctx, int mode);
int gdrf_get_mfma_mode(const gdrf_ctx* ctx);
/* How the K_nm parts of the kernel hyper-parameter gradients (sum Kbar_nm o K_nm, sum Kbar_nm o dK_nm/dlog ls) are formed:
 *   0  (default) Kbar_nm = Wbar L^-1 tile by tile in the solve precision (gemm_nt<BwdKnmProb>, f64 matrix pipe), never stored;
 *   1  no backward solve: sum Kbar o K = sum Wbar o W, and sum Kbar o dK = sum_{i>=j} Linv[i][j] Hd[j][i] with Hd = dK^T Wbar, an M x M
 *      contraction over the rows on the split-fp16 TN kernel (csrc/hyper_tn.h), its contraction with L^-1 in double.  Applies to float arrays
 *      in mode f16x3 with fixed inducing inputs and a kernel other than RationalQuadratic; every other configuration runs form 0.  Faster
 *      (no f64 GEMM), but Hd carries float32-level rounding INTO the cancelling contraction with L^-1: d loss / d log lengthscale is then
 *      accurate to ~4e-4 instead of ~1e-7 at the headline conditioning (tests/test_gpu_round3.py), hence opt-in.
 * gdrf_get_hyper_backward reports the form the next step will actually use. */
int gdrf_set_hyper_backward(gdrf_ctx* ctx, int mode);
int gdrf_get_hyper_backward(const gdrf_ctx* ctx);
/* whiten = 0: the unwhitened branch of pyro's gp.util.conditional (gdrf/models/sparse_gdrf.py:30,175-185: the
 * constructor's `whiten` argument): u_loc and u_scale_tril parameterise q(f(Z)) itself, the predictive uses
 * L^-1 u_loc and L^-1 u_scale_tril.  Default 1 (whitened), which is what the reference's train() always runs. */
int gdrf_set_whiten(gdrf_ctx* ctx, int whiten);
/* mean_function (gdrf/models/abstract_gdrf.py:17-18,33-48; applied as `f_loc + self._mean_function(xs)` in model and guide,
 * gdrf/models/sparse_gdrf.py:346,395): `mean` is a BORROWED device array of the context's element type holding the values
 * the callable returned for the rows of the next gdrf_step_local call(s), addressed as mean[k*stride_k + n*stride_n]
 * (a stride of 0 broadcasts: the reference's (N,) return value is stride_k = 0, stride_n = 1).  NULL = zero_mean (default).
 * The mean shifts the sampled mu that enters the link; the Normal site terms depend on mu - f_loc only, and - as in the
 * reference - log_topic_probs / the predictive path do not add it (sparse_gdrf.py:161-186). */
int gdrf_set_mean(gdrf_ctx* ctx, const void* mean, int64_t stride_k, int64_t stride_n);
/* Learnable inducing inputs (gdrf/models/sparse_gdrf.py:79-88, fixed_inducing_points=False: a PyroParam under
 * stack([interval(0, 1)] * D), i.e. Z = sigmoid(unconstrained)).  The flat parameter vector always carries an (M, D)
 * block for the unconstrained values, gdrf_inducing_layout -> {offset, M*D}; the caller evaluates Z = sigmoid(block) and
 * passes it as Z_dev as before.  With on = 1, gdrf_step_local additionally accumulates
 * G[j][d] = sum_n Kbar_nm[n][j] dk/dr2 (z_jd - x_nd) into red_d[8 + j*D + d] (all-reduced with the rest) and
 * gdrf_step_finish writes d loss / d unconstrained into the block of grads (K_nm and K_uu paths, sigmoid Jacobian);
 * with on = 0 (default) that block of grads stays zero. */
int gdrf_set_learn_inducing(gdrf_ctx* ctx, int on);
int gdrf_inducing_layout(const gdrf_ctx* ctx, int64_t out[2]);
void gdrf_ctx_destroy(gdrf_ctx* ctx);

/* Flat unconstrained-parameter vector (the PyroParam storage of gdrf/models/sparse_gdrf.py:96-122
 * and the pyro kernel's lengthscale/variance): out = {off_log_lengthscale, off_log_variance,
 * off_log_noise, off_u_loc (K*M), off_phi_unc (K*V), off_u_scale_tril_unc (K*M*M), total}.  Element 3 of the vector
 * (between log_noise and u_loc) is log(scale_mixture) of the RationalQuadratic kernel (pyro 1.8.0
 * kernels.isotropic.RationalQuadratic: variance * (1 + r2 / (2 scale_mixture))^(-scale_mixture)); the other kernels
 * ignore it and its gradient is 0. */
int gdrf_param_layout(const gdrf_ctx* ctx, int64_t out[7]);
/* Per-step all-reduce payload: out = {off_ubar, off_phibar, off_A, off_GT, total_T, total_d}. */
int gdrf_red_layout(const gdrf_ctx* ctx, int64_t out[6]);
/* The step's ONE collective ("one ncclAllReduce(sum) per step over a flat buffer"): gdrf_payload_pack copies the
 * 8 + M*D doubles of red_d into the tail of red_T (total_T of gdrf_red_layout includes it) in red_T's element type - as they
 * are for double contexts, as four float pieces each (12 + 12 + 12 + 24 mantissa bits: sums over <= 8 ranks are exact while the
 * ranks' values of an entry lie within 2^9 of each other, as the loss sums do, and accurate to 2^-24 of the largest summand otherwise)
 * for float ones -; the caller all-reduces red_T alone and gdrf_payload_unpack restores red_d from the reduced tail. */
int gdrf_payload_pack(gdrf_ctx* ctx, void* red_T_dev, const double* red_d_dev, void* stream);
int gdrf_payload_unpack(gdrf_ctx* ctx, const void* red_T_dev, double* red_d_dev, void* stream);
/* The collective behind the C ABI, for hosts that do not go through torch.distributed: the caller registers ONE function that sums
 * `count` elements of `buf` (device memory; float when is_double = 0, double otherwise) in place over its ranks on `stream` - e.g. a
 * wrapper of ncclAllReduce(buf, buf, count, ncclFloat / ncclDouble, ncclSum, comm, stream) on the RCCL communicator it owns (`user`) - and
 * returns 0 on success.  gdrf_payload_allreduce then is the step's single collective: pack, that function on the whole flat payload
 * (total_T elements of gdrf_red_layout), unpack.  With no function registered it is a no-op (one rank).  gdrf_amd.Engine uses it when it is
 * constructed with allreduce_fn=...; its default remains torch.distributed.all_reduce between gdrf_payload_pack and gdrf_payload_unpack. */
typedef int (*gdrf_allreduce_fn)(void* buf_dev, int64_t count, int is_double, void* stream, void* user);
int gdrf_set_allreduce(gdrf_ctx* ctx, gdrf_allreduce_fn fn, void* user);
int gdrf_payload_allreduce(gdrf_ctx* ctx, void* red_T_dev, double* red_d_dev, void* stream);
/* Dirichlet concentration (K*V doubles, host): validate_dirichlet_param, gdrf/models/utils.py:6-24. */
int gdrf_set_dirichlet(gdrf_ctx* ctx, const double* alpha_host);

/* K_nm = k(X, Z), row-major (n, ldo): pyro RBF/Matern52.forward(X, Z) as used inside
 * gp.util.conditional (gdrf/models/sparse_gdrf.py:334-344).  The HBM-roofline kernel. */
int gdrf_knm(gdrf_ctx* ctx, const void* X_dev, int64_t n, const void* Z_dev, const void* params_dev,
             void* out_dev, int64_t ldo, void* stream);

/* eps[k][n] ~ N(0,1), Philox keyed by (seed; global row n_offset+n, k, step): the rsample of the
 * guide's mu site (gdrf/models/sparse_gdrf.py:403-405). */
int gdrf_fill_eps(gdrf_ctx* ctx, uint64_t seed, uint32_t step, int64_t n_offset, int64_t n, void* eps_dev, void* stream);

/* sum_n [lgamma(sum_v w+1) - sum_v lgamma(w+1)]: the data-only part of Multinomial.log_prob
 * (gdrf/models/sparse_gdrf.py:363-372).  Synchronises the stream. */
int gdrf_ll_const(gdrf_ctx* ctx, const int32_t* ws_dev, int64_t n, double* out_host, void* stream);
/* The same constant written to a device double (e.g. red_d + 7, where gdrf_step_finish reads it when its ll_const argument
 * is NaN), without synchronising: a mini-batch step (gdrf/train_script.py:461-465) then needs no host round trip for it. */
int gdrf_ll_const_dev(gdrf_ctx* ctx, const int32_t* ws_dev, int64_t n, double* out_dev, void* stream);

/* jittercholesky's retry loop (gdrf/models/utils.py:27-40) on kernel(inducing_points)
 * (gdrf/models/sparse_gdrf.py:327-328,382-383): nlev (<= 8) attempts with the cumulative jitters
 * jitters_host[0..nlev) factorised concurrently in ONE launch, in the array precision -- the precision in
 * which the reference's torch.linalg.cholesky decides whether more jitter is needed.  failed_host[l] != 0 when
 * attempt l hit a non-positive pivot; the caller takes the first success (same outcome as trying them in order).
 * Synchronises the stream.  Uses its own workspace and flag slots: it may run on a second stream beside
 * gdrf_factorize()/gdrf_step_local() of the same parameters (gdrf_amd.Engine overlaps it that way). */
int gdrf_probe(gdrf_ctx* ctx, const void* Z_dev, const void* params_dev, const double* jitters_host, int nlev,
               int* failed_host, void* stream);
/* The same in two calls: the launch alone (asynchronous, so that the caller can enqueue the step it speculates on beside it) and
 * the read of the flags of the last launch (waits for `stream`). */
int gdrf_probe_launch(gdrf_ctx* ctx, const void* Z_dev, const void* params_dev, const double* jitters_host, int nlev, void* stream);
int gdrf_probe_read(gdrf_ctx* ctx, int nlev, int* failed_host, void* stream);

/* K_uu + jitter_total*I, its Cholesky factor L and L^{-1} in the solve precision, kept in the context for
 * the calls below. */
int gdrf_factorize(gdrf_ctx* ctx, const void* Z_dev, const void* params_dev, double jitter_total, void* stream);
/* The same with a mode: 0 = gdrf_factorize.  1 = factorise AHEAD of the step that will use the result (call it right behind the optimizer
 * update: the chain then runs while the host reads the loss and enqueues the next step) and keep a copy of its inputs (the kernel
 * hyper-parameters and Z).  2 = the step's own call: if a mode-1 factorisation with this jitter is waiting, its inputs are only compared
 * with the current ones on the device - gdrf_chol_failed then reports a mismatch like a failure, and the caller redoes the step with
 * mode 0; otherwise as mode 0. */
int gdrf_factorize_mode(gdrf_ctx* ctx, const void* Z_dev, const void* params_dev, double jitter_total, void* stream, int mode);

/* Forward + backward over this rank's n_local observations: everything of one
 * SVI.step(xs, ws) (gdrf/train_script.py:467 -> sparse_gdrf.py:323-409) that is a sum over
 * observations.  X (n,D), ws (n,V) int32, eps (K,n).  Writes the partial sums to red_T
 * (dtype elements) and red_d (doubles); the caller all-reduces both over ranks.  Needs gdrf_factorize() on the same params. */
int gdrf_step_local(gdrf_ctx* ctx, const void* X_dev, const int32_t* ws_dev, const void* eps_dev, int64_t n_local,
                    const void* Z_dev, const void* params_dev, void* red_T_dev, double* red_d_dev, void* stream);

/* One step around a caller-supplied link function (the reference's `link_function` constructor argument, gdrf/models/abstract_gdrf.py:34-50,
 * used as `self._link_function(mu).transpose(-2, -1)` in gdrf/models/sparse_gdrf.py:361), which the caller evaluates itself between three calls:
 *   phase 0: everything of gdrf_step_local up to mu = f_loc + f_var eps (+ mean) -> workspace 14 (K, ldk);
 *   phase 1: ext = theta = link(mu), (K, ext_ld) -> d loglik / d theta in workspace 6, the word-topic gradient and the log-likelihood sum;
 *   phase 2: ext = mubar = J_link^T thetabar, (K, ext_ld) -> the Normal sites, the row-local backward and the rest of gdrf_step_local.
 * theta need not sum to one over the topics (Multinomial normalises p = theta^T Phi, as torch does).  Not a fused path. */
int gdrf_step_local_link(gdrf_ctx* ctx, const void* xs_dev, const int32_t* ws_dev, const void* eps_dev, int64_t n_local, const void* Z_dev,
                         const void* params_dev, void* red_T_dev, double* red_d_dev, void* stream, int phase, const void* ext_dev, int64_t ext_ld);

/* The same with the guide and the model evaluated at DIFFERENT inputs: the reference's guide scales its inputs twice
 * (gdrf/models/sparse_gdrf.py:376 @scale_decorator and :380 `xs = self.scale(xs)`), its model once (:324), so for a world other
 * than the unit cube the guide's gp.util.conditional sees X_guide = scale(scale(xs)) and the model's X_model = scale(xs).
 * mu is drawn from the guide-side predictive, log q uses it, log p(mu) the model-side one; both are differentiated.  About 2.5 x
 * the work of gdrf_step_local (two forwards and backwards, the guide's forward twice); the mean_function values of the guide side
 * come from gdrf_set_mean_guide, the model side's from gdrf_set_mean. */
int gdrf_step_local2(gdrf_ctx* ctx, const void* X_model_dev, const void* X_guide_dev, const int32_t* ws_dev, const void* eps_dev,
                     int64_t n_local, const void* Z_dev, const void* params_dev, void* red_T_dev, double* red_d_dev, void* stream);
int gdrf_set_mean_guide(gdrf_ctx* ctx, const void* mean, int64_t stride_k, int64_t stride_n);

/* Replicated epilogue: Cholesky / kernel hyper-parameter backward, constraint Jacobians, Dirichlet
 * term, loss.  grads (same layout as params) = d loss / d unconstrained.  out_d (8 doubles) =
 * {loss, cholesky_failed, site_sum, loglik_sum, log_prior_phi, ...}.  ll_const = the data-only constant of the
 * Multinomial log-likelihood (gdrf_ll_const, summed over ranks); pass NaN to take it from red_d[7] on the device. */
int gdrf_step_finish(gdrf_ctx* ctx, const void* Z_dev, const void* params_dev, const void* red_T_dev,
                     const double* red_d_dev, double n_global, double ll_const, void* grads_dev, double* out_d_dev,
                     void* stream);

/* pyro.optim.{Adam,AdamW,ClippedAdam} on every unconstrained tensor at once (train_script.py:73-87);
 * skipped on the device when the step's Cholesky failed. t = 1-based step count. */
int gdrf_adam(gdrf_ctx* ctx, int mode, void* params_dev, const void* grads_dev, void* m_dev, void* v_dev, int64_t t,
              double lr, double beta1, double beta2, double eps, double weight_decay, double clip, void* stream);

/* Predictive mean path: log_topic_probs / topic_probs / word_probs / perplexity
 * (gdrf/models/sparse_gdrf.py:161-186, abstract_gdrf.py:113-139).  mode 0: out (K,n) ;
 * 1: out (n,K) ; 2: out (n,V) ; 3: out_d_dev[0..1] = {sum w log p, sum w}.  Modes 0-3 never form the variance (the reference
 * computes and discards it, quirk Q4): K_nm is generated tile by tile as the A operand of the solve-precision matrix instruction and
 * multiplied with L^-T u (csrc/predict.h); any n.
 * mode 4: out (2,K,n) = {f_loc, f_var} of gp.util.conditional(full_cov=False) as SparseGDRF.forward(Xnew) returns them
 * (gdrf/models/sparse_gdrf.py:277-319; the mean_function is added by the caller): the step's forward (K_nm, W = K_nm L^-T,
 * loc = W U^T, tt = |S_k^T w|^2) plus one pass for var = clamp(variance - |w|^2, 0) + tt; n <= n_cap.  Needs gdrf_factorize(). */
int gdrf_predict(gdrf_ctx* ctx, const void* X_dev, int64_t n, const void* Z_dev, const void* params_dev,
                 const int32_t* ws_dev, int mode, void* out_dev, double* out_d_dev, void* stream);

/* Did the last gdrf_factorize() hit a non-positive pivot?  Synchronises the stream. */
int gdrf_chol_failed(gdrf_ctx* ctx, int* failed_host, void* stream);

/* Borrowed pointers into the workspace (for parity tests): which = 0 W, 1 Wbar, 2 q, 3 loc, 4 tt,
 * 5 vbar, 6 locbar, 7 asum, 8 Kuu, 9 L, 10 Linv, 11 S, 12 B, 13 phi, 14 mu, 15 LinvT, 16 ST, 17 the step's K_nm. */
int gdrf_ws_ptr(gdrf_ctx* ctx, int which, void** ptr, int64_t* nelem);
/* Element size (4 or 8 bytes) of that buffer: Kuu, L, Linv, LinvT and the step's K_nm live in the solve precision. */
int gdrf_ws_elem_size(gdrf_ctx* ctx, int which);
/* Device-to-device copy of the first nelem elements of that buffer into dst_dev. */
int gdrf_ws_copy(gdrf_ctx* ctx, int which, void* dst_dev, int64_t nelem, void* stream);

/* Per-kernel timing with HIP events recorded on the launch stream (off by default).  Slots:
 * 0 probe, 1 k_nm, 2 transforms+B_k, 3 fwd_w, 4 loc (W U^T), 5 fwd_t, 6 elbo_rows, 7 bwd_wbar,
 * 8 bwd_knm, 9 tn_sym (A_k), 10 tn_gt, 11 slab reductions, 12 ubar, 13 step_finish, 14 adam, 15 factorize.
 * gdrf_get_timing synchronises on the recorded events and returns accumulated ms and counts. */
int gdrf_set_timing(gdrf_ctx* ctx, int enable);
int gdrf_get_timing(gdrf_ctx* ctx, double* ms_out, int64_t* count_out, int nslots);

#ifdef __cplusplus
}
#endif
#endif

// libgdrf_hip: C-ABI entry points (include/gdrf_hip.h) and launch sequencing.  gfx950 only.
#include "../../include/gdrf_hip.h"
#include "common.h"
#include "gemm_nt.h"
#include "gemm_tn.h"
#include "gemm_split.h"
#include "gemm_tn_topics.h"
#include "gemm_tn_topics1.h"
#include "gemm_fwd_t1.h"
#include "kernels_mm.h"
#include "kernels_n.h"
#include "predict.h"
#include "rows_mfma.h"
#include "hyper_tn.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <vector>

using namespace gdrf;

static thread_local std::string g_err;
static int fail(int code, const char* what, const char* detail) {
  g_err = std::string(what) + ": " + detail;
  return code;
}
#define HIPCHK(x)                                                             \
  do {                                                                        \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess) return fail(-(int)e_ - 1000, #x, hipGetErrorString(e_)); \
  } while (0)
#define LAUNCHCHK(name)                                                       \
  do {                                                                        \
    hipError_t e_ = hipGetLastError();                                        \
    if (e_ != hipSuccess) return fail(-(int)e_ - 1000, name, hipGetErrorString(e_)); \
  } while (0)

#define GDRF_NSLOTS 16
// dtype of a context: N-side element type T / solve element type TS
//   GDRF_F32 (0): T = float, TS = double  (default fp32 mode: K-fold contractions on f32 MFMA, the ill-conditioned
//                 pieces -- K_uu, Cholesky, L^-1, the solve W = K_nm L^-T, its backward, the M x M epilogue -- in f64)
//   GDRF_F64 (1): T = TS = double
//   GDRF_F32_PURE (2): T = TS = float (everything in fp32, as the reference's .float() casts do; for A/B comparisons)
struct gdrf_ctx {
  int dev, M, Mp, K, V, D, dtype, kind;
  int64_t ncap, ldk;          // ldk = leading dimension of the (K, n) arrays
  size_t esz, ssz;            // element sizes: N side, solve side
  int nt;                     // 128-wide tiles over Mp
  int nsplit_cap;
  // solve precision, M x M (ld Mp)
  void *mmslab; size_t mmslab_bytes;    // split-K slabs of the small M x M products: [slices][batch][Mp][Mp]
  void *Kuu, *Lw, *Lo, *L, *LT, *Linv, *LinvT, *Dinv, *t0, *t1, *t2, *Cf, *CfT, *Zs, *GTs;      // Lo: the panel-wise factorisation's output (Lw is its work matrix)
  void *Knm;                  // [ncap][Mp] K_nm in the solve precision (forward A operand, backward epilogue)
  // probe (N-side precision) scratch, only when T != TS
  void *pK, *pL;
  // N-side precision
  void *S, *ST, *Bm, *Sbar, *phi, *Upad, *qpart;
  void *W, *Wbar, *q, *loc, *tt, *vbar, *locbar, *asum, *mu;
  void *g_loc, *g_tt, *g_qpart, *g_vbar, *g_locbar, *g_asum, *g_redT; double* g_redd;   // two-point evaluation (quirk Q3), allocated on first use
  const void* mean_g; int64_t mean_g_sk, mean_g_sn;
  void *Wd;                   // W' = (dK_nm / d log lengthscale) Linv^T, allocated on first use (fixed inducing inputs, kernels without a third hyper-parameter)
  double* wdpart; hipEvent_t ev_wd;
  void* dKh; hipStream_t side2; hipEvent_t ev_ak, ev_ak_done;    // pieces of dK_nm / d log ls (hyper_tn.h), allocated on first use; third stream: A_k beside the f64 backward GEMM
  gdrf_allreduce_fn allreduce; void* allreduce_user;      // the caller's collective (gdrf_set_allreduce), or null
  int hyper_tn; double* hpart;   // K_nm parts of the hyper-parameter gradients through Hd = dK^T Wbar on the TN kernel (hyper_tn.h) instead of the f64 backward GEMM
  void* vbs = nullptr;        // vbar x block scale, zero-padded to a multiple of 64 rows (gemm_tn_topics1.h)
  void *Bh, *STh, *Wh;        // 16-bit pieces of B_k, S_k^T and W (f32 contexts; split-operand MFMA forms, gemm_split.h)
  int split;                  // 0: native f32 MFMA; 1: "bf16x6" (3 bf16 pieces, 6 products); 2: "f16x3" (2 fp16 pieces, 3 products, block scales)
  int wh_pieces;              // pieces Wh has room for
  float* ssc; unsigned* smx;  // block scales (SplitLay pairs) and the maxima they come from
  void *Tst;                  // T_k = W S_k kept for the backward, or nullptr (dense W B_k form instead)
  int64_t t_bs, t_ts;         // its per-topic / per-row-tile strides in elements
  void *slab, *ubar_part, *phibar_part;
  double *dpart, *dsmall;     // dsmall: [0..2] kuu sums, [8] ll_const scratch
  double *llpart;             // per-workgroup partials of the data constant (own buffer: it may be queued beside a step)
  int64_t dpart_len, ubar_blocks_cap, erows_grid_cap;
  double* alpha_dev; double lgam_const;
  Hyper *hyp, *hyp_probe; int* flag;        // flag[0]: solve factorisation failed; flag[8..16): probe levels failed; flag[16]: a reused factorisation's inputs changed (own
                                            // 64-byte line, written only by the compare kernel and by reuse_clear below: the side stream's memset of flag[0..8) never touches it)
  void* snap; int prefact_valid; double prefact_jitter;     // the inputs of a factorisation made ahead of its step (gdrf_factorize_mode)
  hipStream_t side;           // small, tail-heavy kernels run here beside the big GEMMs (fork/join with events)
  hipEvent_t ev_fork, ev_loc, ev_fork2, ev_join, ev_fact0, ev_fact;
  int fact_pending;           // a factorisation has been queued on the side stream: consumers wait for ev_fact
  const void* mean;           // borrowed (K, n) mean_function values for the next gdrf_step_local calls, or null (zero_mean)
  int64_t mean_sk, mean_sn;
  int unwhitened;             // whiten = False: u' = L^-1 u, S' = L^-1 S (solve-precision scratch below, allocated on demand)
  void *uS, *uSb, *uSc, *uU, *uUb, *Uw;
  int learn_z; double* zpart; // learnable inducing inputs: per-row-tile partial sums [ceil(ncap/128)][M][D]
  std::vector<void*> allocs;
  // optional per-kernel HIP-event timing (gdrf_set_timing): events recorded on the launch stream
  int timing;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> tev;   // (slot, (start, stop)) pending
  std::vector<hipEvent_t> ev_pool;
  double t_ms[GDRF_NSLOTS]; int64_t t_cnt[GDRF_NSLOTS];
};

struct ScopedTimer {
  gdrf_ctx* c; int slot; hipStream_t s; hipEvent_t e0, e1; bool on;
  ScopedTimer(gdrf_ctx* c_, int slot_, hipStream_t s_) : c(c_), slot(slot_), s(s_), on(c_->timing != 0) {
    if (!on) return;
    auto get = [&]() { hipEvent_t e; if (!c->ev_pool.empty()) { e = c->ev_pool.back(); c->ev_pool.pop_back(); } else (void)hipEventCreate(&e); return e; };
    e0 = get(); e1 = get();
    (void)hipEventRecord(e0, s);
  }
  ~ScopedTimer() {
    if (!on) return;
    (void)hipEventRecord(e1, s);
    c->tev.push_back({slot, {e0, e1}});
  }
};

const char* gdrf_last_error(void) { return g_err.c_str(); }
int gdrf_version(void) { return 1; }

static int64_t poff(const gdrf_ctx* c, int which) {
  // flat parameter layout; every segment starts on a multiple of 4 elements
  const int64_t o_uloc = 4, o_phi = round_up(o_uloc + (int64_t)c->K * c->M, 4);
  const int64_t o_S = round_up(o_phi + (int64_t)c->K * c->V, 4);
  const int64_t o_Z = round_up(o_S + (int64_t)c->K * c->M * c->M, 4);          // unconstrained inducing inputs (M, D)
  const int64_t total = round_up(o_Z + (int64_t)c->M * c->D, 4);
  switch (which) { case 0: return 0; case 1: return 1; case 2: return 2; case 3: return o_uloc; case 4: return o_phi;
                   case 5: return o_S; case 7: return o_Z; default: return total; }
}
static int64_t roff(const gdrf_ctx* c, int which) {
  const int64_t mm = (int64_t)c->Mp * c->Mp;
  const int64_t o_ubar = 0, o_phib = round_up((int64_t)c->K * c->Mp, 4), o_A = round_up(o_phib + (int64_t)c->K * c->V, 4);
  const int64_t o_GT = o_A + c->K * mm, o_tail = o_GT + mm;
  // tail: the doubles of red_d for the step's single all-reduce (gdrf_payload_pack): as they are in f64 contexts, four float
  // pieces each in f32 ones
  const int64_t nd = 8 + (int64_t)c->M * c->D, total = o_tail + round_up(c->esz == 8 ? nd : 4 * nd, 4);
  switch (which) { case 0: return o_ubar; case 1: return o_phib; case 2: return o_A; case 3: return o_GT; case 5: return o_tail; default: return total; }
}

int gdrf_param_layout(const gdrf_ctx* c, int64_t out[7]) { for (int i = 0; i < 7; ++i) out[i] = poff(c, i); return 0; }
int gdrf_red_layout(const gdrf_ctx* c, int64_t out[6]) {
  for (int i = 0; i < 5; ++i) out[i] = roff(c, i);
  out[5] = 8 + (int64_t)c->M * c->D;          // 8 scalars, then the (M, D) inducing-input sums (learnable inducing points)
  return 0;
}
int gdrf_inducing_layout(const gdrf_ctx* c, int64_t out[2]) { out[0] = poff(c, 7); out[1] = (int64_t)c->M * c->D; return 0; }

// row blocks of the ubar partial kernel: ~1024 workgroups, multiples of its 256-row staging step
static int64_t ubar_rows_per_block(int64_t n) { return std::max<int64_t>(256, round_up((n + 1023) / 1024, 256)); }

// number of row splits of the TN kernels: fill the chip's resident-workgroup slots (256 CUs x 3) with as
// little last-round idling as possible, keep >= 8 chunks per split, cap the slab memory
static int tn_nsplit(const gdrf_ctx* c, int64_t n, int BR, int wg_per_cu = 3) {
  if (const char* e = getenv("GDRF_TN_NSPLIT")) { const int v = atoi(e); if (v > 0) return v; }   // tuning knob (tools/)
  const int tiles = c->K * c->nt * (c->nt + 1) / 2;
  const double slots = 256.0 * wg_per_cu;
  int64_t maxs = (n + 8 * BR - 1) / (8 * BR);
  if (maxs > 64) maxs = 64;
  if (maxs < 1) maxs = 1;
  int best = 1; double best_eff = 0;
  for (int ns = 1; ns <= maxs; ++ns) {
    const double rounds = tiles * (double)ns / slots;
    const double eff = rounds / std::ceil(rounds);
    if (eff > best_eff + 1e-9 || (eff > best_eff - 0.02 && rounds >= 2.0 && tiles * (double)best / slots < 2.0)) { best_eff = eff; best = ns; }
  }
  return best;
}

// Row splits of the G^T = W^T Wbar contraction on the bf16 TN kernel.  Its nt*nt tiles are few, so the split count decides both the
// fill and the block -> XCD map: with a multiple of 8 every XCD owns whole row slabs and the 16 tiles of a slab share their W / Wbar
// panels through that XCD's L2 (measured at N = 1e6, M = 512: 18.8 -> 8.2 GB fetched, 4.46 -> 3.22 ms at 64 splits).
static int tn_nsplit_gt(const gdrf_ctx* c, int64_t n, int BR) {
  int64_t maxs = (n + 8 * BR - 1) / (8 * BR);
  if (maxs > 64) maxs = 64;
  if (maxs < 8) return tn_nsplit(c, n, BR, 2);
  if (const char* e = getenv("GDRF_TN_NSPLIT")) { const int v = atoi(e); if (v > 0) return v; }
  const int tiles = c->nt * c->nt;
  int best = 8; double best_eff = 0;
  for (int ns = 8; ns <= maxs; ns += 8) {
    const double rounds = tiles * (double)ns / 512.0;
    const double eff = rounds / std::ceil(rounds);
    if (eff > best_eff + 1e-9 || (eff > best_eff - 0.02 && rounds >= 2.0 && tiles * (double)best / 512.0 < 2.0)) { best_eff = eff; best = ns; }
  }
  return best;
}

// Row splits of the all-topics A_k kernel (gemm_tn_topics.h): one 512-thread workgroup per CU, tiles x topic groups x splits
// workgroups; fill the 256 CUs' rounds, >= 16 chunks per split, at most 64 splits
static bool tn_topics_on(const gdrf_ctx* c) {
  if (c->split != 2) return false;
  const char* e = getenv("GDRF_TN_TOPICS");
  return !(e && e[0] == '0');
}
// K_nm parts of the hyper-parameter gradients through W' = (dK/dlog ls) Linv^T (forward-shaped GEMM + two dot products with Wbar)
// instead of the backward GEMM Kbar = Wbar Linv: fixed inducing inputs (their gradient needs Kbar itself) and kernels whose only
// shape parameter is the lengthscale
// Measured at the headline size (profiles/r02): correct, but NOT faster - the step is work-conserving on the matrix pipe, so the second
// forward-shaped f64 GEMM (6.7 ms alone, 13.8 ms beside fwd_t) costs what the backward GEMM with G^T inside its stalls did (53.0 vs
// 52.5 ms/step).  Kept as an opt-in (GDRF_WD_PATH=1, covered by a parity test); the backward GEMM is the default.
static bool wd_path(const gdrf_ctx* c) {
  if (c->learn_z || c->kind == GDRF_RATIONALQUADRATIC) return false;
  const char* e = getenv("GDRF_WD_PATH");
  return e && e[0] == '1';
}
// K_nm parts of the hyper-parameter gradients through Hd = dK^T Wbar on the split-fp16 TN kernel (hyper_tn.h): f16x3 contexts with the f64
// solve, fixed inducing inputs (their gradient needs Kbar itself), kernels whose only shape parameter is the lengthscale
static bool hyper_tn_on(const gdrf_ctx* c) {
  return c->hyper_tn && c->split == 2 && c->ssz == 8 && !c->learn_z && c->kind != GDRF_RATIONALQUADRATIC && !c->Tst && (c->Mp / 8) <= 256 && !wd_path(c);
}
static int tn_topics_nsplit(const gdrf_ctx* c, int64_t n) {
  if (const char* e = getenv("GDRF_TNT_NSPLIT")) { const int v = atoi(e); if (v > 0) return v; }
  const int units = tnt_ntiles(c->Mp) * ((c->K + TNT_KT - 1) / TNT_KT);
  int64_t maxs = (n + 16 * 32 - 1) / (16 * 32);
  if (maxs > 64) maxs = 64;
  if (maxs < 1) maxs = 1;
  // cost model (relative units): rounds of one workgroup per CU, each walking n / ns rows, plus the slab traffic that grows with ns (K Mp^2
  // floats written and read per split: 0.13 ms at 64 splits, M = 512, K = 10 - a fixed cost that does not shrink with a rank's share
  // of the rows; at the 8-GPU per-rank size of the headline workload the optimum moves from 64 to ~25 splits)
  const double chunk_us = 3.7;                                   // one 32-row chunk of a (tile, topic group) unit: 9.0 ms at N = 1e6, M = 512, K = 10
  const double slab_us_per_split = 2.0 * (double)c->K * c->Mp * c->Mp * 4.0 / 5.0e6;       // bytes at ~5 TB/s, in us
  int best = 1; double best_t = 1e300;
  for (int ns = 1; ns <= maxs; ++ns) {
    const double rounds = std::ceil(units * (double)ns / 256.0);
    const double t = rounds * std::ceil((double)n / (32.0 * ns)) * chunk_us + ns * slab_us_per_split;
    if (t < best_t * 0.995) { best_t = t; best = ns; }
  }
  return best;
}

int gdrf_ctx_create(gdrf_ctx** out, int device, int64_t n_cap, int M, int K, int V, int D, int dtype, int kernel_id) {
  return gdrf_ctx_create_ex(out, device, n_cap, M, K, V, D, dtype, kernel_id, GDRF_STORE_T_OFF);
}

int gdrf_ctx_create_ex(gdrf_ctx** out, int device, int64_t n_cap, int M, int K, int V, int D, int dtype, int kernel_id, int store_t) {
  if (!out || n_cap < 1 || M < 1 || K < 1 || V < 1 || D < 1) return fail(-1, "gdrf_ctx_create", "bad size");
  if (K > GDRF_TILE) return fail(-1, "gdrf_ctx_create", "num_topic_categories > 128 not supported (loc = W U^T runs as one 128-wide column tile)");
  if (D > GDRF_DMAX) return fail(-1, "gdrf_ctx_create", "more than 4 input dimensions not supported");
  if (dtype != GDRF_F32 && dtype != GDRF_F64 && dtype != GDRF_F32_PURE) return fail(-1, "gdrf_ctx_create", "dtype");
  if (kernel_id < GDRF_RBF || kernel_id > GDRF_RATIONALQUADRATIC) return fail(-1, "gdrf_ctx_create", "kernel_id");
  HIPCHK(hipSetDevice(device));
  gdrf_ctx* c = new gdrf_ctx();
  c->dev = device; c->M = M; c->Mp = (int)round_up(M, GDRF_MPAD); c->K = K; c->V = V; c->D = D;
  c->dtype = dtype; c->kind = kernel_id; c->ncap = n_cap; c->ldk = round_up(n_cap, 4);
  c->esz = dtype == GDRF_F64 ? 8 : 4;
  c->ssz = dtype == GDRF_F32_PURE ? 4 : 8;
  c->nt = (c->Mp + GDRF_TILE - 1) / GDRF_TILE;
  c->lgam_const = 0; c->alpha_dev = nullptr; c->timing = 0;
  c->pK = c->pL = nullptr; c->Tst = nullptr; c->side = nullptr; c->Bh = c->STh = c->Wh = nullptr; c->split = 0; c->wh_pieces = 0; c->ssc = nullptr; c->smx = nullptr;
  c->ev_fork = c->ev_loc = c->ev_fork2 = c->ev_join = c->ev_fact0 = c->ev_fact = nullptr; c->fact_pending = 0; c->learn_z = 0; c->zpart = nullptr; c->unwhitened = 0; c->mean = nullptr; c->mean_sk = c->mean_sn = 0;
  c->allreduce = nullptr; c->allreduce_user = nullptr; c->hyper_tn = 0; c->hpart = nullptr; c->dKh = nullptr; c->side2 = nullptr; c->ev_ak = c->ev_ak_done = nullptr; c->uS = c->uSb = c->uSc = c->uU = c->uUb = c->Uw = nullptr; c->Wd = nullptr; c->wdpart = nullptr; c->ev_wd = nullptr; c->g_loc = nullptr; c->mean_g = nullptr; c->mean_g_sk = c->mean_g_sn = 0;
  for (int i = 0; i < GDRF_NSLOTS; ++i) { c->t_ms[i] = 0; c->t_cnt[i] = 0; }
  const size_t mm = (size_t)c->Mp * c->Mp * c->esz, mms = (size_t)c->Mp * c->Mp * c->ssz;
  auto A = [&](void** p, size_t bytes) -> int {
    hipError_t e = hipMalloc(p, bytes ? bytes : 16);
    if (e != hipSuccess) return fail(-(int)e - 1000, "hipMalloc", hipGetErrorString(e));
    c->allocs.push_back(*p);
    return 0;
  };
  int rc = 0;
#define AL(ptr, bytes) if ((rc = A((void**)&(ptr), (bytes)))) { gdrf_ctx_destroy(c); return rc; }
  AL(c->Kuu, mms) AL(c->Lw, mms) AL(c->Lo, mms)
  c->mmslab_bytes = 8 * mms;
  c->prefact_valid = 0; c->prefact_jitter = 0;
  AL(c->snap, (size_t)(4 + (size_t)c->M * c->D) * c->esz)
  AL(c->mmslab, c->mmslab_bytes) AL(c->L, mms) AL(c->LT, mms) AL(c->Linv, mms) AL(c->LinvT, mms)
  AL(c->Dinv, (size_t)(c->Mp / 32) * 1024 * c->ssz)
  AL(c->t0, mms) AL(c->t1, mms) AL(c->t2, mms) AL(c->GTs, mms)
  AL(c->Cf, (size_t)K * M * c->ssz) AL(c->CfT, (size_t)c->Mp * 32 * c->ssz) AL(c->Zs, (size_t)c->Mp * D * c->ssz)
  AL(c->Knm, (size_t)n_cap * c->Mp * c->ssz)
  AL(c->pK, mm) AL(c->pL, mm * 8)          // probe scratch: K_uu without jitter, 8 level copies
  AL(c->S, mm * K) AL(c->ST, mm * K) AL(c->Bm, mm * K) AL(c->Sbar, mm * K)
  if (c->esz == 4) { AL(c->Bh, (size_t)3 * K * c->Mp * c->Mp * 2) AL(c->STh, (size_t)3 * K * c->Mp * c->Mp * 2) }
  if (c->esz == 4) { AL(c->vbs, (size_t)K * round_up(n_cap, 64) * sizeof(float)) }
  AL(c->phi, (size_t)K * V * c->esz)
  AL(c->Upad, (size_t)GDRF_TILE * c->Mp * c->esz) AL(c->qpart, (size_t)((c->Mp + 63) / 64) * c->ldk * c->esz)
  AL(c->W, (size_t)n_cap * c->Mp * c->esz) AL(c->Wbar, (size_t)n_cap * c->Mp * c->esz)
  AL(c->q, (size_t)c->ldk * c->esz) AL(c->asum, (size_t)c->ldk * c->esz)
  AL(c->loc, (size_t)K * c->ldk * c->esz) AL(c->tt, (size_t)K * c->ldk * c->esz) AL(c->vbar, (size_t)K * c->ldk * c->esz)
  AL(c->locbar, (size_t)K * c->ldk * c->esz) AL(c->mu, (size_t)K * c->ldk * c->esz)
  {
    // GDRF_STORE_T_AUTO would keep T_k when it fits comfortably (<= 64 GB and <= 40 % of the free HBM); measured on
    // MI355X the stored-T form is currently SLOWER (47 vs 43 ms at the headline size: its A operand is a cold HBM
    // stream on every chunk and the register prefetch is drained by vmcnt(0) waits), so AUTO resolves to OFF.
    c->t_ts = (int64_t)GDRF_TILE * c->Mp + 1536;                                   // + 6 KB (f32): not a power of two
    c->t_bs = c->t_ts * ((n_cap + GDRF_TILE - 1) / GDRF_TILE) + 40960 + 512;
    const size_t tbytes = (size_t)K * c->t_bs * c->esz;
    if (store_t == GDRF_STORE_T_ON) { AL(c->Tst, tbytes) }
  }
  c->nsplit_cap = tn_nsplit(c, n_cap, (int)(128 / c->esz));
  if (c->esz == 4) c->nsplit_cap = std::max({c->nsplit_cap, tn_nsplit(c, n_cap, 32, 2), tn_nsplit_gt(c, n_cap, 32), 64});   // the split TN forms run 2 workgroups per CU; the all-topics form up to 64 splits
  AL(c->slab, (size_t)c->nsplit_cap * (K + 2) * mm)          // K batches of A_k, one of G^T, one of Hd = dK^T Wbar
  AL(c->hpart, (size_t)std::max(4096, c->Mp + 2048) * sizeof(double))
  c->ubar_blocks_cap = std::min<int64_t>(1025, (n_cap + 255) / 256);     // upper bound of ubar_blocks(n) over n <= n_cap
  AL(c->ubar_part, (size_t)c->ubar_blocks_cap * K * c->Mp * c->esz)
  c->erows_grid_cap = 1024;
  AL(c->phibar_part, (size_t)c->erows_grid_cap * K * V * c->esz)
  const int64_t rtiles = (n_cap + GDRF_TILE - 1) / GDRF_TILE;
  // users: bwd_knm 3 per workgroup, kuu_bar_reduce 3 per inducing point, elbo_rows 4 per workgroup, predict / ll_const <= 2 x 2048
  c->dpart_len = std::max<int64_t>({((rtiles + 8) * ((c->Mp + 63) / 64) + 16) * 3, (int64_t)3 * c->Mp + 16, 4 * c->erows_grid_cap, (int64_t)8192});
  AL(c->dpart, (size_t)c->dpart_len * sizeof(double))
  AL(c->dsmall, 16 * sizeof(double))
  AL(c->llpart, 2048 * sizeof(double))
  AL(c->alpha_dev, (size_t)K * V * sizeof(double))
  AL(c->hyp, sizeof(Hyper)) AL(c->hyp_probe, sizeof(Hyper)) AL(c->flag, 128)
  AL(c->ssc, (size_t)SplitLay{K}.nfloats() * sizeof(float)) AL(c->smx, (size_t)SplitLay{K}.nmax() * sizeof(unsigned))
#undef AL
  {
    // the side stream carries the factorisation chain (small launches the step's first GEMM waits for): highest priority, so that its
    // workgroups are placed ahead of the K_nm kernel's 16 384 on the main stream (GDRF_SIDE_PRIO=0: default priority)
    int least = 0, greatest = 0;
    const char* sp = getenv("GDRF_SIDE_PRIO");
    if (!(sp && sp[0] == '0') && hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least) {
      HIPCHK(hipStreamCreateWithPriority(&c->side, hipStreamNonBlocking, greatest));
    } else {
      HIPCHK(hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking));
    }
  }
  for (hipEvent_t* e : {&c->ev_fork, &c->ev_loc, &c->ev_fork2, &c->ev_join, &c->ev_fact0, &c->ev_fact, &c->ev_wd, &c->ev_ak, &c->ev_ak_done})
    HIPCHK(hipEventCreateWithFlags(e, hipEventDisableTiming));
  HIPCHK(hipStreamCreateWithFlags(&c->side2, hipStreamNonBlocking));
  HIPCHK(hipMemset(c->flag, 0, 128));
  HIPCHK(hipMemset(c->W, 0, (size_t)n_cap * c->Mp * c->esz));
  std::vector<double> a((size_t)K * V, 1.0);
  *out = c;
  return gdrf_set_dirichlet(c, a.data());
}

void gdrf_ctx_destroy(gdrf_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->dev);
  for (void* p : c->allocs) (void)hipFree(p);
  for (auto& t : c->tev) { (void)hipEventDestroy(t.second.first); (void)hipEventDestroy(t.second.second); }
  for (auto e : c->ev_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : {c->ev_fork, c->ev_loc, c->ev_fork2, c->ev_join, c->ev_fact0, c->ev_fact, c->ev_wd, c->ev_ak, c->ev_ak_done}) if (e) (void)hipEventDestroy(e);
  if (c->side) (void)hipStreamDestroy(c->side);
  if (c->side2) (void)hipStreamDestroy(c->side2);
  delete c;
}

int gdrf_set_dirichlet(gdrf_ctx* c, const double* alpha) {
  HIPCHK(hipSetDevice(c->dev));
  double lg = 0;
  for (int k = 0; k < c->K; ++k) {
    double s = 0;
    for (int v = 0; v < c->V; ++v) {
      const double a = alpha[(size_t)k * c->V + v];
      if (!(a > 0)) return fail(-1, "gdrf_set_dirichlet", "b must be positive");
      s += a; lg -= std::lgamma(a);
    }
    lg += std::lgamma(s);
  }
  c->lgam_const = lg;
  HIPCHK(hipMemcpy(c->alpha_dev, alpha, (size_t)c->K * c->V * sizeof(double), hipMemcpyHostToDevice));
  return 0;
}

// which = 0 W, 1 Wbar, 2 q, 3 loc, 4 tt, 5 vbar, 6 locbar, 7 asum, 8 Kuu, 9 L, 10 Linv, 11 S, 12 B, 13 phi, 14 mu, 15 LinvT, 16 ST, 17 Knm (solve precision)
static int ws_lookup(gdrf_ctx* c, int which, void** ptr, int64_t* nelem, int* esz) {
  const int64_t mm = (int64_t)c->Mp * c->Mp, kn = (int64_t)c->K * c->ldk;
  void* p = nullptr; int64_t n = 0; int e = (int)c->esz;
  switch (which) {
    case 0: p = c->W; n = c->ncap * c->Mp; break;     case 1: p = c->Wbar; n = c->ncap * c->Mp; break;
    case 2: p = c->q; n = c->ldk; break;              case 3: p = c->loc; n = kn; break;
    case 4: p = c->tt; n = kn; break;                 case 5: p = c->vbar; n = kn; break;
    case 6: p = c->locbar; n = kn; break;             case 7: p = c->asum; n = c->ldk; break;
    case 8: p = c->Kuu; n = mm; e = (int)c->ssz; break;   case 9: p = c->L; n = mm; e = (int)c->ssz; break;
    case 10: p = c->Linv; n = mm; e = (int)c->ssz; break; case 11: p = c->S; n = mm * c->K; break;
    case 12: p = c->Bm; n = mm * c->K; break;         case 13: p = c->phi; n = (int64_t)c->K * c->V; break;
    case 14: p = c->mu; n = kn; break;                case 15: p = c->LinvT; n = mm; e = (int)c->ssz; break;
    case 16: p = c->ST; n = mm * c->K; break;
    case 17: p = c->Knm; n = c->ncap * c->Mp; e = (int)c->ssz; break;
    default: return fail(-1, "gdrf_ws_ptr", "unknown buffer id");
  }
  *ptr = p; *nelem = n; *esz = e;
  return 0;
}
int gdrf_ws_ptr(gdrf_ctx* c, int which, void** ptr, int64_t* nelem) { int e; return ws_lookup(c, which, ptr, nelem, &e); }
int gdrf_stores_t(const gdrf_ctx* c) { return c->Tst != nullptr; }
int gdrf_set_mfma_mode(gdrf_ctx* c, int mode) {
  if (mode < 0 || mode > 2) return fail(-1, "gdrf_set_mfma_mode", "mode");
  if (mode != 0 && (c->esz != 4 || c->Tst)) return fail(-1, "gdrf_set_mfma_mode", "the split modes need float arrays and the dense Wbar form");
  // f16x3 scales W by the bound |w| <= sqrt(variance), which the f64 solve guarantees (4x headroom); an all-fp32 solve of an ill-conditioned
  // K_uu can break it and the fp16 pieces would overflow to inf without a trace: bf16x6 (f32's exponent range) is the split mode there
  if (mode == 2 && c->ssz != 8) return fail(-1, "gdrf_set_mfma_mode", "f16x3 needs the f64 solve (GDRF_F32); use bf16x6 or f32 with GDRF_F32_PURE");
  HIPCHK(hipSetDevice(c->dev));
  const int np = mode == 1 ? 3 : (mode == 2 ? 2 : 0);
  if (np > c->wh_pieces) {            // 16-bit pieces of W: np x n_cap x Mp halfwords, allocated on first use
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, (size_t)np * c->ncap * c->Mp * 2);
    if (e != hipSuccess) return fail(-(int)e - 1000, "hipMalloc(Wh)", hipGetErrorString(e));
    c->Wh = p; c->wh_pieces = np; c->allocs.push_back(p);          // a smaller earlier block stays owned by the context until destroy
  }
  c->split = mode;
  return 0;
}
int gdrf_set_mean(gdrf_ctx* c, const void* mean, int64_t stride_k, int64_t stride_n) {
  if (stride_k < 0 || stride_n < 0) return fail(-1, "gdrf_set_mean", "strides must be >= 0 (0 broadcasts)");
  c->mean = mean; c->mean_sk = stride_k; c->mean_sn = stride_n;
  return 0;
}

int gdrf_set_mean_guide(gdrf_ctx* c, const void* mean, int64_t stride_k, int64_t stride_n) {
  if (stride_k < 0 || stride_n < 0) return fail(-1, "gdrf_set_mean_guide", "strides must be >= 0 (0 broadcasts)");
  c->mean_g = mean; c->mean_g_sk = stride_k; c->mean_g_sn = stride_n;
  return 0;
}

int gdrf_set_whiten(gdrf_ctx* c, int whiten) {
  if (whiten != 0 && whiten != 1) return fail(-1, "gdrf_set_whiten", "whiten must be 0 or 1");
  if (!whiten && c->Tst) return fail(-1, "gdrf_set_whiten", "whiten = 0 needs the dense Wbar form (GDRF_STORE_T_OFF)");
  HIPCHK(hipSetDevice(c->dev));
  if (!whiten && !c->uS) {
    const size_t kmm = (size_t)c->K * c->Mp * c->Mp * c->ssz, kv = (size_t)c->K * c->Mp * c->ssz;
    void** ps[] = {&c->uS, &c->uSb, &c->uSc, &c->uU, &c->uUb, &c->Uw};
    const size_t sz[] = {kmm, kmm, kmm, kv, kv, (size_t)c->K * c->M * c->esz};
    for (int i = 0; i < 6; ++i) {
      void* p = nullptr;
      hipError_t e = hipMalloc(&p, sz[i]);
      if (e != hipSuccess) return fail(-(int)e - 1000, "hipMalloc(unwhitened scratch)", hipGetErrorString(e));
      HIPCHK(hipMemset(p, 0, sz[i]));
      *ps[i] = p; c->allocs.push_back(p);
    }
  }
  c->unwhitened = !whiten;
  return 0;
}
int gdrf_set_learn_inducing(gdrf_ctx* c, int on) {
  if (on != 0 && on != 1) return fail(-1, "gdrf_set_learn_inducing", "on must be 0 or 1");
  HIPCHK(hipSetDevice(c->dev));
  if (on && !c->zpart) {
    void* p = nullptr;
    const size_t bytes = (size_t)((c->ncap + GDRF_TILE - 1) / GDRF_TILE) * c->M * c->D * sizeof(double);
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess) return fail(-(int)e - 1000, "hipMalloc(zpart)", hipGetErrorString(e));
    c->zpart = (double*)p; c->allocs.push_back(p);
  }
  c->learn_z = on;
  return 0;
}
int gdrf_get_mfma_mode(const gdrf_ctx* c) { return c->split; }
int gdrf_set_hyper_backward(gdrf_ctx* c, int mode) {
  if (mode != 0 && mode != 1) return fail(-1, "gdrf_set_hyper_backward", "mode must be 0 (f64 backward GEMM) or 1 (Hd = dK^T Wbar on the TN kernel)");
  c->hyper_tn = mode;
  return 0;
}
int gdrf_get_hyper_backward(const gdrf_ctx* c) { return hyper_tn_on(c) ? 1 : 0; }
int gdrf_ws_elem_size(gdrf_ctx* c, int which) { void* p; int64_t n; int e; return ws_lookup(c, which, &p, &n, &e) ? -1 : e; }

int gdrf_set_timing(gdrf_ctx* c, int enable) {
  c->timing = enable;
  for (int i = 0; i < GDRF_NSLOTS; ++i) { c->t_ms[i] = 0; c->t_cnt[i] = 0; }
  return 0;
}
int gdrf_get_timing(gdrf_ctx* c, double* ms_out, int64_t* cnt_out, int nslots) {
  HIPCHK(hipSetDevice(c->dev));
  for (auto& t : c->tev) {
    HIPCHK(hipEventSynchronize(t.second.second));
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, t.second.first, t.second.second));
    c->t_ms[t.first] += ms; c->t_cnt[t.first] += 1;
    c->ev_pool.push_back(t.second.first); c->ev_pool.push_back(t.second.second);
  }
  c->tev.clear();
  for (int i = 0; i < nslots && i < GDRF_NSLOTS; ++i) { ms_out[i] = c->t_ms[i]; cnt_out[i] = c->t_cnt[i]; }
  return 0;
}

// make stream s wait for the factorisation queued on the side stream (no-op before the first gdrf_factorize)
static int join_fact(gdrf_ctx* c, hipStream_t s) {
  if (c->fact_pending) HIPCHK(hipStreamWaitEvent(s, c->ev_fact, 0));
  return 0;
}

int gdrf_ws_copy(gdrf_ctx* c, int which, void* dst, int64_t nelem, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  void* p; int64_t n; int e;
  int rc = ws_lookup(c, which, &p, &n, &e);
  if (rc) return rc;
  if (nelem > n) return fail(-1, "gdrf_ws_copy", "nelem exceeds the buffer");
  if ((rc = join_fact(c, (hipStream_t)stream))) return rc;
  HIPCHK(hipMemcpyAsync(dst, p, (size_t)nelem * e, hipMemcpyDeviceToDevice, (hipStream_t)stream));
  return 0;
}

// ------------------------------------------------------------------------------------------------
// T = N-side element type, TS = solve element type
template <typename T, typename TS> struct Impl {
  using C = NTCfg<T>;
  using CS = NTCfg<TS>;
  static constexpr bool kSame = std::is_same<T, TS>::value;
  static T* P(void* p) { return reinterpret_cast<T*>(p); }
  static const T* P(const void* p) { return reinterpret_cast<const T*>(p); }
  static TS* Q(void* p) { return reinterpret_cast<TS*>(p); }

  // column tiles of the NT core for element type E (128 wide for f32, 64 for f64); row tiles are always 128
  template <typename E> static int nct(const gdrf_ctx* c) { return (c->Mp + NTCfg<E>::CW - 1) / NTCfg<E>::CW; }

  template <typename E>
  static int mm_nt(gdrf_ctx* c, const E* A, int64_t abs_, const E* Bt, int64_t bbs, E* Cm, int64_t cbs, E alpha, int batch,
                   hipStream_t s) {
    // a single M x M product is 16 workgroups of the NT core (45 us at M = 512 in double, four of them in a row in every step's
    // Cholesky backward), a batch of K = 10 is 160 (one per CU for the whole reduction): the reduction is split over S slices into
    // slabs, which are added in double in a fixed order.  S: 8 for a single product, 4 for small batches.
    static const bool splitk = !(getenv("GDRF_MM_SPLITK") && getenv("GDRF_MM_SPLITK")[0] == '0');
    const int64_t mm = (int64_t)c->Mp * c->Mp;
    const int tiles = c->nt * nct<E>(c);
    const int S = batch == 1 ? 8 : 4, kw = c->Mp / S;
    // (batches: measured SLOWER - 160 workgroups already fill most CUs and the slab sum of K matrices costs more than the split saves)
    if (splitk && batch == 1 && c->mmslab && tiles * batch <= 256 && c->Mp >= 256 && c->Mp % S == 0 && kw % NTCfg<E>::BK == 0 &&
        (size_t)S * batch * mm * sizeof(E) <= c->mmslab_bytes) {
      MMProb<E> p{{}, {}, {}, A, abs_, Bt, bbs, (E*)c->mmslab, mm, c->Mp, alpha, kw, batch};
      dim3 grid(tiles, S * batch);
      hipLaunchKernelGGL((gemm_nt_kernel<E, MMProb<E>>), grid, dim3(256), NTCfg<E>::LDS_BYTES, s, p);
      hipLaunchKernelGGL(reduce_slabs_kernel<E>, dim3((c->Mp + 255) / 256, c->Mp, batch), dim3(256), 0, s, (const E*)c->mmslab, S, batch, c->Mp, 0, Cm, GDRF_TILE / 2);
      LAUNCHCHK("mm_nt split-K");
      return 0;
    }
    MMProb<E> p{{}, {}, {}, A, abs_, Bt, bbs, Cm, cbs, c->Mp, alpha, 0, 1};
    dim3 grid(c->nt * nct<E>(c), batch);
    hipLaunchKernelGGL((gemm_nt_kernel<E, MMProb<E>>), grid, dim3(256), NTCfg<E>::LDS_BYTES, s, p);
    LAUNCHCHK("mm_nt");
    return 0;
  }

  // nlev (<= 8) Cholesky attempts in ONE launch, in the N-side precision (what the reference's fp32
  // torch.linalg.cholesky would see): K_uu built once, one workgroup per cumulative jitter; flags in c->flag[8..8+nlev)
  // (slot 0 is the solve factorisation's, so a probe may run on another stream beside gdrf_factorize)
  static int probe(gdrf_ctx* c, const T* Z, const T* params, const double* jitters, int nlev, hipStream_t s) {
    const int Mp = c->Mp, M = c->M;
    ScopedTimer tm(c, 0, s);
    HIPCHK(hipMemsetAsync(c->flag + 8, 0, 32, s));
    hipLaunchKernelGGL(prep_hyper_kernel<T>, dim3(1), dim3(64), 0, s, params, c->hyp_probe);
    dim3 g2((Mp + 255) / 256, Mp);
    hipLaunchKernelGGL(kuu_kernel<T>, g2, dim3(256), 0, s, Z, M, Mp, c->D, c->kind, c->hyp_probe, 0.0, P(c->pK));
    JitterLevels jl;
    for (int l = 0; l < 8; ++l) jl.v[l] = l < nlev ? jitters[l] : 0.0;
    dim3 g3((Mp + 255) / 256, Mp, nlev);
    hipLaunchKernelGGL(level_copies_kernel<T>, g3, dim3(256), 0, s, (const T*)P(c->pK), M, Mp, jl, P(c->pL));
    if (chol_lds_bytes<T>(M) > 48 * 1024)
      HIPCHK(hipFuncSetAttribute((const void*)chol_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_lds_bytes<T>(M)));
    hipLaunchKernelGGL(chol_kernel<T>, dim3(nlev), dim3(1024), chol_lds_bytes<T>(M), s, P(c->pL), M, Mp, c->flag + 8, (int64_t)Mp * Mp);
    LAUNCHCHK("probe");
    return 0;
  }

  // K_uu(+jitter) -> Cholesky(flag) -> L, LT -> Linv, LinvT in the solve precision.  The hyper-parameters and the cast of Z
  // are refreshed on the caller's stream; the factorisation chain (one workgroup for most of its 1.7 ms) runs on the
  // context's side stream so that what follows on the caller's stream and does not need L (the S / B_k transforms and the
  // solve-precision K_nm of gdrf_step_local) overlaps it.  Every consumer of L calls join_fact() first.
  // mode 0: factorise.  mode 1: factorise AHEAD of the step that will use it (right behind the optimizer update, so that the chain runs while
  // the host reads the loss and enqueues the next step) and keep a copy of its inputs.  mode 2: the step's own call - if a mode-1
  // factorisation with this jitter is waiting, only compare its inputs with the current ones on the device (flag[16], read with the
  // failure flag by gdrf_chol_failed: a mismatch makes the caller redo the step, like a wrong jitter guess); otherwise as mode 0.
  static int factorize(gdrf_ctx* c, const T* Z, const T* params, double jitter, hipStream_t s, int mode = 0) {
    const int Mp = c->Mp, M = c->M;
    const int64_t nzs = (int64_t)M * c->D;
    if (mode == 2 && c->prefact_valid && jitter == c->prefact_jitter) {
      hipLaunchKernelGGL(fact_snapshot_kernel<T>, dim3(1), dim3(256), 0, s, params, Z, nzs, P(c->snap), 1, c->flag + 16);
      // the factorisation stays valid for further calls with the same inputs (a predictive evaluation between two steps): every reuse
      // compares again, and any fresh factorisation below invalidates it first
      LAUNCHCHK("factorize (reuse)");
      return 0;
    }
    c->prefact_valid = 0;
    if (mode != 1) HIPCHK(hipMemsetAsync(c->flag + 16, 0, sizeof(int), s));     // this stream's own factorisation: nothing reused, no mismatch to report
    dim3 g2((Mp + 255) / 256, Mp);
    hipLaunchKernelGGL(prep_hyper_kernel<T>, dim3(1), dim3(64), 0, s, params, c->hyp);
    const int64_t nz = (int64_t)M * c->D;
    hipLaunchKernelGGL((cast_kernel<T, TS>), dim3((unsigned)((nz + 255) / 256)), dim3(256), 0, s, nz, Z, Q(c->Zs));
    HIPCHK(hipEventRecord(c->ev_fact0, s));
    hipStream_t f = c->side;
    HIPCHK(hipStreamWaitEvent(f, c->ev_fact0, 0));
    {
      ScopedTimer tm(c, 15, f);
      HIPCHK(hipMemsetAsync(c->flag, 0, 32, f));
      hipLaunchKernelGGL(kuu_kernel<TS>, g2, dim3(256), 0, f, (const TS*)Q(c->Zs), M, Mp, c->D, c->kind, c->hyp, jitter, Q(c->Kuu));
      HIPCHK(hipMemcpyAsync(c->Lw, c->Kuu, (size_t)Mp * Mp * sizeof(TS), hipMemcpyDeviceToDevice, f));
      bool have_dinv = false;
      static const bool panelwise = !(getenv("GDRF_CHOL_PANEL") && getenv("GDRF_CHOL_PANEL")[0] == '0');    // A/B knob: 0 = the single-workgroup kernel
      if (panelwise) {
        // one launch per 32-column panel, one small workgroup per trailing tile (kernels_mm.h: chol_panel_kernel)
        const int nt = (M + 31) / 32;
        for (int k = 0; k < nt; ++k) {
          const int n = nt - k - 1, wgs = n > 0 ? n * (n + 1) / 2 : 1;
          hipLaunchKernelGGL(chol_panel_kernel<TS>, dim3((unsigned)wgs), dim3(128), 0, f, Q(c->Lw), Q(c->Lo), Q(c->Dinv), M, Mp, k, c->flag);
        }
        hipLaunchKernelGGL(finalize_l_kernel<TS>, g2, dim3(256), 0, f, (const TS*)Q(c->Lo), M, Mp, Q(c->L), Q(c->LT));
        have_dinv = true;            // the inverses of the diagonal blocks came out of the panel launches
      } else {
        if (chol_lds_bytes<TS>(M) > 48 * 1024)
          HIPCHK(hipFuncSetAttribute((const void*)chol_kernel<TS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)chol_lds_bytes<TS>(M)));
        hipLaunchKernelGGL(chol_kernel<TS>, dim3(1), dim3(1024), chol_lds_bytes<TS>(M), f, Q(c->Lw), M, Mp, c->flag, (int64_t)0);
        hipLaunchKernelGGL(finalize_l_kernel<TS>, g2, dim3(256), 0, f, (const TS*)Q(c->Lw), M, Mp, Q(c->L), Q(c->LT));
      }
      if (!have_dinv) hipLaunchKernelGGL(trinv_diag_kernel<TS>, dim3(Mp / 32), dim3(64), 0, f, (const TS*)Q(c->L), M, Mp, Q(c->Dinv));
      hipLaunchKernelGGL(trinv_cols_kernel<TS>, dim3(Mp / 32), dim3(1024), 0, f, (const TS*)Q(c->L), (const TS*)Q(c->Dinv), M, Mp, Q(c->Linv),
                         Q(c->LinvT));
    }
    HIPCHK(hipEventRecord(c->ev_fact, f));
    c->fact_pending = 1;
    if (mode == 1) {
      hipLaunchKernelGGL(fact_snapshot_kernel<T>, dim3(1), dim3(256), 0, s, params, Z, nzs, P(c->snap), 0, (int*)nullptr);
      c->prefact_valid = 1; c->prefact_jitter = jitter;
    }
    LAUNCHCHK("factorize");
    return 0;
  }

  static int knm(gdrf_ctx* c, const T* X, int64_t n, const T* Z, const T* params, T* out, int64_t ldo, hipStream_t s) {
    hipLaunchKernelGGL(prep_hyper_kernel<T>, dim3(1), dim3(64), 0, s, params, c->hyp);
    ScopedTimer tm(c, 1, s);
    const int VE = Vec16<T>::N;
    const int vpr = (c->M + VE - 1) / VE, rpp = vpr <= 256 ? 256 / vpr : 1;
    int64_t blocks = (n + 4 * rpp - 1) / (4 * rpp);
    int64_t cap = 256 * 64;          // swept on MI355X (tools/knm_sweep.py): 2048 blocks 0.60-0.65 of HBM peak, 16384 0.67-0.79
    if (const char* e = getenv("GDRF_KNM_BLOCKS")) cap = atoll(e);
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    const char* plain = getenv("GDRF_KNM_PLAIN_STORES");
    if (plain && plain[0] == '1')
      hipLaunchKernelGGL((knm_kernel<T, T, true, false>), dim3((unsigned)blocks), dim3(256), 0, s, X, n, Z, c->M, c->D, c->kind, c->hyp, out, ldo);
    else
      hipLaunchKernelGGL((knm_kernel<T, T, true, true>), dim3((unsigned)blocks), dim3(256), 0, s, X, n, Z, c->M, c->D, c->kind, c->hyp, out, ldo);
    LAUNCHCHK("knm");
    return 0;
  }

  // block scales of the split modes (gemm_split.h): `what` = SPLIT_SC_* groups to refresh from hyp / the maxima
  static void split_scales(gdrf_ctx* c, int what, hipStream_t s) {
    hipLaunchKernelGGL(split_scales_kernel, dim3(1), dim3(64), 0, s, c->split == 2 ? 1 : 0, (const Hyper*)c->hyp, (const unsigned*)c->smx, c->ssc, c->K, what);
  }

  // pieces of S_k^T (and of W in the all-fp32 mode), then tt on the 16-bit matrix path (f32 contexts only)
  template <class SP>
  static int fwd_t_split(gdrf_ctx* c, int64_t n, int64_t rtiles, hipStream_t s) {
    if constexpr (std::is_same<T, float>::value) {
      using E = typename SP::E;
      const int Mp = c->Mp, K = c->K;
      const SplitLay SL{K};
      const int64_t nb = (int64_t)K * Mp * Mp, nw = n * Mp;
      {
        ScopedTimer tm(c, 2, s);
        hipLaunchKernelGGL(split_blocked_kernel<SP>, dim3((unsigned)((nb / 4 + 255) / 256)), dim3(256), 0, s, (const float*)c->ST, nb, Mp, Mp, (E*)c->STh, nb,
                           (const float*)c->ssc + SL.st(0));
        if (c->ssz != 8)       // with the f64 solve, fwd_w's epilogue has already written the pieces of W
          hipLaunchKernelGGL(split_kernel<SP>, dim3((unsigned)((nw / 4 + 255) / 256)), dim3(256), 0, s, (const float*)c->W, nw, (E*)c->Wh,
                             (int64_t)c->ncap * Mp, (const float*)c->ssc + SL.w());
      }
      ScopedTimer tm(c, 5, s);
      // topics per group: as many lower-triangular S^T piece panels (NP x ~0.6 Mp^2 halfwords each) as fit in 2 MB
      const double panel = 0.625 * 2.0 * SP::NP * (double)Mp * Mp;
      int KG = std::max(1, std::min(K, (int)(2.0 * 1024 * 1024 / panel)));
      if (const char* e = getenv("GDRF_FWDT_KG")) { const int v = atoi(e); if (v > 0) KG = std::min(K, v); }   // tuning knob (tools/)
      // two row tiles per 512-thread workgroup (two phase-shifted wave groups, LDS-DMA staging): 6 operand images
      const int64_t pairs = (rtiles + 1) / 2;
      const int rt8 = (int)((pairs + 7) / 8);
      FwdTSplitArgs<SP> a{(const E*)c->Wh, (int64_t)c->ncap * Mp, n, Mp, K, KG, rt8, (const E*)c->STh, nb, (float*)c->tt, c->ldk, (const float*)c->ssc, nullptr};
      constexpr int lds2 = 6 * SplitCfg<SP>::IMG * 2;
#ifdef GDRF_DIAG   // diagnostic builds only (make DIAG=1): these paths synchronise the stream and allocate inside the step
      if (getenv("GDRF_STAMP")) {           // diagnostic: per-phase s_memtime stamps of one workgroup (tools/), never in a timed run
        unsigned long long* d = nullptr;
        HIPCHK(hipMalloc((void**)&d, 2 * 64 * 4 * 8)); HIPCHK(hipMemset(d, 0, 2 * 64 * 4 * 8));
        a.stamps = d;
        HIPCHK(hipFuncSetAttribute((const void*)fwd_t_split_2g_kernel<SP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2 + 4096));
        hipLaunchKernelGGL((fwd_t_split_2g_kernel<SP, true>), dim3((unsigned)(8 * K * rt8)), dim3(512), lds2 + 4096, s, a);
        std::vector<unsigned long long> h(2 * 64 * 4);
        HIPCHK(hipStreamSynchronize(s));
        HIPCHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost)); (void)hipFree(d);
        for (int gpi = 0; gpi < 2; ++gpi) {
          double sm = 0, sv = 0, sb = 0, so = 0; int cnt = 0; std::string line;
          for (int t = 1; t < 40; ++t) {
            const unsigned long long* q = &h[(gpi * 64 + t) * 4];
            const long long m = q[1] - q[0], v = q[2] - q[1], b = q[3] - q[2], o = q[0] - h[(gpi * 64 + t - 1) * 4 + 3];
            sm += m; sv += v; sb += b; so += o; ++cnt;
            line += " " + std::to_string(m) + "/" + std::to_string(o);
          }
          fprintf(stderr, "fwd_t stamps group %d: mean mult %.0f vmcnt %.0f barrier %.0f other-group-phase %.0f | mult/other per chunk:%s\n", gpi, sm / cnt, sv / cnt,
                  sb / cnt, so / cnt, line.c_str());
        }
        return 0;
      }
      if constexpr (SP::NP == 2) {
        if (const char* sw = getenv("GDRF_STAMP_Q4")) {      // diagnostic: s_memtime stamps of wave <value> of one workgroup of the 256 x 256 form
          unsigned long long* d = nullptr;
          HIPCHK(hipMalloc((void**)&d, 64 * 5 * 8)); HIPCHK(hipMemset(d, 0, 64 * 5 * 8));
          FwdTSplitArgs<SP> as = a; as.stamps = d; as.KG = KG | (atoi(sw) << 16);
          constexpr int lds4 = 8 * SplitCfg<SP>::IMG * 2 + 8 * GDRF_TILE * 4 + 64 * 5 * 8;
          HIPCHK(hipFuncSetAttribute((const void*)fwd_t_split_q4_kernel<SP, 17>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4));
          hipLaunchKernelGGL((fwd_t_split_q4_kernel<SP, 17>), dim3((unsigned)(8 * K * rt8)), dim3(1024), lds4, s, as);
          std::vector<unsigned long long> h(64 * 5);
          HIPCHK(hipStreamSynchronize(s));
          HIPCHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost)); (void)hipFree(d);
          double m[5] = {0, 0, 0, 0, 0}; int cnt = 0; std::string line;
          for (int t = 1; t < 24; ++t) {
            const unsigned long long* q = &h[t * 5];
            if (!q[4]) break;
            m[0] += q[1] - q[0]; m[1] += q[2] - q[1]; m[2] += q[3] - q[2]; m[3] += q[4] - q[3]; m[4] += q[0] - h[(t - 1) * 5 + 4]; ++cnt;
            line += " " + std::to_string(q[4] - h[(t - 1) * 5 + 4]);
          }
          if (cnt) fprintf(stderr, "fwd_t q4 stamps wave %d (%d phases): dma+frag issue %.0f  mfma loop %.0f  vmcnt+lgkm %.0f  barrier %.0f  between %.0f | phase lengths:%s\n",
                           atoi(sw), cnt, m[0] / cnt, m[1] / cnt, m[2] / cnt, m[3] / cnt, m[4] / cnt, line.c_str());
          return 0;
        }
      }
#endif
      const char* alt = getenv("GDRF_FWDT_ALTERNATING");          // A/B knob: the phase-alternating form
      if (alt && alt[0] == '1') {
        HIPCHK(hipFuncSetAttribute((const void*)fwd_t_split_2g_kernel<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
        hipLaunchKernelGGL(fwd_t_split_2g_kernel<SP>, dim3((unsigned)(8 * K * rt8)), dim3(512), lds2, s, a);
      } else if (std::is_same<SP, SplitF16>::value && (Mp % 256) == 0 && getenv("GDRF_FWDT_W1") && getenv("GDRF_FWDT_W1")[0] == '1') {     // opt-in: measured slower than the 16-wave form (gemm_fwd_t1.h)
        if constexpr (std::is_same<SP, SplitF16>::value) {       // one wave per SIMD, 64 rows x 256 columns per wave (gemm_fwd_t1.h)
          HIPCHK(hipFuncSetAttribute((const void*)fwd_t_w1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, ft1_lds_bytes()));
          hipLaunchKernelGGL(fwd_t_w1_kernel, dim3((unsigned)(8 * K * rt8)), dim3(256), ft1_lds_bytes(), s, a);
        }
      } else if (SP::NP == 2 && !(getenv("GDRF_FWDT_Q4") && getenv("GDRF_FWDT_Q4")[0] == '0')) {
        if constexpr (SP::NP == 2) {       // 256 x 256 workgroup tiles (two-piece modes: eight operand images in 128 KB)
          constexpr int lds4 = 8 * SplitCfg<SP>::IMG * 2 + 8 * GDRF_TILE * 4;      // + the row-sum slots
          static const int var = getenv("GDRF_Q4_VAR") ? atoi(getenv("GDRF_Q4_VAR")) : 1;      // A/B knob; 1 (requests first) measured best
#define GDRF_Q4(X) { HIPCHK(hipFuncSetAttribute((const void*)fwd_t_split_q4_kernel<SP, X>, hipFuncAttributeMaxDynamicSharedMemorySize, lds4)); \
                     hipLaunchKernelGGL((fwd_t_split_q4_kernel<SP, X>), dim3((unsigned)(8 * K * rt8)), dim3(1024), lds4, s, a); }
          if (var == 1) GDRF_Q4(1) else if (var == 2) GDRF_Q4(2) else if (var == 3) GDRF_Q4(3) else if (var == 5) GDRF_Q4(5) else if (var == 33) GDRF_Q4(33) else GDRF_Q4(0)
#undef GDRF_Q4
        }
      } else {
        HIPCHK(hipFuncSetAttribute((const void*)fwd_t_split_cc_kernel<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds2));
        hipLaunchKernelGGL(fwd_t_split_cc_kernel<SP>, dim3((unsigned)(8 * K * rt8)), dim3(512), lds2, s, a);
      }
      LAUNCHCHK("fwd_t_split");
      return 0;
    } else {
      return fail(-1, "fwd_t_split", "float arrays only");
    }
  }

  // C = Wh^T diag(scale) B over the observations on the 16-bit matrix path (f32 contexts only); A side = the pieces of W.
  // sidx_b / sidx_b_stride: SplitLay pair index of the block scale of the row-scaled B operand (per batch)
  template <class SP>
  static int tn_split(gdrf_ctx* c, const float* B, const float* scale, int64_t scale_bs, int64_t n, int64_t rps, int sym, float* slab,
                      int nbatch, int ns, int ntiles, int sidx_b, int sidx_b_stride, hipStream_t s, const void* Ah = nullptr, int sidx_a = -1) {
    if constexpr (std::is_same<T, float>::value) {
      using E = typename SP::E;
      // stored (pre-split) operand: the pieces of W unless the caller names another [piece][row][Mp] array and its block scale
      TNSplitArgs<SP> a{(const E*)(Ah ? Ah : c->Wh), (int64_t)c->ncap * c->Mp, c->Mp, B, c->Mp, scale, scale_bs, n, rps, c->Mp, sym, slab, nbatch, ns,
                        (const float*)c->ssc, sidx_a >= 0 ? sidx_a : SplitLay{c->K}.w(), sidx_b, sidx_b_stride};
      constexpr int lds = 3 * SP::NP * 32 * 128 * 2;            // double-buffered A image + B image
      HIPCHK(hipFuncSetAttribute((const void*)gemm_tn_split_kernel<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      hipLaunchKernelGGL(gemm_tn_split_kernel<SP>, dim3((unsigned)(ntiles * nbatch * ns)), dim3(256), lds, s, a);
      return 0;
    } else {
      return fail(-1, "tn_split", "float arrays only");
    }
  }

  // Wbar on the 16-bit matrix path (f32 contexts only)
  template <class SP>
  static int wbar_split(gdrf_ctx* c, int64_t n, const T* U, int64_t rtiles, hipStream_t s) {
    if constexpr (std::is_same<T, float>::value) {
      using E = typename SP::E;
      const int Mp = c->Mp, K = c->K;
      const SplitLay SL{K};
      const int64_t nb = (int64_t)K * Mp * Mp;
      hipLaunchKernelGGL(split_blocked_kernel<SP>, dim3((unsigned)((nb / 4 + 255) / 256)), dim3(256), 0, s, (const float*)c->Bm, nb, Mp, Mp, (E*)c->Bh, nb,
                         (const float*)c->ssc + SL.b(0));
      BwdWbarSplitArgs<SP> a{(const float*)c->W, (const E*)c->Wh, (int64_t)c->ncap * Mp, n, c->M, Mp, K, (const E*)c->Bh, nb,
                             (const float*)c->vbar, (const float*)c->locbar, c->ldk, (const float*)c->asum, (const float*)U, (float*)c->Wbar,
                             (const float*)c->ssc, c->smx + SL.mx_wbar()};
      const size_t grp = SplitCfg<SP>::LDS_BYTES + (((size_t)K * GDRF_TILE * sizeof(float) + 15) & ~(size_t)15);
      const int nct_ = (Mp + GDRF_TILE - 1) / GDRF_TILE;
      if (K >= 2 && 2 * grp <= 160 * 1024) {   // two phase-shifted wave groups per workgroup (own A images, shared double-buffered B, LDS-DMA staging)
        const int64_t pairs = (rtiles + 1) / 2;
        const char* alt = getenv("GDRF_WBAR_ALTERNATING");          // A/B knob: the phase-alternating form
        const size_t tabb = ((size_t)K * GDRF_TILE * sizeof(float) + 15) & ~(size_t)15;
        const size_t lds_cc = 6 * (size_t)SplitCfg<SP>::IMG * 2 + 2 * tabb;
#ifdef GDRF_DIAG
        if (getenv("GDRF_STAMP_WBAR")) {      // diagnostic: per-phase s_memtime stamps of one workgroup, never in a timed run
          unsigned long long* d = nullptr;
          HIPCHK(hipMalloc((void**)&d, 2 * 64 * 4 * 8)); HIPCHK(hipMemset(d, 0, 2 * 64 * 4 * 8));
          a.stamps = d;
          HIPCHK(hipFuncSetAttribute((const void*)bwd_wbar_split_cc_kernel<SP, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lds_cc + 4096)));
          hipLaunchKernelGGL((bwd_wbar_split_cc_kernel<SP, true>), dim3((unsigned)round_up(pairs * nct_, 8)), dim3(512), lds_cc + 4096, s, a);
          std::vector<unsigned long long> h(2 * 64 * 4);
          HIPCHK(hipStreamSynchronize(s));
          HIPCHK(hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost)); (void)hipFree(d);
          for (int gpi = 0; gpi < 2; ++gpi) {
            double sd = 0, sm = 0, sv = 0, sb = 0; int cnt = 0; std::string line;
            for (int t = 1; t < 63; ++t) {
              const unsigned long long* q = &h[(gpi * 64 + t) * 4];
              const long long dd = q[1] - q[0], m = q[2] - q[1], v = q[3] - q[2], b = h[(gpi * 64 + t + 1) * 4] - q[3];
              sd += dd; sm += m; sv += v; sb += b; ++cnt;
              if (t < 24) line += " " + std::to_string(dd) + "/" + std::to_string(m) + "/" + std::to_string(v) + "/" + std::to_string(b);
            }
            fprintf(stderr, "wbar_cc stamps group %d: mean dma-issue %.0f mult %.0f vmcnt %.0f barrier %.0f | dma/mult/vmcnt/barrier per phase:%s\n", gpi, sd / cnt,
                    sm / cnt, sv / cnt, sb / cnt, line.c_str());
          }
        } else
#endif
        if (std::is_same<SP, SplitF16>::value && !(alt && alt[0] != '0') && K >= 2 && (Mp % 64) == 0 &&
                   8 * (size_t)SplitCfg<SP>::IMG * 2 + 2 * tabb <= 160 * 1024) {
          if constexpr (std::is_same<SP, SplitF16>::value) {
            const size_t lds64 = 8 * (size_t)SplitCfg<SP>::IMG * 2 + 2 * tabb;
#ifdef GDRF_DIAG
            const char* ab = getenv("GDRF_WBAR_ABLATE");         // timing-only variants with WRONG results: diagnostic builds only
            const int abl = ab ? atoi(ab) : 0;
            if (abl) { static bool warned = false; if (!warned) { warned = true; fprintf(stderr, "libgdrf_hip: GDRF_WBAR_ABLATE=%d is active - Wbar and every gradient are WRONG (timing-only build)\n", abl); } }
#else
            constexpr int abl = 0;
#endif
            // few rows (streaming mini-batches): a handful of workgroups would each walk all K x Mp / 64 chunks one after the other
            // (0.19 ms at n = 64); the reduction blocks are split over gridDim.y slices into slabs, summed in a fixed order
            const int nkb = Mp / 64;
            static const bool wsplit = !(getenv("GDRF_WBAR_SLICES") && getenv("GDRF_WBAR_SLICES")[0] == '0');
            int nslice = 1;
            if (wsplit && pairs * nct_ <= 32 && nkb >= 2 && !abl) nslice = std::min(nkb, 8);
            if ((size_t)nslice * n * Mp * sizeof(float) > (size_t)c->nsplit_cap * (K + 1) * Mp * Mp * sizeof(float)) nslice = 1;   // the TN slab buffer is idle now
            if (nslice > 1) { a.slab = (float*)c->slab; a.slab_stride = (int64_t)round_up(n, 256) * Mp; a.nslice = nslice; }
#define GDRF_K64(X) { HIPCHK(hipFuncSetAttribute((const void*)bwd_wbar_f16_k64_kernel<X>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64)); \
                      hipLaunchKernelGGL(bwd_wbar_f16_k64_kernel<X>, dim3((unsigned)round_up(pairs * nct_, 8), (unsigned)nslice), dim3(512), lds64, s, a); }
#ifdef GDRF_DIAG
            if (abl == 1) GDRF_K64(1) else if (abl == 2) GDRF_K64(2) else if (abl == 3) GDRF_K64(3) else if (abl == 4) GDRF_K64(4)
            else if (abl == 7) GDRF_K64(7) else if (abl == 8) GDRF_K64(8) else if (abl == 16) GDRF_K64(16) else if (abl == 64) GDRF_K64(64) else if (abl == 256) GDRF_K64(256) else
#endif
            GDRF_K64(0)
#undef GDRF_K64
            if (nslice > 1) {
              const int64_t n4 = n * Mp / 4;
              hipLaunchKernelGGL(wbar_slab_sum_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, s, (const float*)a.slab, a.slab_stride, nslice, n4,
                                 (float*)c->Wbar, a.wbar_max);
            }
          }
        } else if (!(alt && alt[0] == '1') && lds_cc <= 160 * 1024) {
          HIPCHK(hipFuncSetAttribute((const void*)bwd_wbar_split_cc_kernel<SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cc));
          hipLaunchKernelGGL((bwd_wbar_split_cc_kernel<SP>), dim3((unsigned)round_up(pairs * nct_, 8)), dim3(512), lds_cc, s, a);
        } else {
          const size_t lds = std::max<size_t>(2 * grp, 8 * 32 * 68 * sizeof(float));      // the epilogue's transposition tiles
          HIPCHK(hipFuncSetAttribute((const void*)bwd_wbar_split_kernel<SP, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL((bwd_wbar_split_kernel<SP, 2>), dim3((unsigned)round_up(pairs * nct_, 8)), dim3(512), lds, s, a);
        }
      } else {
        if (grp > 160 * 1024) return fail(-1, "wbar_split", "too many topics for the LDS scale table");
        const size_t lds = std::max<size_t>(grp, 4 * 32 * 68 * sizeof(float));
        HIPCHK(hipFuncSetAttribute((const void*)bwd_wbar_split_kernel<SP, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((bwd_wbar_split_kernel<SP, 1>), dim3((unsigned)round_up(rtiles * nct_, 8)), dim3(256), lds, s, a);
      }
      LAUNCHCHK("wbar_split");
      return 0;
    } else {
      return fail(-1, "wbar_split", "float arrays only");
    }
  }

  // K_nm in the solve precision with the exact exponential, zero-padded to Mp columns
  static int knm_solve(gdrf_ctx* c, const T* X, int64_t n, hipStream_t s) {
    ScopedTimer tm(c, 1, s);
    const int VE = Vec16<TS>::N;
    const int vpr = (c->Mp + VE - 1) / VE, rpp = vpr <= 256 ? 256 / vpr : 1;
    int64_t blocks = (n + 4 * rpp - 1) / (4 * rpp);
    static const int64_t kcap = getenv("GDRF_KNM_SOLVE_BLOCKS") ? atoll(getenv("GDRF_KNM_SOLVE_BLOCKS")) : 256 * 64;
    if (blocks > kcap) blocks = kcap;
    if (blocks < 1) blocks = 1;
    if constexpr (sizeof(TS) == 8) {
      if (c->kind == 0 && vpr <= 256) {
        if (c->D <= 2) hipLaunchKernelGGL((knm_rbf_f64_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, s, X, n, (const double*)Q(c->Zs), c->M, c->D, c->hyp,
                                          (double*)Q(c->Knm), (int64_t)c->Mp);
        else hipLaunchKernelGGL((knm_rbf_f64_kernel<T, GDRF_DMAX>), dim3((unsigned)blocks), dim3(256), 0, s, X, n, (const double*)Q(c->Zs), c->M, c->D, c->hyp,
                                (double*)Q(c->Knm), (int64_t)c->Mp);
        LAUNCHCHK("knm_solve");
        return 0;
      }
    }
    hipLaunchKernelGGL((knm_kernel<TS, T, false>), dim3((unsigned)blocks), dim3(256), 0, s, X, n, (const TS*)Q(c->Zs), c->M, c->D, c->kind,
                       c->hyp, Q(c->Knm), (int64_t)c->Mp);
    LAUNCHCHK("knm_solve");
    return 0;
  }

  // whiten = False: S' = L^-1 S (into c->S / c->ST, and kept in the solve precision in c->uS), u' = L^-1 u (c->Uw, c->uU)
  static int unwhiten_forward(gdrf_ctx* c, const T* U, hipStream_t s) {
    const int Mp = c->Mp, M = c->M, K = c->K;
    const int64_t mm = (int64_t)Mp * Mp;
    int rc;
    if ((rc = join_fact(c, s))) return rc;
    hipLaunchKernelGGL((cast_kernel<T, TS>), dim3((unsigned)((K * mm + 255) / 256)), dim3(256), 0, s, K * mm, (const T*)P(c->ST), Q(c->uSb));
    if ((rc = mm_nt<TS>(c, Q(c->Linv), 0, Q(c->uSb), mm, Q(c->uS), mm, TS(1), K, s))) return rc;        // S'[i][j] = sum_q Linv[i][q] S[q][j]
    dim3 g3((Mp + 255) / 256, Mp, K);
    hipLaunchKernelGGL((cast_with_transpose_kernel<TS, T>), g3, dim3(256), 0, s, (const TS*)Q(c->uS), Mp, P(c->S), P(c->ST));
    hipLaunchKernelGGL((lower_matvec_kernel<TS, T>), dim3((M + 127) / 128, K), dim3(128), 0, s, (const TS*)Q(c->Linv), U, M, Mp, Q(c->uU),
                       (T*)c->Uw);
    return 0;
  }

  // parts of one evaluation: the parameter transforms, the forward over the rows (K_nm, W, loc, tt), the per-row terms, the backward
  // over the rows.  gdrf_step_local runs them all; the two-point evaluation (step_local2: guide and model on different inputs) runs
  // them selectively.
  enum { SL_TRANSFORMS = 1, SL_FORWARD = 2, SL_ROWS = 4, SL_BACKWARD = 8, SL_ALL = 15, SL_NO_DK = 16 /* forward for the predictive path: no backward follows */ };
  static int step_local(gdrf_ctx* c, const T* X, const int32_t* ws, const T* eps, int64_t n, const T* Z, const T* params,
                        T* redT, double* redd, hipStream_t s, int mask = SL_ALL) {
    const int Mp = c->Mp, M = c->M, K = c->K, V = c->V;
    const int64_t mm = (int64_t)Mp * Mp, ldk = c->ldk;
    int rc;
    const T* U = params + poff(c, 3);
    const T* phi_unc = params + poff(c, 4);
    const T* Sunc = params + poff(c, 5);
    if (c->unwhitened && !(mask & SL_TRANSFORMS)) U = (const T*)c->Uw;
    if (mask & SL_TRANSFORMS) {
      ScopedTimer tm(c, 2, s);
      dim3 g3((Mp + 255) / 256, Mp, K);
      hipLaunchKernelGGL(build_s_kernel<T>, g3, dim3(256), 0, s, Sunc, M, Mp, P(c->S), P(c->ST));
      hipLaunchKernelGGL(build_phi_kernel<T>, dim3(K), dim3(64), 0, s, phi_unc, K, V, P(c->phi));
      if (c->unwhitened) {
        if ((rc = unwhiten_forward(c, U, s))) return rc;     // S, ST now hold S' = L^-1 S; c->Uw holds u' = L^-1 u
        U = (const T*)c->Uw;
      }
      hipLaunchKernelGGL(build_upad_kernel<T>, dim3((Mp + 255) / 256, GDRF_TILE), dim3(256), 0, s, U, K, M, Mp, P(c->Upad));
      if (!c->Tst && (rc = mm_nt<T>(c, P(c->S), mm, P(c->S), mm, P(c->Bm), mm, T(1), K, s))) return rc;      // B_k = S_k S_k^T
      if constexpr (std::is_same<T, float>::value) {
        if (c->split) {            // block scales of W (from the variance), B_k and S_k^T (from their maxima)
          const SplitLay SL{K};
          HIPCHK(hipMemsetAsync(c->smx, 0, (size_t)SL.nmax() * sizeof(unsigned), s));
          if (c->split == 2) {
            hipLaunchKernelGGL(absmax_batched_kernel, dim3(64, K), dim3(256), 0, s, (const float*)c->Bm, mm, mm, c->smx + SL.mx_b(0));
            hipLaunchKernelGGL(absmax_batched_kernel, dim3(64, K), dim3(256), 0, s, (const float*)c->ST, mm, mm, c->smx + SL.mx_st(0));
          }
          split_scales(c, SPLIT_SC_W | SPLIT_SC_BST, s);
        }
      }
    }
    const int64_t rtiles = (n + GDRF_TILE - 1) / GDRF_TILE;
    const bool use_wd = wd_path(c);
    if (mask & SL_FORWARD) {
    // (1) W = Knm Linv^T in the solve precision, stored in the N-side precision
    {
      if ((rc = knm_solve(c, X, n, s))) return rc;
      if ((rc = join_fact(c, s))) return rc;          // W needs L^-1
      ScopedTimer tm(c, 3, s);
      FwdWProb<TS, T> p{{}, {}, (const TS*)Q(c->Knm), n, Mp, (const TS*)Q(c->Linv), P(c->W), P(c->qpart), ldk};
      if (c->split && sizeof(TS) == 8 && sizeof(T) == 4) {      // pieces of W from the same epilogue
        p.Wh = c->Wh; p.wh_stride = (int64_t)c->ncap * Mp; p.wh_mode = c->split; p.wh_scale = c->ssc + SplitLay{K}.w();
      }
#ifdef GDRF_NT_TRACE   // diagnostic builds only: per-workgroup phase stamps of this launch written to $GDRF_NT_TRACE_FILE (tools/nt_trace.py)
      unsigned long long* trace_d = nullptr; const size_t trace_n = (size_t)nt_xcd_row_grid(rtiles, nct<TS>(c)) * 8;
      if (getenv("GDRF_NT_TRACE_FILE") && sizeof(TS) == 8) {
        HIPCHK(hipMalloc((void**)&trace_d, trace_n * 8)); HIPCHK(hipMemset(trace_d, 0, trace_n * 8));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_nt_trace), &trace_d, sizeof(trace_d)));
        HIPCHK(hipDeviceSynchronize());
      }
#endif
      hipLaunchKernelGGL((gemm_nt_kernel<TS, FwdWProb<TS, T>>), dim3(nt_xcd_row_grid(rtiles, nct<TS>(c))), dim3(256), CS::LDS_BYTES, s, p);
#ifdef GDRF_NT_TRACE
      if (trace_d) {
        HIPCHK(hipDeviceSynchronize());
        std::vector<unsigned long long> h(trace_n);
        HIPCHK(hipMemcpy(h.data(), trace_d, trace_n * 8, hipMemcpyDeviceToHost));
        if (FILE* f = fopen(getenv("GDRF_NT_TRACE_FILE"), "wb")) { fwrite(h.data(), 8, trace_n, f); fclose(f); }
        unsigned long long* z = nullptr;
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_nt_trace), &z, sizeof(z)));
        (void)hipFree(trace_d);
      }
#endif
    }
    // loc = W U^T, on the side stream beside fwd_t (both only read W)
    HIPCHK(hipEventRecord(c->ev_fork, s));
    HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
    {
      ScopedTimer tm(c, 4, c->side);
      static const bool loc_rows = !(getenv("GDRF_LOC_ROWS") && getenv("GDRF_LOC_ROWS")[0] == '0');      // A/B knob: 0 = the NT core
      const size_t ulds = (size_t)K * Mp * sizeof(T);
      if (loc_rows && K <= LOC_KMAX && Mp % (16 * Vec16<T>::N) == 0 && ulds <= 150 * 1024) {
        if (ulds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)loc_rows_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ulds));
        const int64_t blocks = std::min<int64_t>((n + 31) / 32, 256 * 8);
        hipLaunchKernelGGL(loc_rows_kernel<T>, dim3((unsigned)blocks), dim3(256), ulds, c->side, (const T*)P(c->W), n, Mp, K, (const T*)P(c->Upad), P(c->loc), ldk);
      } else {
        LocProb<T> p{{}, {}, {}, P(c->W), n, Mp, K, P(c->Upad), P(c->loc), ldk};
        hipLaunchKernelGGL((gemm_nt_kernel<T, LocProb<T>>), dim3((unsigned)rtiles), dim3(256), C::LDS_BYTES, c->side, p);
      }
    }
    HIPCHK(hipEventRecord(c->ev_loc, c->side));
    if constexpr (std::is_same<T, float>::value && sizeof(TS) == 8) {
      if (hyper_tn_on(c) && !(mask & SL_NO_DK)) {
        // pieces of dK_nm / d log(lengthscale) for the backward's Hd = dK^T Wbar; on the side stream behind loc: the main stream does not wait for it
        const int vpr = Mp / 8, rpp = 256 / vpr;
        const int64_t blocks = std::min<int64_t>((n + rpp - 1) / rpp, 256 * 16);
        if (!c->dKh) {
          void* pw = nullptr;
          hipError_t e = hipMalloc(&pw, (size_t)2 * c->ncap * Mp * 2);
          if (e != hipSuccess) return fail(-(int)e - 1000, "hipMalloc(dKh)", hipGetErrorString(e));
          c->dKh = pw; c->allocs.push_back(pw);
        }
        _Float16* dkh = (_Float16*)c->dKh;
        const float* dsc = (const float*)c->ssc + SplitLay{K}.dk();
        if (c->D <= 2) hipLaunchKernelGGL((dk_pieces_kernel<T, 2>), dim3((unsigned)blocks), dim3(256), 0, c->side, X, n, (const double*)c->Zs, M, c->D, c->kind, c->hyp, dkh,
                                          (int64_t)c->ncap * Mp, Mp, dsc);
        else hipLaunchKernelGGL((dk_pieces_kernel<T, GDRF_DMAX>), dim3((unsigned)blocks), dim3(256), 0, c->side, X, n, (const double*)c->Zs, M, c->D, c->kind, c->hyp, dkh,
                                (int64_t)c->ncap * Mp, Mp, dsc);
        LAUNCHCHK("dk_pieces");
      }
    }
    // W' = (dK_nm / d log lengthscale) Linv^T on the side stream, beside the K-fold contractions (its f64 MFMAs fill their stalls);
    // with it the K_nm parts of the hyper-parameter gradients are two dot products with Wbar (wbar_dot_kernel) and the backward
    // GEMM Kbar = Wbar Linv with its pass over K_nm is not needed
    if (use_wd) {
      if (!c->Wd) {
        void* pw = nullptr;
        hipError_t e = hipMalloc(&pw, (size_t)c->ncap * Mp * c->esz);
        if (e != hipSuccess) return fail(-(int)e - 1000, "hipMalloc(Wd)", hipGetErrorString(e));
        c->Wd = pw; c->allocs.push_back(pw);
        HIPCHK(hipMalloc(&pw, (size_t)2 * 2048 * sizeof(double))); c->wdpart = (double*)pw; c->allocs.push_back(pw);
      }
      ScopedTimer tm(c, 8, c->side);
      const size_t zl = (size_t)Mp * (c->D <= 2 ? 2 : 4) * sizeof(TS);
      if (c->D <= 2) {
        FwdWProb<TS, T, true, 2> p{{}, {}, (const TS*)Q(c->Knm), n, Mp, (const TS*)Q(c->Linv), P(c->Wd), nullptr, 0, X, (const TS*)Q(c->Zs), c->hyp, M, c->D, c->kind};
        HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_kernel<TS, FwdWProb<TS, T, true, 2>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CS::LDS_BYTES + zl)));
        hipLaunchKernelGGL((gemm_nt_kernel<TS, FwdWProb<TS, T, true, 2>>), dim3(nt_xcd_row_grid(rtiles, nct<TS>(c))), dim3(256), CS::LDS_BYTES + zl, c->side, p);
      } else {
        FwdWProb<TS, T, true, 4> p{{}, {}, (const TS*)Q(c->Knm), n, Mp, (const TS*)Q(c->Linv), P(c->Wd), nullptr, 0, X, (const TS*)Q(c->Zs), c->hyp, M, c->D, c->kind};
        HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_kernel<TS, FwdWProb<TS, T, true, 4>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(CS::LDS_BYTES + zl)));
        hipLaunchKernelGGL((gemm_nt_kernel<TS, FwdWProb<TS, T, true, 4>>), dim3(nt_xcd_row_grid(rtiles, nct<TS>(c))), dim3(256), CS::LDS_BYTES + zl, c->side, p);
      }
      LAUNCHCHK("fwd_wd");
    }
    // (2) tt_kn = ||S_k^T w_n||^2
    if (c->split) {
      if ((rc = (c->split == 2 ? fwd_t_split<SplitF16>(c, n, rtiles, s) : fwd_t_split<SplitBf16>(c, n, rtiles, s)))) return rc;
    } else {
      ScopedTimer tm(c, 5, s);
      FwdTProb<T> p{{K}, {}, {}, P(c->W), n, Mp, P(c->ST), P(c->tt), ldk, P(c->Tst), c->t_bs, c->t_ts};
      hipLaunchKernelGGL((gemm_nt_kernel<T, FwdTProb<T>>), dim3((unsigned)(8 * K * ((rtiles + 7) / 8))), dim3(256), C::LDS_BYTES, s, p);
    }
    LAUNCHCHK("forward");
    HIPCHK(hipStreamWaitEvent(s, c->ev_loc, 0));
    }
    // per-row ELBO terms and row-local backward
    int egrid;
    if (mask & SL_ROWS) {
      ScopedTimer tm(c, 6, s);
      // matrix-core form (rows_mfma.h): 16 rows per wave, the three K x V products of a row block as 16x16x4 matrix instructions on
      // register-resident operands; K <= 32, V <= 64.  GDRF_ROWS_MFMA=0: the one-thread-per-row kernel (which serves the other sizes)
      static const bool rows_mfma = !(getenv("GDRF_ROWS_MFMA") && getenv("GDRF_ROWS_MFMA")[0] == '0');
      if (rows_mfma && K <= 32 && V <= 64) {
        const int nkt = K <= 16 ? 1 : 2, nvt = V <= 32 ? 2 : 4;
        const size_t lds = rows_mfma_lds<T>(K, V, nkt, nvt, 4);
        const int64_t groups = (n + 15) / 16;
        egrid = (int)std::min<int64_t>((groups + 3) / 4, c->erows_grid_cap);
#define GDRF_RM(KT, VT) { if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)elbo_rows_mfma_kernel<T, KT, VT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                          hipLaunchKernelGGL((elbo_rows_mfma_kernel<T, KT, VT>), dim3(egrid), dim3(256), lds, s, n, K, V, c->hyp, P(c->qpart), nct<TS>(c), P(c->loc), P(c->tt), eps, ldk, n, ws, \
                                             P(c->phi), (const T*)c->mean, c->mean_sk, c->mean_sn, P(c->q), P(c->vbar), P(c->locbar), P(c->asum), P(c->mu), c->dpart, P(c->phibar_part)); }
        if (nkt == 1) { if (nvt == 2) GDRF_RM(1, 2) else GDRF_RM(1, 4) } else { if (nvt == 2) GDRF_RM(2, 2) else GDRF_RM(2, 4) }
#undef GDRF_RM
        LAUNCHCHK("elbo_rows (mfma)");
        hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->dpart, (int64_t)egrid, 4, redd);
        hipLaunchKernelGGL(reduce_parts_kernel<T>, dim3((K * V + 255) / 256), dim3(256), 0, s, P(c->phibar_part), (int64_t)egrid,
                           (int64_t)K * V, redT + roff(c, 1));
      } else {
      const bool kreg = K <= GDRF_KMAX;
      auto lds_for = [&](int rb, bool wsep) {
        return 128 + ((size_t)2 * K * V + (size_t)rb * (K + 1) * (kreg ? 1 : 2) + (size_t)rb * (V + 1) * (wsep ? 2 : 1)) * sizeof(T);
      };
      int RB = 128;
      const char* wst = getenv("GDRF_ROWS_STAGE_WS");
      bool wsep = kreg && wst && wst[0] == '1' && lds_for(128, true) <= 150 * 1024;      // A/B knob: the counts staged through LDS (measured slower)
      while (RB > 32 && lds_for(RB, wsep) > 150 * 1024) RB >>= 1;
      const size_t lds = lds_for(RB, wsep);
      if (lds > 150 * 1024)
        return fail(-1, "gdrf_step_local", "num_topic_categories x num_observation_categories too large: the row kernel keeps the "
                                           "(K, V) word-topic matrix and its gradient in LDS (2*K*V + 32*(2K + V + 3) elements <= 150 KB)");
      const void* kfn = !kreg ? (const void*)elbo_rows_kernel<T, false, false>
                              : (wsep ? (const void*)elbo_rows_kernel<T, true, true> : (const void*)elbo_rows_kernel<T, true, false>);
      if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute(kfn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      int64_t nblk = (n + RB - 1) / RB;
      egrid = (int)std::min<int64_t>(nblk, c->erows_grid_cap);
#define GDRF_ROWS_ARGS n, K, V, c->hyp, P(c->qpart), nct<TS>(c), P(c->loc), P(c->tt), eps, ldk, n, ws, P(c->phi), (const T*)c->mean, c->mean_sk, c->mean_sn, \
                       P(c->q), P(c->vbar), P(c->locbar), P(c->asum), P(c->mu), c->dpart, P(c->phibar_part)
      if (!kreg) hipLaunchKernelGGL((elbo_rows_kernel<T, false, false>), dim3(egrid), dim3(RB), lds, s, GDRF_ROWS_ARGS);
      else if (wsep) hipLaunchKernelGGL((elbo_rows_kernel<T, true, true>), dim3(egrid), dim3(RB), lds, s, GDRF_ROWS_ARGS);
      else hipLaunchKernelGGL((elbo_rows_kernel<T, true, false>), dim3(egrid), dim3(RB), lds, s, GDRF_ROWS_ARGS);
#undef GDRF_ROWS_ARGS
      LAUNCHCHK("elbo_rows");
      hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->dpart, (int64_t)egrid, 4, redd);
      hipLaunchKernelGGL(reduce_parts_kernel<T>, dim3((K * V + 255) / 256), dim3(256), 0, s, P(c->phibar_part), (int64_t)egrid,
                         (int64_t)K * V, redT + roff(c, 1));
      }
    }
    if (!(mask & SL_BACKWARD)) return 0;
    if constexpr (std::is_same<T, float>::value) {
      if (c->split) {            // block scale of diag(vbar_k) W, the scaled operand of the A_k contraction
        const SplitLay SL{K};
        if (c->split == 2) {
          HIPCHK(hipMemsetAsync(c->smx + SL.mx_v(0), 0, (size_t)K * sizeof(unsigned), s));
          HIPCHK(hipMemsetAsync(c->smx + SL.mx_wbar(), 0, sizeof(unsigned), s));
          hipLaunchKernelGGL(absmax_batched_kernel, dim3(64, K), dim3(256), 0, s, (const float*)c->vbar, n, ldk, c->smx + SL.mx_v(0));
        }
        split_scales(c, SPLIT_SC_V, s);
      }
    }
    // ubar = locbar W needs only the row kernel's locbar: on the side stream BESIDE the Wbar contraction (a vector / HBM pass next to a
    // matrix-pipe kernel that leaves half of each SIMD's registers free) instead of inside bwd_knm with G^T (GDRF_UBAR_EARLY=0: there)
    static const bool ubar_early = !(getenv("GDRF_UBAR_EARLY") && getenv("GDRF_UBAR_EARLY")[0] == '0');
    auto launch_ubar = [&](hipStream_t ss) -> int {
      ScopedTimer tm(c, 12, ss);
      const int64_t rpb = ubar_rows_per_block(n), nb = (n + rpb - 1) / rpb;
      if (nb > c->ubar_blocks_cap) return fail(-1, "gdrf_step_local", "ubar partial buffer too small");
      const int kq = K <= 16 ? (K + 3) / 4 : 4;
#define GDRF_UBAR(Q4) hipLaunchKernelGGL((ubar_part_kernel<T, Q4>), dim3((unsigned)nb, (Mp + 255) / 256), dim3(256), 0, ss, P(c->W), n, Mp, K, P(c->locbar), \
                                          ldk, rpb, P(c->ubar_part))
      if (kq == 1) GDRF_UBAR(1); else if (kq == 2) GDRF_UBAR(2); else if (kq == 3) GDRF_UBAR(3); else GDRF_UBAR(4);
#undef GDRF_UBAR
      hipLaunchKernelGGL(reduce_parts_kernel<T>, dim3((K * Mp + 255) / 256), dim3(256), 0, ss, P(c->ubar_part), nb, (int64_t)K * Mp,
                         redT + roff(c, 0));
      return 0;
    };
    if (ubar_early) {
      HIPCHK(hipEventRecord(c->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
      if ((rc = launch_ubar(c->side))) return rc;
    }
    // (3) Wbar
    {
      ScopedTimer tm(c, 7, s);
      const size_t lds = C::LDS_BYTES + (size_t)K * GDRF_TILE * sizeof(T);
      const dim3 grid(c->Tst ? nt_xcd_pair_grid(rtiles, nct<T>(c)) : (unsigned)round_up(rtiles * nct<T>(c), 8));
      if (c->Tst) {
        BwdWbarTProb<T> p{{}, {}, P(c->Tst), c->t_bs, c->t_ts, P(c->W), n, M, Mp, K, P(c->S), P(c->vbar), P(c->locbar), ldk,
                          P(c->asum), U, P(c->Wbar)};
        if (lds > 48 * 1024)
          HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_kernel<T, BwdWbarTProb<T>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gemm_nt_kernel<T, BwdWbarTProb<T>>), grid, dim3(256), lds, s, p);
      } else if (c->split) {
        if ((rc = (c->split == 2 ? wbar_split<SplitF16>(c, n, U, rtiles, s) : wbar_split<SplitBf16>(c, n, U, rtiles, s)))) return rc;
        split_scales(c, SPLIT_SC_WBAR, s);           // max |Wbar| came out of the epilogue: scale of the G^T contraction's operand
      } else {
        BwdWbarProb<T> p{{}, {}, P(c->W), n, M, Mp, K, P(c->Bm), P(c->vbar), P(c->locbar), ldk, P(c->asum), U, P(c->Wbar)};
        if (lds > 48 * 1024)
          HIPCHK(hipFuncSetAttribute((const void*)gemm_nt_kernel<T, BwdWbarProb<T>>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL((gemm_nt_kernel<T, BwdWbarProb<T>>), grid, dim3(256), lds, s, p);
      }
    }
    // Wbar is ready: G^T = W^T Wbar and ubar = locbar W go to the side stream, where they fill the last-round tails of
    // bwd_knm and the A_k contraction on the main stream
    HIPCHK(hipEventRecord(c->ev_fork2, s));
    HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork2, 0));
    {
      hipStream_t ss = c->side;
      const int BR = TNCfg<T>::BR;
      const int ns = std::min(c->split ? tn_nsplit_gt(c, n, BR) : tn_nsplit(c, n, BR, 3), c->nsplit_cap);
      const int64_t rps = round_up((n + ns - 1) / ns, BR);
      T* slab_gt = P(c->slab) + (int64_t)c->nsplit_cap * K * mm;           // the (K+1)-th batch region of the slab buffer
      TNArgs<T> b{P(c->W), Mp, P(c->Wbar), Mp, nullptr, 0, n, rps, Mp, 0, slab_gt, 1, ns};
      { ScopedTimer tm(c, 10, ss);
        if (c->split) {
          const int ib = SplitLay{K}.wbar();
          if ((rc = (c->split == 2 ? tn_split<SplitF16>(c, (const float*)c->Wbar, nullptr, 0, n, rps, 0, (float*)slab_gt, 1, ns, c->nt * c->nt, ib, 0, ss)
                                   : tn_split<SplitBf16>(c, (const float*)c->Wbar, nullptr, 0, n, rps, 0, (float*)slab_gt, 1, ns, c->nt * c->nt, ib, 0, ss)))) return rc;
        } else {
          hipLaunchKernelGGL(gemm_tn_kernel<T>, dim3((unsigned)(c->nt * c->nt * ns)), dim3(256), TNCfg<T>::LDS_BYTES, ss, b);
        } }
      { ScopedTimer tm(c, 11, ss);
        dim3 gr1((Mp + 255) / 256, Mp, 1);
        hipLaunchKernelGGL(reduce_slabs_kernel<T>, gr1, dim3(256), 0, ss, (const T*)slab_gt, ns, 1, Mp, 0, redT + roff(c, 3)); }
      if (!ubar_early && (rc = launch_ubar(ss))) return rc;
    }
    HIPCHK(hipEventRecord(c->ev_join, c->side));
    // (4) kernel hyper-parameter partials through K_nm
    bool hyper_done = false;
    if constexpr (std::is_same<T, float>::value && sizeof(TS) == 8) {
      if (hyper_tn_on(c)) {
        // on the side stream, behind G^T: Hd = dK^T Wbar on the same TN kernel (slab batch K + 1), its contraction with L^-1 in double ->
        // red_d[5]; sum Wbar o W -> red_d[4]; red_d[6] = 0.  No backward solve GEMM (hyper_tn.h).
        hipStream_t ss = c->side;
        ScopedTimer tm(c, 8, ss);
        const int BR = TNCfg<T>::BR;
        const int ns = std::min(tn_nsplit_gt(c, n, BR), c->nsplit_cap);
        const int64_t rps = round_up((n + ns - 1) / ns, BR);
        float* slab_hd = (float*)c->slab + (int64_t)c->nsplit_cap * (K + 1) * mm;
        const SplitLay SL{K};
        if ((rc = tn_split<SplitF16>(c, (const float*)c->Wbar, nullptr, 0, n, rps, 0, slab_hd, 1, ns, c->nt * c->nt, SL.wbar(), 0, ss, c->dKh, SL.dk()))) return rc;
        hipLaunchKernelGGL(slab_linvt_dot_kernel<TS>, dim3((unsigned)M), dim3(256), 0, ss, (const float*)slab_hd, ns, Mp, M, (const TS*)Q(c->LinvT), c->hpart);
        hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, ss, c->hpart, (int64_t)M, 1, redd + 5);
        hipLaunchKernelGGL(wbar_w_dot_kernel<T>, dim3(2048), dim3(256), 0, ss, (const T*)P(c->Wbar), (const T*)P(c->W), n, Mp, c->hpart + Mp);
        hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, ss, c->hpart + Mp, (int64_t)2048, 1, redd + 4);
        HIPCHK(hipMemsetAsync(redd + 6, 0, sizeof(double), ss));
        LAUNCHCHK("hyper_tn");
        HIPCHK(hipEventRecord(c->ev_join, c->side));           // the join event now also covers these
        hyper_done = true;
      }
    }
    if (hyper_done) {
    } else if (use_wd) {
      // on the side stream, behind W' and G^T / ubar: sum Wbar o W -> red_d[4], sum Wbar o W' -> red_d[5]; red_d[6] = 0
      hipStream_t ss = c->side;
      hipLaunchKernelGGL(wbar_dot_kernel<T>, dim3(2048), dim3(256), 0, ss, (const T*)P(c->Wbar), (const T*)P(c->W), (const T*)P(c->Wd), n, Mp, c->wdpart);
      hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, ss, c->wdpart, (int64_t)2048, 2, redd + 4);
      HIPCHK(hipMemsetAsync(redd + 6, 0, sizeof(double), ss));
      HIPCHK(hipEventRecord(c->ev_join, c->side));           // the join event now also covers these
    } else {
      ScopedTimer tm(c, 8, s);
      const int64_t nb = nt_xcd_row_grid(rtiles, nct<TS>(c));
      if (3 * nb > c->dpart_len) return fail(-1, "gdrf_step_local", "n_local exceeds the context capacity");
      if (c->learn_z) {
        BwdKnmProb<TS, T, true> p{{}, {}, P(c->Wbar), n, M, Mp, c->D, c->kind, (const TS*)Q(c->LinvT), (const TS*)Q(c->Knm), X, (const TS*)Q(c->Zs),
                                  c->hyp, c->dpart, c->zpart};
        hipLaunchKernelGGL((gemm_nt_kernel<TS, BwdKnmProb<TS, T, true>>), dim3((unsigned)nb), dim3(256), CS::LDS_BYTES, s, p);
        hipLaunchKernelGGL(reduce_parts_kernel<double>, dim3((unsigned)((M * c->D + 255) / 256)), dim3(256), 0, s, (const double*)c->zpart, rtiles,
                           (int64_t)M * c->D, redd + 8);
      } else {
        BwdKnmProb<TS, T> p{{}, {}, P(c->Wbar), n, M, Mp, c->D, c->kind, (const TS*)Q(c->LinvT), (const TS*)Q(c->Knm), X, (const TS*)Q(c->Zs), c->hyp,
                            c->dpart, nullptr};
        // capped at 160 registers where the LDS-transposed f64 epilogue runs: two of its waves then share a SIMD with one wave of
        // G^T's TN contraction on the side stream (192 registers), which fills this kernel's stalls instead of queueing behind it
        if (sizeof(TS) == 8) hipLaunchKernelGGL((gemm_nt_kernel_v160<TS, BwdKnmProb<TS, T>>), dim3((unsigned)nb), dim3(256), CS::LDS_BYTES, s, p);
        else hipLaunchKernelGGL((gemm_nt_kernel<TS, BwdKnmProb<TS, T>>), dim3((unsigned)nb), dim3(256), CS::LDS_BYTES, s, p);
      }
      hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->dpart, nb, 3, redd + 4);      // red_d[4..6]
    }
    LAUNCHCHK("backward");
    // (5) A_k = W^T diag(vbar_k) W.  It needs W and vbar only - not Wbar - and could run on a third stream BESIDE the f64 backward GEMM
    // (GDRF_AK_ASIDE=1).  Measured (profiles/r03): no gain - 48.57 vs 48.71 ms/step; the two kernels time-slice the CUs (A_k 9.2 -> 18.0 ms,
    // bwd_knm 8.7 -> 10.7 ms beside each other) instead of filling each other's stalls as the register-capped G^T kernel does.  Off.
    static const bool ak_aside = getenv("GDRF_AK_ASIDE") && getenv("GDRF_AK_ASIDE")[0] == '1';
    const bool ak_on_side = ak_aside && !hyper_done && !use_wd && c->split == 2 && sizeof(TS) == 8;
    hipStream_t s_main = s;
    if (ak_on_side) {
      HIPCHK(hipStreamWaitEvent(c->side2, c->ev_fork2, 0));       // recorded on the caller's stream right behind the Wbar contraction
      s = c->side2;
    }
    {
      const int BR = TNCfg<T>::BR;
      const int ns = std::min(tn_nsplit(c, n, BR, c->split ? 2 : 3), c->nsplit_cap);
      const int64_t rps = round_up((n + ns - 1) / ns, BR);
      TNArgs<T> a{P(c->W), Mp, P(c->W), Mp, P(c->vbar), ldk, n, rps, Mp, 1, P(c->slab), K, ns};
      int red_ns = ns, red_qd = GDRF_TILE / 2;
      { ScopedTimer tm(c, 9, s);
        if (tn_topics_on(c)) {
          if constexpr (std::is_same<T, float>::value) {
            const SplitLay SL{K};
            const int nst = std::min(tn_topics_nsplit(c, n), c->nsplit_cap);
            const int64_t rpst = round_up((n + nst - 1) / nst, 32);
            const int ntl = tnt_ntiles(Mp), kgroups = (K + TNT_KT - 1) / TNT_KT;
            TNTopicsArgs ta{(const _Float16*)c->Wh, (int64_t)c->ncap * Mp, Mp, (const float*)c->W, Mp, (const float*)c->vbar, ldk, n, rpst, Mp,
                            (float*)c->slab, K, nst, ntl, (const float*)c->ssc, SL.w(), SL.v(0)};
            // The one-wave forms run every stage of their 10-topic groups whatever K: with more than 40 % of the topic slots empty (K <= 5, K = 11 .. 12)
            // the two-wave form, which walks the topic pairs that exist, is the faster one.
            const int w1 = getenv("GDRF_TNT_W1") ? atoi(getenv("GDRF_TNT_W1")) : (10 * K >= 6 * kgroups * TNT_KT ? 2 : 0);   // one-wave-per-SIMD forms (gemm_tn_topics1.h): 2 = 128 rows per wave, 1 = 64; 0: the two-wave form
            if (w1) {
              int ns1 = c->nsplit_cap < 64 ? c->nsplit_cap : 64;                            // 8 k splits: every XCD owns whole splits
              if (const char* e = getenv("GDRF_TNT_NSPLIT")) { const int v = atoi(e); if (v > 0 && v <= c->nsplit_cap) ns1 = v; }
              while (ns1 > 8 && (n + ns1 - 1) / ns1 < 8 * TN1_CH) ns1 -= 8;                 // at least 8 chunks per split
              if (ns1 >= 8) ns1 &= ~7;
              const int64_t rps1 = round_up((n + ns1 - 1) / ns1, TN1_CH);
              const int ntl1 = w1 == 2 ? tnt_ntiles(Mp) : tn1_ntiles(Mp);
              const int64_t lds64 = round_up(c->ncap, 64);
              hipLaunchKernelGGL(tn1_scale_rows_kernel, dim3(256, K), dim3(256), 0, s, (const float*)c->vbar, ldk, n, (float*)c->vbs, lds64,
                                 (const float*)c->ssc, SL.v(0));
              TNTopicsArgs t1{(const _Float16*)c->Wh, (int64_t)c->ncap * Mp, Mp, (const float*)c->W, Mp, (const float*)c->vbs, lds64, n, rps1, Mp,
                              (float*)c->slab, K, ns1, ntl1, (const float*)c->ssc, SL.w(), SL.v(0)};
              if (w1 == 2) {
                // two launches: the tiles whose upper 64 rows lie above the diagonal (J = 2 I + 1) multiply half the A tiles (gemm_tn_topics1.h)
                const int nth = tn2_ntiles_half(Mp);
                TNTopicsArgs t0 = t1, t4 = t1;
                t0.ntiles = ntl1 - nth; t4.ntiles = nth;
                HIPCHK(hipFuncSetAttribute((const void*)tn_topics_w2_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, tn2_lds_bytes()));
                HIPCHK(hipFuncSetAttribute((const void*)tn_topics_w2_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, tn2_lds_bytes()));
                if (t0.ntiles > 0) hipLaunchKernelGGL(tn_topics_w2_kernel<0>, dim3((unsigned)(t0.ntiles * kgroups * ns1)), dim3(256), tn2_lds_bytes(), s, t0);
                if (t4.ntiles > 0) hipLaunchKernelGGL(tn_topics_w2_kernel<4>, dim3((unsigned)(t4.ntiles * kgroups * ns1)), dim3(256), tn2_lds_bytes(), s, t4);
              } else {
                HIPCHK(hipFuncSetAttribute((const void*)tn_topics_w1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, tn1_lds_bytes()));
                hipLaunchKernelGGL(tn_topics_w1_kernel, dim3((unsigned)(ntl1 * kgroups * ns1)), dim3(256), tn1_lds_bytes(), s, t1);
              }
              LAUNCHCHK("tn_topics_w1");
              red_ns = ns1; red_qd = 32;
            } else {
            static const int pph = getenv("GDRF_TNT_PPH") ? atoi(getenv("GDRF_TNT_PPH")) : 3;     // topic pairs per phase (A/B knob)
#define GDRF_TNT(X) { HIPCHK(hipFuncSetAttribute((const void*)tn_topics_f16_kernel<X>, hipFuncAttributeMaxDynamicSharedMemorySize, tnt_lds_bytes(X))); \
                      hipLaunchKernelGGL(tn_topics_f16_kernel<X>, dim3((unsigned)(ntl * kgroups * nst)), dim3(512), tnt_lds_bytes(X), s, ta); }
            if (pph == 1) GDRF_TNT(1) else if (pph == 2) GDRF_TNT(2) else GDRF_TNT(3)
#undef GDRF_TNT
            LAUNCHCHK("tn_topics");
            red_ns = nst; red_qd = 32;
            }
          }
        } else if (c->split) {
          const int ib = SplitLay{K}.v(0), ntl = c->nt * (c->nt + 1) / 2;
          if ((rc = (c->split == 2 ? tn_split<SplitF16>(c, (const float*)c->W, (const float*)c->vbar, ldk, n, rps, 1, (float*)c->slab, K, ns, ntl, ib, 2, s)
                                   : tn_split<SplitBf16>(c, (const float*)c->W, (const float*)c->vbar, ldk, n, rps, 1, (float*)c->slab, K, ns, ntl, ib, 2, s)))) return rc;
        } else {
          hipLaunchKernelGGL(gemm_tn_kernel<T>, dim3((unsigned)(c->nt * (c->nt + 1) / 2 * K * ns)), dim3(256), TNCfg<T>::LDS_BYTES, s, a);
        } }
      { ScopedTimer tm(c, 11, s);
        dim3 gr((Mp + 255) / 256, Mp, K);
        hipLaunchKernelGGL(reduce_slabs_kernel<T>, gr, dim3(256), 0, s, P(c->slab), red_ns, K, Mp, 1, redT + roff(c, 2), red_qd); }
    }
    if (ak_on_side) {
      HIPCHK(hipEventRecord(c->ev_ak_done, c->side2));
      s = s_main;
      HIPCHK(hipStreamWaitEvent(s, c->ev_ak_done, 0));
    }
    HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
    LAUNCHCHK("reductions");
    return 0;
  }

  // Guide and model evaluated at DIFFERENT inputs (the reference's quirk Q3, sparse_gdrf.py:376-380): forward at the guide's inputs,
  // keep its (loc, tt, q); forward at the model's; two-point row terms; backward through the model-side predictive; forward at the
  // guide's inputs again (the N x M intermediates are not kept twice) and backward through the guide-side predictive; the two
  // payloads add.  ~2.5 x the cost of a step - this path exists for parity with the reference on non-unit worlds, not for speed.
  static int step_local2(gdrf_ctx* c, const T* Xm, const T* Xg, const int32_t* ws, const T* eps, int64_t n, const T* Z, const T* params,
                         T* redT, double* redd, hipStream_t s) {
    const int K = c->K, V = c->V;
    const int64_t ldk = c->ldk, kn = (int64_t)K * ldk, nq = (int64_t)((c->Mp + 63) / 64) * ldk;
    const int64_t nT = roff(c, 4), nd = 8 + (int64_t)c->M * c->D;
    int rc;
    if (!c->g_loc) {
      void** ps[] = {&c->g_loc, &c->g_tt, &c->g_qpart, &c->g_vbar, &c->g_locbar, &c->g_asum, &c->g_redT, (void**)&c->g_redd};
      const size_t sz[] = {(size_t)kn * c->esz, (size_t)kn * c->esz, (size_t)nq * c->esz, (size_t)kn * c->esz, (size_t)kn * c->esz,
                           (size_t)ldk * c->esz, (size_t)nT * c->esz, (size_t)nd * sizeof(double)};
      for (int i = 0; i < 8; ++i) {
        void* p = nullptr;
        hipError_t e = hipMalloc(&p, sz[i]);
        if (e != hipSuccess) return fail(-(int)e - 1000, "hipMalloc(two-point scratch)", hipGetErrorString(e));
        *ps[i] = p; c->allocs.push_back(p);
      }
    }
    if ((rc = step_local(c, Xg, ws, eps, n, Z, params, redT, redd, s, SL_TRANSFORMS | SL_FORWARD))) return rc;
    HIPCHK(hipMemcpyAsync(c->g_loc, c->loc, (size_t)kn * c->esz, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->g_tt, c->tt, (size_t)kn * c->esz, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->g_qpart, c->qpart, (size_t)nq * c->esz, hipMemcpyDeviceToDevice, s));
    if ((rc = step_local(c, Xm, ws, eps, n, Z, params, redT, redd, s, SL_FORWARD))) return rc;
    {
      ScopedTimer tm(c, 6, s);
      const int RB = 64;
      const size_t lds = 128 + ((size_t)2 * K * V + (size_t)2 * RB * (K + 1) + (size_t)RB * (V + 1)) * sizeof(T);
      if (lds > 150 * 1024) return fail(-1, "gdrf_step_local2", "num_topic_categories x num_observation_categories too large for the row kernel's LDS");
      if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)elbo_rows2_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int egrid = (int)std::min<int64_t>((n + RB - 1) / RB, c->erows_grid_cap);
      hipLaunchKernelGGL(elbo_rows2_kernel<T>, dim3(egrid), dim3(RB), lds, s, n, K, V, c->hyp, nct<TS>(c), P(c->qpart), P(c->loc), P(c->tt),
                         (const T*)c->g_qpart, (const T*)c->g_loc, (const T*)c->g_tt, eps, ldk, n, ws, P(c->phi), (const T*)c->mean, c->mean_sk,
                         c->mean_sn, (const T*)c->mean_g, c->mean_g_sk, c->mean_g_sn, P(c->q), P(c->vbar), P(c->locbar), P(c->asum),
                         (T*)c->g_vbar, (T*)c->g_locbar, (T*)c->g_asum, P(c->mu), c->dpart, P(c->phibar_part));
      LAUNCHCHK("elbo_rows2");
      hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->dpart, (int64_t)egrid, 4, redd);
      hipLaunchKernelGGL(reduce_parts_kernel<T>, dim3((K * V + 255) / 256), dim3(256), 0, s, P(c->phibar_part), (int64_t)egrid,
                         (int64_t)K * V, redT + roff(c, 1));
    }
    // backward through the model-side predictive (the buffers hold its W), payload aside
    if ((rc = step_local(c, Xm, ws, eps, n, Z, params, redT, redd, s, SL_BACKWARD))) return rc;
    HIPCHK(hipMemcpyAsync(c->g_redT, redT, (size_t)nT * c->esz, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(c->g_redd, redd, (size_t)nd * sizeof(double), hipMemcpyDeviceToDevice, s));
    // the guide side: its forward again, its row-local gradients in place of the model's
    if ((rc = step_local(c, Xg, ws, eps, n, Z, params, redT, redd, s, SL_FORWARD))) return rc;
    std::swap(c->vbar, c->g_vbar); std::swap(c->locbar, c->g_locbar); std::swap(c->asum, c->g_asum);
    rc = step_local(c, Xg, ws, eps, n, Z, params, redT, redd, s, SL_BACKWARD);
    std::swap(c->vbar, c->g_vbar); std::swap(c->locbar, c->g_locbar); std::swap(c->asum, c->g_asum);
    if (rc) return rc;
    // payload = model side + guide side: ubar, A_k, G^T (not the phibar block, which the row kernel wrote once) and red_d[4..]
    auto add = [&](int64_t off, int64_t len) {
      hipLaunchKernelGGL(add_into_kernel<T>, dim3((unsigned)((len + 255) / 256)), dim3(256), 0, s, len, (const T*)c->g_redT + off, redT + off);
    };
    add(roff(c, 0), roff(c, 1) - roff(c, 0));
    add(roff(c, 2), roff(c, 5) - roff(c, 2));
    hipLaunchKernelGGL(add_into_kernel<double>, dim3((unsigned)((nd - 4 + 255) / 256)), dim3(256), 0, s, nd - 4, (const double*)c->g_redd + 4, redd + 4);
    LAUNCHCHK("step_local2");
    return 0;
  }

  // one svi.step around a caller-supplied link function, in three calls (kernels_n.h: elbo_rows_link_kernel):
  //   phase 0: transforms + forward + mu -> workspace 14;  phase 1: ext = theta (K, ext_ld) -> thetabar in workspace 6 (locbar), Phi-bar,
  //   the log-likelihood sum;  phase 2: ext = mubar (K, ext_ld) -> the Normal sites, the row-local backward and the rest of gdrf_step_local
  static int step_local_link(gdrf_ctx* c, const T* X, const int32_t* ws, const T* eps, int64_t n, const T* Z, const T* params,
                             T* redT, double* redd, hipStream_t s, int phase, const T* ext, int64_t ext_ld) {
    const int K = c->K, V = c->V;
    const int64_t ldk = c->ldk;
    int rc;
    if (phase == 0 && (rc = step_local(c, X, ws, eps, n, Z, params, redT, redd, s, SL_TRANSFORMS | SL_FORWARD))) return rc;
    const int RB = 64;
    const size_t lds = 128 + ((size_t)2 * K * V + (size_t)RB * (K + 1) + (size_t)RB * (V + 1)) * sizeof(T);
    if (lds > 150 * 1024) return fail(-1, "gdrf_step_local_link", "num_topic_categories x num_observation_categories too large for the row kernel's LDS");
    if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)elbo_rows_link_kernel<T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int egrid = (int)std::min<int64_t>((n + RB - 1) / RB, c->erows_grid_cap);
    hipLaunchKernelGGL(elbo_rows_link_kernel<T>, dim3(egrid), dim3(RB), lds, s, phase, n, K, V, c->hyp, P(c->qpart), nct<TS>(c), P(c->loc), P(c->tt),
                       eps, ldk, n, ws, P(c->phi), (const T*)c->mean, c->mean_sk, c->mean_sn, ext, ext_ld, P(c->q), P(c->vbar), P(c->locbar),
                       P(c->asum), P(c->mu), c->dpart, P(c->phibar_part));
    LAUNCHCHK("elbo_rows_link");
    if (phase == 1)
      hipLaunchKernelGGL(reduce_parts_kernel<T>, dim3((K * V + 255) / 256), dim3(256), 0, s, P(c->phibar_part), (int64_t)egrid, (int64_t)K * V,
                         redT + roff(c, 1));
    if (phase == 2) {
      hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->dpart, (int64_t)egrid, 4, redd);
      return step_local(c, X, ws, eps, n, Z, params, redT, redd, s, SL_BACKWARD);
    }
    return 0;
  }

  static int step_finish(gdrf_ctx* c, const T* Z, const T* params, const T* redT, const double* redd, double n_global,
                         double ll_const, T* grads, double* out_d, hipStream_t s) {
    const int Mp = c->Mp, M = c->M, K = c->K, V = c->V;
    const int64_t mm = (int64_t)Mp * Mp;
    int rc;
    const T* ubar = redT + roff(c, 0);
    const T* phib = redT + roff(c, 1);
    const T* Ak = redT + roff(c, 2);
    const T* GT = redT + roff(c, 3);
    if ((rc = join_fact(c, s))) return rc;
    ScopedTimer tm(c, 13, s);
    dim3 g2((Mp + 255) / 256, Mp);
    dim3 g3((M + 255) / 256, M, K);
    // Sbar_k = 2 A_k S_k and the u_scale_tril gradient do not depend on the Cholesky backward below (a chain of four dependent M x M
    // products): they run beside it on the side stream (whitened form; the unwhitened one chains them through L^-T further down)
    const bool sbar_aside = !c->unwhitened;
    if (sbar_aside) {
      HIPCHK(hipEventRecord(c->ev_fork, s));
      HIPCHK(hipStreamWaitEvent(c->side, c->ev_fork, 0));
      if ((rc = mm_nt<T>(c, Ak, mm, P(c->ST), mm, P(c->Sbar), mm, T(2), K, c->side))) return rc;
      hipLaunchKernelGGL(grad_s_kernel<T>, g3, dim3(256), 0, c->side, P(c->Sbar), P(c->S), M, Mp, -1.0 / n_global, grads + poff(c, 5));
      HIPCHK(hipEventRecord(c->ev_join, c->side));
    }
    // Cholesky / inverse backward in the solve precision
    hipLaunchKernelGGL((cast_kernel<T, TS>), dim3((unsigned)((mm + 255) / 256)), dim3(256), 0, s, mm, GT, Q(c->GTs));
    // HT = GT Linv ; LbarT = -triu(HT)
    if ((rc = mm_nt<TS>(c, Q(c->GTs), 0, Q(c->LinvT), 0, Q(c->t0), 0, TS(1), 1, s))) return rc;
    if (c->unwhitened) {
      // Sbar'^T = 2 S'^T A_k (A_k symmetric) -> Sbar = L^-T Sbar' -> P_k = S'_k Sbar_k^T ; ubar = L^-T ubar' ; HT += sum_k (P_k + u'_k ubar_k^T)
      if ((rc = mm_nt<T>(c, P(c->ST), mm, Ak, mm, P(c->Sbar), mm, T(2), K, s))) return rc;
      hipLaunchKernelGGL((cast_kernel<T, TS>), dim3((unsigned)((K * mm + 255) / 256)), dim3(256), 0, s, K * mm, (const T*)P(c->Sbar), Q(c->uSb));
      if ((rc = mm_nt<TS>(c, Q(c->LinvT), 0, Q(c->uSb), mm, Q(c->uSc), mm, TS(1), K, s))) return rc;   // Sbar[q][j] = sum_i LinvT[q][i] Sbar'[i][j]
      if ((rc = mm_nt<TS>(c, Q(c->uS), mm, Q(c->uSc), mm, Q(c->uSb), mm, TS(1), K, s))) return rc;      // P_k[i][j] = sum_q S'[i][q] Sbar[j][q]
      hipLaunchKernelGGL((upper_matvec_kernel<TS, T>), dim3((M + 127) / 128, K), dim3(128), 0, s, (const TS*)Q(c->Linv), ubar, M, Mp, Q(c->uUb));
      hipLaunchKernelGGL(add_et_kernel<TS>, dim3((M + 255) / 256, M), dim3(256), 0, s, (const TS*)Q(c->uSb), (const TS*)Q(c->uU),
                         (const TS*)Q(c->uUb), K, M, Mp, Q(c->t0));
    }
    hipLaunchKernelGGL(lbar_t_kernel<TS>, g2, dim3(256), 0, s, (const TS*)Q(c->t0), Mp, Q(c->t1));
    // Q = L^T Lbar ; P = Phi(Q)
    if ((rc = mm_nt<TS>(c, Q(c->LT), 0, Q(c->t1), 0, Q(c->t0), 0, TS(1), 1, s))) return rc;
    hipLaunchKernelGGL(phi_tril_kernel<TS>, g2, dim3(256), 0, s, (const TS*)Q(c->t0), Mp, Q(c->t2));
    // YT = Linv^T P^T ; S' = Linv^T Y
    if ((rc = mm_nt<TS>(c, Q(c->LinvT), 0, Q(c->t2), 0, Q(c->t0), 0, TS(1), 1, s))) return rc;
    if ((rc = mm_nt<TS>(c, Q(c->LinvT), 0, Q(c->t0), 0, Q(c->t1), 0, TS(1), 1, s))) return rc;
    hipLaunchKernelGGL(kuu_bar_reduce_kernel<TS>, dim3(M), dim3(256), 0, s, (const TS*)Q(c->t1), (const TS*)Q(c->Zs), M, Mp, c->D, c->kind,
                       c->hyp, c->dpart);
    hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->dpart, (int64_t)M, 3, c->dsmall);
    if (c->learn_z)
      hipLaunchKernelGGL((grad_z_kernel<TS, T>), dim3(M), dim3(256), 0, s, (const TS*)Q(c->t1), (const TS*)Q(c->Zs), M, Mp, c->D, c->kind, c->hyp,
                         redd + 8, -1.0 / n_global, grads + poff(c, 7));
    // Sbar_k = 2 A_k S_k (N-side precision: well conditioned)
    if (sbar_aside) {
      HIPCHK(hipStreamWaitEvent(s, c->ev_join, 0));
    } else {
      if ((rc = mm_nt<T>(c, Ak, mm, P(c->ST), mm, P(c->Sbar), mm, T(2), K, s))) return rc;
      hipLaunchKernelGGL(grad_s_kernel<T>, g3, dim3(256), 0, s, P(c->Sbar), P(c->S), M, Mp, -1.0 / n_global, grads + poff(c, 5));
    }
    hipLaunchKernelGGL(grad_small_kernel<T>, dim3(1), dim3(256), 0, s, M, Mp, K, V, c->hyp, redd, c->dsmall, ubar, phib, P(c->phi),
                       c->alpha_dev, c->lgam_const, ll_const, n_global, grads, grads + poff(c, 3), grads + poff(c, 4), c->flag, out_d);
    if (c->unwhitened)       // overwrite the u_loc / u_scale_tril blocks with the gradients chained through L^-T
      hipLaunchKernelGGL((grad_unwhitened_kernel<TS, T>), g3, dim3(256), 0, s, (const TS*)Q(c->uSc), (const TS*)Q(c->uUb), params + poff(c, 5), K, M,
                         Mp, -1.0 / n_global, grads + poff(c, 5), grads + poff(c, 3));
    LAUNCHCHK("step_finish");
    return 0;
  }

  static int predict(gdrf_ctx* c, const T* X, int64_t n, const T* Z, const T* params, const int32_t* ws, int mode,
                     T* out, double* out_d, hipStream_t s) {
    const int Mp = c->Mp, M = c->M, K = c->K, V = c->V;
    const T* U = params + poff(c, 3);
    if (int rcj = join_fact(c, s)) return rcj;
    if (c->unwhitened) {       // loc = K_nm L^-T (L^-1 u)
      hipLaunchKernelGGL((lower_matvec_kernel<TS, T>), dim3((M + 127) / 128, K), dim3(128), 0, s, (const TS*)Q(c->Linv), U, M, Mp, Q(c->uU),
                         (T*)c->Uw);
      U = (const T*)c->Uw;
    }
    if (mode == 4) {
      // (f_loc, f_var) of gp.util.conditional(full_cov=False) (gdrf/models/sparse_gdrf.py:277-319): the step's own forward - transforms,
      // K_nm, W = K_nm L^-T with its row norms, loc = W U^T, tt = |S_k^T w|^2 - and one pass that assembles the variance
      if (n > c->ncap) return fail(-1, "gdrf_predict", "mode 4 (loc, var) needs n <= n_cap");
      if (int rc = step_local(c, X, nullptr, nullptr, n, Z, params, nullptr, nullptr, s, SL_TRANSFORMS | SL_FORWARD | SL_NO_DK)) return rc;
      hipLaunchKernelGGL(predict_var_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, K, c->hyp, (const T*)P(c->qpart), nct<TS>(c),
                         (const T*)P(c->loc), (const T*)P(c->tt), c->ldk, out);
      LAUNCHCHK("predict (loc, var)");
      return 0;
    }
    if (mode >= 2) hipLaunchKernelGGL(build_phi_kernel<T>, dim3(K), dim3(64), 0, s, params + poff(c, 4), K, V, P(c->phi));
    {
      // matrix-core form (predict.h): a wave owns 16 rows, one covariance value per lane and step is the A operand of the 16x16x4
      // matrix instruction, the padded transposed coefficients CfT its B operand.  K <= 32, the scaled inducing inputs in LDS.
      static const bool mfma_on = !(getenv("GDRF_PREDICT_MFMA") && getenv("GDRF_PREDICT_MFMA")[0] == '0');     // A/B knob: 0 = one thread per row
      const int M4 = (int)round_up(M, 4), NB = K <= 16 ? 1 : 2, DDt = c->D <= 2 ? 2 : GDRF_DMAX;
      const size_t lds = 128 + ((size_t)M4 * DDt + (size_t)K * V + (size_t)4 * 16 * (16 * NB + 1)) * sizeof(TS);
      if (mfma_on && K <= 32 && lds <= 64 * 1024) {
        const int ldc = 16 * NB;
        hipLaunchKernelGGL((predict_coeff_t_kernel<TS, T>), dim3(M4), dim3(64), 0, s, (const TS*)Q(c->LinvT), U, M, Mp, M4, K, ldc, Q(c->CfT));
        const int64_t groups = (n + 15) / 16;
        int64_t blocks = (groups + 3) / 4; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
        const int64_t ldo = mode == 0 ? n : (mode == 1 ? K : V);
#define GDRF_PM(DDv, NBv) { if (lds > 48 * 1024) HIPCHK(hipFuncSetAttribute((const void*)predict_mfma_kernel<TS, T, DDv, NBv>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                            hipLaunchKernelGGL((predict_mfma_kernel<TS, T, DDv, NBv>), dim3((unsigned)blocks), dim3(256), lds, s, X, n, (const TS*)Q(c->Zs), M, M4, c->D, c->kind, \
                                               c->hyp, (const TS*)Q(c->CfT), K, V, (const T*)P(c->phi), ws, mode, out, ldo, c->dpart); }
        if (DDt == 2) { if (NB == 1) GDRF_PM(2, 1) else GDRF_PM(2, 2) } else { if (NB == 1) GDRF_PM(GDRF_DMAX, 1) else GDRF_PM(GDRF_DMAX, 2) }
#undef GDRF_PM
        if (mode == 3) hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->dpart, blocks, 2, out_d);
        LAUNCHCHK("predict (mfma)");
        return 0;
      }
    }
    hipLaunchKernelGGL((predict_coeff_kernel<TS, T>), dim3((M + 127) / 128, K), dim3(128), 0, s, (const TS*)Q(c->Linv), U, M, Mp, K, Q(c->Cf));
    int64_t blocks = (n + 127) / 128; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    const int64_t ldo = mode == 0 ? n : (mode == 1 ? K : V);
    if (K > GDRF_KMAX) {
      const size_t lds = 128 + ((size_t)M * c->D + (size_t)K * V + (size_t)128 * (V + 1)) * sizeof(TS);
      if (lds > 150 * 1024) return fail(-1, "gdrf_predict", "M*D + K*V + 128*(V+1) solve-precision elements exceed the LDS budget (150 KB)");
      if (lds > 48 * 1024)
        HIPCHK(hipFuncSetAttribute((const void*)predict_rows_bigk_kernel<TS, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((predict_rows_bigk_kernel<TS, T>), dim3((unsigned)blocks), dim3(128), lds, s, X, n, (const TS*)Q(c->Zs), M, c->D, c->kind,
                         c->hyp, (const TS*)Q(c->Cf), K, V, (const T*)P(c->phi), ws, mode, out, ldo, c->dpart);
    } else {
      size_t lds = 128 + ((size_t)M * c->D + (size_t)K * V + (size_t)K * M) * sizeof(TS);
      int in_lds = 1;
      if (lds > 64 * 1024) { in_lds = 0; lds -= (size_t)K * M * sizeof(TS); }
      if (lds > 150 * 1024) return fail(-1, "gdrf_predict", "M*D + K*V solve-precision elements exceed the LDS budget (150 KB)");
      if (lds > 48 * 1024)
        HIPCHK(hipFuncSetAttribute((const void*)predict_rows_kernel<TS, T>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL((predict_rows_kernel<TS, T>), dim3((unsigned)blocks), dim3(128), lds, s, X, n, (const TS*)Q(c->Zs), M, c->D, c->kind,
                         c->hyp, (const TS*)Q(c->Cf), K, V, (const T*)P(c->phi), ws, mode, out, ldo, c->dpart, in_lds);
    }
    if (mode == 3) hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->dpart, blocks, 2, out_d);
    LAUNCHCHK("predict");
    return 0;
  }
};


int gdrf_knm(gdrf_ctx* c, const void* X, int64_t n, const void* Z, const void* params, void* out, int64_t ldo, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t s = (hipStream_t)stream;
  if (c->dtype == GDRF_F64) return Impl<double, double>::knm(c, (const double*)X, n, (const double*)Z, (const double*)params, (double*)out, ldo, s);
  return Impl<float, float>::knm(c, (const float*)X, n, (const float*)Z, (const float*)params, (float*)out, ldo, s);
}

int gdrf_fill_eps(gdrf_ctx* c, uint64_t seed, uint32_t step, int64_t n_offset, int64_t n, void* eps, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t s = (hipStream_t)stream;
  dim3 grid((unsigned)((n + 255) / 256), c->K);
  if (c->esz == 4) hipLaunchKernelGGL(fill_eps_kernel<float>, grid, dim3(256), 0, s, seed, step, n_offset, n, c->K, (float*)eps, n);
  else hipLaunchKernelGGL(fill_eps_kernel<double>, grid, dim3(256), 0, s, seed, step, n_offset, n, c->K, (double*)eps, n);
  LAUNCHCHK("fill_eps");
  return 0;
}

// red_d (8 + M*D doubles) -> tail of red_T, and back after the all-reduce
int gdrf_payload_pack(gdrf_ctx* c, void* redT, const double* redd, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  const int nd = 8 + c->M * c->D;
  if (c->esz == 8) hipLaunchKernelGGL(payload_pack_kernel<double>, dim3((nd + 255) / 256), dim3(256), 0, (hipStream_t)stream, redd, nd, (double*)redT + roff(c, 5));
  else hipLaunchKernelGGL(payload_pack_kernel<float>, dim3((nd + 255) / 256), dim3(256), 0, (hipStream_t)stream, redd, nd, (float*)redT + roff(c, 5));
  LAUNCHCHK("payload_pack");
  return 0;
}
int gdrf_payload_unpack(gdrf_ctx* c, const void* redT, double* redd, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  const int nd = 8 + c->M * c->D;
  if (c->esz == 8) hipLaunchKernelGGL(payload_unpack_kernel<double>, dim3((nd + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const double*)redT + roff(c, 5), nd, redd);
  else hipLaunchKernelGGL(payload_unpack_kernel<float>, dim3((nd + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)redT + roff(c, 5), nd, redd);
  LAUNCHCHK("payload_unpack");
  return 0;
}

int gdrf_set_allreduce(gdrf_ctx* c, gdrf_allreduce_fn fn, void* user) {
  c->allreduce = fn; c->allreduce_user = user;
  return 0;
}
// the step's single collective: pack red_d into the tail of red_T, the caller's sum over the ranks on the whole flat buffer, unpack
int gdrf_payload_allreduce(gdrf_ctx* c, void* redT, double* redd, void* stream) {
  if (!c->allreduce) return 0;
  if (int rc = gdrf_payload_pack(c, redT, redd, stream)) return rc;
  const int rc = c->allreduce(redT, roff(c, 4), c->esz == 8 ? 1 : 0, stream, c->allreduce_user);
  if (rc) return fail(-2, "gdrf_payload_allreduce", "the registered all-reduce function reported an error");
  return gdrf_payload_unpack(c, redT, redd, stream);
}

int gdrf_ll_const_dev(gdrf_ctx* c, const int32_t* ws, int64_t n, double* out_dev, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t s = (hipStream_t)stream;
  int64_t blocks = (n + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(ll_const_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ws, n, c->V, c->llpart);
  hipLaunchKernelGGL(reduce_dparts_kernel, dim3(1), dim3(1024), 0, s, c->llpart, blocks, 1, out_dev);
  LAUNCHCHK("ll_const");
  return 0;
}

int gdrf_ll_const(gdrf_ctx* c, const int32_t* ws, int64_t n, double* out_host, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = gdrf_ll_const_dev(c, ws, n, c->dsmall + 8, stream)) return rc;
  HIPCHK(hipMemcpyAsync(out_host, c->dsmall + 8, sizeof(double), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

#define TYPED3(c, fn, ...)                                                                                         \
  do {                                                                                                             \
    if ((c)->dtype == GDRF_F32) { using T = float; using I = Impl<float, double>; return I::fn(__VA_ARGS__); }     \
    if ((c)->dtype == GDRF_F64) { using T = double; using I = Impl<double, double>; return I::fn(__VA_ARGS__); }   \
    { using T = float; using I = Impl<float, float>; return I::fn(__VA_ARGS__); }                                  \
  } while (0)

static int probe_dispatch(gdrf_ctx* c, const void* Z, const void* params, const double* jitters, int nlev, hipStream_t s) {
  TYPED3(c, probe, c, (const T*)Z, (const T*)params, jitters, nlev, s);
}

int gdrf_probe(gdrf_ctx* c, const void* Z, const void* params, const double* jitters, int nlev, int* failed_host, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  if (nlev < 1 || nlev > 8) return fail(-1, "gdrf_probe", "nlev must be in [1, 8]");
  hipStream_t s = (hipStream_t)stream;
  int rc = probe_dispatch(c, Z, params, jitters, nlev, s);
  if (rc) return rc;
  HIPCHK(hipMemcpyAsync(failed_host, c->flag + 8, sizeof(int) * nlev, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

// the two halves of gdrf_probe: the launch alone (asynchronous), and the read of its flags (waits for the stream)
int gdrf_probe_launch(gdrf_ctx* c, const void* Z, const void* params, const double* jitters, int nlev, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  if (nlev < 1 || nlev > 8) return fail(-1, "gdrf_probe_launch", "nlev must be in [1, 8]");
  return probe_dispatch(c, Z, params, jitters, nlev, (hipStream_t)stream);
}
int gdrf_probe_read(gdrf_ctx* c, int nlev, int* failed_host, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  if (nlev < 1 || nlev > 8) return fail(-1, "gdrf_probe_read", "nlev must be in [1, 8]");
  hipStream_t s = (hipStream_t)stream;
  HIPCHK(hipMemcpyAsync(failed_host, c->flag + 8, sizeof(int) * nlev, hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  return 0;
}

int gdrf_factorize(gdrf_ctx* c, const void* Z, const void* params, double jitter, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t s = (hipStream_t)stream;
  TYPED3(c, factorize, c, (const T*)Z, (const T*)params, jitter, s);
}

int gdrf_factorize_mode(gdrf_ctx* c, const void* Z, const void* params, double jitter, void* stream, int mode) {
  HIPCHK(hipSetDevice(c->dev));
  if (mode < 0 || mode > 2) return fail(-1, "gdrf_factorize_mode", "mode must be 0, 1 or 2");
  hipStream_t s = (hipStream_t)stream;
  TYPED3(c, factorize, c, (const T*)Z, (const T*)params, jitter, s, mode);
}

int gdrf_step_local(gdrf_ctx* c, const void* X, const int32_t* ws, const void* eps, int64_t n, const void* Z, const void* params,
                    void* redT, double* redd, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  if (n < 1 || n > c->ncap) return fail(-1, "gdrf_step_local", "n_local outside [1, n_cap]");
  hipStream_t s = (hipStream_t)stream;
  TYPED3(c, step_local, c, (const T*)X, ws, (const T*)eps, n, (const T*)Z, (const T*)params, (T*)redT, redd, s);
}

int gdrf_step_local_link(gdrf_ctx* c, const void* X, const int32_t* ws, const void* eps, int64_t n, const void* Z, const void* params,
                         void* redT, double* redd, void* stream, int phase, const void* ext, int64_t ext_ld) {
  HIPCHK(hipSetDevice(c->dev));
  if (n < 1 || n > c->ncap) return fail(-1, "gdrf_step_local_link", "n_local outside [1, n_cap]");
  if (phase < 0 || phase > 2) return fail(-1, "gdrf_step_local_link", "phase must be 0, 1 or 2");
  if (phase > 0 && (!ext || ext_ld < n)) return fail(-1, "gdrf_step_local_link", "phases 1 and 2 take a (K, ext_ld >= n) array");
  hipStream_t s = (hipStream_t)stream;
  TYPED3(c, step_local_link, c, (const T*)X, ws, (const T*)eps, n, (const T*)Z, (const T*)params, (T*)redT, redd, s, phase, (const T*)ext, ext_ld);
}

int gdrf_step_local2(gdrf_ctx* c, const void* X_model, const void* X_guide, const int32_t* ws, const void* eps, int64_t n, const void* Z,
                     const void* params, void* redT, double* redd, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  if (n < 1 || n > c->ncap) return fail(-1, "gdrf_step_local2", "n_local outside [1, n_cap]");
  if (c->Tst) return fail(-1, "gdrf_step_local2", "needs the dense Wbar form (GDRF_STORE_T_OFF)");
  hipStream_t s = (hipStream_t)stream;
  TYPED3(c, step_local2, c, (const T*)X_model, (const T*)X_guide, ws, (const T*)eps, n, (const T*)Z, (const T*)params, (T*)redT, redd, s);
}

int gdrf_step_finish(gdrf_ctx* c, const void* Z, const void* params, const void* redT, const double* redd, double n_global,
                     double ll_const, void* grads, double* out_d, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t s = (hipStream_t)stream;
  TYPED3(c, step_finish, c, (const T*)Z, (const T*)params, (const T*)redT, redd, n_global, ll_const, (T*)grads, out_d, s);
}

int gdrf_adam(gdrf_ctx* c, int mode, void* params, const void* grads, void* m, void* v, int64_t t, double lr, double b1, double b2,
              double eps, double wd, double clip, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t s = (hipStream_t)stream;
  const int64_t n = poff(c, 6);
  const double bc1 = 1.0 - std::pow(b1, (double)t), bc2 = 1.0 - std::pow(b2, (double)t);
  dim3 grid((unsigned)((n + 255) / 256));
  ScopedTimer tm(c, 14, s);
  if (c->esz == 4)
    hipLaunchKernelGGL(adam_kernel<float>, grid, dim3(256), 0, s, n, (float*)params, (const float*)grads, (float*)m, (float*)v, mode, lr,
                       b1, b2, eps, wd, clip, bc1, bc2, (const int*)c->flag);
  else
    hipLaunchKernelGGL(adam_kernel<double>, grid, dim3(256), 0, s, n, (double*)params, (const double*)grads, (double*)m, (double*)v, mode,
                       lr, b1, b2, eps, wd, clip, bc1, bc2, (const int*)c->flag);
  LAUNCHCHK("adam");
  return 0;
}

int gdrf_predict(gdrf_ctx* c, const void* X, int64_t n, const void* Z, const void* params, const int32_t* ws, int mode,
                 void* out, double* out_d, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  if (mode < 0 || mode > 4) return fail(-1, "gdrf_predict", "mode");
  if (n < 1) return fail(-1, "gdrf_predict", "n must be >= 1");
  if (mode == 3 && !ws) return fail(-1, "gdrf_predict", "perplexity needs ws");
  hipStream_t s = (hipStream_t)stream;
  TYPED3(c, predict, c, (const T*)X, n, (const T*)Z, (const T*)params, ws, mode, (T*)out, out_d, s);
}

int gdrf_chol_failed(gdrf_ctx* c, int* failed, void* stream) {
  HIPCHK(hipSetDevice(c->dev));
  hipStream_t s = (hipStream_t)stream;
  if (int rc = join_fact(c, s)) return rc;
  int f17[17];                               // [0] the factorisation failed, [16] a reused factorisation's inputs had changed
  HIPCHK(hipMemcpyAsync(f17, c->flag, sizeof(f17), hipMemcpyDeviceToHost, s));
  HIPCHK(hipStreamSynchronize(s));
  *failed = f17[0] | f17[16];
  return 0;
}

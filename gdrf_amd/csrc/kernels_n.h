// Observation-sized kernels of one SVI step (everything that is a sum over rows n).
//
// Reference behaviour restated (paths under /root/reference; arithmetic in SURVEY.md A.3/A.4, App. C):
//   pyro.contrib.gp.util.conditional as called at gdrf/models/sparse_gdrf.py:334-344,384-394
//   guide/model sample sites                      gdrf/models/sparse_gdrf.py:354-372,396-409
//   softmax link + topic_probs @ phi              gdrf/models/abstract_gdrf.py:21-22, sparse_gdrf.py:361-362
#pragma once
#include "common.h"
#include "gemm_nt.h"
#include "kernels_mm.h"
#include "gemm_split.h"


namespace gdrf {

template <typename T> __device__ __forceinline__ typename Vec16<T>::type vzero() {
  typename Vec16<T>::type z;
#pragma unroll
  for (int e = 0; e < Vec16<T>::N; ++e) z[e] = 0;
  return z;
}

// =====================================================================================
// k_nm: the standalone pairwise covariance block K_nm (N x M, row-major).  HBM-write bound:
// algorithmic bytes = N*M*s + N*D*s + M*D*s.  One 16-byte store per lane, rows contiguous.
// =====================================================================================
template <typename T> __device__ __forceinline__ T cov_fast(int kind, T r2, T var, T alpha);

// 2^t in double for t <= ~1000 (0 below -1100), straight-line: 2^t = ldexp(P(t - rint t), rint t), P the degree-12 Taylor
// polynomial of 2^r on [-1/2, 1/2] (remainder (ln2 / 2)^13 / 13! = 1.7e-16 relative).  The coefficients ln2^i / i! come from
// constant memory, i.e. scalar registers: a Horner step is then ONE v_fma_f64 (with literal constants the compiler emits
// v_mov_b64 + v_fmac_f64 per step, which made the K_nm kernel VALU-bound).
__constant__ double exp2_coef[13] = {1.0, 0.6931471805599453, 0.2402265069591007, 0.055504108664821576, 0.009618129107628477, 0.0013333558146428441, 0.00015403530393381606, 1.5252733804059838e-05, 1.3215486790144305e-06, 1.0178086009239696e-07, 7.054911620801121e-09, 4.44553827187081e-10, 2.5678435993488196e-11};
__device__ __forceinline__ double exp2_poly(double t) {
  t = fmax(t, -1100.0);
  const double k = __builtin_rint(t), r = t - k;             // exact: |r| <= 1/2
  double p = exp2_coef[12];
#pragma unroll
  for (int i = 11; i >= 0; --i) p = fma(p, r, exp2_coef[i]);
  return __builtin_ldexp(p, (int)k);
}

// RBF K_nm in double (the in-step solve-precision copy), its own kernel so that it does not carry the generic kernel's
// 132 registers (3 workgroups per CU): the generic loop branches on the kernel kind per element around the library exp().
// Straight-line form, 4 rows x 2 columns = 8 independent chains per lane: k = 2^t, t = log2 var - sum_d (a x_d - a z_d)^2
// with the coordinates pre-scaled by a = sqrt(log2(e) / 2) / ls (exp2_poly above).  Needs ldo % 2 == 0 and ldo / 2 <= 256.
template <typename TX, int DD, bool NT = true>
__global__ __launch_bounds__(256, 4) void knm_rbf_f64_kernel(const TX* __restrict__ X, int64_t N, const double* __restrict__ Z, int M, int D,
                                                          const Hyper* __restrict__ h, double* __restrict__ out, int64_t ldo) {
  typedef double V __attribute__((ext_vector_type(2)));
  const int vpr = (int)(ldo / 2), rpp = 256 / vpr;
  const int rsub = (int)threadIdx.x / vpr, cv = (int)threadIdx.x % vpr, i0 = cv * 2;
  if (rsub >= rpp) return;
  // a workgroup owns a contiguous block of rows: addresses are a uniform base plus a 32-bit offset (the host keeps
  // rows-per-workgroup x ldo below 2^31), and a pass writes 4 rpp adjacent rows
  const int64_t rows_per = (N + gridDim.x - 1) / gridDim.x, rb = (int64_t)blockIdx.x * rows_per;
  if (rb >= N) return;
  const int nr = (int)((N - rb < rows_per) ? N - rb : rows_per);
  double* __restrict__ ob = out + rb * ldo;
  const TX* __restrict__ xb = X + rb * D;
  const double a = sqrt(0.5 * 1.4426950408889634074 * h->inv_ls2), lv = log2(h->var);
  double z[2][DD];       // coordinates beyond D are zero on both sides; a padding column sits at 1e160: t = -inf, clamped, 2^-1100 = 0
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int d = 0; d < DD; ++d) z[e][d] = (i0 + e < M) ? ((d < D) ? a * Z[(int64_t)(i0 + e) * D + d] : 0.0) : ((d == 0) ? 1e160 : 0.0);
  const int ldo32 = (int)ldo;
  for (int r0 = rsub; r0 < nr; r0 += 4 * rpp) {
    double x[4][DD];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r0 + u * rpp;
#pragma unroll
      for (int d = 0; d < DD; ++d) x[u][d] = (rr < nr && d < D) ? a * (double)xb[rr * D + d] : 0.0;
    }
    V o[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        double t = lv;
#pragma unroll
        for (int d = 0; d < DD; ++d) { const double dd = x[u][d] - z[e][d]; t = fma(-dd, dd, t); }
        o[u][e] = exp2_poly(t);
      }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int rr = r0 + u * rpp;
      if (rr < nr) {
        V* dst = reinterpret_cast<V*>(ob + (unsigned)(rr * ldo32 + i0));
        if (NT) __builtin_nontemporal_store(o[u], dst); else *dst = o[u];
      }
    }
  }
}

// T: output / arithmetic type, TX: type of X, FAST: v_exp_f32-based exponential (roofline kernel) or exact exp
// (the solve-precision copy of K_nm that feeds W = K_nm L^-T).  Columns M..ldo-1 of every row are written as zeros.
template <typename T, typename TX, bool FAST, bool NT = true>
__global__ __launch_bounds__(256) void knm_kernel(const TX* __restrict__ X, int64_t N, const T* __restrict__ Z, int M, int D,
                                                  int kind, const Hyper* __restrict__ h, T* __restrict__ out, int64_t ldo) {
  using V = typename Vec16<T>::type;
  constexpr int VE = Vec16<T>::N;
  const T var = (T)h->var, ils2 = (T)h->inv_ls2, al = (T)h->alpha;
  const int vpr = (int)((ldo + VE - 1) / VE);          // 16-byte vectors per output row (ldo >= M)
  const bool aligned = (ldo % VE) == 0;
  if (vpr <= 256) {
    // fast path: a thread owns ONE column vector for the whole launch, so its VE inducing points live in
    // registers; a workgroup pass writes rpp whole rows (contiguous bytes); rows unrolled x4 to keep
    // several 16-byte stores in flight per lane.  No division inside the loop.
    const int rpp = 256 / vpr;
    const int rsub = (int)threadIdx.x / vpr, cv = (int)threadIdx.x % vpr, i0 = cv * VE;
    if (rsub >= rpp) return;
    T z[VE][GDRF_DMAX];
#pragma unroll
    for (int e = 0; e < VE; ++e)
#pragma unroll
      for (int d = 0; d < GDRF_DMAX; ++d) z[e][d] = (i0 + e < M && d < D) ? Z[(int64_t)(i0 + e) * D + d] : T(0);
    const int64_t stride = (int64_t)gridDim.x * rpp;
    const bool vec_ok = aligned && (i0 + VE <= ldo);
#ifndef GDRF_KNM_NO_RBF_FAST
    if constexpr (FAST && sizeof(T) == 4) {
      if (kind == 0) {
        // RBF on the roofline path: k = var * exp(-r2 / (2 ls^2)) = 2^(log2 var - sum_d (a x_d - a z_d)^2), a = sqrt(log2(e) / 2) / ls:
        // with the coordinates pre-scaled by a (the inducing points once per thread, x once per row) an element costs
        // D subtractions, D fused multiply-adds and one v_exp_f32 instead of ~9 VALU operations - the kernel is a 2 GB
        // streaming store, but at ~5 TB/s its VALU work was a third of its time.
        const float a = sqrtf(0.5f * 1.44269504088896f * (float)ils2), lv = __log2f((float)var);
#pragma unroll
        for (int e = 0; e < VE; ++e)
#pragma unroll
          for (int d = 0; d < GDRF_DMAX; ++d) z[e][d] *= a;
        for (int64_t row0 = (int64_t)blockIdx.x * rpp + rsub; row0 < N; row0 += 4 * stride) {
          float x[4][GDRF_DMAX];
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t row = row0 + u * stride;
#pragma unroll
            for (int d = 0; d < GDRF_DMAX; ++d) x[u][d] = (row < N && d < D) ? a * (float)X[row * D + d] : 0.0f;
          }
#pragma unroll
          for (int u = 0; u < 4; ++u) {
            const int64_t row = row0 + u * stride;
            if (row >= N) break;
            V o;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
              float t = -lv;
#pragma unroll
              for (int d = 0; d < GDRF_DMAX; ++d) if (d < D) { const float dd = x[u][d] - z[e][d]; t = fmaf(dd, dd, t); }
              o[e] = (i0 + e < M) ? __builtin_amdgcn_exp2f(-t) : 0.0f;
            }
            T* orow = out + row * ldo;
            if (vec_ok) { if (NT) __builtin_nontemporal_store(o, reinterpret_cast<V*>(orow + i0)); else *reinterpret_cast<V*>(orow + i0) = o; }
            else for (int e = 0; e < VE; ++e) if (i0 + e < ldo) orow[i0 + e] = o[e];
          }
        }
        return;
      }
    }
#endif
    for (int64_t row0 = (int64_t)blockIdx.x * rpp + rsub; row0 < N; row0 += 4 * stride) {
      T x[4][GDRF_DMAX];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = row0 + u * stride;
#pragma unroll
        for (int d = 0; d < GDRF_DMAX; ++d) x[u][d] = (row < N && d < D) ? (T)X[row * D + d] : T(0);
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int64_t row = row0 + u * stride;
        if (row >= N) break;
        V o;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
          T r2 = 0;
#pragma unroll
          for (int d = 0; d < GDRF_DMAX; ++d) if (d < D) { const T t = x[u][d] - z[e][d]; r2 += t * t; }
          o[e] = (i0 + e < M) ? (FAST ? cov_fast<T>(kind, r2 * ils2, var, al) : cov_from_r2<T>(kind, r2 * ils2, var, al)) : T(0);
        }
        T* orow = out + row * ldo;
        if (vec_ok) { if (NT) __builtin_nontemporal_store(o, reinterpret_cast<V*>(orow + i0)); else *reinterpret_cast<V*>(orow + i0) = o; }
        else for (int e = 0; e < VE; ++e) if (i0 + e < ldo) orow[i0 + e] = o[e];
      }
    }
    return;
  }
  // wide rows (M > 256 vectors): one row per workgroup pass, inducing points read through L1
  for (int64_t row = blockIdx.x; row < N; row += gridDim.x) {
    T x[GDRF_DMAX];
#pragma unroll
    for (int d = 0; d < GDRF_DMAX; ++d) x[d] = (d < D) ? (T)X[row * D + d] : T(0);
    T* orow = out + row * ldo;
    for (int cv = threadIdx.x; cv < vpr; cv += 256) {
      const int i0 = cv * VE;
      V o;
#pragma unroll
      for (int e = 0; e < VE; ++e) {
        const int i = i0 + e;
        T r2 = 0;
        if (i < M) {
#pragma unroll
          for (int d = 0; d < GDRF_DMAX; ++d) if (d < D) { const T t = x[d] - Z[(int64_t)i * D + d]; r2 += t * t; }
        }
        o[e] = (i < M) ? (FAST ? cov_fast<T>(kind, r2 * ils2, var, al) : cov_from_r2<T>(kind, r2 * ils2, var, al)) : T(0);
      }
      if (aligned && i0 + VE <= ldo) __builtin_nontemporal_store(o, reinterpret_cast<V*>(orow + i0));
      else for (int e = 0; e < VE; ++e) if (i0 + e < ldo) orow[i0 + e] = o[e];
    }
  }
}

// =====================================================================================
// NT-core problems
// =====================================================================================
// fast exponential for the f32 GEMM operand generators (v_exp_f32); f64 keeps exp()
template <typename T> __device__ __forceinline__ T cov_fast(int kind, T r2, T var, T alpha) { return cov_from_r2<T>(kind, r2, var, alpha); }
template <> __device__ __forceinline__ float cov_fast<float>(int kind, float r2, float var, float alpha) {
  if (kind == 0) return var * __expf(-0.5f * r2);
  if (kind == 4) return var * __expf(-alpha * __logf(1.0f + r2 * (0.5f / alpha)));
  const float r = sqrtf(r2 + 1e-12f);
  if (kind == 3) return var * __expf(-r);
  if (kind == 2) { const float a3 = 1.7320508076f * r; return var * (1.0f + a3) * __expf(-a3); }
  const float a = 2.2360679775f * r;
  return var * (1.0f + a + (5.0f / 3.0f) * r * r) * __expf(-a);
}

// (1) W = K_nm Linv^T : A = the solve-precision copy of K_nm (knm_kernel), Bt = Linv, triangular k range;
//     stores W and the per-column-tile partial of q_n = ||w_n||^2.
// T = solve precision (f64 in the default fp32 mode: the triangular solve cancels terms ~|Linv||k| >> |w|),
// TN = precision of the N-sized outputs (W, qpart)
// DERIV: the A operand is dK_nm / d log(lengthscale) instead of K_nm, formed from the K_nm chunk when it is written to LDS
// (dcov_dlogls_from_k: no second exponential); the product W' = K' Linv^T gives the K_nm part of the lengthscale gradient as
// sum W' o Wbar, and sum W o Wbar is the variance part (Kbar = Wbar Linv, so sum_m Kbar_nm K_nm = sum_j Wbar_nj W_nj): no
// backward GEMM, no second pass over K_nm.  Z (solve precision, [M][D]) sits in the problem-owned LDS behind the tiles.
template <typename T, typename TN, bool DERIV = false, int DD = 2> struct FwdWProb : NTXcdRowMap, NTNoExtra {
  using V = typename Vec16<T>::type;
  using AVec = V;
  static constexpr bool SCALE_A = false;
  static constexpr bool A_PER_REP = false;
  static constexpr int DEPTH = (sizeof(T) == 8 && !DERIV) ? GDRF_FWDW_DEPTH : 1;
  static constexpr int MIN_WGS = (sizeof(T) == 8 && !DERIV) ? GDRF_FWDW_WGS : 2;
#ifdef GDRF_NT_TRACE
  static constexpr bool TRACE = sizeof(T) == 8 && !DERIV;
#endif
#ifndef GDRF_NO_TRI
  static constexpr int TRI = 1;            // W[n][col] = sum_{k <= col} K_nm[n][k] Linv[col][k]
#endif
  const T* Knm; int64_t nrows; int Mp;
  const T* Linv; TN* W; TN* qpart; int64_t ldq;      // qpart [col_tiles][ldq]
  const TN* X = nullptr; const T* Z = nullptr; const Hyper* h = nullptr; int M = 0, D = 0, kind = 0;   // DERIV only
  void* Wh = nullptr; int64_t wh_stride = 0;          // optional: the 16-bit pieces of W (gemm_split.h), written by the same epilogue
  int wh_mode = 0; const float* wh_scale = nullptr;   // 1: three bf16 pieces; 2: two fp16 pieces of W * wh_scale[0]
  struct ACtx { const T* p[NTCfg<T>::VPT]; T x[DERIV ? NTCfg<T>::VPT : 1][DD]; const T* zS; T ils2, al; };
  struct ECtx { T rs[4][4]; int ct; };
  __device__ __forceinline__ int col_tiles() const { return (Mp + NTCfg<T>::CW - 1) / NTCfg<T>::CW; }
  __device__ __forceinline__ bool loop_cols() const { return false; }
  __device__ __forceinline__ int a_reuse() const { return 1; }
  __device__ __forceinline__ void krange(int64_t, int n0, int, int& kb, int& ke) const {
    kb = 0; ke = n0 + NTCfg<T>::CW; if (ke > Mp) ke = Mp;
  }
  __device__ __forceinline__ void prepA(ACtx& c, int64_t m0, int, char* extra) const {
#pragma unroll
    for (int i = 0; i < NTCfg<T>::VPT; ++i) {
      const int64_t r = m0 + nt_stage_row<T>(i);
      c.p[i] = (r < nrows) ? Knm + r * Mp : nullptr;
      if constexpr (DERIV) {
#pragma unroll
        for (int d = 0; d < DD; ++d) c.x[i][d] = (r < nrows && d < D) ? (T)X[r * D + d] : T(0);
      }
    }
    if constexpr (DERIV) {
      T* zs = reinterpret_cast<T*>(extra);                  // [Mp][DD], zero beyond (M, D)
      for (int e = threadIdx.x; e < Mp * DD; e += 256) { const int m = e / DD, d = e - m * DD; zs[e] = (m < M && d < D) ? Z[m * D + d] : T(0); }
      c.zS = zs; c.ils2 = (T)h->inv_ls2; c.al = (T)h->alpha;
      __syncthreads();
    }
  }
  __device__ __forceinline__ V a_to_lds(const V& v, const ACtx& c, int i, int k) const {
    if constexpr (!DERIV) return v;
    else {
      V o;
#pragma unroll
      for (int e = 0; e < Vec16<T>::N; ++e) {
        T r2 = 0;
#pragma unroll
        for (int d = 0; d < DD; ++d) { const T t = c.x[i][d] - c.zS[(k + e) * DD + d]; r2 += t * t; }
        o[e] = dcov_dlogls_from_k<T>(kind, v[e], r2 * c.ils2, c.al);
      }
      return o;
    }
  }
  __device__ __forceinline__ void prepE(ECtx& e, int64_t, int) const {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) e.rs[a][r] = 0;
    e.ct = 0;
  }
  __device__ __forceinline__ V loadA(const ACtx& c, int i, int k, int, int) const {
    return c.p[i] ? *reinterpret_cast<const V*>(c.p[i] + k) : vzero<T>();
  }
  __device__ __forceinline__ V loadB(int n0, int i, int k, int, int) const {
    const int c = n0 + nt_stage_row<T>(i);
    return (c < Mp) ? *reinterpret_cast<const V*>(Linv + (int64_t)c * Mp + k) : vzero<T>();
  }
  template <class Acc, int NB_>
  __device__ __forceinline__ void tile_done(Acc (&acc)[4][NB_], int64_t m0, int n0, int, ECtx& ec, int wr, int wc, int lane) const {
    ec.ct = n0 / NTCfg<T>::CW;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int b = 0; b < NTCfg<T>::NB; ++b) ec.rs[a][r] += acc[a][b][r] * acc[a][b][r];
    if constexpr (sizeof(T) == 8 && sizeof(TN) == 4) {
      {
        // f64 solve, f32 W, split contractions: each wave transposes its 64 x 32 quadrant through a private LDS tile, 32 rows
        // at a time, and stores whole 128-byte row segments - W as float4 and its three bf16 pieces as 8-byte vectors - instead
        // of 32 scalar stores per lane followed by a separate pass that re-reads all of W to split it.
        extern __shared__ __attribute__((aligned(16))) char nt_smem[];
        constexpr int TS_ = 36;                               // tile row stride in floats (144 B)
        float* tile = reinterpret_cast<float*>(nt_smem) + (threadIdx.x >> 6) * (32 * TS_);
        const int lr = lane & 15, lg = lane >> 4;
        const float wsc = wh_scale ? wh_scale[0] : 1.0f;
        __syncthreads();                                      // every wave is done with the operand images of the last chunk
#pragma unroll
        for (int h = 0; h < 2; ++h) {
#pragma unroll
          for (int a2 = 0; a2 < 2; ++a2)
#pragma unroll
            for (int b = 0; b < NTCfg<T>::NB; ++b)
#pragma unroll
              for (int r = 0; r < 4; ++r) tile[(a2 * 16 + lg + 4 * r) * TS_ + b * 16 + lr] = (float)acc[2 * h + a2][b][r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave writes and reads: LDS is in order per wave
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const int rr = (lane >> 3) + 8 * i, cv = (lane & 7) * 4;
            const int64_t m = m0 + wr * 64 + h * 32 + rr;
            const int n = n0 + wc * (NTCfg<T>::CW / 2) + cv;
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4 t = *reinterpret_cast<const f4*>(tile + rr * TS_ + cv);
            if (m < nrows && n < Mp) {                        // Mp is a multiple of 32: a vector never straddles the edge
              *reinterpret_cast<f4*>(W + m * Mp + n) = t;
              auto put = [&](auto sp) {
                if (!Wh) return;
                using SP = decltype(sp);
                typename SP::V4 pv[SP::NP];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                  typename SP::E pc[SP::NP];
                  SP::split(t[e] * wsc, pc);
#pragma unroll
                  for (int q = 0; q < SP::NP; ++q) pv[q][e] = pc[q];
                }
#pragma unroll
                for (int q = 0; q < SP::NP; ++q)
                  *reinterpret_cast<typename SP::V4*>(reinterpret_cast<typename SP::E*>(Wh) + q * wh_stride + m * Mp + n) = pv[q];
              };
              if (wh_mode == 2) put(SplitF16{}); else put(SplitBf16{});
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        return;
      }
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + nt_acc_row<T>(wr, a, lane, r);
#pragma unroll
        for (int b = 0; b < NTCfg<T>::NB; ++b) {
          const int n = n0 + nt_acc_col<T>(wc, b, lane);
          if (m < nrows && n < Mp) W[m * Mp + n] = (TN)acc[a][b][r];
        }
      }
  }
  __device__ __forceinline__ void finish(int64_t m0, int, ECtx& ec, char* smem, int wr, int wc, int lane) const {
    if (!qpart) return;                                     // the derivative product has no row norms to report
    T* rsum = reinterpret_cast<T*>(smem);
    nt_rowsum_finish<T>(ec.rs, rsum, wr, wc, lane);
    if (threadIdx.x < GDRF_TILE) {
      const int64_t m = m0 + threadIdx.x;
      if (m < nrows) qpart[(int64_t)ec.ct * ldq + m] = (TN)rsum[threadIdx.x];
    }
  }
};

// sum_n,j Wbar[n][j] W[n][j] and sum Wbar[n][j] Wd[n][j] (per-workgroup partials [grid][2], deterministic order): the K_nm parts of the
// variance / lengthscale gradients when W' = (dK_nm / d log ls) Linv^T is available (FwdWProb<.., DERIV>)
template <typename T>
__global__ __launch_bounds__(256) void wbar_dot_kernel(const T* __restrict__ Wbar, const T* __restrict__ W, const T* __restrict__ Wd, int64_t nrows,
                                                       int Mp, double* __restrict__ part) {
  using V = typename Vec16<T>::type;
  constexpr int VE = Vec16<T>::N;
  __shared__ double scratch[16];
  const int64_t nvec = nrows * Mp / VE;                     // Mp is a multiple of 32; padded columns hold zeros
  double s1 = 0, s2 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const V b = reinterpret_cast<const V*>(Wbar)[i], w = reinterpret_cast<const V*>(W)[i], d = reinterpret_cast<const V*>(Wd)[i];
    T a1 = 0, a2 = 0;
#pragma unroll
    for (int e = 0; e < VE; ++e) { a1 += b[e] * w[e]; a2 += b[e] * d[e]; }
    s1 += (double)a1; s2 += (double)a2;
  }
  const double t1 = block_sum(s1, scratch), t2 = block_sum(s2, scratch);
  if (threadIdx.x == 0) { part[2 * (int64_t)blockIdx.x] = t1; part[2 * (int64_t)blockIdx.x + 1] = t2; }
}

// (1b) loc = W U^T on the matrix cores: Bt = zero-padded u_loc [128][Mp]; stores loc[k][n] for k < K
// loc = W U^T for a handful of topics as a streaming pass over W (K <= LOC_KMAX): on the NT core the K columns are padded to a 128-wide
// tile, so 12.8x the useful MFMAs run beside fwd_t (1.1 ms of f32 matrix work there).  Here 16 lanes share a row (16-byte loads, 256 contiguous
// bytes per step), two row groups per pass reuse every U vector read from LDS, the 16 partial sums meet by xor-shuffles: the 2 GB of W once.
#define LOC_KMAX 16
template <typename T>
__global__ __launch_bounds__(256) void loc_rows_kernel(const T* __restrict__ W, int64_t nrows, int Mp, int K, const T* __restrict__ U /*[>=K][Mp]*/,
                                                       T* __restrict__ loc, int64_t ldk) {
  using VT = typename Vec16<T>::type;
  constexpr int VE = Vec16<T>::N;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Us = reinterpret_cast<T*>(smem);                   // [K][Mp]
  for (int e = threadIdx.x * VE; e < K * Mp; e += blockDim.x * VE) *reinterpret_cast<VT*>(Us + e) = *reinterpret_cast<const VT*>(U + e);
  __syncthreads();
  const int lane = threadIdx.x & 63, sub = lane & 15, grp = lane >> 4, wave = threadIdx.x >> 6;
  const int nsteps = Mp / (16 * VE);                    // Mp is a multiple of 32; the host checks Mp % (16 VE) == 0
  const int64_t rows_per_pass = 8 * 4;                  // 4 waves x 2 row groups x 4 rows
  for (int64_t r0 = (int64_t)blockIdx.x * rows_per_pass + wave * 8; r0 < nrows; r0 += (int64_t)gridDim.x * rows_per_pass) {
    const int64_t ra = r0 + grp, rb = r0 + 4 + grp;
    const T* wa = W + (ra < nrows ? ra : 0) * Mp + sub * VE;
    const T* wb = W + (rb < nrows ? rb : 0) * Mp + sub * VE;
    T acca[LOC_KMAX], accb[LOC_KMAX];
#pragma unroll
    for (int k = 0; k < LOC_KMAX; ++k) { acca[k] = 0; accb[k] = 0; }
    for (int j = 0; j < nsteps; ++j) {
      const VT xa = *reinterpret_cast<const VT*>(wa + j * 16 * VE), xb = *reinterpret_cast<const VT*>(wb + j * 16 * VE);
#pragma unroll
      for (int k = 0; k < LOC_KMAX; ++k) if (k < K) {
        const VT u = *reinterpret_cast<const VT*>(Us + k * Mp + j * 16 * VE + sub * VE);
#pragma unroll
        for (int e = 0; e < VE; ++e) { acca[k] += xa[e] * u[e]; accb[k] += xb[e] * u[e]; }
      }
    }
#pragma unroll
    for (int k = 0; k < LOC_KMAX; ++k) if (k < K) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) { acca[k] += __shfl_xor(acca[k], o, 64); accb[k] += __shfl_xor(accb[k], o, 64); }
      if (sub == 0) {
        if (ra < nrows) loc[(int64_t)k * ldk + ra] = acca[k];
        if (rb < nrows) loc[(int64_t)k * ldk + rb] = accb[k];
      }
    }
  }
}

template <typename T> struct LocProb : NTDefaultMap, NTPlainA<T>, NTNoExtra {
  using V = typename Vec16<T>::type;
  static constexpr bool SCALE_A = false;
  static constexpr bool A_PER_REP = false;
  static constexpr int DEPTH = 1;
  const T* W; int64_t nrows; int Mp, K;
  const T* Upad; T* loc; int64_t ldk;
  struct ACtx { const T* p[NTCfg<T>::VPT]; };
  struct ECtx {};
  __device__ __forceinline__ int col_tiles() const { return 1; }
  __device__ __forceinline__ bool loop_cols() const { return false; }
  __device__ __forceinline__ int a_reuse() const { return 1; }
  __device__ __forceinline__ void krange(int64_t, int, int, int& kb, int& ke) const { kb = 0; ke = Mp; }
  __device__ __forceinline__ void prepA(ACtx& c, int64_t m0, int, char*) const {
#pragma unroll
    for (int i = 0; i < NTCfg<T>::VPT; ++i) {
      const int64_t r = m0 + nt_stage_row<T>(i);
      c.p[i] = (r < nrows) ? W + r * Mp : nullptr;
    }
  }
  __device__ __forceinline__ void prepE(ECtx&, int64_t, int) const {}
  __device__ __forceinline__ V loadA(const ACtx& c, int i, int k, int, int) const {
    return c.p[i] ? *reinterpret_cast<const V*>(c.p[i] + k) : vzero<T>();
  }
  __device__ __forceinline__ V loadB(int, int i, int k, int, int) const {
    return *reinterpret_cast<const V*>(Upad + (int64_t)nt_stage_row<T>(i) * Mp + k);
  }
  template <class Acc, int NB_>
  __device__ __forceinline__ void tile_done(Acc (&acc)[4][NB_], int64_t m0, int, int, ECtx&, int wr, int wc, int lane) const {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + nt_acc_row<T>(wr, a, lane, r);
        if (m >= nrows) continue;
#pragma unroll
        for (int b = 0; b < NTCfg<T>::NB; ++b) {
          const int k = nt_acc_col<T>(wc, b, lane);
          if (k < K) loc[(int64_t)k * ldk + m] = acc[a][b][r];
        }
      }
  }
  __device__ __forceinline__ void finish(int64_t, int, ECtx&, char*, int, int, int) const {}
};

// (2) T_k = W S_k (never stored) -> tt[k][n] = sum_j T_k[n][j]^2 ; one workgroup walks all column tiles
template <typename T> struct FwdTProb : NTXcdRowBatchMap, NTPlainA<T>, NTNoExtra {
  using V = typename Vec16<T>::type;
  static constexpr bool SCALE_A = false;
  static constexpr bool A_PER_REP = false;
  static constexpr int DEPTH = 1;
  const T* W; int64_t nrows; int Mp;
  const T* ST;                     // [K][Mp][Mp], ST[k][j][i] = S_k[i][j]
  T* tt; int64_t ldt;              // [K][ldt]
  T* Tst; int64_t t_bs, t_ts;      // optional T_k kept for the backward (nullptr: not stored), chunk-major blocks:
                                   // [K][row tile][k chunk][128 rows][BK]; t_bs / t_ts = elements per topic / row tile
                                   // (both padded away from powers of two: HBM channel interleave)
  struct ACtx { const T* p[NTCfg<T>::VPT]; };
  struct ECtx { T rs[4][4]; };
  __device__ __forceinline__ int col_tiles() const { return (Mp + NTCfg<T>::CW - 1) / NTCfg<T>::CW; }
  __device__ __forceinline__ bool loop_cols() const { return true; }
  __device__ __forceinline__ int a_reuse() const { return 1; }
  __device__ __forceinline__ void krange(int64_t, int n0, int, int& kb, int& ke) const { kb = n0; ke = Mp; }
  __device__ __forceinline__ void prepA(ACtx& c, int64_t m0, int, char*) const {
#pragma unroll
    for (int i = 0; i < NTCfg<T>::VPT; ++i) {
      const int64_t r = m0 + nt_stage_row<T>(i);
      c.p[i] = (r < nrows) ? W + r * Mp : nullptr;
    }
  }
  __device__ __forceinline__ void prepE(ECtx& e, int64_t, int) const {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) e.rs[a][r] = 0;
  }
  __device__ __forceinline__ V loadA(const ACtx& c, int i, int k, int, int) const {
    return c.p[i] ? *reinterpret_cast<const V*>(c.p[i] + k) : vzero<T>();
  }
  __device__ __forceinline__ V loadB(int n0, int i, int k, int, int bz) const {
    const int c = n0 + nt_stage_row<T>(i);
    return (c < Mp) ? *reinterpret_cast<const V*>(ST + ((int64_t)bz * Mp + c) * Mp + k) : vzero<T>();
  }
  template <class Acc, int NB_>
  __device__ __forceinline__ void tile_done(Acc (&acc)[4][NB_], int64_t m0, int n0, int bz, ECtx& e, int wr, int wc, int lane) const {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < NTCfg<T>::NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) e.rs[a][r] += acc[a][b][r] * acc[a][b][r];
    if (Tst && m0 < nrows) {          // padding workgroups of the rounded-up grid own no tile
      constexpr int BK = NTCfg<T>::BK;
      T* base = Tst + (int64_t)bz * t_bs + (m0 / GDRF_TILE) * t_ts;     // this row tile's blocks
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int row = nt_acc_row<T>(wr, a, lane, r);
#pragma unroll
          for (int b = 0; b < NTCfg<T>::NB; ++b) {
            const int n = n0 + nt_acc_col<T>(wc, b, lane);
            if (n < Mp) base[((int64_t)(n / BK) * GDRF_TILE + row) * BK + (n % BK)] = acc[a][b][r];
          }
        }
    }
  }
  __device__ __forceinline__ void finish(int64_t m0, int bz, ECtx& e, char* smem, int wr, int wc, int lane) const {
    T* rsum = reinterpret_cast<T*>(smem);      // [128]
    nt_rowsum_finish<T>(e.rs, rsum, wr, wc, lane);
    if (threadIdx.x < GDRF_TILE) {
      const int64_t m = m0 + threadIdx.x;
      if (m < nrows) tt[(int64_t)bz * ldt + m] = rsum[threadIdx.x];
    }
  }
};

// (3) Wbar = sum_k diag(2 vbar_k) W B_k + locbar^T U - 2 diag(asum) W.  One staged chunk of W serves all K
//     topics (a_reuse = K): the per-(topic,row) factor 2 vbar_kn sits in an LDS table and scales the A fragments.
template <typename T> struct BwdWbarProb : NTXcdMap, NTPlainA<T> {
  using V = typename Vec16<T>::type;
  static constexpr bool SCALE_A = true;
  static constexpr bool A_PER_REP = false;
  static constexpr int DEPTH = 1;
  const T* W; int64_t nrows; int M, Mp, K;
  const T* Bm;                     // [K][Mp][Mp] symmetric B_k = S_k S_k^T
  const T* vbar; const T* locbar; int64_t ldk;   // [K][ldk]
  const T* asum;                   // [nrows] a_n * sum_k vbar_kn
  const T* U;                      // [K][M] u_loc (unpadded)
  T* Wbar;
  struct ACtx { const T* p[NTCfg<T>::VPT]; };
  struct ECtx {};
  __device__ __forceinline__ int col_tiles() const { return (Mp + NTCfg<T>::CW - 1) / NTCfg<T>::CW; }
  __device__ __forceinline__ bool loop_cols() const { return false; }
  __device__ __forceinline__ int a_reuse() const { return K; }
  __device__ __forceinline__ void krange(int64_t, int, int, int& kb, int& ke) const { kb = 0; ke = Mp; }
  __device__ __forceinline__ void prepA(ACtx& c, int64_t m0, int, char* extra) const {
    T* sc = reinterpret_cast<T*>(extra);          // [K][128]
    for (int e = threadIdx.x; e < K * GDRF_TILE; e += 256) {
      const int k = e / GDRF_TILE, r = e - k * GDRF_TILE;
      sc[e] = (m0 + r < nrows) ? T(2) * vbar[(int64_t)k * ldk + m0 + r] : T(0);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NTCfg<T>::VPT; ++i) {
      const int64_t r = m0 + nt_stage_row<T>(i);
      c.p[i] = (r < nrows) ? W + r * Mp : nullptr;
    }
  }
  __device__ __forceinline__ void prepE(ECtx&, int64_t, int) const {}
  __device__ __forceinline__ V loadA(const ACtx& c, int i, int k, int, int) const {
    return c.p[i] ? *reinterpret_cast<const V*>(c.p[i] + k) : vzero<T>();
  }
  __device__ __forceinline__ V loadB(int n0, int i, int k, int rep, int) const {
    const int c = n0 + nt_stage_row<T>(i);
    return (c < Mp) ? *reinterpret_cast<const V*>(Bm + ((int64_t)rep * Mp + c) * Mp + k) : vzero<T>();
  }
  // rank-K epilogue term locbar^T U as extra MFMA chunk(s): As[row][kk] = locbar[k][m0+row], Bs[col][kk] = U[k][n0+col]
  __device__ __forceinline__ int extra_chunks() const { return (K + NTCfg<T>::BK - 1) / NTCfg<T>::BK; }
  __device__ __forceinline__ void fill_extra(T* As, T* Bs, int x, int64_t m0, int n0) const {
    constexpr int BK = NTCfg<T>::BK, LDK = NTCfg<T>::LDK, CW = NTCfg<T>::CW;
    for (int e = threadIdx.x; e < GDRF_TILE * BK; e += 256) {
      const int kk = e / GDRF_TILE, row = e - kk * GDRF_TILE, k = x * BK + kk;
      As[row * LDK + kk] = (k < K && m0 + row < nrows) ? locbar[(int64_t)k * ldk + m0 + row] : T(0);
    }
    for (int e = threadIdx.x; e < CW * BK; e += 256) {
      const int kk = e / CW, col = e - kk * CW, k = x * BK + kk;
      Bs[col * LDK + kk] = (k < K && n0 + col < M) ? U[(int64_t)k * M + n0 + col] : T(0);
    }
  }
  template <class Acc, int NB_>
  __device__ __forceinline__ void tile_done(Acc (&acc)[4][NB_], int64_t m0, int n0, int, ECtx&, int wr, int wc, int lane) const {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + nt_acc_row<T>(wr, a, lane, r);
        if (m >= nrows) continue;
        const T as2 = T(2) * asum[m];
#pragma unroll
        for (int b = 0; b < NTCfg<T>::NB; ++b) {
          const int n = n0 + nt_acc_col<T>(wc, b, lane);
          if (n >= Mp) continue;
          T v = acc[a][b][r] - as2 * W[m * Mp + n];
          Wbar[m * Mp + n] = v;
        }
      }
  }
  __device__ __forceinline__ void finish(int64_t, int, ECtx&, char*, int, int, int) const {}
};

// (3') the same Wbar from the stored T_k:  Wbar = sum_k diag(2 vbar_k) T_k S_k^T + locbar^T U - 2 diag(asum) W.
//      Triangular (j <= i): half the flops of (3).  A = T_k (changes with the topic: A_PER_REP), Bt = S_k row-major.
template <typename T> struct BwdWbarTProb : NTXcdPairMap, NTPlainA<T> {
  using V = typename Vec16<T>::type;
  static constexpr bool SCALE_A = true;
  static constexpr bool A_PER_REP = true;
  static constexpr int DEPTH = 1;   // 2 compiles, but hipcc drains it with vmcnt(0) at every wait: no gain
  const T* Tst; int64_t t_bs, t_ts; const T* W; int64_t nrows; int M, Mp, K;
  const T* S;                      // [K][Mp][Mp] lower triangular
  const T* vbar; const T* locbar; int64_t ldk;
  const T* asum; const T* U; T* Wbar;
  struct ACtx { int64_t tile_off; bool ok; };
  struct ECtx {};
  __device__ __forceinline__ int col_tiles() const { return (Mp + NTCfg<T>::CW - 1) / NTCfg<T>::CW; }
  __device__ __forceinline__ bool loop_cols() const { return false; }
  __device__ __forceinline__ int a_reuse() const { return K; }
  __device__ __forceinline__ void krange(int64_t, int n0, int, int& kb, int& ke) const {
    kb = 0; ke = n0 + NTCfg<T>::CW; if (ke > Mp) ke = Mp;
  }
  __device__ __forceinline__ void prepA(ACtx& c, int64_t m0, int, char* extra) const {
    T* sc = reinterpret_cast<T*>(extra);          // [K][128]
    for (int e = threadIdx.x; e < K * GDRF_TILE; e += 256) {
      const int k = e / GDRF_TILE, r = e - k * GDRF_TILE;
      sc[e] = (m0 + r < nrows) ? T(2) * vbar[(int64_t)k * ldk + m0 + r] : T(0);
    }
    __syncthreads();
    c.tile_off = (m0 / GDRF_TILE) * t_ts;
    c.ok = m0 < nrows;       // the launch grid is rounded up to a multiple of 8: padding workgroups own no tile
  }
  __device__ __forceinline__ void prepE(ECtx&, int64_t, int) const {}
  // chunk-major blocks [k chunk][128 rows][BK]: one staged chunk is 16 KB of contiguous memory (rows beyond nrows
  // of the last tile were written by the forward as zeros-times-S = 0)
  __device__ __forceinline__ V loadA(const ACtx& c, int i, int k, int rep, int) const {
    constexpr int BK = NTCfg<T>::BK;
    if (!c.ok) return vzero<T>();
    return *reinterpret_cast<const V*>(Tst + (int64_t)rep * t_bs + c.tile_off +
                                       ((int64_t)(k / BK) * GDRF_TILE + nt_stage_row<T>(i)) * BK + (k % BK));
  }
  __device__ __forceinline__ V loadB(int n0, int i, int k, int rep, int) const {
    const int c = n0 + nt_stage_row<T>(i);
    return (c < Mp) ? *reinterpret_cast<const V*>(S + ((int64_t)rep * Mp + c) * Mp + k) : vzero<T>();
  }
  // rank-K epilogue term locbar^T U as extra MFMA chunk(s): As[row][kk] = locbar[k][m0+row], Bs[col][kk] = U[k][n0+col]
  __device__ __forceinline__ int extra_chunks() const { return (K + NTCfg<T>::BK - 1) / NTCfg<T>::BK; }
  __device__ __forceinline__ void fill_extra(T* As, T* Bs, int x, int64_t m0, int n0) const {
    constexpr int BK = NTCfg<T>::BK, LDK = NTCfg<T>::LDK, CW = NTCfg<T>::CW;
    for (int e = threadIdx.x; e < GDRF_TILE * BK; e += 256) {
      const int kk = e / GDRF_TILE, row = e - kk * GDRF_TILE, k = x * BK + kk;
      As[row * LDK + kk] = (k < K && m0 + row < nrows) ? locbar[(int64_t)k * ldk + m0 + row] : T(0);
    }
    for (int e = threadIdx.x; e < CW * BK; e += 256) {
      const int kk = e / CW, col = e - kk * CW, k = x * BK + kk;
      Bs[col * LDK + kk] = (k < K && n0 + col < M) ? U[(int64_t)k * M + n0 + col] : T(0);
    }
  }
  template <class Acc, int NB_>
  __device__ __forceinline__ void tile_done(Acc (&acc)[4][NB_], int64_t m0, int n0, int, ECtx&, int wr, int wc, int lane) const {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + nt_acc_row<T>(wr, a, lane, r);
        if (m >= nrows) continue;
        const T as2 = T(2) * asum[m];
#pragma unroll
        for (int b = 0; b < NTCfg<T>::NB; ++b) {
          const int n = n0 + nt_acc_col<T>(wc, b, lane);
          if (n >= Mp) continue;
          T v = acc[a][b][r] - as2 * W[m * Mp + n];
          Wbar[m * Mp + n] = v;
        }
      }
  }
  __device__ __forceinline__ void finish(int64_t, int, ECtx&, char*, int, int, int) const {}
};

// (4) Knm_bar = Wbar Linv (never stored) -> sum Knm_bar*Knm and sum Knm_bar*dKnm/dlog(ls) per workgroup
// LZ: also the per-inducing-point sums  G[j][d] = sum_n Kbar[n][j] * dk/dr2(x_n, z_j) * (z_jd - x_nd)  that the gradient of
// learnable inducing inputs needs (sparse_gdrf.py:79-88, fixed_inducing_points=False); one partial per (row tile, column).
template <typename T, typename TN, bool LZ = false> struct BwdKnmProb : NTXcdRowMap, NTNoExtra {
  using V = typename Vec16<T>::type;
  static constexpr int MIN_WGS = (sizeof(T) == 8 && !LZ) ? 3 : 2;   // f64: three workgroups per CU hide each other's epilogues
#ifndef GDRF_NO_TRI
  static constexpr int TRI = 2;            // Kbar[n][col] = sum_{k >= col} Wbar[n][k] LinvT[col][k]
#endif
  static constexpr bool SCALE_A = false;
  static constexpr bool A_PER_REP = false;
  static constexpr int DEPTH = 1;
  const TN* Wbar; int64_t nrows; int M, Mp, D, kind;
  const T* LinvT;                  // [Mp][Mp], LinvT[i][j] = Linv[j][i]
  const T* Knm;                    // [nrows][Mp] solve-precision K_nm (same buffer the forward consumed)
  const TN* X; const T* Z; const Hyper* h;
  double* part;                    // [gridDim.x][3]: sum Kbar*K, sum Kbar*dK/dlog(ls), sum Kbar*dK/dlog(alpha)
  double* zpart;                   // LZ: [row tiles][M][D]
  struct ACtx { const TN* p[NTCfg<T>::VPT]; };
  struct ECtx { T s1, s2, s3; T zs[LZ ? NTCfg<T>::NB : 1][LZ ? GDRF_DMAX : 1]; int n0; };
  __device__ __forceinline__ int col_tiles() const { return (Mp + NTCfg<T>::CW - 1) / NTCfg<T>::CW; }
  __device__ __forceinline__ bool loop_cols() const { return false; }
  __device__ __forceinline__ int a_reuse() const { return 1; }
  __device__ __forceinline__ void krange(int64_t, int n0, int, int& kb, int& ke) const { kb = n0; ke = Mp; }
  __device__ __forceinline__ void prepA(ACtx& c, int64_t m0, int, char*) const {
#pragma unroll
    for (int i = 0; i < NTCfg<T>::VPT; ++i) {
      const int64_t r = m0 + nt_stage_row<T>(i);
      c.p[i] = (r < nrows) ? Wbar + r * Mp : nullptr;
    }
  }
  __device__ __forceinline__ void prepE(ECtx& e, int64_t, int) const {
    e.s1 = 0; e.s2 = 0; e.s3 = 0; e.n0 = 0;
    if constexpr (LZ) {
#pragma unroll
      for (int b = 0; b < NTCfg<T>::NB; ++b)
#pragma unroll
        for (int d = 0; d < GDRF_DMAX; ++d) e.zs[b][d] = 0;
    }
  }
  using AVec = TN __attribute__((ext_vector_type(Vec16<T>::N)));        // same element count, N-side element type
  __device__ __forceinline__ AVec loadA(const ACtx& c, int i, int k, int, int) const {
    AVec t;
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) t[e] = 0;
    if (c.p[i]) t = *reinterpret_cast<const AVec*>(c.p[i] + k);
    return t;
  }
  template <class... X> __device__ __forceinline__ V a_to_lds(const AVec& t, X&&...) const {
    V v;
#pragma unroll
    for (int e = 0; e < Vec16<T>::N; ++e) v[e] = (T)t[e];
    return v;
  }
  __device__ __forceinline__ V loadB(int n0, int i, int k, int, int) const {
    const int c = n0 + nt_stage_row<T>(i);
    return (c < Mp) ? *reinterpret_cast<const V*>(LinvT + (int64_t)c * Mp + k) : vzero<T>();
  }
  // per-strip arithmetic of the f64 epilogue for one kernel kind (straight-line: no per-element branch on the kind)
  typedef T KT2 __attribute__((ext_vector_type(2)));
  template <int KD, int DD, int NR>
  __device__ __forceinline__ void strip_math(const KT2 (&kv)[NR], const T* __restrict__ trow, int tstride, const TN (&x)[NR][DD],
                                             const T (&zc)[2][DD], const bool (&rok)[NR], const bool (&cok)[2],
                                             T ils2, T al, ECtx& e) const {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const KT2 kb = *reinterpret_cast<const KT2*>(trow + 4 * i * tstride);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const T kbar = (rok[i] && cok[j]) ? kb[j] : T(0);
        const T kvv = kv[i][j];
        T r2 = 0;
#pragma unroll
        for (int d = 0; d < DD; ++d) { const T t = (d < D ? (T)x[i][d] : T(0)) - zc[j][d]; r2 += t * t; }
        r2 *= ils2;
        e.s1 += kbar * kvv;
        e.s2 += kbar * dcov_dlogls_from_k<T>(KD, kvv, r2, al);
        if constexpr (KD == 4) e.s3 += kbar * dcov_dlogalpha_from_k<T>(KD, kvv, r2, al);
        if constexpr (LZ) {
          const T w = kbar * dcov_dr2_from_k<T>(KD, kvv, r2, al);
#pragma unroll
          for (int d = 0; d < DD; ++d) e.zs[j][d] += w * (zc[j][d] - (d < D ? (T)x[i][d] : T(0)));
        }
      }
    }
  }
  // f64 solve.  The accumulator layout (lane = column, 4 scattered rows) makes the K_nm reads of the straightforward
  // epilogue 32 dependent 8-byte loads per lane, which took as long as the GEMM loop.  Each wave instead transposes its
  // 64 x 32 quadrant through a private LDS tile, 16 rows at a time, so that a lane owns two adjacent columns: K_nm comes in
  // as four independent 16-byte loads per strip (a row segment of 256 B per 16 lanes).  Addresses are 32-bit offsets from
  // the tile's uniform base pointers (64-bit per-row addresses for the whole tile, hoisted above the strips, spilled).
  template <int DD, class Acc, int NB_>
  __device__ __forceinline__ void epi_f64(Acc (&acc)[4][NB_], int64_t m0, int n0, ECtx& e, int wr, int wc, int lane) const {
    if (m0 >= nrows) return;                                // a padding workgroup of the XCD-aware grid (uniform: no barrier is skipped by a part of it)
    const T ils2 = (T)h->inv_ls2, al = (T)h->alpha;
    extern __shared__ __attribute__((aligned(16))) char nt_smem[];
    constexpr int TS_ = 48;                                 // tile row stride in doubles: consecutive rows 128 B apart mod 256
    T* tile = reinterpret_cast<T*>(nt_smem) + (threadIdx.x >> 6) * (16 * TS_);
    const int lr = lane & 15, lg = lane >> 4;
    const int nc = n0 + wc * (NTCfg<T>::CW / 2) + 2 * lr;  // this lane's columns nc, nc + 1
    const int ncl = nc < Mp ? nc : 0;                       // Mp is even: a pair never straddles the padded edge
    const bool cok[2] = {nc < M, nc + 1 < M};
    T zc[2][DD];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int nz = cok[j] ? nc + j : 0;
#pragma unroll
      for (int d = 0; d < DD; ++d) { const T v = Z[nz * D + (d < D ? d : 0)]; zc[j][d] = d < D ? v : T(0); }
    }
    const T* __restrict__ Kt = Knm + m0 * Mp;              // uniform bases of this row tile
    const TN* __restrict__ Xt = X + m0 * D;
    const int rmax = (int)((nrows - m0 < GDRF_TILE ? nrows - m0 : GDRF_TILE) - 1);   // last valid row of the tile (m0 < nrows)
    const int r0 = wr * 64 + lg;                            // strip a, slot i: tile row r0 + a * 16 + 4 * i
    __syncthreads();                                        // every wave is done with the operand images of the last chunk
    constexpr int NR = 2;                                   // rows per load batch: 4 cost 8 more registers (166: no longer two waves beside a TN wave)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
      for (int hb = 0; hb < 4 / NR; ++hb) {
        KT2 kv[NR]; TN x[NR][DD]; bool rok[NR];
#pragma unroll
        for (int i = 0; i < NR; ++i) {
          const int rl = r0 + a * 16 + 4 * (hb * NR + i);
          rok[i] = rl <= rmax;
          const unsigned rc = (unsigned)(rok[i] ? rl : rmax);
          kv[i] = *reinterpret_cast<const KT2*>(Kt + (rc * (unsigned)Mp + (unsigned)ncl));
#pragma unroll
          for (int d = 0; d < DD; ++d) x[i][d] = Xt[rc * (unsigned)D + (unsigned)(d < D ? d : 0)];
        }
        if (hb == 0) {
#pragma unroll
          for (int b = 0; b < NTCfg<T>::NB; ++b)
#pragma unroll
            for (int r = 0; r < 4; ++r) tile[(lg + 4 * r) * TS_ + b * 16 + lr] = acc[a][b][r];
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // same wave writes and reads: LDS is in order per wave
        }
        const T* trow = tile + (lg + 4 * hb * NR) * TS_ + 2 * lr;   // slot i: row lg + 4 * (hb * NR + i) of the strip
        switch (kind) {
          case 0: strip_math<0, DD, NR>(kv, trow, TS_, x, zc, rok, cok, ils2, al, e); break;
          case 1: strip_math<1, DD, NR>(kv, trow, TS_, x, zc, rok, cok, ils2, al, e); break;
          case 2: strip_math<2, DD, NR>(kv, trow, TS_, x, zc, rok, cok, ils2, al, e); break;
          case 3: strip_math<3, DD, NR>(kv, trow, TS_, x, zc, rok, cok, ils2, al, e); break;
          default: strip_math<4, DD, NR>(kv, trow, TS_, x, zc, rok, cok, ils2, al, e); break;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);                  // one batch at a time
      }
    }
  }
  template <class Acc, int NB_>
  __device__ __forceinline__ void tile_done(Acc (&acc)[4][NB_], int64_t m0, int n0, int, ECtx& e, int wr, int wc, int lane) const {
    const T ils2 = (T)h->inv_ls2, al = (T)h->alpha;
    e.n0 = n0;
#ifndef GDRF_BWDKNM_SCALAR_EPILOGUE
    if constexpr (sizeof(T) == 8 && !LZ) {
      if (D <= 2) {            // the reference's worlds are 1-D (time) or 2-D (space); more dimensions take the generic form below
        epi_f64<2>(acc, m0, n0, e, wr, wc, lane);
        return;
      }
    }
#endif
    T z[4][GDRF_DMAX];
#pragma unroll
    for (int b = 0; b < NTCfg<T>::NB; ++b) {
      const int n = n0 + nt_acc_col<T>(wc, b, lane);
#pragma unroll
      for (int d = 0; d < GDRF_DMAX; ++d) z[b][d] = (n < M && d < D) ? Z[(int64_t)n * D + d] : T(0);
    }
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int64_t m = m0 + nt_acc_row<T>(wr, a, lane, r);
        if (m >= nrows) continue;
        T x[GDRF_DMAX];
#pragma unroll
        for (int d = 0; d < GDRF_DMAX; ++d) x[d] = (d < D) ? (T)X[m * D + d] : T(0);
#pragma unroll
        for (int b = 0; b < NTCfg<T>::NB; ++b) {
          const int n = n0 + nt_acc_col<T>(wc, b, lane);
          if (n >= M) continue;
          T r2 = 0;
#pragma unroll
          for (int d = 0; d < GDRF_DMAX; ++d) if (d < D) { const T t = x[d] - z[b][d]; r2 += t * t; }
          r2 *= ils2;
          const T kv = Knm[m * Mp + n];
          e.s1 += acc[a][b][r] * kv;
          e.s2 += acc[a][b][r] * dcov_dlogls_from_k<T>(kind, kv, r2, al);
          if (kind == 4) e.s3 += acc[a][b][r] * dcov_dlogalpha_from_k<T>(kind, kv, r2, al);
          if constexpr (LZ) {
            const T w = acc[a][b][r] * dcov_dr2_from_k<T>(kind, kv, r2, al);
#pragma unroll
            for (int d = 0; d < GDRF_DMAX; ++d) if (d < D) e.zs[b][d] += w * (z[b][d] - x[d]);
          }
        }
      }
  }
  // column (inside the workgroup tile) of the lane's bb-th inducing-point gradient sum: the accumulator column, or the
  // adjacent pair the LDS-transposed f64 epilogue gives the lane
  static __device__ __forceinline__ int zcol(int wc, int bb, int lane) {
    return nt_acc_col<T>(wc, bb, lane);
  }
  __device__ __forceinline__ void finish(int64_t m0, int, ECtx& e, char* smem, int wr, int wc, int lane) const {
    double* scratch = reinterpret_cast<double*>(smem);
    __syncthreads();
    const double a = block_sum((double)e.s1, scratch);
    const double b = block_sum((double)e.s2, scratch);
    const double c3 = block_sum((double)e.s3, scratch);
    if (threadIdx.x == 0) { part[3 * (int64_t)blockIdx.x] = a; part[3 * (int64_t)blockIdx.x + 1] = b; part[3 * (int64_t)blockIdx.x + 2] = c3; }
    if constexpr (LZ) {
      // column sums: the 4 lane groups of a wave hold 16 rows each, the two waves wr = 0, 1 the two halves of the row tile
      double* zb = reinterpret_cast<double*>(smem + 1024);       // [CW][DMAX]
#pragma unroll
      for (int bb = 0; bb < NTCfg<T>::NB; ++bb)
#pragma unroll
        for (int d = 0; d < GDRF_DMAX; ++d) {
          T v = e.zs[bb][d];
          v += __shfl_xor(v, 16, 64);
          v += __shfl_xor(v, 32, 64);
          e.zs[bb][d] = v;
        }
      __syncthreads();
      if (wr == 1 && (lane >> 4) == 0) {
#pragma unroll
        for (int bb = 0; bb < NTCfg<T>::NB; ++bb)
#pragma unroll
          for (int d = 0; d < GDRF_DMAX; ++d) zb[zcol(wc, bb, lane) * GDRF_DMAX + d] = (double)e.zs[bb][d];
      }
      __syncthreads();
      if (wr == 0 && (lane >> 4) == 0 && m0 < nrows) {
        const int64_t rt = m0 / GDRF_TILE;
#pragma unroll
        for (int bb = 0; bb < NTCfg<T>::NB; ++bb) {
          const int cl = zcol(wc, bb, lane), n = e.n0 + cl;
          if (n < M)
            for (int d = 0; d < D; ++d) zpart[(rt * M + n) * D + d] = (double)e.zs[bb][d] + zb[cl * GDRF_DMAX + d];
        }
      }
    }
  }
};

// =====================================================================================
// elbo_rows: per observation: variance, reparameterised draw, softmax link, Multinomial
// log-likelihood, both Normal site terms, and the row-local part of the backward.
// One thread per row.  Each lane walks its own row of counts (V int32, contiguous) straight from global memory: the lanes of a wave are
// V * 4 bytes apart, but every lane consumes whole lines over its V iterations, so the 200 MB stream is read once (0.87 ms at the
// headline size); staging the workgroup's rows through LDS with whole-line 16-byte loads (WSTAGE) was measured SLOWER (1.1-1.5 ms: an
// extra pass and barrier in a kernel that lives on occupancy), it is kept for A/B runs only.  KREG: K <= GDRF_KMAX, the per-topic values of a row live in registers; otherwise
// they are re-read from the (K, n) arrays (coalesced over the rows) and the softmax pull-back goes through LDS - any K.
// =====================================================================================
#define GDRF_KMAX 32

template <typename T, bool KREG, bool WSTAGE = false>
__global__ __launch_bounds__(128) void elbo_rows_kernel(
    int64_t nrows, int K, int V, const Hyper* __restrict__ h,
    const T* __restrict__ qpart, int nqpart, const T* __restrict__ loc, const T* __restrict__ tt, const T* __restrict__ eps,
    int64_t ldk, int64_t lde,
    const int32_t* __restrict__ ws, const T* __restrict__ phi,
    const T* __restrict__ mean /*may be null*/, int64_t mean_sk, int64_t mean_sn,
    T* __restrict__ qout, T* __restrict__ vbar, T* __restrict__ locbar, T* __restrict__ asum, T* __restrict__ mu_out,
    double* __restrict__ dpart /*[grid][4]*/, T* __restrict__ phibar_part /*[grid][K*V]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int RB = blockDim.x;
  constexpr int KR = KREG ? GDRF_KMAX : 1;
  double* scratch = reinterpret_cast<double*>(smem);    // [16]
  T* phiS = reinterpret_cast<T*>(smem + 128);           // [K*V]
  T* accS = phiS + K * V;                               // [K*V] phibar accumulator (owner-thread only)
  T* thS = accS + K * V;                                // [RB][K+1]
  T* pbS = thS + RB * (K + 1);                          // [RB][V+1]
  T* tbS = pbS + RB * (V + 1);                          // [RB][K+1], !KREG only
  T* wS = tbS + (KREG ? 0 : RB * (K + 1));              // [RB][V+1] the counts, WSTAGE only
  for (int e = threadIdx.x; e < K * V; e += RB) { phiS[e] = phi[e]; accS[e] = 0; }
  __syncthreads();
  const T var = (T)h->var, eta = (T)h->noise;
  const T feps = t_eps<T>();
  double s_site = 0, s_llw = 0, s_noise = 0, s_vd = 0;
  const int64_t nblk = (nrows + RB - 1) / RB;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t n = blk * RB + threadIdx.x;
    const bool ok = n < nrows;
    T* th = thS + threadIdx.x * (K + 1);
    T* pb = pbS + threadIdx.x * (V + 1);
    T* tbl = tbS + threadIdx.x * (K + 1);
    const T* wrow = wS + threadIdx.x * (V + 1);
    if constexpr (WSTAGE) {   // counts of this block's rows -> LDS (as T) with whole-line loads
      const int64_t e0 = blk * RB * (int64_t)V;
      int64_t cnt = (nrows - blk * RB < RB ? nrows - blk * RB : (int64_t)RB) * V;
      const int32_t* src = ws + e0;
      const int head = (int)((4 - (e0 & 3)) & 3);         // elements in front of the first 16-byte aligned one
      for (int64_t e = threadIdx.x; e < head && e < cnt; e += RB) wS[(e / V) * (V + 1) + e % V] = (T)src[e];
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      const int64_t nvec = cnt > head ? (cnt - head) >> 2 : 0;
      for (int64_t q = threadIdx.x; q < nvec; q += RB) {
        const i32x4 w4 = *reinterpret_cast<const i32x4*>(src + head + 4 * q);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const int64_t e = head + 4 * q + j; wS[(e / V) * (V + 1) + e % V] = (T)w4[j]; }
      }
      for (int64_t e = head + 4 * nvec + threadIdx.x; e < cnt; e += RB) wS[(e / V) * (V + 1) + e % V] = (T)src[e];
      __syncthreads();
    }
    T v[KR], mu[KR], ep[KR];
    T a = 0, vd = 0;
    if (ok) {
      T qn = 0;
      for (int c = 0; c < nqpart; ++c) qn += qpart[(int64_t)c * ldk + n];
      qout[n] = qn;
      a = (var - qn > T(0)) ? T(1) : T(0);
      const T v0 = a * (var - qn);
      auto topic = [&](int k, T& vk, T& ek, T& mk) {
        ek = eps[(int64_t)k * lde + n];
        vk = v0 + tt[(int64_t)k * ldk + n];
        mk = loc[(int64_t)k * ldk + n] + vk * ek;
        if (mean) mk += mean[(int64_t)k * mean_sk + n * mean_sn];   // f_loc + mean_function(xs): the site terms only see mu - f_loc
      };
      T mx = -3.0e38f;
      if constexpr (KREG) {
#pragma unroll
        for (int k = 0; k < KR; ++k) if (k < K) { topic(k, v[k], ep[k], mu[k]); mx = fmax(mx, mu[k]); }
      } else {
        for (int k = 0; k < K; ++k) { T vk, ek, mk; topic(k, vk, ek, mk); th[k] = mk; mx = fmax(mx, mk); }
      }
      T se = 0;
      if constexpr (KREG) {
#pragma unroll
        for (int k = 0; k < KR; ++k) if (k < K) { const T e = t_exp<T>(mu[k] - mx); th[k] = e; se += e; }
      } else {
        for (int k = 0; k < K; ++k) { const T e = t_exp<T>(th[k] - mx); th[k] = e; se += e; }
      }
      const T ise = T(1) / se;
      for (int k = 0; k < K; ++k) th[k] *= ise;
      // pass 1: p_v = sum_k theta_k phi_kv ; pass 2: log-likelihood and pbar = w * mask / p
      T ps = 0;
      for (int vv = 0; vv < V; ++vv) {
        T p = 0;
        for (int k = 0; k < K; ++k) p += th[k] * phiS[k * V + vv];
        pb[vv] = p;
        ps += p;
      }
      const T ips = T(1) / ps;
      T llw = 0;
      for (int vv = 0; vv < V; ++vv) {
        const T p = pb[vv];
        const T ph = p * ips;
        const T wv = WSTAGE ? wrow[vv] : (T)ws[n * V + vv];
        const bool inr = (ph > feps) && (ph < T(1) - feps);
        const T phc = fmin(fmax(ph, feps), T(1) - feps);
        llw += wv * t_log<T>(phc);
        pb[vv] = inr ? wv / p : T(0);
      }
      s_llw += (double)llw;
      // thetabar_k = sum_v phi_kv pbar_v ; softmax Jacobian
      // softmax pull-back mubar_k = theta_k (thetabar_k - sum_j theta_j thetabar_j).  mu = loc + v eps has a spread of tens of units (the
      // reference passes the predictive VARIANCE as the Normal's scale, quirk Q1), so one theta is 1 - O(1e-4) and the literal form
      // subtracts two numbers of size thetabar ~ sum_v w_v that agree to 4 digits: 1e-3 relative error in float.  With ANY constant c,
      // thetabar_k - dot = (thetabar_k - c) + sum_j theta_j (c - thetabar_j)   (sum_j theta_j = 1);  c = thetabar of the dominant topic
      // removes the large term from the sum - every product then carries a small theta_j or is exactly zero.
      T tb[KR];
      T cref = 0;
      if constexpr (KREG) {
#pragma unroll
        for (int k = 0; k < KR; ++k) if (k < K) {
          T s = 0;
          for (int vv = 0; vv < V; ++vv) s += phiS[k * V + vv] * pb[vv];
          tb[k] = s;
          if (mu[k] == mx) cref = s;
        }
      } else {
        T tmax = -1;
        for (int k = 0; k < K; ++k) {
          T s = 0;
          for (int vv = 0; vv < V; ++vv) s += phiS[k * V + vv] * pb[vv];
          tbl[k] = s;
          if (th[k] > tmax) { tmax = th[k]; cref = s; }
        }
      }
      T dot = 0;                                     // = sum_j theta_j (cref - thetabar_j)
      if constexpr (KREG) {
#pragma unroll
        for (int k = 0; k < KR; ++k) if (k < K) dot += th[k] * (cref - tb[k]);
      } else {
        for (int k = 0; k < K; ++k) dot += th[k] * (cref - tbl[k]);
      }
      T site = 0, ng = 0, vsum = 0;
      auto finish_topic = [&](int k, T vk, T ek, T mk, T tbk) {
        const T mub = th[k] * ((tbk - cref) + dot);
        const T s = vk + eta, r = vk / s, e2 = ek * ek;
        site += -t_log<T>(s) + t_log<T>(vk) - T(0.5) * e2 * r * r + T(0.5) * e2;
        const T dcdv = -T(1) / s + T(1) / vk - e2 * r * eta / (s * s);
        ng += -T(1) / s + e2 * r * r / s;
        const T vb = mub * ek + dcdv;
        vbar[(int64_t)k * ldk + n] = vb;
        locbar[(int64_t)k * ldk + n] = mub;
        if (mu_out) mu_out[(int64_t)k * ldk + n] = mk;
        vsum += vb;
      };
      if constexpr (KREG) {
#pragma unroll
        for (int k = 0; k < KR; ++k) if (k < K) finish_topic(k, v[k], ep[k], mu[k], tb[k]);
      } else {
        for (int k = 0; k < K; ++k) { T vk, ek, mk; topic(k, vk, ek, mk); finish_topic(k, vk, ek, mk, tbl[k]); }
      }
      vd = a * vsum;
      asum[n] = vd;
      s_site += (double)site; s_noise += (double)ng; s_vd += (double)vd;
    } else {
      for (int k = 0; k < K; ++k) th[k] = 0;
      for (int vv = 0; vv < V; ++vv) pb[vv] = 0;
    }
    __syncthreads();
    // phibar_kv += sum_rows theta_k pbar_v : each (k,v) pair has one owner thread
    for (int e = threadIdx.x; e < K * V; e += RB) {
      const int k = e / V, vv = e - k * V;
      T s = 0;
      for (int r = 0; r < RB; ++r) s += thS[r * (K + 1) + k] * pbS[r * (V + 1) + vv];
      accS[e] += s;
    }
    __syncthreads();
  }
  const double b0 = block_sum(s_site, scratch), b1 = block_sum(s_llw, scratch);
  const double b2 = block_sum(s_noise, scratch), b3 = block_sum(s_vd, scratch);
  if (threadIdx.x == 0) {
    dpart[4 * (int64_t)blockIdx.x + 0] = b0; dpart[4 * (int64_t)blockIdx.x + 1] = b1;
    dpart[4 * (int64_t)blockIdx.x + 2] = b2; dpart[4 * (int64_t)blockIdx.x + 3] = b3;
  }
  for (int e = threadIdx.x; e < K * V; e += RB) phibar_part[(int64_t)blockIdx.x * K * V + e] = accS[e];
}

// The per-row terms around a CALLER-SUPPLIED link function (the reference's `link_function` constructor argument,
// gdrf/models/abstract_gdrf.py:34-50, applied as `self._link_function(mu).transpose(-2, -1)` in sparse_gdrf.py:361): the link and
// its Jacobian are evaluated by the host between three launches of this kernel (one thread per row, any K; not a hot path).
//   phase 0: q, v, mu = loc + v eps (+ mean)            -> mu_out (K, ldk)
//   phase 1: theta = ext (K, ext_ld) as returned by the link: p = theta^T Phi, Multinomial log-likelihood of the normalised p,
//            thetabar_k = sum_v Phi_kv w_v / p_v - sum_v w_v / sum_v p_v  (theta need not sum to one) -> locbar (K, ldk), Phi-bar partials
//   phase 2: mubar = ext (K, ext_ld), the pull-back of thetabar through the link: both Normal sites and the row-local backward
// dpart[block][4]: phase 1 writes slot 1 (sum w log p), phase 2 slots 0, 2, 3 (site, d/d noise, a sum vbar) - same grid in both.
template <typename T>
__global__ __launch_bounds__(64) void elbo_rows_link_kernel(
    int phase, int64_t nrows, int K, int V, const Hyper* __restrict__ h,
    const T* __restrict__ qpart, int nqpart, const T* __restrict__ loc, const T* __restrict__ tt, const T* __restrict__ eps,
    int64_t ldk, int64_t lde, const int32_t* __restrict__ ws, const T* __restrict__ phi,
    const T* __restrict__ mean /*may be null*/, int64_t mean_sk, int64_t mean_sn,
    const T* __restrict__ ext, int64_t ext_ld,
    T* __restrict__ qout, T* __restrict__ vbar, T* __restrict__ locbar, T* __restrict__ asum, T* __restrict__ mu_out,
    double* __restrict__ dpart /*[grid][4]*/, T* __restrict__ phibar_part /*[grid][K*V]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int RB = blockDim.x;
  double* scratch = reinterpret_cast<double*>(smem);    // [16]
  T* phiS = reinterpret_cast<T*>(smem + 128);           // [K*V]
  T* accS = phiS + K * V;                               // [K*V]
  T* thS = accS + K * V;                                // [RB][K+1]
  T* pbS = thS + RB * (K + 1);                          // [RB][V+1]
  if (phase == 1) {
    for (int e = threadIdx.x; e < K * V; e += RB) { phiS[e] = phi[e]; accS[e] = 0; }
    __syncthreads();
  }
  const T var = (T)h->var, eta = (T)h->noise;
  const T feps = t_eps<T>();
  double s_site = 0, s_llw = 0, s_noise = 0, s_vd = 0;
  const int64_t nblk = (nrows + RB - 1) / RB;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t n = blk * RB + threadIdx.x;
    const bool ok = n < nrows;
    T* th = thS + threadIdx.x * (K + 1);
    T* pb = pbS + threadIdx.x * (V + 1);
    if (phase != 1) {
      if (ok) {
        T qn = 0;
        for (int c = 0; c < nqpart; ++c) qn += qpart[(int64_t)c * ldk + n];
        const T a = (var - qn > T(0)) ? T(1) : T(0);
        const T v0 = a * (var - qn);
        if (phase == 0) qout[n] = qn;
        T site = 0, ng = 0, vsum = 0;
        for (int k = 0; k < K; ++k) {
          const T ek = eps[(int64_t)k * lde + n];
          const T vk = v0 + tt[(int64_t)k * ldk + n];
          if (phase == 0) {
            T mk = loc[(int64_t)k * ldk + n] + vk * ek;
            if (mean) mk += mean[(int64_t)k * mean_sk + n * mean_sn];
            mu_out[(int64_t)k * ldk + n] = mk;
          } else {
            const T mub = ext[(int64_t)k * ext_ld + n];
            const T sk = vk + eta, r = vk / sk, e2 = ek * ek;
            site += -t_log<T>(sk) + t_log<T>(vk) - T(0.5) * e2 * r * r + T(0.5) * e2;
            const T dcdv = -T(1) / sk + T(1) / vk - e2 * r * eta / (sk * sk);
            ng += -T(1) / sk + e2 * r * r / sk;
            const T vb = mub * ek + dcdv;
            vbar[(int64_t)k * ldk + n] = vb;
            locbar[(int64_t)k * ldk + n] = mub;
            vsum += vb;
          }
        }
        if (phase == 2) {
          const T vd = a * vsum;
          asum[n] = vd;
          s_site += (double)site; s_noise += (double)ng; s_vd += (double)vd;
        }
      }
      continue;
    }
    // ---- phase 1
    if (ok) {
      for (int k = 0; k < K; ++k) th[k] = ext[(int64_t)k * ext_ld + n];
      T ps = 0;
      for (int vv = 0; vv < V; ++vv) {
        T p = 0;
        for (int k = 0; k < K; ++k) p += th[k] * phiS[k * V + vv];
        pb[vv] = p;
        ps += p;
      }
      const T ips = T(1) / ps;
      T llw = 0, wsum = 0;
      for (int vv = 0; vv < V; ++vv) {
        const T p = pb[vv];
        const T ph = p * ips;
        const T wv = (T)ws[n * V + vv];
        const bool inr = (ph > feps) && (ph < T(1) - feps);
        const T phc = fmin(fmax(ph, feps), T(1) - feps);
        llw += wv * t_log<T>(phc);
        pb[vv] = inr ? wv / p : T(0);
        wsum += inr ? wv : T(0);
      }
      s_llw += (double)llw;
      const T cn = wsum * ips;                          // d/dp_v' of -sum_v w_v log(sum p): the same for every v'
      for (int k = 0; k < K; ++k) {
        T sv = 0, rowsum = 0;
        for (int vv = 0; vv < V; ++vv) { sv += phiS[k * V + vv] * pb[vv]; rowsum += phiS[k * V + vv]; }
        locbar[(int64_t)k * ldk + n] = sv - cn * rowsum;
      }
      // the Phi gradient's constant part: theta_k (pbar_v - cn)
      for (int vv = 0; vv < V; ++vv) pb[vv] -= cn;
    } else {
      for (int k = 0; k < K; ++k) th[k] = 0;
      for (int vv = 0; vv < V; ++vv) pb[vv] = 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * V; e += RB) {
      const int k = e / V, vv = e - k * V;
      T sacc = 0;
      for (int r = 0; r < RB; ++r) sacc += thS[r * (K + 1) + k] * pbS[r * (V + 1) + vv];
      accS[e] += sacc;
    }
    __syncthreads();
  }
  if (phase == 0) return;
  if (phase == 1) {
    const double b1 = block_sum(s_llw, scratch);
    if (threadIdx.x == 0) dpart[4 * (int64_t)blockIdx.x + 1] = b1;
    for (int e = threadIdx.x; e < K * V; e += RB) phibar_part[(int64_t)blockIdx.x * K * V + e] = accS[e];
  } else {
    const double b0 = block_sum(s_site, scratch), b2 = block_sum(s_noise, scratch), b3 = block_sum(s_vd, scratch);
    if (threadIdx.x == 0) {
      dpart[4 * (int64_t)blockIdx.x + 0] = b0; dpart[4 * (int64_t)blockIdx.x + 2] = b2; dpart[4 * (int64_t)blockIdx.x + 3] = b3;
    }
  }
}

// The same per-row terms when the guide and the model evaluate the GP predictive at DIFFERENT inputs - the reference's quirk Q3
// (gdrf/models/sparse_gdrf.py:376-380: for a world other than the unit cube the guide scales its inputs twice, the model once).
// Guide side (subscript g): mu = loc_g + v_g eps, log q = -log v_g - eps^2 / 2.  Model side (m): log p = -log s_m - (d / s_m)^2 / 2 with
// s_m = v_m + noise, d = mu - loc_m (- the model-side mean).  With mub the pull-back of the likelihood through the softmax link:
//   locbar_g = mub - d / s_m^2 ;  vbar_g = locbar_g eps + 1 / v_g ;  locbar_m = d / s_m^2 ;  vbar_m = -1 / s_m + d^2 / s_m^3 (= d/d noise)
// (for identical inputs their sums are the single-point formulas of elbo_rows_kernel).  One thread per row, any K; not a hot path.
template <typename T>
__global__ __launch_bounds__(64) void elbo_rows2_kernel(
    int64_t nrows, int K, int V, const Hyper* __restrict__ h, int nqpart,
    const T* __restrict__ qpart_m, const T* __restrict__ loc_m, const T* __restrict__ tt_m,
    const T* __restrict__ qpart_g, const T* __restrict__ loc_g, const T* __restrict__ tt_g,
    const T* __restrict__ eps, int64_t ldk, int64_t lde, const int32_t* __restrict__ ws, const T* __restrict__ phi,
    const T* __restrict__ mean_m, int64_t mm_sk, int64_t mm_sn, const T* __restrict__ mean_g, int64_t mg_sk, int64_t mg_sn,
    T* __restrict__ qout, T* __restrict__ vbar_m, T* __restrict__ locbar_m, T* __restrict__ asum_m,
    T* __restrict__ vbar_g, T* __restrict__ locbar_g, T* __restrict__ asum_g, T* __restrict__ mu_out,
    double* __restrict__ dpart /*[grid][4]*/, T* __restrict__ phibar_part /*[grid][K*V]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int RB = blockDim.x;
  double* scratch = reinterpret_cast<double*>(smem);
  T* phiS = reinterpret_cast<T*>(smem + 128);           // [K*V]
  T* accS = phiS + K * V;                               // [K*V]
  T* thS = accS + K * V;                                // [RB][K+1]
  T* pbS = thS + RB * (K + 1);                          // [RB][V+1]
  T* tbS = pbS + RB * (V + 1);                          // [RB][K+1]
  for (int e = threadIdx.x; e < K * V; e += RB) { phiS[e] = phi[e]; accS[e] = 0; }
  __syncthreads();
  const T var = (T)h->var, eta = (T)h->noise, feps = t_eps<T>();
  double s_site = 0, s_llw = 0, s_noise = 0, s_vd = 0;
  const int64_t nblk = (nrows + RB - 1) / RB;
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t n = blk * RB + threadIdx.x;
    const bool ok = n < nrows;
    T* th = thS + threadIdx.x * (K + 1);
    T* pb = pbS + threadIdx.x * (V + 1);
    T* tb = tbS + threadIdx.x * (K + 1);
    if (ok) {
      T qm = 0, qg = 0;
      for (int c = 0; c < nqpart; ++c) { qm += qpart_m[(int64_t)c * ldk + n]; qg += qpart_g[(int64_t)c * ldk + n]; }
      qout[n] = qm;
      const T am = (var - qm > T(0)) ? T(1) : T(0), ag = (var - qg > T(0)) ? T(1) : T(0);
      const T v0m = am * (var - qm), v0g = ag * (var - qg);
      auto mu_of = [&](int k, T& vg, T& ek) {
        ek = eps[(int64_t)k * lde + n];
        vg = v0g + tt_g[(int64_t)k * ldk + n];
        T m = loc_g[(int64_t)k * ldk + n] + vg * ek;
        if (mean_g) m += mean_g[(int64_t)k * mg_sk + n * mg_sn];
        return m;
      };
      T mx = -3.0e38f;
      for (int k = 0; k < K; ++k) { T vg, ek; const T m = mu_of(k, vg, ek); th[k] = m; mx = fmax(mx, m); }
      T se = 0;
      for (int k = 0; k < K; ++k) { const T e = t_exp<T>(th[k] - mx); th[k] = e; se += e; }
      const T ise = T(1) / se;
      for (int k = 0; k < K; ++k) th[k] *= ise;
      T ps = 0;
      for (int vv = 0; vv < V; ++vv) { T p = 0; for (int k = 0; k < K; ++k) p += th[k] * phiS[k * V + vv]; pb[vv] = p; ps += p; }
      const T ips = T(1) / ps;
      T llw = 0;
      for (int vv = 0; vv < V; ++vv) {
        const T p = pb[vv], ph = p * ips, wv = (T)ws[n * V + vv];
        const bool inr = (ph > feps) && (ph < T(1) - feps);
        llw += wv * t_log<T>(fmin(fmax(ph, feps), T(1) - feps));
        pb[vv] = inr ? wv / p : T(0);
      }
      s_llw += (double)llw;
      // cancellation-free softmax pull-back, as in elbo_rows_kernel: reference value = thetabar of the dominant topic
      T cref = 0, tmax = -1;
      for (int k = 0; k < K; ++k) {
        T sum = 0;
        for (int vv = 0; vv < V; ++vv) sum += phiS[k * V + vv] * pb[vv];
        tb[k] = sum;
        if (th[k] > tmax) { tmax = th[k]; cref = sum; }
      }
      T dot = 0;                                     // = sum_j theta_j (cref - thetabar_j)
      for (int k = 0; k < K; ++k) dot += th[k] * (cref - tb[k]);
      T site = 0, ng = 0, vsm = 0, vsg = 0;
      for (int k = 0; k < K; ++k) {
        T vg, ek;
        const T mu = mu_of(k, vg, ek);
        const T vm = v0m + tt_m[(int64_t)k * ldk + n], sm = vm + eta;
        T lm = loc_m[(int64_t)k * ldk + n];
        if (mean_m) lm += mean_m[(int64_t)k * mm_sk + n * mm_sn];
        const T d = mu - lm, mub = th[k] * ((tb[k] - cref) + dot);
        site += -t_log<T>(sm) - T(0.5) * (d / sm) * (d / sm) + t_log<T>(vg) + T(0.5) * ek * ek;
        const T lbm = d / (sm * sm), vbm = -T(1) / sm + d * d / (sm * sm * sm);
        const T lbg = mub - lbm, vbg = lbg * ek + T(1) / vg;
        ng += vbm;
        vbar_m[(int64_t)k * ldk + n] = vbm; locbar_m[(int64_t)k * ldk + n] = lbm;
        vbar_g[(int64_t)k * ldk + n] = vbg; locbar_g[(int64_t)k * ldk + n] = lbg;
        if (mu_out) mu_out[(int64_t)k * ldk + n] = mu;
        vsm += vbm; vsg += vbg;
      }
      asum_m[n] = am * vsm; asum_g[n] = ag * vsg;
      s_site += (double)site; s_noise += (double)ng; s_vd += (double)(am * vsm + ag * vsg);
    } else {
      for (int k = 0; k < K; ++k) th[k] = 0;
      for (int vv = 0; vv < V; ++vv) pb[vv] = 0;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < K * V; e += RB) {
      const int k = e / V, vv = e - k * V;
      T sum = 0;
      for (int r = 0; r < RB; ++r) sum += thS[r * (K + 1) + k] * pbS[r * (V + 1) + vv];
      accS[e] += sum;
    }
    __syncthreads();
  }
  const double b0 = block_sum(s_site, scratch), b1 = block_sum(s_llw, scratch);
  const double b2 = block_sum(s_noise, scratch), b3 = block_sum(s_vd, scratch);
  if (threadIdx.x == 0) {
    dpart[4 * (int64_t)blockIdx.x + 0] = b0; dpart[4 * (int64_t)blockIdx.x + 1] = b1;
    dpart[4 * (int64_t)blockIdx.x + 2] = b2; dpart[4 * (int64_t)blockIdx.x + 3] = b3;
  }
  for (int e = threadIdx.x; e < K * V; e += RB) phibar_part[(int64_t)blockIdx.x * K * V + e] = accS[e];
}

// y[i] += x[i]
template <typename T>
__global__ void add_into_kernel(int64_t n, const T* __restrict__ x, T* __restrict__ y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] += x[i];
}

// vmax[k] = bits of max_n |x[k][n]| (floats are ordered like their bit patterns when non-negative); vmax zeroed by the caller
__global__ __launch_bounds__(256) void absmax_rows_kernel(const float* __restrict__ x, int64_t n, int64_t ld, unsigned* __restrict__ vmax) {
  const int k = blockIdx.y;
  float m = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
    m = fmaxf(m, fabsf(x[(int64_t)k * ld + i]));
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((threadIdx.x & 63) == 0 && m > 0) atomicMax(vmax + k, __float_as_uint(m));
}

// =====================================================================================
// ubar partials: ubar[k][i] = sum_n locbar[k][n] W[n][i].  A workgroup covers 256 columns x rows_per_blk rows: lane = 4 adjacent
// columns (one 16-byte W load per row, 1 KB per wave), wave = every fourth row, 16 topics at a time; locbar goes through LDS as
// [row][16 topics] so that a row's 16 factors are four broadcast 16-byte reads (the one-column form issued 16 LDS reads and one 4-byte
// load per 16 FMAs and ran at 1 TB/s).  The four waves' sums meet in LDS in a fixed order.
// =====================================================================================
// KQ: topic quads per pass (K = 10 runs as 3 quads = 12 topics, not 16: the kernel is VALU-bound)
template <typename T, int KQ = 4>
__global__ __launch_bounds__(256) void ubar_part_kernel(const T* __restrict__ W, int64_t nrows, int Mp, int K,
                                                        const T* __restrict__ locbar, int64_t ldk, int64_t rows_per_blk,
                                                        T* __restrict__ part /*[gridDim.x][K][Mp]*/) {
  typedef T T4 __attribute__((ext_vector_type(4)));
  __shared__ __attribute__((aligned(16))) T lb[256][16];
  const int cq = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int col = blockIdx.y * 256 + 4 * cq;               // Mp is a multiple of 32: a quad is inside or outside as a whole
  const bool cok = col < Mp;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_blk;
  int64_t r1 = r0 + rows_per_blk; if (r1 > nrows) r1 = nrows;
  for (int k0 = 0; k0 < K; k0 += 4 * KQ) {
    T4 acc[4 * KQ];
#pragma unroll
    for (int kk = 0; kk < 4 * KQ; ++kk) acc[kk] = T4{0, 0, 0, 0};
    for (int64_t rs = r0; rs < r1; rs += 256) {
      __syncthreads();
      {
        const int64_t n = rs + threadIdx.x;
        T v[4 * KQ];
#pragma unroll
        for (int kk = 0; kk < 4 * KQ; ++kk) v[kk] = (k0 + kk < K && n < r1) ? locbar[(int64_t)(k0 + kk) * ldk + n] : T(0);
#pragma unroll
        for (int q = 0; q < KQ; ++q) *reinterpret_cast<T4*>(&lb[threadIdx.x][4 * q]) = T4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
      }
      __syncthreads();
      const int cnt = (int)((r1 - rs < 256) ? (r1 - rs) : 256);
      if (cok) {
        const T* wp = W + rs * Mp + col;
#pragma unroll 4
        for (int r = rg; r < cnt; r += 4) {
          const T4 w = *reinterpret_cast<const T4*>(wp + (int64_t)r * Mp);
#pragma unroll
          for (int q = 0; q < KQ; ++q) {
            const T4 l = *reinterpret_cast<const T4*>(&lb[r][4 * q]);
            acc[4 * q] += l[0] * w; acc[4 * q + 1] += l[1] * w; acc[4 * q + 2] += l[2] * w; acc[4 * q + 3] += l[3] * w;
          }
        }
      }
    }
    // waves 1..3 hand their sums to wave 0 through LDS (lb is free: 4 topics x 3 waves x 64 lanes x 16 B = 12 KB per pass)
    T4* red = reinterpret_cast<T4*>(&lb[0][0]);
#pragma unroll
    for (int g4 = 0; g4 < KQ; ++g4) {
      __syncthreads();
      if (rg > 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) red[((rg - 1) * 4 + j) * 64 + cq] = acc[4 * g4 + j];
      }
      __syncthreads();
      if (rg == 0 && cok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int kk = 4 * g4 + j;
          T4 t = acc[kk];
#pragma unroll
          for (int w = 0; w < 3; ++w) t += red[(w * 4 + j) * 64 + cq];
          if (k0 + kk < K) *reinterpret_cast<T4*>(part + ((int64_t)blockIdx.x * K + k0 + kk) * Mp + col) = t;
        }
      }
    }
  }
}

// =====================================================================================
// reductions of per-workgroup partials (deterministic order)
// =====================================================================================
// out[e] = sum_p part[p][e]  (accumulated in double)
template <typename T>
__global__ void reduce_parts_kernel(const T* __restrict__ part, int64_t nparts, int64_t len, T* __restrict__ out) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= len) return;
  double s = 0;
#pragma unroll 16
  for (int64_t p = 0; p < nparts; ++p) s += (double)part[p * len + e];      // (independent loads in flight; the additions stay in order)
  out[e] = (T)s;
}
// out[c] = sum_p part[p][c], c < ncomp ; one block
__global__ void reduce_dparts_kernel(const double* __restrict__ part, int64_t nparts, int ncomp, double* __restrict__ out) {
  __shared__ double scratch[16];
  for (int c = 0; c < ncomp; ++c) {
    double s = 0;
#pragma unroll 8
    for (int64_t p = threadIdx.x; p < nparts; p += blockDim.x) s += part[p * ncomp + c];      // 1024 threads, independent loads in flight
    s = block_sum(s, scratch);
    if (threadIdx.x == 0) out[c] = s;
  }
}
// The doubles of red_d travelling in the element type of the flat payload, so that ONE all-reduce carries the whole step
// (SURVEY.md 8(e)).  double payload: copied.  float payload: d = p0 + p1 + p2 + p3 with 12-bit-mantissa pieces p0..p2 (the
// remainder p3 keeps 24 bits).  While the ranks' values of an entry have similar magnitude (within 2^9: the loss sums red_d[0..8),
// which every rank accumulates over its share of the rows) the float sums of the pieces over <= 8 ranks are exact and the reduced
// value is the f64 sum to ~2^-48.  When they differ by more - the inducing-input sums red_d[8..) of spatially sharded rows can, and
// can cancel - the sum of the leading pieces rounds at 2^-24 of the LARGEST summand and depends on the reduction order: float32
// resolution of the largest contribution, which is what the float32 gradient it ends in resolves anyway
// (tests/test_gpu_round3.py::test_packed_payload_with_rank_dependent_magnitudes).
template <typename T>
__global__ void payload_pack_kernel(const double* __restrict__ d, int nd, T* __restrict__ tail) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nd) return;
  if constexpr (sizeof(T) == 8) { tail[i] = d[i]; }
  else {
    double r = d[i];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      const float f = __uint_as_float(__float_as_uint((float)r) & 0xFFFFF000u);
      tail[p * nd + i] = f;
      r -= (double)f;
    }
    tail[3 * nd + i] = (float)r;
  }
}
template <typename T>
__global__ void payload_unpack_kernel(const T* __restrict__ tail, int nd, double* __restrict__ d) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nd) return;
  if constexpr (sizeof(T) == 8) { d[i] = tail[i]; }
  else d[i] = (((double)tail[3 * nd + i] + (double)tail[2 * nd + i]) + (double)tail[nd + i]) + (double)tail[i];
}
// TN slabs: out[b][i][j] = sum_sp slab[sp][b][i][j]; symmetric batches only hold tiles ti >= tj and are mirrored
template <typename T>
__global__ void reduce_slabs_kernel(const T* __restrict__ slab, int nsplit, int nbatch, int Mp, int sym, T* __restrict__ out, int QD = GDRF_TILE / 2) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, b = blockIdx.z;
  if (j >= Mp) return;
  // sym: QD x QD blocks above the diagonal are mirrored from below (the split TN kernels do not compute them: 64 x 64 quadrants
  // of diagonal tiles in gemm_tn_split_kernel, 32 x 32 sub-tiles in tn_topics_f16_kernel; the f32 kernel computes them, to the
  // same values up to the summation order)
  if (sym && (j / QD) > (i / QD)) return;
  const int64_t mm = (int64_t)Mp * Mp;
  double s = 0;
  const T* src = slab + (int64_t)b * mm + (int64_t)i * Mp + j;
  const int64_t step = (int64_t)nbatch * mm;
#pragma unroll 8
  for (int sp = 0; sp < nsplit; ++sp) s += (double)src[(int64_t)sp * step];      // (8 independent loads in flight; the additions stay in order)
  out[(int64_t)b * mm + (int64_t)i * Mp + j] = (T)s;
  if (sym && (j / QD) < (i / QD)) out[(int64_t)b * mm + (int64_t)j * Mp + i] = (T)s;
}

// =====================================================================================
// small gradients, Dirichlet term and the loss (one workgroup)
// red_d: [0] site [1] llw [2] noise_g [3] var_direct [4] knm_k [5] knm_dls ; kuu: [0] kuu_k [1] kuu_dls
// =====================================================================================
template <typename T>
__global__ void grad_small_kernel(int M, int Mp, int K, int V, const Hyper* __restrict__ h, const double* __restrict__ red_d,
                                  const double* __restrict__ kuu_d, const T* __restrict__ ubar, const T* __restrict__ phibar_lik,
                                  const T* __restrict__ phi, const double* __restrict__ alpha, double lgam_const,
                                  double ll_const, double n_global, T* __restrict__ g_hyper3, T* __restrict__ g_uloc,
                                  T* __restrict__ g_phi, const int* __restrict__ flag, double* __restrict__ out_d) {
  __shared__ double scratch[16];
  const double sc = -1.0 / n_global;
  for (int e = threadIdx.x; e < K * M; e += blockDim.x) {
    const int k = e / M, i = e - k * M;
    g_uloc[e] = (T)(sc * (double)ubar[(int64_t)k * Mp + i]);
  }
  double lp = 0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double dot = 0;
    for (int v = 0; v < V; ++v) {
      const double ph = (double)phi[k * V + v];
      const double pb = (double)phibar_lik[k * V + v] + (alpha[k * V + v] - 1.0) / ph;
      dot += ph * pb;
      lp += (alpha[k * V + v] - 1.0) * log(ph);
    }
    for (int v = 0; v < V; ++v) {
      const double ph = (double)phi[k * V + v];
      const double pb = (double)phibar_lik[k * V + v] + (alpha[k * V + v] - 1.0) / ph;
      g_phi[k * V + v] = (T)(sc * ph * (pb - dot));
    }
  }
  lp = block_sum(lp, scratch);
  if (threadIdx.x == 0) {
    const double lp_phi = lp + lgam_const;
    if (ll_const != ll_const) ll_const = red_d[7];          // NaN: the (all-reduced) data constant travels in the payload
    const double elbo_n = red_d[0] + red_d[1] + ll_const + lp_phi;
    out_d[0] = -elbo_n / n_global;            // loss
    out_d[1] = (double)(*flag);
    out_d[2] = red_d[0]; out_d[3] = red_d[1] + ll_const; out_d[4] = lp_phi;
    g_hyper3[0] = (T)(sc * (red_d[5] + kuu_d[1]));                              // d/d log lengthscale
    g_hyper3[1] = (T)(sc * (red_d[4] + kuu_d[0] + h->var * red_d[3]));          // d/d log variance
    g_hyper3[2] = (T)(sc * (h->noise * red_d[2]));                              // d/d log noise
    g_hyper3[3] = (T)(sc * (red_d[6] + kuu_d[2]));                              // d/d log scale_mixture (RationalQuadratic; else 0)
  }
}

// =====================================================================================
// optimizers on the flat unconstrained parameter vector; predicated on the Cholesky flag
// mode 0 Adam, 1 AdamW (decoupled decay), 2 ClippedAdam
// =====================================================================================
template <typename T>
__global__ void adam_kernel(int64_t n, T* __restrict__ p, const T* __restrict__ g, T* __restrict__ m, T* __restrict__ v,
                            int mode, double lr, double b1, double b2, double eps, double wd, double clip,
                            double bc1, double bc2, const int* __restrict__ flag) {
  if (flag && *flag) return;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double gi = (double)g[i], pi = (double)p[i];
  if (mode == 2) gi = fmin(fmax(gi, -clip), clip);
  if (mode == 1) pi *= (1.0 - lr * wd);
  const double mi = b1 * (double)m[i] + (1.0 - b1) * gi;
  const double vi = b2 * (double)v[i] + (1.0 - b2) * gi * gi;
  m[i] = (T)mi; v[i] = (T)vi;
  if (mode == 2) pi -= lr * sqrt(bc2) / bc1 * mi / (sqrt(vi) + eps);
  else pi -= (lr / bc1) * mi / (sqrt(vi) / sqrt(bc2) + eps);
  p[i] = (T)pi;
}

// =====================================================================================
// predictive path (mean only): loc_kn = k(x_n, Z) c_k,  c_k = Linv^T u_k
// =====================================================================================
template <typename T, typename TN>
__global__ void predict_coeff_kernel(const T* __restrict__ Linv, const TN* __restrict__ U, int M, int Mp, int K, T* __restrict__ Cf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (i >= M) return;
  double s = 0;
  for (int j = i; j < M; ++j) s += (double)Linv[(int64_t)j * Mp + i] * (double)U[(int64_t)k * M + j];
  Cf[(int64_t)k * M + i] = (T)s;
}

// mode 0: write loc (K,N) ; 1: topic_probs (N,K) ; 2: word_probs (N,V) ; 3: perplexity partial sums
template <typename T, typename TN>
__global__ __launch_bounds__(128) void predict_rows_kernel(const TN* __restrict__ X, int64_t nrows, const T* __restrict__ Z, int M, int D,
                                                           int kind, const Hyper* __restrict__ h, const T* __restrict__ Cf, int K, int V,
                                                           const TN* __restrict__ phi, const int32_t* __restrict__ ws, int mode,
                                                           TN* __restrict__ out, int64_t ldo, double* __restrict__ dpart,
                                                           int cf_in_lds) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* scratch = reinterpret_cast<double*>(smem);       // [16]
  T* Zs = reinterpret_cast<T*>(smem + 128);                // [M*D]
  T* phiS = Zs + M * D;                                    // [K*V]
  T* CsL = phiS + K * V;                                   // [K*M] when it fits
  const T* Cs = cf_in_lds ? CsL : Cf;
  for (int e = threadIdx.x; e < M * D; e += blockDim.x) Zs[e] = Z[e];
  if (cf_in_lds) for (int e = threadIdx.x; e < K * M; e += blockDim.x) CsL[e] = Cf[e];
  if (mode >= 2) for (int e = threadIdx.x; e < K * V; e += blockDim.x) phiS[e] = (T)phi[e];
  __syncthreads();
  const T var = (T)h->var, ils2 = (T)h->inv_ls2;
  double s_wlp = 0, s_w = 0;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < nrows; n += (int64_t)gridDim.x * blockDim.x) {
    T x[GDRF_DMAX];
#pragma unroll
    for (int d = 0; d < GDRF_DMAX; ++d) x[d] = (d < D) ? (T)X[n * D + d] : T(0);
    T lc[GDRF_KMAX];
#pragma unroll
    for (int k = 0; k < GDRF_KMAX; ++k) lc[k] = 0;
    bool done = false;
    if constexpr (sizeof(T) == 8) {
      if (kind == 0) {
        // RBF in double: k = 2^(log2 var - sum_d (a x_d - a z_d)^2) through the straight-line polynomial of knm_rbf_f64_kernel instead of
        // the library exp() per (row, inducing point) - this loop is N x M of them
        const double a = sqrt(0.5 * 1.4426950408889634074 * (double)ils2), lv = log2((double)var);
        double xa[GDRF_DMAX];
#pragma unroll
        for (int d = 0; d < GDRF_DMAX; ++d) xa[d] = a * (double)x[d];
        for (int i = 0; i < M; ++i) {
          double t = lv;
#pragma unroll
          for (int d = 0; d < GDRF_DMAX; ++d) if (d < D) { const double dd = xa[d] - a * (double)Zs[i * D + d]; t = fma(-dd, dd, t); }
          const double kv = exp2_poly(t);
#pragma unroll
          for (int k = 0; k < GDRF_KMAX; ++k) if (k < K) lc[k] += kv * Cs[k * M + i];
        }
        done = true;
      }
    }
    if (!done)
    for (int i = 0; i < M; ++i) {
      T r2 = 0;
#pragma unroll
      for (int d = 0; d < GDRF_DMAX; ++d) if (d < D) { const T t = x[d] - Zs[i * D + d]; r2 += t * t; }
      const T kv = cov_from_r2<T>(kind, r2 * ils2, var, (T)h->alpha);
#pragma unroll
      for (int k = 0; k < GDRF_KMAX; ++k) if (k < K) lc[k] += kv * Cs[k * M + i];
    }
    if (mode == 0) {
#pragma unroll
      for (int k = 0; k < GDRF_KMAX; ++k) if (k < K) out[(int64_t)k * ldo + n] = (TN)lc[k];
      continue;
    }
    T mx = -3.0e38f;
#pragma unroll
    for (int k = 0; k < GDRF_KMAX; ++k) if (k < K) mx = fmax(mx, lc[k]);
    T se = 0;
#pragma unroll
    for (int k = 0; k < GDRF_KMAX; ++k) if (k < K) { lc[k] = t_exp<T>(lc[k] - mx); se += lc[k]; }
    const T ise = T(1) / se;
#pragma unroll
    for (int k = 0; k < GDRF_KMAX; ++k) if (k < K) lc[k] *= ise;
    if (mode == 1) {
#pragma unroll
      for (int k = 0; k < GDRF_KMAX; ++k) if (k < K) out[n * ldo + k] = (TN)lc[k];
      continue;
    }
    for (int v = 0; v < V; ++v) {
      T p = 0;
#pragma unroll
      for (int k = 0; k < GDRF_KMAX; ++k) if (k < K) p += lc[k] * phiS[k * V + v];
      if (mode == 2) out[n * ldo + v] = (TN)p;
      else { const double w = (double)ws[n * V + v]; s_wlp += w * (double)t_log<T>(p); s_w += w; }
    }
  }
  if (mode == 3) {
    const double a = block_sum(s_wlp, scratch), b = block_sum(s_w, scratch);
    if (threadIdx.x == 0) { dpart[2 * (int64_t)blockIdx.x] = a; dpart[2 * (int64_t)blockIdx.x + 1] = b; }
  }
}

// the same for K > GDRF_KMAX: topics in chunks of GDRF_KMAX (the covariance row is re-evaluated per chunk), softmax over all K
// in two sweeps (running maximum and sum, then the normalised values); word_probs accumulate in an LDS row per thread
template <typename T, typename TN>
__global__ __launch_bounds__(128) void predict_rows_bigk_kernel(const TN* __restrict__ X, int64_t nrows, const T* __restrict__ Z, int M, int D,
                                                                int kind, const Hyper* __restrict__ h, const T* __restrict__ Cf, int K, int V,
                                                                const TN* __restrict__ phi, const int32_t* __restrict__ ws, int mode,
                                                                TN* __restrict__ out, int64_t ldo, double* __restrict__ dpart) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* scratch = reinterpret_cast<double*>(smem);       // [16]
  T* Zs = reinterpret_cast<T*>(smem + 128);                // [M*D]
  T* phiS = Zs + M * D;                                    // [K*V]
  T* pbS = phiS + K * V;                                   // [blockDim][V+1]
  for (int e = threadIdx.x; e < M * D; e += blockDim.x) Zs[e] = Z[e];
  if (mode >= 2) for (int e = threadIdx.x; e < K * V; e += blockDim.x) phiS[e] = (T)phi[e];
  __syncthreads();
  const T var = (T)h->var, ils2 = (T)h->inv_ls2;
  T* pb = pbS + threadIdx.x * (V + 1);
  double s_wlp = 0, s_w = 0;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < nrows; n += (int64_t)gridDim.x * blockDim.x) {
    T x[GDRF_DMAX];
#pragma unroll
    for (int d = 0; d < GDRF_DMAX; ++d) x[d] = (d < D) ? (T)X[n * D + d] : T(0);
    T lc[GDRF_KMAX];
    auto chunk = [&](int k0) {
#pragma unroll
      for (int k = 0; k < GDRF_KMAX; ++k) lc[k] = 0;
      for (int i = 0; i < M; ++i) {
        T r2 = 0;
#pragma unroll
        for (int d = 0; d < GDRF_DMAX; ++d) if (d < D) { const T t = x[d] - Zs[i * D + d]; r2 += t * t; }
        const T kv = cov_from_r2<T>(kind, r2 * ils2, var, (T)h->alpha);
#pragma unroll
        for (int k = 0; k < GDRF_KMAX; ++k) if (k0 + k < K) lc[k] += kv * Cf[(int64_t)(k0 + k) * M + i];
      }
    };
    T mx = -3.0e38f, se = 0;
    for (int k0 = 0; k0 < K; k0 += GDRF_KMAX) {
      chunk(k0);
      if (mode == 0) {
#pragma unroll
        for (int k = 0; k < GDRF_KMAX; ++k) if (k0 + k < K) out[(int64_t)(k0 + k) * ldo + n] = (TN)lc[k];
        continue;
      }
      T cm = mx;
#pragma unroll
      for (int k = 0; k < GDRF_KMAX; ++k) if (k0 + k < K) cm = fmax(cm, lc[k]);
      se *= t_exp<T>(mx - cm);
#pragma unroll
      for (int k = 0; k < GDRF_KMAX; ++k) if (k0 + k < K) se += t_exp<T>(lc[k] - cm);
      mx = cm;
    }
    if (mode == 0) continue;
    const T ise = T(1) / se;
    if (mode >= 2) for (int v = 0; v < V; ++v) pb[v] = 0;
    for (int k0 = 0; k0 < K; k0 += GDRF_KMAX) {
      chunk(k0);
#pragma unroll
      for (int k = 0; k < GDRF_KMAX; ++k) if (k0 + k < K) {
        const T th = t_exp<T>(lc[k] - mx) * ise;
        if (mode == 1) out[n * ldo + k0 + k] = (TN)th;
        else for (int v = 0; v < V; ++v) pb[v] += th * phiS[(k0 + k) * V + v];
      }
    }
    if (mode == 2) for (int v = 0; v < V; ++v) out[n * ldo + v] = (TN)pb[v];
    if (mode == 3) for (int v = 0; v < V; ++v) { const double w = (double)ws[n * V + v]; s_wlp += w * (double)t_log<T>(pb[v]); s_w += w; }
  }
  if (mode == 3) {
    const double a = block_sum(s_wlp, scratch), b = block_sum(s_w, scratch);
    if (threadIdx.x == 0) { dpart[2 * (int64_t)blockIdx.x] = a; dpart[2 * (int64_t)blockIdx.x + 1] = b; }
  }
}

// =====================================================================================
// data-only Multinomial constant: sum_n [lgamma(sum_v w + 1) - sum_v lgamma(w + 1)]  
// =====================================================================================
__global__ void ll_const_kernel(const int32_t* __restrict__ ws, int64_t nrows, int V, double* __restrict__ dpart) {
  __shared__ double scratch[16];
  double s = 0;
  for (int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; n < nrows; n += (int64_t)gridDim.x * blockDim.x) {
    double tot = 0, sl = 0;
    for (int v = 0; v < V; ++v) { const double w = (double)ws[n * V + v]; tot += w; sl += lgamma(w + 1.0); }
    s += lgamma(tot + 1.0) - sl;
  }
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) dpart[blockIdx.x] = s;
}

// =====================================================================================
// counter-based standard normals: Philox4x32-10 keyed by (seed), counter (global row, topic, step)
// -> identical draws for any sharding of the rows over ranks
// =====================================================================================
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
  const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
  const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n1 = (uint32_t)p1;
  const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1, n3 = (uint32_t)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
template <typename T>
__global__ void fill_eps_kernel(uint64_t seed, uint32_t step, int64_t n_offset, int64_t nrows, int K, T* __restrict__ eps, int64_t ldk) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int k = blockIdx.y;
  if (n >= nrows) return;
  const uint64_t gn = (uint64_t)(n + n_offset);
  uint32_t c[4] = {(uint32_t)gn, (uint32_t)(gn >> 32), (uint32_t)k, step};
  uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  const double u1 = ((double)c[0] + 0.5) * (1.0 / 4294967296.0);
  const double u2 = ((double)c[1] + 0.5) * (1.0 / 4294967296.0);
  eps[(int64_t)k * ldk + n] = (T)(sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2));
}

}  // namespace gdrf

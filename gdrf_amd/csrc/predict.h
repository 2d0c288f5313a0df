// Predictive path on the matrix cores: loc = K_nm Cf, Cf = L^-T u_k (gdrf/models/sparse_gdrf.py:161-186: log_topic_probs evaluates
// gp.util.conditional and keeps f_loc only), then the softmax link, theta Phi and the perplexity sums of gdrf/models/abstract_gdrf.py:113-139.
//
// The one-thread-per-row kernel (kernels_n.h: predict_rows_kernel) walked the M inducing points with K multiply-adds and K LDS reads per
// covariance value on the vector pipe (7.3 ms at N = 1e6, M = 512, K = 10).  Here a wave owns 16 rows: every lane evaluates ONE
// covariance value per step - k(x_row, z_i), row = lane & 15, i = 4 step + (lane >> 4), which is exactly the A operand of the 16x16x4
// matrix instruction - and the (4 inducing points) x (16 topics) slab of Cf is its B operand, read as one contiguous 64-element line of
// the padded, transposed coefficient array CfT[i][16 NB].  K_nm is never stored; the arithmetic stays in the solve precision (Cf has
// the magnitude of L^-1 and cancels, DESIGN.md "precision"), i.e. v_mfma_f64_16x16x4_f64 in the default build.  The vector work left
// is the covariance itself (RBF in double: the straight-line 2^t polynomial of knm_rbf_f64_kernel).
#pragma once
#include "common.h"
#include "kernels_mm.h"
#include "kernels_n.h"

namespace gdrf {

// CfT[i][c] = (L^-T u_c)[i] = sum_{j >= i} LinvT[i][j] u_c[j] for c < K, 0 for the padding (i >= M or c >= K); ldc = 16 NB <= 32.
// One wave per inducing point: the lanes stride along the contiguous row i of LinvT, K running sums per lane, one wave reduction each.
template <typename T, typename TN>
__global__ __launch_bounds__(64) void predict_coeff_t_kernel(const T* __restrict__ LinvT, const TN* __restrict__ U, int M, int Mp, int M4, int K, int ldc,
                                                             T* __restrict__ CfT) {
  const int i = blockIdx.x, lane = threadIdx.x;
  double acc[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) acc[k] = 0;
  if (i < M)
    for (int j = i + lane; j < M; j += 64) {
      const double l = (double)LinvT[(int64_t)i * Mp + j];
#pragma unroll
      for (int k = 0; k < 32; ++k) if (k < K) acc[k] += l * (double)U[(int64_t)k * M + j];
    }
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    if (k < ldc) {
      const double s = (k < K) ? wave_sum(acc[k]) : 0.0;
      if (lane == 0) CfT[(int64_t)i * ldc + k] = (T)s;
    }
  }
}

// mode 0: loc (K, n) ; 1: topic_probs (n, K) ; 2: word_probs (n, V) ; 3: perplexity partial sums {sum w log p, sum w} per workgroup
// NB: 16-topic column blocks (K <= 16 NB).  LDS: scratch[16] doubles | Zs[M4][DD] | phiS[K V] | per-wave tile [16][16 NB + 1]
template <typename T, typename TN, int DD, int NB>
__global__ __launch_bounds__(256) void predict_mfma_kernel(const TN* __restrict__ X, int64_t nrows, const T* __restrict__ Z, int M, int M4, int D,
                                                           int kind, const Hyper* __restrict__ h, const T* __restrict__ CfT, int K, int V,
                                                           const TN* __restrict__ phi, const int32_t* __restrict__ ws, int mode,
                                                           TN* __restrict__ out, int64_t ldo, double* __restrict__ dpart) {
  using MF = Mfma<T>;
  using acc_t = typename MF::acc_t;
  constexpr int LDC = 16 * NB, TLD = LDC + 1;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  double* scratch = reinterpret_cast<double*>(smem);       // [16]
  T* Zs = reinterpret_cast<T*>(smem + 128);                // [M4][DD], pre-scaled for the RBF form
  T* phiS = Zs + (size_t)M4 * DD;                          // [K*V]
  T* tiles = phiS + K * V;                                 // [waves][16][TLD]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
  const int lr = lane & 15, lg = lane >> 4;
  T* tile = tiles + (size_t)wave * 16 * TLD;
  const bool rbf64 = (sizeof(T) == 8) && kind == 0;
  const T var = (T)h->var, ils2 = (T)h->inv_ls2, al = (T)h->alpha;
  const double a = sqrt(0.5 * 1.4426950408889634074 * h->inv_ls2), lv = log2(h->var);
  for (int e = threadIdx.x; e < M4 * DD; e += blockDim.x) {
    const int i = e / DD, d = e - i * DD;
    T z = (i < M && d < D) ? Z[(int64_t)i * D + d] : T(0);
    if (rbf64) z = (T)(a * (double)z);
    Zs[e] = z;
  }
  if (mode >= 2) for (int e = threadIdx.x; e < K * V; e += blockDim.x) phiS[e] = (T)phi[e];
  __syncthreads();
  const int64_t ngroups = (nrows + 15) / 16;
  const int nsteps = M4 / 4;
  double s_wlp = 0, s_w = 0;
  for (int64_t g = (int64_t)blockIdx.x * nwaves + wave; g < ngroups; g += (int64_t)gridDim.x * nwaves) {
    const int64_t n0 = g * 16, n = n0 + lr;
    T x[DD];
#pragma unroll
    for (int d = 0; d < DD; ++d) {
      x[d] = (n < nrows && d < D) ? (T)X[n * D + d] : T(0);
      if (rbf64) x[d] = (T)(a * (double)x[d]);
    }
    acc_t acc[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) acc[b] = acc_t{0, 0, 0, 0};
    const T* cf = CfT + (size_t)lg * LDC + lr;
    const T* zs = Zs + lg * DD;
    if (rbf64) {
      for (int s0 = 0; s0 < nsteps; s0 += 4) {
        T kv[4], bv[4][NB];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int s = (s0 + u < nsteps) ? s0 + u : nsteps - 1;        // nsteps need not be a multiple of 4: the tail repeats its last step with a zero B
          double t = lv;
#pragma unroll
          for (int d = 0; d < DD; ++d) { const double dd = (double)x[d] - (double)zs[s * 4 * DD + d]; t = fma(-dd, dd, t); }
          kv[u] = (T)exp2_poly(t);
#pragma unroll
          for (int b = 0; b < NB; ++b) bv[u][b] = (s0 + u < nsteps) ? cf[(size_t)s * 4 * LDC + b * 16] : T(0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
          for (int b = 0; b < NB; ++b) acc[b] = MF::mma(kv[u], bv[u][b], acc[b]);
      }
    } else {
      for (int s = 0; s < nsteps; ++s) {
        T r2 = 0;
#pragma unroll
        for (int d = 0; d < DD; ++d) { const T t = x[d] - zs[s * 4 * DD + d]; r2 += t * t; }
        const T kv = cov_from_r2<T>(kind, r2 * ils2, var, al);
#pragma unroll
        for (int b = 0; b < NB; ++b) acc[b] = MF::mma(kv, cf[(size_t)s * 4 * LDC + b * 16], acc[b]);
      }
    }
    // accumulators -> this wave's tile [row][topic]  (LDS is in order per wave: no barrier, waits only)
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
      for (int r = 0; r < 4; ++r) tile[MF::crow(lane, r) * TLD + b * 16 + lr] = acc[b][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (mode == 0) {
      for (int k = lg; k < K; k += 4)
        if (n < nrows) out[(int64_t)k * ldo + n] = (TN)tile[lr * TLD + k];
    } else {
      // softmax over the topics of row lr: lane group lg takes the topics k = lg (mod 4); the four groups meet by shuffles
      T mx = -3.0e38f;
      for (int k = lg; k < K; k += 4) mx = fmax(mx, tile[lr * TLD + k]);
      mx = fmax(mx, __shfl_xor(mx, 16, 64));
      mx = fmax(mx, __shfl_xor(mx, 32, 64));
      T se = 0;
      for (int k = lg; k < K; k += 4) { const T e = t_exp<T>(tile[lr * TLD + k] - mx); tile[lr * TLD + k] = e; se += e; }
      se += __shfl_xor(se, 16, 64);
      se += __shfl_xor(se, 32, 64);
      const T ise = T(1) / se;
      for (int k = lg; k < K; k += 4) tile[lr * TLD + k] *= ise;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int rows = (int)((nrows - n0 < 16) ? nrows - n0 : 16);
      if (mode == 1) {               // (n, K) row-major with ldo == K: the 16 rows are one contiguous run
        for (int e = lane; e < rows * K; e += 64) { const int r = e / K, k = e - r * K; out[(n0 + r) * ldo + k] = (TN)tile[r * TLD + k]; }
      } else {
        for (int e = lane; e < rows * V; e += 64) {
          const int r = e / V, v = e - r * V;
          T p = 0;
          for (int k = 0; k < K; ++k) p += tile[r * TLD + k] * phiS[k * V + v];
          if (mode == 2) out[(n0 + r) * ldo + v] = (TN)p;
          else { const double w = (double)ws[(n0 + r) * V + v]; s_wlp += w * (double)t_log<T>(p); s_w += w; }
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the tile is rewritten by the next group
  }
  if (mode == 3) {
    const double sa = block_sum(s_wlp, scratch), sb = block_sum(s_w, scratch);
    if (threadIdx.x == 0) { dpart[2 * (int64_t)blockIdx.x] = sa; dpart[2 * (int64_t)blockIdx.x + 1] = sb; }
  }
}

// f_var of gp.util.conditional(full_cov=False) from the step's forward by-products: var_kn = clamp(variance - q_n, 0) + tt_kn,
// q_n = sum of the partial row norms of W, tt = |S_k^T w_n|^2; out = [f_loc (K, n) | f_var (K, n)]
template <typename T>
__global__ void predict_var_kernel(int64_t nrows, int K, const Hyper* __restrict__ h, const T* __restrict__ qpart, int nqpart, const T* __restrict__ loc,
                                   const T* __restrict__ tt, int64_t ldk, T* __restrict__ out) {
  const int64_t n = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= nrows) return;
  T qn = 0;
  for (int c = 0; c < nqpart; ++c) qn += qpart[(int64_t)c * ldk + n];
  const T var = (T)h->var;
  const T v0 = (var - qn > T(0)) ? var - qn : T(0);
  for (int k = 0; k < K; ++k) {
    out[(int64_t)k * nrows + n] = loc[(int64_t)k * ldk + n];
    out[((int64_t)K + k) * nrows + n] = v0 + tt[(int64_t)k * ldk + n];
  }
}

}  // namespace gdrf

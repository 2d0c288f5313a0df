// The K_nm parts of the kernel hyper-parameter gradients WITHOUT the backward solve GEMM Kbar = Wbar L^-1 (SURVEY.md App. C:
// "d/d log var = sum Kbar o K + ...;  d/d log ls = sum Kbar o dK/dlog ls + ...").
//
// Kbar_nm is needed for two sums only (fixed inducing inputs, kernels whose shape parameter is the lengthscale).  With W = K L^-T,
//   sum_nj Kbar_nj K_nj              = sum_ni Wbar_ni W_ni                                   (one streaming dot product), and
//   sum_nj Kbar_nj dK_nj             = sum_{i >= j} Linv[i][j] Hd[j][i],   Hd = dK^T Wbar   (an M x M contraction over the rows).
// Hd has exactly the shape of G^T = W^T Wbar and runs on the same split-fp16 TN kernel (gemm_tn_split_kernel) with the pieces of
// dK = dK_nm / d log(lengthscale) as its stored operand; the ill-conditioned contraction with L^-1 is then M x M and done in double.
// This removes the f64 GEMM gemm_nt<BwdKnmProb> (8.5 ms of the 47.6 ms step at the headline size: 18 % of the step for 4.5 % of its
// flops, on the f64 matrix pipe) for a 2.6 ms fp16 contraction that shares the side stream with G^T.  Price: Hd carries float32-level
// rounding BEFORE the cancelling contraction with L^-1 (the f64 GEMM only saw the float32 rounding of Wbar, which that contraction
// does not amplify): measured on the host at the headline conditioning, 1e-5 ... 7e-4 relative on d loss / d log lengthscale,
// none on d / d log variance (the W route is exact); DESIGN.md section 3, tests/test_gpu_round3.py.  gdrf_set_hyper_backward selects.
#pragma once
#include "common.h"
#include "kernels_mm.h"
#include "kernels_n.h"
#include "gemm_split.h"

namespace gdrf {

// dK[n][j] = d k(x_n, z_j) / d log(lengthscale) as two block-scaled fp16 pieces, out[p][n][ldo] (the layout of the pieces of W);
// columns M..ldo-1 are zeros.  Evaluated in double (RBF: k = 2^t through exp2_poly, dK = k r^2), 8 columns per thread.
template <typename TX, int DD>
__global__ __launch_bounds__(256) void dk_pieces_kernel(const TX* __restrict__ X, int64_t N, const double* __restrict__ Z, int M, int D, int kind,
                                                        const Hyper* __restrict__ h, _Float16* __restrict__ out, int64_t piece_stride, int ldo,
                                                        const float* __restrict__ scale) {
  const int vpr = ldo / 8, rpp = 256 / vpr;                  // ldo a multiple of 32, ldo / 8 <= 256 (checked by the host)
  const int rsub = (int)threadIdx.x / vpr, cv = (int)threadIdx.x % vpr, i0 = cv * 8;
  if (rsub >= rpp) return;
  const double ils2 = h->inv_ls2, var = h->var, al = h->alpha;
  const double a = sqrt(0.5 * 1.4426950408889634074 * ils2), lv = log2(var);
  const double s = (double)scale[0];
  double z[8][DD];
#pragma unroll
  for (int e = 0; e < 8; ++e)
#pragma unroll
    for (int d = 0; d < DD; ++d) z[e][d] = (i0 + e < M && d < D) ? Z[(int64_t)(i0 + e) * D + d] : 0.0;
  const bool rbf = kind == 0;
  for (int64_t row = (int64_t)blockIdx.x * rpp + rsub; row < N; row += (int64_t)gridDim.x * rpp) {
    double x[DD];
#pragma unroll
    for (int d = 0; d < DD; ++d) x[d] = d < D ? (double)X[row * D + d] : 0.0;
    f16x8 hi, lo;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      double u = 0;
#pragma unroll
      for (int d = 0; d < DD; ++d) { const double dd = x[d] - z[e][d]; u = fma(dd, dd, u); }
      double dk;
      if (rbf) {
        const double ua = u * (a * a);                              // = r2 / (2 ln 2)
        dk = exp2_poly(lv - ua) * (ua * 1.3862943611198906);        // k r2,  r2 = 2 ln2 ua
      } else {
        const double r2 = u * ils2, k = cov_from_r2<double>(kind, r2, var, al);
        dk = dcov_dlogls_from_k<double>(kind, k, r2, al);
      }
      const float y = (i0 + e < M) ? (float)(dk * s) : 0.0f;
      const _Float16 hh = (_Float16)y;
      hi[e] = hh; lo[e] = (_Float16)(y - (float)hh);
    }
    *reinterpret_cast<f16x8*>(out + row * ldo + i0) = hi;
    *reinterpret_cast<f16x8*>(out + piece_stride + row * ldo + i0) = lo;
  }
}

// part[block] = sum over this block's share of Wbar o W (double accumulation of float4 partial products; deterministic)
template <typename T>
__global__ __launch_bounds__(256) void wbar_w_dot_kernel(const T* __restrict__ Wbar, const T* __restrict__ W, int64_t nrows, int Mp, double* __restrict__ part) {
  using V = typename Vec16<T>::type;
  constexpr int VE = Vec16<T>::N;
  __shared__ double scratch[16];
  const int64_t nvec = nrows * Mp / VE;
  double s1 = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * blockDim.x) {
    const V b = reinterpret_cast<const V*>(Wbar)[i], w = reinterpret_cast<const V*>(W)[i];
    double a1 = 0;
#pragma unroll
    for (int e = 0; e < VE; ++e) a1 += (double)b[e] * (double)w[e];
    s1 += a1;
  }
  const double t1 = block_sum(s1, scratch);
  if (threadIdx.x == 0) part[blockIdx.x] = t1;
}

// part[a] = sum_{b >= a} LinvT[a][b] * (sum_sp slab[sp][a][b]):  row a of the contraction of Hd = dK^T Wbar (split-row slabs of the TN
// kernel, summed here in double in slab order) with L^-1.  One workgroup per row a < M.
template <typename TS>
__global__ __launch_bounds__(256) void slab_linvt_dot_kernel(const float* __restrict__ slab, int nsplit, int Mp, int M, const TS* __restrict__ LinvT,
                                                            double* __restrict__ part) {
  __shared__ double scratch[16];
  const int a = blockIdx.x;
  const int64_t mm = (int64_t)Mp * Mp;
  double s = 0;
  for (int b = a + (int)threadIdx.x; b < M; b += blockDim.x) {
    double hsum = 0;
    const float* src = slab + (int64_t)a * Mp + b;
#pragma unroll 8
    for (int sp = 0; sp < nsplit; ++sp) hsum += (double)src[(int64_t)sp * mm];
    s += (double)LinvT[(int64_t)a * Mp + b] * hsum;
  }
  s = block_sum(s, scratch);
  if (threadIdx.x == 0) part[a] = s;
}

}  // namespace gdrf

// Replicated M x M algebra of one SVI step: K_uu, jitter-Cholesky with failure flag, explicit
// triangular inverse, parameter transforms and the Cholesky backward.  All matrices are stored
// with leading dimension Mp = round_up(M, 32) and zero padding.
//
// Reference behaviour restated (paths under /root/reference):
//   gdrf/models/utils.py:27-40            jittercholesky (failure -> host retries with more jitter)
//   gdrf/models/sparse_gdrf.py:327-328    Kuu / Luu rebuilt on every model/guide call
//   pyro LowerCholeskyTransform / softmax  
#pragma once
#include "common.h"
#include "gemm_nt.h"

namespace gdrf {

// hyper-parameter block kept on the device (double): filled by prep_hyper
struct Hyper { double ls, var, noise, inv_ls2; };

template <typename T>
__global__ void prep_hyper_kernel(const T* __restrict__ params, Hyper* h) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double ls = exp((double)params[0]), var = exp((double)params[1]), noise = exp((double)params[2]);
    h->ls = ls; h->var = var; h->noise = noise; h->inv_ls2 = 1.0 / (ls * ls);
  }
}

// elementwise precision change (inducing points, all-reduced G^T) between the N-side and the solve precision
template <typename TA, typename TB>
__global__ void cast_kernel(int64_t n, const TA* __restrict__ in, TB* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (TB)in[i];
}

// K_uu[i][j] = k(z_i, z_j) + jitter * (i == j), zero outside M
template <typename T>
__global__ void kuu_kernel(const T* __restrict__ Z, int M, int Mp, int D, int kind, const Hyper* __restrict__ h,
                           double jitter, T* __restrict__ Kuu) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= Mp) return;
  T v = 0;
  if (i < M && j < M) {
    const T r2 = sqdist<T>(Z + (int64_t)i * D, Z + (int64_t)j * D, D) * (T)h->inv_ls2;
    v = cov_from_r2<T>(kind, r2, (T)h->var);
    if (i == j) v += (T)jitter;
  }
  Kuu[(int64_t)i * Mp + j] = v;
}

// ---- single-workgroup blocked left-looking Cholesky (panel 32), in place on the lower triangle.
// A pivot <= 0 (or NaN) sets *flag and is replaced by 1 so that everything downstream stays finite;
// the optimizer update is predicated on the flag and the host retries with the reference's
// cumulative jitter schedule.
template <typename T>
__global__ __launch_bounds__(1024) void chol_kernel(T* __restrict__ A, int M, int ld, int* __restrict__ flag) {
  __shared__ T Sa[32][33];
  __shared__ T Sb[32][33];
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  for (int c0 = 0; c0 < M; c0 += 32) {
    // (a) panel update: A[i][c0+tx] -= sum_{p<c0} A[i][p] * A[c0+tx][p] for every row block i0 >= c0
    for (int i0 = c0; i0 < M; i0 += 32) {
      T acc = 0;
      for (int p0 = 0; p0 < c0; p0 += 32) {
        Sa[ty][tx] = (i0 + ty < M) ? A[(int64_t)(i0 + ty) * ld + p0 + tx] : T(0);
        Sb[ty][tx] = (c0 + ty < M) ? A[(int64_t)(c0 + ty) * ld + p0 + tx] : T(0);
        __syncthreads();
#pragma unroll
        for (int p = 0; p < 32; ++p) acc += Sa[ty][p] * Sb[tx][p];
        __syncthreads();
      }
      if (c0 > 0 && i0 + ty < M && c0 + tx < M) A[(int64_t)(i0 + ty) * ld + c0 + tx] -= acc;
    }
    __syncthreads();
    // (b) factor the 32x32 diagonal block in LDS
    {
      const bool in = (c0 + ty < M) && (c0 + tx < M);
      Sa[ty][tx] = in ? A[(int64_t)(c0 + ty) * ld + c0 + tx] : ((ty == tx) ? T(1) : T(0));
    }
    __syncthreads();
    for (int j = 0; j < 32; ++j) {
      if (tid == 0) {
        T d = Sa[j][j];
        if (!(d > T(0))) { if (c0 + j < M) *flag = 1; d = T(1); }
        Sa[j][j] = t_sqrt<T>(d);
      }
      __syncthreads();
      if (ty == 0 && tx > j) Sa[tx][j] /= Sa[j][j];
      __syncthreads();
      if (tx > j && ty >= tx) Sa[ty][tx] -= Sa[ty][j] * Sa[tx][j];
      __syncthreads();
    }
    if (c0 + ty < M && c0 + tx < M) A[(int64_t)(c0 + ty) * ld + c0 + tx] = (tx <= ty) ? Sa[ty][tx] : T(0);
    // (c) rows below the block: X = A_panel * Ld^{-T}.  Ld^{-1} by forward substitution (one column per
    // thread) into Sb, then a 32-deep product per element.
    Sb[ty][tx] = 0;
    __syncthreads();
    if (tid < 32) {
      const int c = tid;
      Sb[c][c] = T(1) / Sa[c][c];
      for (int r = c + 1; r < 32; ++r) {
        T s = 0;
        for (int p = c; p < r; ++p) s += Sa[r][p] * Sb[p][c];
        Sb[r][c] = -s / Sa[r][r];
      }
    }
    __syncthreads();
    for (int i0 = c0 + 32; i0 < M; i0 += 32) {
      // reuse Sa for the panel rows (the factored block already went back to global memory)
      __syncthreads();
      Sa[ty][tx] = (i0 + ty < M) ? A[(int64_t)(i0 + ty) * ld + c0 + tx] : T(0);
      __syncthreads();
      T o = 0;
#pragma unroll
      for (int p = 0; p < 32; ++p) o += Sa[ty][p] * Sb[tx][p];     // x_j = sum_p a_p Linv[j][p]
      if (i0 + ty < M) A[(int64_t)(i0 + ty) * ld + c0 + tx] = o;
    }
    __syncthreads();
  }
}

// L (strict upper and padding zeroed) and its transpose
template <typename T>
__global__ void finalize_l_kernel(const T* __restrict__ A, int M, int Mp, T* __restrict__ L, T* __restrict__ LT) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= Mp) return;
  const T v = (i < M && j <= i) ? A[(int64_t)i * Mp + j] : T(0);
  L[(int64_t)i * Mp + j] = v;
  LT[(int64_t)j * Mp + i] = v;
}

// inverse of every 32x32 diagonal block of L (identity on the padding)
template <typename T>
__global__ __launch_bounds__(64) void trinv_diag_kernel(const T* __restrict__ L, int M, int Mp, T* __restrict__ Dinv) {
  __shared__ T Sl[32][33];
  __shared__ T Sx[32][33];
  const int b = blockIdx.x, c = threadIdx.x;
  for (int e = c; e < 1024; e += 64) {
    const int r = e >> 5, q = e & 31;
    const int gi = b * 32 + r, gj = b * 32 + q;
    Sl[r][q] = (gi < M && gj < M) ? L[(int64_t)gi * Mp + gj] : ((r == q) ? T(1) : T(0));
    Sx[r][q] = 0;
  }
  __syncthreads();
  if (c < 32) {
    Sx[c][c] = T(1) / Sl[c][c];
    for (int r = c + 1; r < 32; ++r) {
      T s = 0;
      for (int p = c; p < r; ++p) s += Sl[r][p] * Sx[p][c];
      Sx[r][c] = -s / Sl[r][r];
    }
  }
  __syncthreads();
  for (int e = c; e < 1024; e += 64) {
    const int r = e >> 5, q = e & 31;
    Dinv[(int64_t)b * 1024 + e] = Sx[r][q];
  }
}

// block column J of Linv = L^{-1}; one 1024-thread workgroup per block column.
//   X[J][J] = Dinv[J];   X[I][J] = -Dinv[I] * sum_{P=J}^{I-1} L[I][P] X[P][J]
template <typename T>
__global__ __launch_bounds__(1024) void trinv_cols_kernel(const T* __restrict__ L, const T* __restrict__ Dinv, int M, int Mp,
                                                         T* __restrict__ X, T* __restrict__ XT) {
  __shared__ T Sa[32][33];
  __shared__ T Sb[32][33];
  const int J = blockIdx.x, nb = Mp / 32;
  const int tid = threadIdx.x, tx = tid & 31, ty = tid >> 5;
  auto put = [&](int I, T v) {
    const int gi = I * 32 + ty, gj = J * 32 + tx;
    if (gi >= M || gj >= M) v = 0;
    X[(int64_t)gi * Mp + gj] = v;
    XT[(int64_t)gj * Mp + gi] = v;
  };
  // blocks above the diagonal are zero
  for (int I = 0; I < J; ++I) put(I, T(0));
  put(J, Dinv[(int64_t)J * 1024 + ty * 32 + tx]);
  __syncthreads();
  for (int I = J + 1; I < nb; ++I) {
    T acc = 0;
    for (int P = J; P < I; ++P) {
      Sa[ty][tx] = L[(int64_t)(I * 32 + ty) * Mp + P * 32 + tx];
      Sb[ty][tx] = X[(int64_t)(P * 32 + ty) * Mp + J * 32 + tx];
      __syncthreads();
#pragma unroll
      for (int p = 0; p < 32; ++p) acc += Sa[ty][p] * Sb[p][tx];
      __syncthreads();
    }
    Sa[ty][tx] = acc;
    Sb[ty][tx] = Dinv[(int64_t)I * 1024 + ty * 32 + tx];
    __syncthreads();
    T o = 0;
#pragma unroll
    for (int p = 0; p < 32; ++p) o += Sb[ty][p] * Sa[p][tx];
    __syncthreads();
    put(I, -o);
    __threadfence_block();
    __syncthreads();
  }
}

// S_k = tril(unc,-1) + diag(exp(diag unc)) (LowerCholeskyTransform), S_k^T ; padded, zero outside M
template <typename T>
__global__ void build_s_kernel(const T* __restrict__ Sunc, int M, int Mp, T* __restrict__ S, T* __restrict__ ST) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, k = blockIdx.z;
  if (j >= Mp) return;
  T v = 0;
  if (i < M && j < M) {
    const T u = Sunc[((int64_t)k * M + i) * M + j];
    v = (j < i) ? u : ((j == i) ? t_exp<T>(u) : T(0));
  }
  S[((int64_t)k * Mp + i) * Mp + j] = v;
  ST[((int64_t)k * Mp + j) * Mp + i] = v;
}

// zero-padded copy of u_loc, [128][Mp], the Bt operand of the loc = W U^T product
template <typename T>
__global__ void build_upad_kernel(const T* __restrict__ U, int K, int M, int Mp, T* __restrict__ Upad) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (j >= Mp) return;
  Upad[(int64_t)k * Mp + j] = (k < K && j < M) ? U[(int64_t)k * M + j] : T(0);
}

// row softmax of the word-topic matrix (stack-of-simplex transform)
template <typename T>
__global__ void build_phi_kernel(const T* __restrict__ phi_unc, int K, int V, T* __restrict__ phi) {
  const int k = blockIdx.x;
  if (threadIdx.x != 0 || k >= K) return;
  T mx = phi_unc[(int64_t)k * V];
  for (int v = 1; v < V; ++v) mx = fmax(mx, phi_unc[(int64_t)k * V + v]);
  T s = 0;
  for (int v = 0; v < V; ++v) s += t_exp<T>(phi_unc[(int64_t)k * V + v] - mx);
  for (int v = 0; v < V; ++v) phi[(int64_t)k * V + v] = t_exp<T>(phi_unc[(int64_t)k * V + v] - mx) / s;
}

// ---- plain batched M x M product on the NT core: C[b] = alpha * A[b] * Bt[b]^T ------------------
template <typename T> struct MMProb : NTDefaultMap, NTPlainA<T> {
  using V = typename Vec16<T>::type;
  static constexpr bool SCALE_A = false;
  const T* A; int64_t a_bs;
  const T* Bt; int64_t b_bs;
  T* C; int64_t c_bs;
  int Mp; T alpha;
  struct ACtx { int64_t m0; }; struct ECtx {};
  __device__ __forceinline__ int col_tiles() const { return (Mp + NTCfg<T>::CW - 1) / NTCfg<T>::CW; }
  __device__ __forceinline__ bool loop_cols() const { return false; }
  __device__ __forceinline__ int a_reuse() const { return 1; }
  __device__ __forceinline__ void krange(int64_t, int, int, int& kb, int& ke) const { kb = 0; ke = Mp; }
  __device__ __forceinline__ void prepA(ACtx& c, int64_t m0, int, char*) const { c.m0 = m0; }
  __device__ __forceinline__ void prepE(ECtx&, int64_t, int) const {}
  __device__ __forceinline__ V zero() const { V z; for (int e = 0; e < Vec16<T>::N; ++e) z[e] = 0; return z; }
  __device__ __forceinline__ V loadA(const ACtx& c, int i, int k, int bz) const {
    const int64_t r = c.m0 + nt_stage_row<T>(i);
    return (r < Mp) ? *reinterpret_cast<const V*>(A + bz * a_bs + r * Mp + k) : zero();
  }
  __device__ __forceinline__ V loadB(int n0, int i, int k, int, int bz) const {
    const int c = n0 + nt_stage_row<T>(i);
    return (c < Mp) ? *reinterpret_cast<const V*>(Bt + bz * b_bs + (int64_t)c * Mp + k) : zero();
  }
  template <class Acc, int NB_>
  __device__ __forceinline__ void tile_done(Acc (&acc)[4][NB_], int64_t m0, int n0, int bz, ECtx&, int wr, int wc, int lane) const {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < NTCfg<T>::NB; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int64_t m = m0 + nt_acc_row<T>(wr, a, lane, r);
          const int n = n0 + nt_acc_col<T>(wc, b, lane);
          if (m < Mp && n < Mp) C[bz * c_bs + m * Mp + n] = alpha * acc[a][b][r];
        }
  }
  __device__ __forceinline__ void finish(int64_t, int, ECtx&, char*, int, int, int) const {}
};

// ---- elementwise pieces of the Cholesky backward --------------------------
// LbarT = -triu(HT)   where HT = G^T Linv  (so Lbar = -tril(Linv^T G))
template <typename T>
__global__ void lbar_t_kernel(const T* __restrict__ HT, int Mp, T* __restrict__ LbarT) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= Mp) return;
  LbarT[(int64_t)i * Mp + j] = (i <= j) ? -HT[(int64_t)i * Mp + j] : T(0);
}
// P = tril(Q) with the diagonal halved
template <typename T>
__global__ void phi_tril_kernel(const T* __restrict__ Q, int Mp, T* __restrict__ P) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= Mp) return;
  const T q = Q[(int64_t)i * Mp + j];
  P[(int64_t)i * Mp + j] = (j < i) ? q : ((j == i) ? T(0.5) * q : T(0));
}
// sum_{ij} Kuu_bar * K0  and  sum_{ij} Kuu_bar * dK0/dlog(ls), Kuu_bar = (S' + S'^T)/2 ; one partial pair per block
template <typename T>
__global__ void kuu_bar_reduce_kernel(const T* __restrict__ Sp, const T* __restrict__ Z, int M, int Mp, int D, int kind,
                                      const Hyper* __restrict__ h, double* __restrict__ part) {
  __shared__ double scratch[16];
  const int i = blockIdx.x;
  double s1 = 0, s2 = 0;
  const T var = (T)h->var, ils2 = (T)h->inv_ls2;
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    const T kb = T(0.5) * (Sp[(int64_t)i * Mp + j] + Sp[(int64_t)j * Mp + i]);
    const T r2 = sqdist<T>(Z + (int64_t)i * D, Z + (int64_t)j * D, D) * ils2;
    const T k0 = cov_from_r2<T>(kind, r2, var);
    s1 += (double)(kb * k0);
    s2 += (double)(kb * dcov_dlogls<T>(kind, k0, r2, var));
  }
  s1 = block_sum(s1, scratch);
  s2 = block_sum(s2, scratch);
  if (threadIdx.x == 0) { part[2 * i] = s1; part[2 * i + 1] = s2; }
}

// gradient of the loss w.r.t. the unconstrained u_scale_tril from Sbar = 2 A_k S_k (already scaled by 2)
template <typename T>
__global__ void grad_s_kernel(const T* __restrict__ Sbar, const T* __restrict__ S, int M, int Mp, double neg_inv_n,
                              T* __restrict__ g) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, k = blockIdx.z;
  if (j >= M) return;
  const int64_t pi = ((int64_t)k * Mp + i) * Mp + j;
  T v = 0;
  if (j < i) v = Sbar[pi];
  else if (j == i) v = Sbar[pi] * S[pi];
  g[((int64_t)k * M + i) * M + j] = (T)(neg_inv_n) * v;
}

}  // namespace gdrf

// Common device/host helpers for libgdrf_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define GDRF_WAVE 64
#define GDRF_TILE 128          // output tile edge of both GEMM cores
#define GDRF_KBYTES 128        // bytes of reduction index staged per row per chunk (NT core)
#ifndef GDRF_FWDW_WGS
#define GDRF_FWDW_WGS 2         // workgroups per CU the f64 W = K_nm L^-T kernel's register allocation leaves room for (2: 144 registers, 3 resident)
#endif
#ifndef GDRF_KBYTES_F64
#define GDRF_KBYTES_F64 0      // the same for f64 operands (0: GDRF_KBYTES).  256 (32 doubles per chunk, half the barriers per MFMA) was
                               // measured SLOWER: 180 registers -> 2 workgroups per CU, fwd_w 6.55 -> 7.03 ms
#endif
#ifndef GDRF_FWDW_DEPTH
#define GDRF_FWDW_DEPTH 2        // register prefetch distance (chunks) of the f64 W = K_nm L^-T kernel
#endif
#define GDRF_MPAD 32           // M is padded to a multiple of this in every workspace matrix
#define GDRF_DMAX 4            // input dimensions supported (index columns of the reference's CSV)

namespace gdrf {

template <typename T> struct Vec16;                 // 16-byte vector of T
template <> struct Vec16<float>  { using type = float  __attribute__((ext_vector_type(4))); static constexpr int N = 4; };
template <> struct Vec16<double> { using type = double __attribute__((ext_vector_type(2))); static constexpr int N = 2; };

// ---- MFMA 16x16x4 wrappers: same A/B lane maps for f32 and f64, different C/D row map
// (cdna_hip_programming.md §3: f32 row = 4*(lane>>4)+reg ; f64 row = (lane>>4)+4*reg).
template <typename T> struct Mfma;
template <> struct Mfma<float> {
  using acc_t = float __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return ((lane >> 4) << 2) + r; }
};
template <> struct Mfma<double> {
  using acc_t = double __attribute__((ext_vector_type(4)));
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + (r << 2); }
};

// A raw s_barrier (no vmcnt drain: LDS-DMA requests stay in flight across it) that the COMPILER cannot move memory accesses across either.
// The s_barrier intrinsic alone does not order ordinary LDS loads: hipcc may hoist a read that follows it in the source above it (after
// the preceding `s_waitcnt` asm) - another wave's LDS stores of the phase before are then read before they were made.  Found in round 3:
// an arithmetic-only edit of tn_topics_f16_kernel changed the schedule and produced run-to-run different results.  The empty asm with a
// memory clobber on either side pins the accesses; the caller still waits for its own counters (lgkmcnt / vmcnt) in front.
__device__ __forceinline__ void gdrf_raw_barrier() {
  asm volatile("" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// ---- wave / block reductions -------------------------------------------------
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// sum over the 16 lanes that share (lane>>4)
template <typename T> __device__ __forceinline__ T group16_sum(T v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// block sum for blockDim.x <= 1024 ; result valid in thread 0 ; scratch >= 16 doubles
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, l = threadIdx.x & 63, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (l == 0) scratch[w] = v;
  __syncthreads();
  double r = 0.0;
  if (threadIdx.x == 0) for (int i = 0; i < nw; ++i) r += scratch[i];
  return r;
}

template <typename T> __device__ __forceinline__ T t_exp(T x);
template <> __device__ __forceinline__ float  t_exp<float>(float x)   { return expf(x); }
template <> __device__ __forceinline__ double t_exp<double>(double x) { return exp(x); }
template <typename T> __device__ __forceinline__ T t_log(T x);
template <> __device__ __forceinline__ float  t_log<float>(float x)   { return logf(x); }
template <> __device__ __forceinline__ double t_log<double>(double x) { return log(x); }
template <typename T> __device__ __forceinline__ T t_sqrt(T x);
template <> __device__ __forceinline__ float  t_sqrt<float>(float x)   { return sqrtf(x); }
template <> __device__ __forceinline__ double t_sqrt<double>(double x) { return sqrt(x); }
template <typename T> __device__ __forceinline__ T t_eps();
template <> __device__ __forceinline__ float  t_eps<float>()  { return 1.1920928955078125e-07f; }
template <> __device__ __forceinline__ double t_eps<double>() { return 2.220446049250313e-16; }

// ---- covariance functions (pyro.contrib.gp.kernels.isotropic semantics, direct (x-z)^2 form)
// kind 0 = RBF, 1 = Matern52.  r2 is the squared distance scaled by 1/lengthscale^2.
struct KParams { double inv_ls2; double var; };     // device-side hyper-parameters (double, cast per use)

// kind: 0 RBF, 1 Matern52, 2 Matern32, 3 Exponential, 4 RationalQuadratic  (pyro.contrib.gp.kernels.isotropic)
// RationalQuadratic: k = var * (1 + r2 / (2 alpha))^(-alpha), alpha = scale_mixture (ignored by the other kinds)
template <typename T> __device__ __forceinline__ T cov_from_r2(int kind, T r2, T var, T alpha = T(1)) {
  if (kind == 0) return var * t_exp<T>(T(-0.5) * r2);
  if (kind == 4) return var * t_exp<T>(-alpha * t_log<T>(T(1) + r2 * (T(0.5) / alpha)));
  const T r = t_sqrt<T>(r2 + T(1e-12));
  if (kind == 3) return var * t_exp<T>(-r);
  if (kind == 2) { const T a = T(1.73205080756887729353) * r; return var * (T(1) + a) * t_exp<T>(-a); }
  const T a = T(2.23606797749978969641) * r;
  return var * (T(1) + a + (T(5) / T(3)) * r * r) * t_exp<T>(-a);
}
// d k / d log(lengthscale)
template <typename T> __device__ __forceinline__ T dcov_dlogls(int kind, T k, T r2, T var, T alpha = T(1)) {
  if (kind == 0) return k * r2;
  if (kind == 4) return k * r2 / (T(1) + r2 * (T(0.5) / alpha));
  const T r = t_sqrt<T>(r2 + T(1e-12));
  if (kind == 3) return k * (r2 / r);
  if (kind == 2) { const T a = T(1.73205080756887729353) * r; return var * t_exp<T>(-a) * a * T(1.73205080756887729353) * (r2 / r); }
  const T a = T(2.23606797749978969641) * r;
  return var * t_exp<T>(-a) * (a / T(3)) * (T(1) + a) * T(2.23606797749978969641) * (r2 / r);
}
// the same derivative from an already evaluated k (no second exponential): exp(-a) = k / (var (1 + a + 5/3 r^2))
template <typename T> __device__ __forceinline__ T dcov_dlogls_from_k(int kind, T k, T r2, T alpha = T(1)) {
  if (kind == 0) return k * r2;
  if (kind == 4) return k * r2 / (T(1) + r2 * (T(0.5) / alpha));
  const T r = t_sqrt<T>(r2 + T(1e-12));
  if (kind == 3) return k * (r2 / r);
  if (kind == 2) { const T a = T(1.73205080756887729353) * r; return k / (T(1) + a) * a * T(1.73205080756887729353) * (r2 / r); }
  const T a = T(2.23606797749978969641) * r;
  return k / (T(1) + a + (T(5) / T(3)) * r * r) * (a / T(3)) * (T(1) + a) * T(2.23606797749978969641) * (r2 / r);
}
// d k / d r2 from an already evaluated k (r2 = squared distance / lengthscale^2): the factor of the gradient with respect
// to an input point, d k(x,z)/d z_d = dcov_dr2 * d r2/d z_d = dcov_dr2 * 2 (z_d - x_d) / lengthscale^2
template <typename T> __device__ __forceinline__ T dcov_dr2_from_k(int kind, T k, T r2, T alpha = T(1)) {
  if (kind == 0) return T(-0.5) * k;
  if (kind == 4) return T(-0.5) * k / (T(1) + r2 * (T(0.5) / alpha));
  const T r = t_sqrt<T>(r2 + T(1e-12));
  if (kind == 3) return -k / (T(2) * r);
  if (kind == 2) { const T a = T(1.73205080756887729353) * r; return T(-1.5) * k / (T(1) + a); }
  const T a = T(2.23606797749978969641) * r;
  return -(T(5) / T(6)) * (T(1) + a) * k / (T(1) + a + (T(5) / T(3)) * r * r);
}
// d k / d log(alpha) of the RationalQuadratic kernel (0 for the others): with u = r2 / (2 alpha),
// d/d alpha [-alpha log(1 + u)] = -log(1 + u) + u / (1 + u)
template <typename T> __device__ __forceinline__ T dcov_dlogalpha_from_k(int kind, T k, T r2, T alpha) {
  if (kind != 4) return T(0);
  const T u = r2 * (T(0.5) / alpha);
  return k * alpha * (u / (T(1) + u) - t_log<T>(T(1) + u));
}
template <typename T> __device__ __forceinline__ T sqdist(const T* __restrict__ x, const T* __restrict__ z, int D) {
  T s = 0;
  for (int d = 0; d < D; ++d) { const T t = x[d] - z[d]; s += t * t; }
  return s;
}

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

}  // namespace gdrf

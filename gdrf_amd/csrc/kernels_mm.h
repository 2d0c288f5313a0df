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
struct Hyper { double ls, var, noise, inv_ls2, alpha; };     // alpha: RationalQuadratic scale_mixture (parameter slot 3)

template <typename T>
__global__ void prep_hyper_kernel(const T* __restrict__ params, Hyper* h) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const double ls = exp((double)params[0]), var = exp((double)params[1]), noise = exp((double)params[2]);
    h->ls = ls; h->var = var; h->noise = noise; h->inv_ls2 = 1.0 / (ls * ls); h->alpha = exp((double)params[3]);
  }
}

// the values a factorisation depends on - the first four parameters (log lengthscale, variance, noise, scale mixture) and the inducing inputs -
// copied aside (mode 0) or compared bit for bit with that copy (mode 1: *mismatch = 1 if any differs, 0 otherwise - written
// unconditionally, so the word needs no clearing by anyone else: a memset queued on another stream could land after this kernel and
// erase a mismatch).  One workgroup.
template <typename T>
__global__ void fact_snapshot_kernel(const T* __restrict__ params, const T* __restrict__ Z, int64_t nz, T* __restrict__ snap, int mode, int* __restrict__ mismatch) {
  if (mode == 1) {
    if (threadIdx.x == 0) *mismatch = 0;
    __syncthreads();
  }
  int bad = 0;
  for (int64_t e = threadIdx.x; e < nz + 4; e += blockDim.x) {
    const T v = e < 4 ? params[e] : Z[e - 4];
    if (mode == 0) snap[e] = v;
    else if (!(v == snap[e]) && !(v != v && snap[e] != snap[e])) bad = 1;     // equal values (a NaN only matches a NaN)
  }
  if (mode == 1 && bad) *mismatch = 1;
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
    v = cov_from_r2<T>(kind, r2, (T)h->var, (T)h->alpha);
    if (i == j) v += (T)jitter;
  }
  Kuu[(int64_t)i * Mp + j] = v;
}

// ---- single-workgroup blocked left-looking Cholesky (panel 32), in place on the lower triangle.
// Per panel c0:
//   (a) U = A[c0:M, c0:c0+32] - L[c0:M, 0:c0] L[c0:c0+32, 0:c0]^T on the matrix cores: the 32 panel rows of L are
//       staged in LDS (B operand), every wave owns 16-row strips of U and streams its A operand (16 rows x 16
//       reduction indices per 16-byte load) straight from L2; 2 MFMA column tiles per strip.
//   (b) the 32x32 diagonal block of U is factored by ONE wave, one row per lane, rows in registers, the current
//       column broadcast through LDS (no workgroup barriers inside the column loop);
//   (c) rows below: X = U Ld^{-T} with Ld^{-1} from forward substitution (one column per lane).
// A pivot <= 0 (or NaN) sets *flag and is replaced by 1 so that everything downstream stays finite; the optimizer
// update is predicated on the flag and the host retries with the reference's cumulative jitter schedule.
// blockIdx.x selects an independent problem: matrix A + blockIdx.x * batch_stride, flag[blockIdx.x] (the jitter
// probe factorises the same K_uu with several cumulative jitters in one launch).
#define CHOL_PC 256
__device__ __forceinline__ float chol_readlane(float v, int lane) {
  return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ double chol_readlane(double v, int lane) {
  const uint64_t u = __builtin_bit_cast(uint64_t, v);
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)u, lane);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(u >> 32), lane);
  return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
template <typename T>
__global__ __launch_bounds__(1024) void chol_kernel(T* __restrict__ A, int M, int ld, int* __restrict__ flag, int64_t batch_stride) {
  using MM = Mfma<T>;
  using acc_t = typename MM::acc_t;
  using V = typename Vec16<T>::type;
  constexpr int VE = Vec16<T>::N;            // elements per 16-byte vector: 4 (f32) / 2 (f64)
  constexpr int KC = 4 * VE;                 // reduction indices per staged step: lane group g owns VE of them
  extern __shared__ __attribute__((aligned(16))) char smem_chol[];
  T* Lp = reinterpret_cast<T*>(smem_chol);   // [32][ldp] one column chunk of the panel rows L[c0:c0+32, pc:pc+CHOL_PC]
  constexpr int ldp = CHOL_PC + VE;          // row stride of Lp (elements), 16-byte aligned, bank-shifted
  T (*Sa)[33] = reinterpret_cast<T (*)[33]>(Lp + 32 * ldp);
  T (*Sb)[33] = Sa + 32;
  T* Scol = reinterpret_cast<T*>(Sb + 32);   // [32]
  A += (int64_t)blockIdx.x * batch_stride;
  flag += blockIdx.x;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6, lr = lane & 15, lg = lane >> 4;
#ifdef GDRF_CHOL_PROF
  unsigned long long tph[4] = {0, 0, 0, 0}, t0 = __builtin_readcyclecounter();
#define CHOL_TICK(i) { const unsigned long long t1 = __builtin_readcyclecounter(); tph[i] += t1 - t0; t0 = t1; }
#else
#define CHOL_TICK(i)
#endif
  for (int c0 = 0; c0 < M; c0 += 32) {
    // ---- (a) panel update.  The panel rows are staged in column chunks of CHOL_PC; a wave keeps the accumulators
    //      of up to NS of its strips across the chunks (more strips: another pass over the chunks).
    if (c0 > 0) {
      const int nstrips = (M - c0 + 15) / 16;
      constexpr int NS = sizeof(T) == 4 ? 4 : 2;             // strips whose accumulators a wave carries at once
      for (int sg = 0; sg * 16 * NS < nstrips; ++sg) {     // strips sg*16*NS + wave + 16*s, s = 0..NS-1
        acc_t acc[NS][2];
#pragma unroll
        for (int s4 = 0; s4 < NS; ++s4) { acc[s4][0] = acc_t{0, 0, 0, 0}; acc[s4][1] = acc_t{0, 0, 0, 0}; }
        for (int pc = 0; pc < c0; pc += CHOL_PC) {
          const int pw = (c0 - pc < CHOL_PC) ? (c0 - pc) : CHOL_PC;      // multiple of 32
          __syncthreads();
          for (int e = tid; e < 32 * pw; e += 1024) {
            const int r = e / pw, q = e - r * pw;
            Lp[r * ldp + q] = (c0 + r < M) ? A[(int64_t)(c0 + r) * ld + pc + q] : T(0);
          }
          __syncthreads();
#pragma unroll
          for (int s4 = 0; s4 < NS; ++s4) {
            const int st = sg * 16 * NS + wave + 16 * s4;
            if (st >= nstrips) continue;
            const int row = c0 + st * 16 + lr;
            const bool rok = row < M;
            const T* arow = A + (int64_t)(rok ? row : 0) * ld + pc;
            // 32 reduction indices per pass: the UN strided global loads of a pass are issued together, so the wave waits for
            // one L2 round trip per 32 columns instead of one per KC (this loop was latency-bound: 22 % of the kernel)
            constexpr int UN = 32 / KC;
            for (int p0 = 0; p0 < pw; p0 += 32) {
              V a[UN];
#pragma unroll
              for (int u = 0; u < UN; ++u) {
                a[u] = *reinterpret_cast<const V*>(arow + p0 + u * KC + lg * VE);
                if (!rok) { for (int e = 0; e < VE; ++e) a[u][e] = 0; }
              }
#pragma unroll
              for (int u = 0; u < UN; ++u) {
                const V b0 = *reinterpret_cast<const V*>(Lp + lr * ldp + p0 + u * KC + lg * VE);
                const V b1 = *reinterpret_cast<const V*>(Lp + (16 + lr) * ldp + p0 + u * KC + lg * VE);
#pragma unroll
                for (int e = 0; e < VE; ++e) { acc[s4][0] = MM::mma(a[u][e], b0[e], acc[s4][0]); acc[s4][1] = MM::mma(a[u][e], b1[e], acc[s4][1]); }
              }
            }
          }
        }
#pragma unroll
        for (int s4 = 0; s4 < NS; ++s4) {
          const int st = sg * 16 * NS + wave + 16 * s4;
          if (st >= nstrips) continue;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = c0 + st * 16 + MM::crow(lane, r);
            if (i < M) {
              if (c0 + lr < M) A[(int64_t)i * ld + c0 + lr] -= acc[s4][0][r];
              if (c0 + 16 + lr < M) A[(int64_t)i * ld + c0 + 16 + lr] -= acc[s4][1][r];
            }
          }
        }
      }
      __syncthreads();
    }
    CHOL_TICK(0)
    // ---- (b) factor the 32x32 diagonal block: one wave, rows in registers
    if (wave == 0 && sizeof(T) == 8) {
      // f64: a row per lane PAIR - lane l holds the even columns of row l, lane l + 32 the odd ones - so that the 31-column update
      // of a pivot step, the bulk of this serial chain (47 % of the kernel), is half as many instructions.  The scaled pivot column
      // reaches every lane through LDS (Scol): its own row's entry as the multiplier, the other rows' entries as broadcasts.
      const int l = lane & 31, half = lane >> 5;
      T xh[16];                                // xh[i] = U[l][2 i + half]
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = 2 * i + half;
        const bool in = (c0 + l < M) && (c0 + c < M);
        xh[i] = in ? A[(int64_t)(c0 + l) * ld + c0 + c] : ((l == c) ? T(1) : T(0));
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        constexpr int dummy = 0; (void)dummy;
        const int jh = j & 1, ji = j >> 1;
        T d = chol_readlane(xh[ji], j + 32 * jh);             // U[j][j]: row j, in the half that owns column j
        if (!(d > T(0))) { if (lane == 0 && c0 + j < M) *flag = 1; d = T(1); }
        double y = __builtin_amdgcn_rsq((double)d);
        y = y * (1.5 - 0.5 * d * y * y);
        y = y * (1.5 - 0.5 * d * y * y);
        double p0 = d * y;
        p0 = p0 + 0.5 * y * (d - p0 * p0);
        const T piv = (T)p0, rp = (T)(y + y * (1.0 - p0 * y));
        if (half == jh) {
          const T colv = (l == j) ? piv : ((l > j) ? xh[ji] * rp : T(0));
          xh[ji] = colv;
          Scol[l] = colv;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const T xj = Scol[l];                                  // L[l][j]
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          if (2 * i + 1 > j) {                                 // compile time: some half still has a column right of j at index i
            const int c = 2 * i + half;
            const T lcj = Scol[c];                             // L[c][j]
            if (l >= c && c > j) xh[i] -= xj * lcj;
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      T dg = T(1);
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int c = 2 * i + half;
        Sa[l][c] = xh[i];
        if (c0 + l < M && c0 + c < M) A[(int64_t)(c0 + l) * ld + c0 + c] = xh[i];
        if (c == l) dg = xh[i];
      }
      if (half == (l & 1)) Scol[l] = T(1) / dg;                // reciprocal pivots for the rows below (after the last step's readers)
    } else if (wave == 0) {
      const int l = lane & 31;                 // lanes 32..63 mirror lanes 0..31 (same values, same writes)
      T x[32];
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        const bool in = (c0 + l < M) && (c0 + j < M);
        x[j] = in ? A[(int64_t)(c0 + l) * ld + c0 + j] : ((l == j) ? T(1) : T(0));
      }
#pragma unroll
      for (int j = 0; j < 32; ++j) {
        T d = chol_readlane(x[j], j);
        if (!(d > T(0))) { if (lane == 0 && c0 + j < M) *flag = 1; d = T(1); }
        T piv, rp;
        if constexpr (sizeof(T) == 8) {
          // sqrt and reciprocal from ONE v_rsq_f64 seed and Newton steps (the library sqrt + divide pair is ~45 dependent
          // instructions on this serial chain): y -> y (1.5 - 0.5 d y^2) twice, then piv = d y with one correction
          double y = __builtin_amdgcn_rsq(d);
          y = y * (1.5 - 0.5 * d * y * y);
          y = y * (1.5 - 0.5 * d * y * y);
          double p0 = d * y;
          p0 = p0 + 0.5 * y * (d - p0 * p0);
          piv = p0; rp = y + y * (1.0 - p0 * y);
        } else {
          piv = t_sqrt<T>(d); rp = T(1) / piv;
        }
        x[j] = (l == j) ? piv : ((l > j) ? x[j] * rp : T(0));
        if constexpr (sizeof(T) == 4) {
          // f32: the column reaches the other lanes through v_readlane (compile-time lane index, no LDS round trip)
          // no (l >= c) predicate: entries above the diagonal are never read (step c zeroes them), and a select per element
          // doubled the instruction count of this serial chain
#pragma unroll
          for (int c = j + 1; c < 32; ++c) {
            const T lcj = chol_readlane(x[j], c);
            x[c] -= x[j] * lcj;
          }
        } else {
          // f64: two readlanes per value plus their hazards measured slower than a broadcast through LDS
          Scol[l] = x[j];
          __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
          __builtin_amdgcn_wave_barrier();
#pragma unroll
          for (int c = j + 1; c < 32; ++c) {
            const T lcj = Scol[c];
            if (l >= c) x[c] -= x[j] * lcj;               // (f64: dropping this predicate as in the f32 branch measured 1.5x SLOWER)
          }
          __builtin_amdgcn_wave_barrier();
        }
      }
      if (lane < 32) {
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          Sa[l][j] = x[j];
          if (c0 + l < M && c0 + j < M) A[(int64_t)(c0 + l) * ld + c0 + j] = x[j];
        }
        T dg = T(1);
#pragma unroll
        for (int j = 0; j < 32; ++j) if (l == j) dg = x[j];
        Scol[l] = T(1) / dg;                   // reciprocal pivots for the rows below
      }
    }
    __syncthreads();
    CHOL_TICK(1)
    CHOL_TICK(2)
    // ---- (c) rows below the block: x Ld^T = u, one row per thread by forward substitution in registers; the factored block
    //      (Sa) and the reciprocal pivots (Scol) are broadcast reads from LDS
    for (int i0 = c0 + 32; i0 < M; i0 += 1024) {
      const int row = i0 + tid;
      if (row < M) {
        T a[32];
        T* arow = A + (int64_t)row * ld + c0;
#pragma unroll
        for (int j = 0; j < 32; ++j) a[j] = arow[j];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
          T sacc = a[j];
#pragma unroll
          for (int q = 0; q < j; ++q) sacc -= a[q] * Sa[j][q];
          a[j] = sacc * Scol[j];
        }
#pragma unroll
        for (int j = 0; j < 32; ++j) if (c0 + j < M) arow[j] = a[j];
      }
    }
    __syncthreads();
    CHOL_TICK(3)
  }
#ifdef GDRF_CHOL_PROF
  if (tid == 0 && blockIdx.x == 0) printf("chol<%d> cycles: update %llu  diag %llu  inverse %llu  apply %llu\n", (int)sizeof(T), tph[0], tph[1], tph[2], tph[3]);
#endif
}

// dynamic LDS bytes of chol_kernel<T>
template <typename T> inline size_t chol_lds_bytes(int) {
  const size_t ldp = CHOL_PC + Vec16<T>::N;
  return (32 * ldp + 2 * 32 * 33 + 32) * sizeof(T);
}

// ---- tile Cholesky, ONE LAUNCH PER 32-COLUMN PANEL (right-looking), for the factorisation on the step's critical path.
// chol_kernel above is one workgroup for the whole matrix: its panel update runs at the matrix rate of a single CU, its 32 x 32
// diagonal factor costs ~1250 cycles per pivot (one LDS all-gather of the scaled column per pivot) and the rows below are LDS-fed
// substitutions: 0.80 ms at M = 512 in double, the fixed cost of every step.  Here launch k factors panel k and applies it to the
// whole trailing matrix with one 2-wave workgroup per trailing tile (i, j), k < j <= i; nothing inside a launch depends on another
// workgroup (the launch boundary is the only synchronisation), because every workgroup REDOES the small serial part itself:
//   wave 0 factors the 64 x 32 panel [A(k,k); A(i,k)], wave 1 the panel [A(k,k); A(j,k)] - the diagonal tile's factor comes out
//   of both, L(i,k) and L(j,k) are the lower halves - then the two waves apply A(i,j) -= L(i,k) L(j,k)^T.
// A panel lives in the accumulator layout of the 16x16x4 matrix instruction and is factored FOUR pivots at a time: the four
// columns go through LDS once; every lane factors the 4 x 4 diagonal block redundantly (the only serial chain: 4 reciprocal
// square roots), solves its own rows against it, and the rank-4 update of the columns to the right is <= 7 MFMAs.  8 such steps
// per launch instead of 32 all-gathers.  L(., k) is written to a separate matrix (other workgroups of the launch still read the
// unfactored column k), by the workgroups of trailing column k + 1; diagonal tiles are kept fully symmetric.
// A pivot <= 0 (or NaN) sets *flag and is replaced by 1, as in chol_kernel.
template <typename T> __device__ __forceinline__ void chol_pivot(T d, T& piv, T& rp) { piv = t_sqrt<T>(d); rp = T(1) / piv; }
template <> __device__ __forceinline__ void chol_pivot<double>(double d, double& piv, double& rp) {
  // one v_rsq_f64 seed (>= 2^-23 relative) and ONE Newton step on it (>= 2^-45); the square root and the reciprocal each take their own
  // correction from there, which squares the error again (the library sqrt + divide is ~45 dependent instructions on the serial chain)
  double y = __builtin_amdgcn_rsq(d);
  y = y * (1.5 - 0.5 * d * y * y);
  double p0 = d * y;
  p0 = p0 + 0.5 * y * (d - p0 * p0);
  piv = p0; rp = y + y * (1.0 - p0 * y);
}
template <typename T>
__global__ __launch_bounds__(128) void chol_panel_kernel(T* __restrict__ A, T* __restrict__ Lout, T* __restrict__ Dinv, int M, int ld, int k, int* __restrict__ flag) {
  using MM = Mfma<T>;
  using acc_t = typename MM::acc_t;
  constexpr int LS = 34;                                     // row stride of the L tiles in LDS (elements)
  __shared__ __attribute__((aligned(16))) T Pn[2][64][4];    // per wave: the four panel columns of the current step
  __shared__ __attribute__((aligned(16))) T Ls[2][32][LS];   // L(i,k), L(j,k)
  __shared__ __attribute__((aligned(16))) T Ld[2][32][LS];   // L(k,k) (each wave's own copy: identical values)
  const int nt = (M + 31) / 32, n = nt - k - 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, lr = lane & 15, lg = lane >> 4;
  int i = k, j = k;
  if (n > 0) {                                               // trailing tiles, column by column
    int b = blockIdx.x, jj = 0;
    while (b >= n - jj) { b -= n - jj; ++jj; }
    j = k + 1 + jj; i = j + b;
  }
  const bool last = n == 0;
  const int t = (w == 0) ? i : j;
  // In the workgroup that stores L(k,k) wave 1 would repeat wave 0's panel (i == j).  It factors [A(k,k); 1] instead: the rows below
  // come out as 1 L(k,k)^-T, the transposed inverse of the diagonal block, which the explicit inverse of L needs (trinv_cols_kernel)
  const bool inv_duty = w == 1 && (last || (i == k + 1 && j == k + 1));
  auto ld_elem = [&](int gr, int gc) -> T {
    return (gr < M && gc < M) ? A[(int64_t)gr * ld + gc] : ((gr == gc) ? T(1) : T(0));
  };
  // the trailing tile's rows 16 w .. 16 w + 15 (both column halves), fetched up front
  acc_t u[2];
  if (!last) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) u[cb][r] = ld_elem(32 * i + 16 * w + MM::crow(lane, r), 32 * j + 16 * cb + lr);
  }
  // the panel: row blocks 0, 1 = A(k,k), 2, 3 = A(t,k); the strictly upper tile (0, 1) is never needed
  acc_t c[4][2];
#pragma unroll
  for (int rb = 0; rb < 4; ++rb)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = (rb < 2 ? 32 * k + 16 * rb : 32 * t + 16 * (rb - 2)) + MM::crow(lane, r);
        if (rb >= 2 && inv_duty) c[rb][cb][r] = (16 * (rb - 2) + MM::crow(lane, r) == 16 * cb + lr) ? T(1) : T(0);
        else c[rb][cb][r] = (rb == 0 && cb == 1) ? T(0) : ld_elem(gr, 32 * k + 16 * cb + lr);
      }
  typedef T V2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    const int j0 = 4 * s, cb0 = s >> 2;
    // (1) the step's four columns -> LDS (16 lanes hold them)
    if ((lr >> 2) == (s & 3)) {
      const int q = lr & 3;
#pragma unroll
      for (int rb = 0; rb < 4; ++rb)
#pragma unroll
        for (int r = 0; r < 4; ++r) Pn[w][16 * rb + MM::crow(lane, r)][q] = c[rb][cb0][r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // (2) the 4 x 4 diagonal block (lower part) and this lane's four rows (row 16 rb + lr of every row block)
    T d[4][4], x[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const V2 lo = *reinterpret_cast<const V2*>(&Pn[w][j0 + q][0]), hi = *reinterpret_cast<const V2*>(&Pn[w][j0 + q][2]);
      d[q][0] = lo[0]; d[q][1] = lo[1]; d[q][2] = hi[0]; d[q][3] = hi[1];
    }
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
      const V2 lo = *reinterpret_cast<const V2*>(&Pn[w][16 * rb + lr][0]), hi = *reinterpret_cast<const V2*>(&Pn[w][16 * rb + lr][2]);
      x[rb][0] = lo[0]; x[rb][1] = lo[1]; x[rb][2] = hi[0]; x[rb][3] = hi[1];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();                         // all reads done before the next step's columns overwrite Pn
    // (3) factor the block: l[q][p], p <= q; ri[q] = 1 / l[q][q]
    T l[4][4], ri[4];
    bool bad = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      T dq = d[q][q];
#pragma unroll
      for (int p = 0; p < q; ++p) dq -= l[q][p] * l[q][p];
      if (!(dq > T(0))) { bad = bad || (32 * k + j0 + q < M); dq = T(1); }
      chol_pivot<T>(dq, l[q][q], ri[q]);
#pragma unroll
      for (int q2 = q + 1; q2 < 4; ++q2) {
        T v = d[q2][q];
#pragma unroll
        for (int p = 0; p < q; ++p) v -= l[q2][p] * l[q][p];
        l[q2][q] = v * ri[q];
      }
    }
    if (bad && lane == 0) *flag = 1;
    // (4) this lane's rows against the block: x L44^T = u.  Rows of the diagonal tile above the block are finished (0), rows inside
    //     it are the block's own rows (exactly l, zero right of the diagonal)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        T v = x[rb][q];
#pragma unroll
        for (int p = 0; p < q; ++p) v -= x[rb][p] * l[q][p];
        x[rb][q] = v * ri[q];
      }
      if (rb < 2) {
        const int rel = 16 * rb + lr - j0;
#pragma unroll
        for (int q = 0; q < 4; ++q) x[rb][q] = (rel < q) ? T(0) : ((rel == q) ? l[q][q] : x[rb][q]);
      }
    }
    // the matrix-instruction operand of row 16 rb + lr: column lg of the four
    T op[4];
#pragma unroll
    for (int rb = 0; rb < 4; ++rb) op[rb] = (lg == 0) ? x[rb][0] : ((lg == 1) ? x[rb][1] : ((lg == 2) ? x[rb][2] : x[rb][3]));
    Ld[w][lr][j0 + lg] = op[0];
    Ld[w][16 + lr][j0 + lg] = op[1];
    Ls[w][lr][j0 + lg] = op[2];
    Ls[w][16 + lr][j0 + lg] = op[3];
    // (5) rank-4 update of the columns from this block on (the block's own columns become ~0 and are not read again)
#pragma unroll
    for (int rb = 0; rb < 4; ++rb)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) {
        if (cb < cb0 || (rb == 0 && (cb == 1 || s >= 4)) ) continue;
        c[rb][cb] = MM::mma(-op[rb], op[cb], c[rb][cb]);
      }
  }
  __syncthreads();
  if (!last) {
    // A(i,j) -= L(i,k) L(j,k)^T: this wave's 16 rows
    const int bsel = (i == j) ? 0 : 1;                        // (wave 1 of a diagonal tile may have been on inverse duty)
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      const T a = -Ls[0][16 * w + lr][4 * s + lg];
#pragma unroll
      for (int cb = 0; cb < 2; ++cb) u[cb] = MM::mma(a, Ls[bsel][16 * cb + lr][4 * s + lg], u[cb]);
    }
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int gr = 32 * i + 16 * w + MM::crow(lane, r), gc = 32 * j + 16 * cb + lr;
        if (gr < ld && gc < ld) A[(int64_t)gr * ld + gc] = u[cb][r];
      }
  }
  // L(i,k) by the workgroups of the first trailing column, L(k,k) by the first of them (or by the last launch's only workgroup)
  const T (*src)[LS] = nullptr; int gi = 0;
  if (last) { if (w == 0) { src = Ld[0]; gi = k; } }
  else if (j == k + 1) {
    if (w == 0) { src = Ls[0]; gi = i; }
    else if (i == k + 1) { src = Ld[1]; gi = k; }
  }
  if (src) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int e = it * 64 + lane, r = e >> 5, q = e & 31;
      const int gr = 32 * gi + r, gc = 32 * k + q;
      if (gr < ld && gc < ld) Lout[(int64_t)gr * ld + gc] = src[r][q];
    }
  }
  if (inv_duty && Dinv) {
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int e = it * 64 + lane, r = e >> 5, q = e & 31;
      Dinv[(int64_t)k * 1024 + e] = (q <= r) ? Ls[1][q][r] : T(0);
    }
  }
}

// nlev copies of K (no jitter) with jitters[lev] added to the diagonal of copy lev
struct JitterLevels { double v[8]; };
template <typename T>
__global__ void level_copies_kernel(const T* __restrict__ K0, int M, int Mp, JitterLevels jl, T* __restrict__ out) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, lev = blockIdx.z;
  if (j >= Mp) return;
  T v = K0[(int64_t)i * Mp + j];
  if (i == j && i < M) v += (T)jl.v[lev];
  out[((int64_t)lev * Mp + i) * Mp + j] = v;
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

// block ROW I of X = L^{-1}; one 256-thread workgroup per block row, one wave per 16x16 quadrant of a 32x32 block.  From X L = 1:
//   X[I][I] = Dinv[I];   X[I][J] = -(sum_{P=J+1}^{I} X[I][P] L[P][J]) Dinv[J],  J = I-1 .. 0
// The 32x32x32 tile products run on the matrix cores with both operands fetched straight from L2 in MFMA fragment order: the right
// operand L[P][J] with its columns across the lanes, the left operand X[I][P] out of the TRANSPOSED copy this kernel writes anyway
// (XT[P][I]: its rows across the lanes) - 128-byte segments on both sides, no LDS staging, PB tile pairs per round trip.  Only the
// product with Dinv[J] needs the sum turned from the accumulator layout into a left operand, through one 32x32 LDS tile.
// (History: the scalar column form - every thread 32 LDS-fed FMAs per tile product - was LDS-bound, 310-335 us at M = 512 f64;
// the MFMA column form with L as the left operand, 32-byte strided segments, 190 us.)
// 16 waves: wave = (slice of the P sum, 16 x 16 quadrant).  The serial chain is the J loop (up to Mp / 32 - 1 dependent steps per block
// row); with one wave per quadrant a step summed up to 15 tile products one after the other (137 us at M = 512 in double, behind
// the 0.23 ms panel-wise factorisation).  The four slices' partial sums meet in LDS and are added in a fixed order.
template <typename T>
__global__ __launch_bounds__(1024) void trinv_cols_kernel(const T* __restrict__ L, const T* __restrict__ Dinv, int M, int Mp,
                                                         T* __restrict__ X, T* __restrict__ XT) {
  using MM = Mfma<T>;
  using acc_t = typename MM::acc_t;
  __shared__ T Sp[4][32][33];
  const int I = blockIdx.x, nb = Mp / 32;
  const int tid = threadIdx.x, lane = tid & 63, w16 = tid >> 6, wave = w16 & 3, sl = w16 >> 2;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
  auto put = [&](int J, int r, int c, T v) {
    const int gi = I * 32 + r, gj = J * 32 + c;
    if (gi >= M || gj >= M) v = 0;
    X[(int64_t)gi * Mp + gj] = v;
    XT[(int64_t)gj * Mp + gi] = v;
  };
  // blocks right of the diagonal are zero
  for (int J = I + 1; J < nb; ++J)
    for (int e = tid; e < 1024; e += 1024) put(J, e >> 5, e & 31, T(0));
  for (int e = tid; e < 1024; e += 1024) put(I, e >> 5, e & 31, Dinv[(int64_t)I * 1024 + e]);
  __threadfence_block();
  __syncthreads();
  constexpr int PB = 2;
  const T* xt = XT + (int64_t)lg * Mp + I * 32 + wr * 16 + lr;         // + (P*32 + 4*kk)*Mp          : left fragment element [row lr][k lg] = X[I][P][row][k]
  const T* lp = L + (int64_t)lg * Mp + wc * 16 + lr;                   // + (P*32 + 4*kk)*Mp + J*32   : right fragment element [k lg][col lr]
  for (int J = I - 1; J >= 0; --J) {
    acc_t acc = acc_t{0, 0, 0, 0};
    // slice sl takes P = J + 1 + sl, + 4, ... in pairs
    for (int P0 = J + 1 + sl; P0 <= I; P0 += 4 * PB) {
      T a[PB][8], b[PB][8];
#pragma unroll
      for (int u = 0; u < PB; ++u) {
        const int P = (P0 + 4 * u <= I) ? P0 + 4 * u : P0;              // clamped: surplus loads repeat a valid tile and are not used
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) {
          a[u][kk] = xt[(int64_t)(P * 32 + 4 * kk) * Mp];
          b[u][kk] = lp[(int64_t)(P * 32 + 4 * kk) * Mp + J * 32];
        }
      }
#pragma unroll
      for (int u = 0; u < PB; ++u)
        if (P0 + 4 * u <= I) {
#pragma unroll
          for (int kk = 0; kk < 8; ++kk) acc = MM::mma(a[u][kk], b[u][kk], acc);
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) Sp[sl][wr * 16 + MM::crow(lane, r)][wc * 16 + lr] = acc[r];
    __syncthreads();
    if (sl == 0) {
      // left operand of the product with Dinv[J]: the sum of the four slices, element [row lr][k 4 kk + lg]
      acc_t o = acc_t{0, 0, 0, 0};
      const T* dj = Dinv + (int64_t)J * 1024 + lg * 32 + wc * 16 + lr;    // [k lg][col lr]
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const int rr = wr * 16 + lr, cc = 4 * kk + lg;
        const T sx = (Sp[0][rr][cc] + Sp[1][rr][cc]) + (Sp[2][rr][cc] + Sp[3][rr][cc]);
        o = MM::mma(sx, dj[4 * kk * 32], o);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) put(J, wr * 16 + MM::crow(lane, r), wc * 16 + lr, -o[r]);
    }
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
template <typename T> struct MMProb : NTDefaultMap, NTPlainA<T>, NTNoExtra {
  using V = typename Vec16<T>::type;
  static constexpr bool SCALE_A = false;
  static constexpr bool A_PER_REP = false;
  static constexpr int DEPTH = 1;
  const T* A; int64_t a_bs;
  const T* Bt; int64_t b_bs;
  T* C; int64_t c_bs;
  int Mp; T alpha;
  int kw;                          // > 0: split-K - the grid's y index is (slice * nb + batch): slice [s kw, (s + 1) kw) of the reduction of batch
  int nb;                          //      member `batch`; C = slabs [slice][batch][Mp][Mp] (c_bs = Mp * Mp), summed by reduce_slabs_kernel
  struct ACtx { int64_t m0; }; struct ECtx {};
  __device__ __forceinline__ int col_tiles() const { return (Mp + NTCfg<T>::CW - 1) / NTCfg<T>::CW; }
  __device__ __forceinline__ bool loop_cols() const { return false; }
  __device__ __forceinline__ int a_reuse() const { return 1; }
  __device__ __forceinline__ void krange(int64_t, int, int bz, int& kb, int& ke) const {
    if (kw > 0) { kb = (bz / nb) * kw; ke = (kb + kw < Mp) ? kb + kw : Mp; } else { kb = 0; ke = Mp; }
  }
  __device__ __forceinline__ void prepA(ACtx& c, int64_t m0, int, char*) const { c.m0 = m0; }
  __device__ __forceinline__ void prepE(ECtx&, int64_t, int) const {}
  __device__ __forceinline__ V zero() const { V z; for (int e = 0; e < Vec16<T>::N; ++e) z[e] = 0; return z; }
  __device__ __forceinline__ V loadA(const ACtx& c, int i, int k, int, int bz) const {
    const int64_t r = c.m0 + nt_stage_row<T>(i);
    return (r < Mp) ? *reinterpret_cast<const V*>(A + (kw > 0 ? bz % nb : bz) * a_bs + r * Mp + k) : zero();
  }
  __device__ __forceinline__ V loadB(int n0, int i, int k, int, int bz) const {
    const int c = n0 + nt_stage_row<T>(i);
    return (c < Mp) ? *reinterpret_cast<const V*>(Bt + (kw > 0 ? bz % nb : bz) * b_bs + (int64_t)c * Mp + k) : zero();
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
// sum_{ij} Kuu_bar * K0,  sum_{ij} Kuu_bar * dK0/dlog(ls)  and  sum_{ij} Kuu_bar * dK0/dlog(alpha), Kuu_bar = (S' + S'^T)/2 ;
// one partial triple per block
template <typename T>
__global__ void kuu_bar_reduce_kernel(const T* __restrict__ Sp, const T* __restrict__ Z, int M, int Mp, int D, int kind,
                                      const Hyper* __restrict__ h, double* __restrict__ part) {
  __shared__ double scratch[16];
  const int i = blockIdx.x;
  double s1 = 0, s2 = 0, s3 = 0;
  const T var = (T)h->var, ils2 = (T)h->inv_ls2, al = (T)h->alpha;
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    const T kb = T(0.5) * (Sp[(int64_t)i * Mp + j] + Sp[(int64_t)j * Mp + i]);
    const T r2 = sqdist<T>(Z + (int64_t)i * D, Z + (int64_t)j * D, D) * ils2;
    const T k0 = cov_from_r2<T>(kind, r2, var, al);
    s1 += (double)(kb * k0);
    s2 += (double)(kb * dcov_dlogls<T>(kind, k0, r2, var, al));
    s3 += (double)(kb * dcov_dlogalpha_from_k<T>(kind, k0, r2, al));
  }
  s1 = block_sum(s1, scratch);
  s2 = block_sum(s2, scratch);
  s3 = block_sum(s3, scratch);
  if (threadIdx.x == 0) { part[3 * i] = s1; part[3 * i + 1] = s2; part[3 * i + 2] = s3; }
}

// ---- whiten = False (pyro conditional's unwhitened branch; sparse_gdrf.py:30): the variational mean and scale factor are
// given in the space of f(Z): the forward uses u' = L^-1 u and S' = L^-1 S, the backward chains through L^-T and adds the
// L-dependence of u', S' to Lbar.  All M x M, in the solve precision.
// out[k][i] = sum_{q <= i} Linv[i][q] U[k][q]   (u' = L^-1 u), written in both precisions
template <typename T, typename TP>
__global__ void lower_matvec_kernel(const T* __restrict__ Linv, const TP* __restrict__ U, int M, int Mp, T* __restrict__ outS,
                                    TP* __restrict__ outP) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (i >= M) return;
  T s = 0;
  for (int q = 0; q <= i; ++q) s += Linv[(int64_t)i * Mp + q] * (T)U[(int64_t)k * M + q];
  outS[(int64_t)k * Mp + i] = s;
  outP[(int64_t)k * M + i] = (TP)s;
}
// out[k][q] = sum_{i >= q} Linv[i][q] ubar'[k][i]   (ubar = L^-T ubar')
template <typename T, typename TP>
__global__ void upper_matvec_kernel(const T* __restrict__ Linv, const TP* __restrict__ ub, int M, int Mp, T* __restrict__ out) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (q >= M) return;
  T s = 0;
  for (int i = q; i < M; ++i) s += Linv[(int64_t)i * Mp + q] * (T)ub[(int64_t)k * Mp + i];
  out[(int64_t)k * Mp + q] = s;
}
// a (solve precision, [K][Mp][Mp]) -> the N-side copies S and S^T
template <typename T, typename TP>
__global__ void cast_with_transpose_kernel(const T* __restrict__ a, int Mp, TP* __restrict__ S, TP* __restrict__ ST) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, k = blockIdx.z;
  if (j >= Mp) return;
  const TP v = (TP)a[((int64_t)k * Mp + i) * Mp + j];
  S[((int64_t)k * Mp + i) * Mp + j] = v;
  ST[((int64_t)k * Mp + j) * Mp + i] = v;
}
// HT[i][j] += sum_k ( P[k][i][j] + u'[k][i] * ubar[k][j] ):  the transposed extra term E^T = sum_k (S'_k Sbar_k^T + u'_k ubar_k^T) of
// Lbar = -tril(L^-T G + E)
template <typename T>
__global__ void add_et_kernel(const T* __restrict__ Pk, const T* __restrict__ up, const T* __restrict__ ub, int K, int M, int Mp,
                              T* __restrict__ HT) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (j >= M || i >= M) return;
  T s = 0;
  for (int k = 0; k < K; ++k) s += Pk[((int64_t)k * Mp + i) * Mp + j] + up[(int64_t)k * Mp + i] * ub[(int64_t)k * Mp + j];
  HT[(int64_t)i * Mp + j] += s;
}
// parameter gradients in the unwhitened case: Sbar (= L^-T Sbar', full) through the lower_cholesky transform (diag = exp(unc)),
// and ubar (= L^-T ubar')
template <typename T, typename TP>
__global__ void grad_unwhitened_kernel(const T* __restrict__ Sbar, const T* __restrict__ ub, const TP* __restrict__ Sunc, int K, int M,
                                       int Mp, double neg_inv_n, TP* __restrict__ gS, TP* __restrict__ gU) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y, k = blockIdx.z;
  if (j >= M) return;
  const int64_t pi = ((int64_t)k * Mp + i) * Mp + j, gi = ((int64_t)k * M + i) * M + j;
  double v = 0;
  if (j < i) v = (double)Sbar[pi];
  else if (j == i) v = (double)Sbar[pi] * exp((double)Sunc[gi]);
  gS[gi] = (TP)(neg_inv_n * v);
  if (i == 0) gU[(int64_t)k * M + j] = (TP)(neg_inv_n * (double)ub[(int64_t)k * Mp + j]);
}

// gradient of the loss w.r.t. the unconstrained inducing inputs (interval(0,1) constraint = sigmoid; sparse_gdrf.py:79-88):
//   Zbar[i][d] = 2/ls^2 * ( G[i][d] + 2 * sum_j Kuu_bar[i][j] * dk/dr2(z_i, z_j) * (z_id - z_jd) ),  Kuu_bar = (S' + S'^T)/2,
// G = the observation-side sums of gemm_nt<BwdKnmProb<.., true>> (all-reduced), then the chain through z = sigmoid(u):
// g = -1/N * Zbar * z (1 - z).  One block per inducing point.
template <typename T, typename TP>
__global__ void grad_z_kernel(const T* __restrict__ Sp, const T* __restrict__ Z, int M, int Mp, int D, int kind,
                              const Hyper* __restrict__ h, const double* __restrict__ G, double neg_inv_n, TP* __restrict__ g) {
  __shared__ double scratch[16];
  const int i = blockIdx.x;
  const T var = (T)h->var, ils2 = (T)h->inv_ls2, al = (T)h->alpha;
  double hs[GDRF_DMAX];
  for (int d = 0; d < GDRF_DMAX; ++d) hs[d] = 0;
  for (int j = threadIdx.x; j < M; j += blockDim.x) {
    if (j == i) continue;
    const T kb = T(0.5) * (Sp[(int64_t)i * Mp + j] + Sp[(int64_t)j * Mp + i]);
    const T r2 = sqdist<T>(Z + (int64_t)i * D, Z + (int64_t)j * D, D) * ils2;
    const T w = kb * dcov_dr2_from_k<T>(kind, cov_from_r2<T>(kind, r2, var, al), r2, al);
    for (int d = 0; d < D; ++d) hs[d] += (double)(w * (Z[(int64_t)i * D + d] - Z[(int64_t)j * D + d]));
  }
  for (int d = 0; d < D; ++d) {
    const double hsum = block_sum(hs[d], scratch);
    if (threadIdx.x == 0) {
      const double z = (double)Z[(int64_t)i * D + d];
      const double zbar = 2.0 * (double)ils2 * (G[(int64_t)i * D + d] + 2.0 * hsum);
      g[(int64_t)i * D + d] = (TP)(neg_inv_n * zbar * z * (1.0 - z));
    }
  }
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

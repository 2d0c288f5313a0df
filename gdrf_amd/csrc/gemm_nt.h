// NT GEMM core on f32/f64 MFMA 16x16x4:  C[m][n] = sum_k A[m][k] * Bt[n][k]
// Both operands have the reduction index contiguous in memory.  128x128 output tile per
// 256-thread workgroup (4 waves as 2x2, each wave 4x4 MFMA tiles), 128 bytes of k per row
// per chunk staged through LDS with a register prefetch of the next chunk.
//
// The problem object P supplies the operands and the epilogue:
//   int  col_tiles()                       number of 128-wide column tiles
//   bool loop_cols()                       true: one workgroup walks every column tile of its row tile
//   void krange(m0, n0, bz, kb, ke)        reduction range [kb, ke), multiples of BK (triangular skipping)
//   void prepA(ctx, m0, bz) / V loadA(ctx, i, k, bz)   i = 0..VPT-1 selects the thread's i-th staged row
//   V    loadB(n0, i, k, bz)
//   void tile_done(acc, m0, n0, bz, ectx)  per output tile
//   void finish(m0, bz, ectx, smem)        once per workgroup (after all its column tiles)
#pragma once
#include "common.h"

namespace gdrf {

template <typename T> struct NTCfg {
  static constexpr int BK = GDRF_KBYTES / (int)sizeof(T);   // 32 (f32) / 16 (f64)
  static constexpr int VE = 16 / (int)sizeof(T);            // elements per 16-byte vector
  static constexpr int LDK = BK + VE;                       // padded LDS row: 144 bytes
  static constexpr int KG = BK / 4;                         // k indices per lane group per chunk
  static constexpr int VPR = BK / VE;                       // 8 vectors per staged row
  static constexpr int VPT = GDRF_TILE * VPR / 256;         // 4 vectors per thread per operand
  static constexpr int LDS_BYTES = 2 * GDRF_TILE * LDK * (int)sizeof(T);
};

// row (0..127) and k offset (elements) of the i-th vector a thread stages
template <typename T> __device__ __forceinline__ int nt_stage_row(int i) { return (threadIdx.x >> 3) + 32 * i; }
template <typename T> __device__ __forceinline__ int nt_stage_k() { return (threadIdx.x & 7) * NTCfg<T>::VE; }

template <typename T, class P>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 2 : 1)) void gemm_nt_kernel(P p) {
  using C = NTCfg<T>;
  using V = typename Vec16<T>::type;
  using MM = Mfma<T>;
  using acc_t = typename MM::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + GDRF_TILE * C::LDK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
  const int nct = p.col_tiles();
  const bool loopc = p.loop_cols();
  const int64_t rtile = loopc ? (int64_t)blockIdx.x : (int64_t)blockIdx.x / nct;
  const int ct_first = loopc ? 0 : (int)(blockIdx.x % nct);
  const int ct_last = loopc ? nct : ct_first + 1;
  const int64_t m0 = rtile * GDRF_TILE;
  const int bz = blockIdx.y;

  typename P::ACtx actx;
  p.prepA(actx, m0, bz);
  typename P::ECtx ectx;
  p.prepE(ectx, m0, bz);

  const int srow_k = nt_stage_k<T>();
  for (int ct = ct_first; ct < ct_last; ++ct) {
    const int n0 = ct * GDRF_TILE;
    acc_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = acc_t{0, 0, 0, 0};
    int kb, ke;
    p.krange(m0, n0, bz, kb, ke);
    V ra[C::VPT], rb[C::VPT];
    if (kb < ke) {
#pragma unroll
      for (int i = 0; i < C::VPT; ++i) { ra[i] = p.loadA(actx, i, kb + srow_k, bz); rb[i] = p.loadB(n0, i, kb + srow_k, bz); }
    }
    for (int k = kb; k < ke; k += C::BK) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < C::VPT; ++i) {
        const int r = nt_stage_row<T>(i);
        *reinterpret_cast<V*>(&As[r * C::LDK + srow_k]) = ra[i];
        *reinterpret_cast<V*>(&Bs[r * C::LDK + srow_k]) = rb[i];
      }
      __syncthreads();
      if (k + C::BK < ke) {
#pragma unroll
        for (int i = 0; i < C::VPT; ++i) { ra[i] = p.loadA(actx, i, k + C::BK + srow_k, bz); rb[i] = p.loadB(n0, i, k + C::BK + srow_k, bz); }
      }
      // fragments: lane (lr, lg) owns k indices lg*KG .. lg*KG+KG-1 of this chunk for row/col lr
      const T* pa = &As[(wr * 64 + lr) * C::LDK + lg * C::KG];
      const T* pb = &Bs[(wc * 64 + lr) * C::LDK + lg * C::KG];
#pragma unroll
      for (int v = 0; v < C::KG / C::VE; ++v) {
        V fa[4], fb[4];
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
          fa[t4] = *reinterpret_cast<const V*>(pa + t4 * 16 * C::LDK + v * C::VE);
          fb[t4] = *reinterpret_cast<const V*>(pb + t4 * 16 * C::LDK + v * C::VE);
        }
#pragma unroll
        for (int e = 0; e < C::VE; ++e)
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = MM::mma(fa[a][e], fb[b][e], acc[a][b]);
      }
    }
    p.tile_done(acc, m0, n0, bz, ectx, wr, wc, lane);
  }
  p.finish(m0, bz, ectx, smem, wr, wc, lane);
}

// element (row, col) of accumulator register r of MFMA tile (a, b) inside the workgroup tile
template <typename T> __device__ __forceinline__ int nt_acc_row(int wr, int a, int lane, int r) {
  return wr * 64 + a * 16 + Mfma<T>::crow(lane, r);
}
__device__ __forceinline__ int nt_acc_col(int wc, int b, int lane) { return wc * 64 + b * 16 + (lane & 15); }

}  // namespace gdrf

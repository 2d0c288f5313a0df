// TN GEMM core (reduction over observations):  C[i][j] = sum_n A[n][i] * s[n] * B[n][j]
// A, B row-major (n slow).  128x128 output tile per 256-thread workgroup; the rows are split
// over gridDim.z workgroups, each writing its partial tile to a slab (summed by reduce_slabs,
// deterministic: no float atomics).
#pragma once
#include "common.h"

namespace gdrf {

template <typename T> struct TNCfg {
  static constexpr int BR = 128 / (int)sizeof(T);           // rows per chunk: 32 (f32) / 16 (f64)
  static constexpr int VE = 16 / (int)sizeof(T);
  static constexpr int LDC = GDRF_TILE + 16;                // LDS row stride (elements): bank-offset 16
  static constexpr int VPR = GDRF_TILE / VE;                // vectors per staged row
  static constexpr int VPT = BR * VPR / 256;                // 4
  static constexpr int LDS_BYTES = 2 * BR * LDC * (int)sizeof(T);
};

template <typename T> struct TNArgs {
  const T* A; int64_t lda;           // [nrows][lda]
  const T* B; int64_t ldb;           // [nrows][ldb]
  const T* scale; int64_t scale_bs;  // optional per-row scale, batch stride (elements); nullptr = 1
  int64_t nrows;                     // rows of this rank
  int64_t rows_per_split;            // multiple of BR
  int ncols;                         // Mp
  int sym;                           // 1: only tiles ti >= tj (blockIdx.x enumerates the lower triangle)
  T* slab;                           // [nsplit][nbatch][ncols][ncols]
  int nbatch;
  int nsplit;                        // gridDim.x = ntiles * nbatch * nsplit
};

template <typename T>
__global__ __launch_bounds__(256, (sizeof(T) == 4 ? 2 : 1)) void gemm_tn_kernel(TNArgs<T> g) {
  using C = TNCfg<T>;
  using V = typename Vec16<T>::type;
  using MM = Mfma<T>;
  using acc_t = typename MM::acc_t;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + C::BR * C::LDC;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
  const int nt = (g.ncols + GDRF_TILE - 1) / GDRF_TILE;
  const int ntiles = g.sym ? nt * (nt + 1) / 2 : nt * nt;
  // block -> (tile, batch, split).  With a multiple of 8 splits, split sp runs on the blocks with
  // blockIdx % 8 == sp % 8 (observed to share an XCD, MI355X_MICROARCH.md): the tiles of one split walk the
  // same rows of W at the same time, so each XCD's L2 serves one row window to all of them.  Speed only.
  int tile, b, sp;
  {
    const unsigned bid = blockIdx.x;
    if ((g.nsplit & 7) == 0) {
      const unsigned xcd = bid & 7u, idx = bid >> 3, per = (unsigned)(ntiles * g.nbatch);
      const unsigned sl = idx / per, r = idx - sl * per;
      sp = (int)(sl * 8u + xcd); b = (int)(r / (unsigned)ntiles); tile = (int)(r % (unsigned)ntiles);
    } else {
      tile = (int)(bid % (unsigned)ntiles);
      const unsigned r = bid / (unsigned)ntiles;
      b = (int)(r % (unsigned)g.nbatch); sp = (int)(r / (unsigned)g.nbatch);
    }
  }
  int ti, tj;
  if (g.sym) {
    int t = tile; ti = 0;
    while (t >= ti + 1) { t -= ti + 1; ++ti; }
    tj = t;
  } else { ti = tile / nt; tj = tile % nt; }
  const int i0 = ti * GDRF_TILE, j0 = tj * GDRF_TILE;
  const int64_t r0 = (int64_t)sp * g.rows_per_split;
  int64_t r1 = r0 + g.rows_per_split; if (r1 > g.nrows) r1 = g.nrows;
  const T* sc = g.scale ? g.scale + (int64_t)b * g.scale_bs : nullptr;

  acc_t acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[a][c] = acc_t{0, 0, 0, 0};

  const int srow = tid / C::VPR;                 // + (256/VPR)*i
  const int scol = (tid % C::VPR) * C::VE;
  const bool a_ok = (i0 + scol) < g.ncols, b_ok = (j0 + scol) < g.ncols;   // ncols multiple of 32 >= VE
  V ra[C::VPT], rb[C::VPT];
  T rs[C::VPT];          // row scales travel with the prefetch and are applied at the LDS store, so that the
                         // loads stay in flight behind the MFMAs instead of being waited for here
  auto gload = [&](int64_t rbase) {
#pragma unroll
    for (int i = 0; i < C::VPT; ++i) {
      const int64_t n = rbase + srow + (256 / C::VPR) * i;
      V va, vb;
#pragma unroll
      for (int e = 0; e < C::VE; ++e) { va[e] = 0; vb[e] = 0; }
      T s = T(1);
      if (n < r1) {
        if (a_ok) va = *reinterpret_cast<const V*>(g.A + n * g.lda + i0 + scol);
        if (b_ok) vb = *reinterpret_cast<const V*>(g.B + n * g.ldb + j0 + scol);
        if (sc) s = sc[n];
      }
      ra[i] = va; rb[i] = vb; rs[i] = s;
    }
  };
  if (r0 < r1) gload(r0);
  for (int64_t r = r0; r < r1; r += C::BR) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < C::VPT; ++i) {
      const int rr = srow + (256 / C::VPR) * i;
      V vb = rb[i];
#pragma unroll
      for (int e = 0; e < C::VE; ++e) vb[e] *= rs[i];
      *reinterpret_cast<V*>(&As[rr * C::LDC + scol]) = ra[i];
      *reinterpret_cast<V*>(&Bs[rr * C::LDC + scol]) = vb;
    }
    __syncthreads();
    if (r + C::BR < r1) gload(r + C::BR);
#pragma unroll
    for (int t = 0; t < C::BR / 4; ++t) {
      const int kk = 4 * t + lg;
      T fa[4], fb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fa[q] = As[kk * C::LDC + wr * 64 + q * 16 + lr];
        fb[q] = Bs[kk * C::LDC + wc * 64 + q * 16 + lr];
      }
#pragma unroll
      for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[a][c] = MM::mma(fa[a], fb[c], acc[a][c]);
    }
  }
  T* out = g.slab + ((int64_t)sp * g.nbatch + b) * (int64_t)g.ncols * g.ncols;
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = i0 + wr * 64 + a * 16 + MM::crow(lane, r);
        const int j = j0 + wc * 64 + c * 16 + lr;
        if (i < g.ncols && j < g.ncols) out[(int64_t)i * g.ncols + j] = acc[a][c][r];
      }
}

}  // namespace gdrf

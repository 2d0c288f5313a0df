// The dominant contraction  Wbar = sum_k diag(2 vbar_k) W B_k  on the bf16 matrix path with exact-split emulation.
//
// gfx950 runs f32-input MFMA at 1/16 of the bf16 rate (MI355X_MICROARCH.md: 157 TF vs ~2.5 PF), and ~90 % of the step is
// bound by it.  Here every f32 operand x is written as x = x1 + x2 + x3 with bf16 pieces (x1 = bf16(x), x2 = bf16(x - x1),
// x3 = bf16(x - x1 - x2): 3 x 8 significand bits, residual <= 2^-24 |x|), and a product a*b is the sum of the SIX cross
// terms of weight >= 2^-16:  a1b1 + (a1b2 + a2b1) + (a1b3 + a2b2 + a3b1).  The dropped terms are <= 2^-24 |ab|, i.e. the
// rounding of one f32 product, so the result has f32-level error while running on mfma_f32_16x16x32_bf16 with f32
// accumulation (16/6 = 2.7x the f32 MFMA rate).  B_k is split once per step (split3_kernel), W chunks are split when they are
// staged to LDS (once per chunk, reused by all K topics), the per-(topic,row) factor 2 vbar_kn scales the f32 partial
// product of each chunk before it is added to the tile accumulator.
//
// Same tiling / block map / epilogue as gemm_nt<BwdWbarProb> (gemm_nt.h, kernels_n.h); parity: tests/test_gpu_parity.py.
#pragma once
#include "common.h"
#include "gemm_nt.h"

namespace gdrf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Bf16x6Cfg {
  static constexpr int BK = 32;                 // f32 reduction indices per chunk = one 16x16x32 MFMA deep
  static constexpr int LDH = BK + 8;            // LDS row stride in halfwords: 80 bytes (16-byte aligned, bank-shifted)
  static constexpr int PIECE = GDRF_TILE * LDH; // halfwords per piece image of a 128-row operand tile
  static constexpr int LDS_BYTES = 2 * 3 * PIECE * 2;    // A and B, 3 pieces each: 61440
};

__device__ __forceinline__ void split3(float x, __bf16& h, __bf16& m, __bf16& l) {
  h = (__bf16)x;
  const float r1 = x - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// out[p][i] = p-th bf16 piece of in[i]
__global__ void split3_kernel(const float* __restrict__ in, int64_t n, __bf16* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  __bf16 h, m, l;
  split3(in[i], h, m, l);
  out[i] = h; out[n + i] = m; out[2 * n + i] = l;
}

struct BwdWbarBf16Args {
  const float* W; int64_t nrows; int M, Mp, K;
  const __bf16* Bh; int64_t piece_stride;     // Bh[p][k][col][i], piece_stride = K*Mp*Mp
  const float* vbar; const float* locbar; int64_t ldk;
  const float* asum; const float* U; float* Wbar;
};

__global__ __launch_bounds__(256, 2) void bwd_wbar_bf16x6_kernel(BwdWbarBf16Args g) {
  using CF = Bf16x6Cfg;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  __bf16* As = reinterpret_cast<__bf16*>(smem);               // [3][128][LDH]
  __bf16* Bs = As + 3 * CF::PIECE;                            // [3][128][LDH]
  float* scaleS = reinterpret_cast<float*>(smem + CF::LDS_BYTES);   // [K][128]  2 vbar_kn

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, lr = lane & 15, lg = lane >> 4;
  const int nct = (g.Mp + GDRF_TILE - 1) / GDRF_TILE;
  int64_t rtile; int ct;
  NTXcdMap{}.map_block(blockIdx.x, nct, false, rtile, ct);
  const int64_t m0 = rtile * GDRF_TILE;
  const int n0 = ct * GDRF_TILE;
  const int K = g.K, Mp = g.Mp;

  for (int e = tid; e < K * GDRF_TILE; e += 256) {
    const int k = e / GDRF_TILE, r = e - k * GDRF_TILE;
    scaleS[e] = (m0 + r < g.nrows) ? 2.0f * g.vbar[(int64_t)k * g.ldk + m0 + r] : 0.0f;
  }
  // staging map.  A: 4 float4 per thread (row (tid>>3)+32i, k offset 4*(tid&7)).  B pieces: 2 x 16-byte vectors of 8 bf16
  // per thread and piece (vector v = tid + 256 j: row v>>2, k offset 8*(v&3)).
  const int a_k = (tid & 7) * 4;
  const float* a_ptr[4]; bool a_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int64_t r = m0 + (tid >> 3) + 32 * i;
    a_ok[i] = r < g.nrows;
    a_ptr[i] = g.W + (a_ok[i] ? r : 0) * Mp;
  }
  f32x4 acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = f32x4{0, 0, 0, 0};

  f32x4 ra[4];
  bf16x8 rb[3][2];
  auto load_a = [&](int kA) {
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = a_ok[i] ? *reinterpret_cast<const f32x4*>(a_ptr[i] + kA + a_k) : f32x4{0, 0, 0, 0};
  };
  auto load_b = [&](int kA, int rep) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = tid + 256 * j, row = v >> 2, kq = (v & 3) * 8, col = n0 + row;
#pragma unroll
      for (int p = 0; p < 3; ++p) {
        bf16x8 z;
#pragma unroll
        for (int e = 0; e < 8; ++e) z[e] = (__bf16)0.0f;
        rb[p][j] = (col < Mp) ? *reinterpret_cast<const bf16x8*>(g.Bh + p * g.piece_stride + ((int64_t)rep * Mp + col) * Mp + kA + kq) : z;
      }
    }
  };
  auto store_a = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      bf16x4 h, m, l;
#pragma unroll
      for (int e = 0; e < 4; ++e) { __bf16 x1, x2, x3; split3(ra[i][e], x1, x2, x3); h[e] = x1; m[e] = x2; l[e] = x3; }
      const int off = ((tid >> 3) + 32 * i) * CF::LDH + a_k;
      *reinterpret_cast<bf16x4*>(As + off) = h;
      *reinterpret_cast<bf16x4*>(As + CF::PIECE + off) = m;
      *reinterpret_cast<bf16x4*>(As + 2 * CF::PIECE + off) = l;
    }
  };
  auto store_b = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int v = tid + 256 * j, off = (v >> 2) * CF::LDH + (v & 3) * 8;
#pragma unroll
      for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(Bs + p * CF::PIECE + off) = rb[p][j];
    }
  };
  // one staged chunk: P = A B^T through the six cross products (small terms first), then acc += diag(scale) P
  auto compute = [&](const float* sc_row /* scaleS + rep*128, or nullptr for scale 1 */) {
    bf16x8 fb[3][4];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        fb[p][b] = *reinterpret_cast<const bf16x8*>(Bs + p * CF::PIECE + (wc * 64 + b * 16 + lr) * CF::LDH + lg * 8);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      bf16x8 fa[3];
#pragma unroll
      for (int p = 0; p < 3; ++p)
        fa[p] = *reinterpret_cast<const bf16x8*>(As + p * CF::PIECE + (wr * 64 + a * 16 + lr) * CF::LDH + lg * 8);
      f32x4 s4 = f32x4{1, 1, 1, 1};
      if (sc_row) s4 = *reinterpret_cast<const f32x4*>(sc_row + wr * 64 + a * 16 + lg * 4);   // rows 4*lg + r of this MFMA tile
      f32x4 P[4];
#pragma unroll
      for (int b = 0; b < 4; ++b) P[b] = f32x4{0, 0, 0, 0};
#pragma unroll
      for (int b = 0; b < 4; ++b) P[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[2][b], P[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) P[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[1][b], P[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) P[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[2], fb[0][b], P[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) P[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[1][b], P[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) P[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[1], fb[0][b], P[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b) P[b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[0], fb[0][b], P[b], 0, 0, 0);
#pragma unroll
      for (int b = 0; b < 4; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[a][b][r] += s4[r] * P[b][r];
    }
  };

  const int nk = Mp / CF::BK, nchunks = nk * K;
  load_a(0);
  load_b(0, 0);
  for (int c = 0; c < nchunks; ++c) {
    const int q = c / K, rep = c - q * K;
    __syncthreads();
    if (rep == 0) store_a();
    store_b();
    __syncthreads();
    if (c + 1 < nchunks) {
      const int q1 = (c + 1) / K, rep1 = (c + 1) - q1 * K;
      if (rep1 == 0) load_a(q1 * CF::BK);
      load_b(q1 * CF::BK, rep1);
    }
    compute(scaleS + rep * GDRF_TILE);
  }
  // rank-K epilogue term locbar^T U as extra chunk(s), split on the fly
  for (int x = 0; x * CF::BK < K; ++x) {
    __syncthreads();
    for (int e = tid; e < GDRF_TILE * CF::BK; e += 256) {
      const int kk = e / GDRF_TILE, row = e - kk * GDRF_TILE, k = x * CF::BK + kk;
      const float va = (k < K && m0 + row < g.nrows) ? g.locbar[(int64_t)k * g.ldk + m0 + row] : 0.0f;
      const float vb = (k < K && n0 + row < g.M) ? g.U[(int64_t)k * g.M + n0 + row] : 0.0f;
      __bf16 h, m, l;
      split3(va, h, m, l);
      As[row * CF::LDH + kk] = h; As[CF::PIECE + row * CF::LDH + kk] = m; As[2 * CF::PIECE + row * CF::LDH + kk] = l;
      split3(vb, h, m, l);
      Bs[row * CF::LDH + kk] = h; Bs[CF::PIECE + row * CF::LDH + kk] = m; Bs[2 * CF::PIECE + row * CF::LDH + kk] = l;
    }
    __syncthreads();
    compute(nullptr);
  }
  // Wbar = acc - 2 asum W
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int64_t m = m0 + wr * 64 + a * 16 + lg * 4 + r;
      if (m >= g.nrows) continue;
      const float as2 = 2.0f * g.asum[m];
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int n = n0 + wc * 64 + b * 16 + lr;
        if (n < Mp) g.Wbar[m * Mp + n] = acc[a][b][r] - as2 * g.W[m * Mp + n];
      }
    }
}

}  // namespace gdrf
